"""Multitaper spectral estimation: host side of DSP.jl ``src/multitaper.jl`` over libmi355dsp (``mdsp_mt_*``).

``MTConfig`` / ``mt_pgram`` / ``MTSpectrogramConfig`` / ``mt_spectrogram`` / ``MTCrossSpectraConfig`` /
``mt_cross_power_spectra`` / ``MTCoherenceConfig`` / ``mt_coherence`` keep the reference's signatures, defaults,
checks and normalisation; tapering, the transforms, ``fft2pow!``, ``cs_inner!`` and ``coherence_from_cs!`` run on the GPU.
Julia's type parameter ``T`` is the first positional argument here (``MTConfig(np.float64, n_samples; ...)``).
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass

import numpy as np

from . import _dev, _lib, _plancache, util, windows
from ._lib import ArgumentError, DimensionMismatch
from .periodograms import Periodogram, Spectrogram, frame_count


def _nextpow2(n: int) -> int:
    return 1 << max(0, (int(n) - 1).bit_length())


class MTConfig:
    """``MTConfig{T}(n_samples; fs=1, nfft=nextpow(2, n_samples), window=nothing, nw=4, ntapers=2nw-1,
    taper_weights=fill(1/ntapers, ntapers), onesided=T<:Real)`` (multitaper.jl:5-49, :112-135)."""

    def __init__(self, T, n_samples: int, *, fs=1, nfft: int | None = None, window=None, nw=4, ntapers: int | None = None,
                 taper_weights=None, onesided: bool | None = None, engine: int = _lib.ENGINE_AUTO):
        T = np.dtype(T)
        if T.kind not in "fc":
            T = util.fftintype(T)
        onesided = (T.kind != "c") if onesided is None else bool(onesided)
        if onesided and T.kind == "c":
            raise ArgumentError("cannot compute one-sided FFT of a complex signal")               # :115-117
        n_samples = int(n_samples)
        if not n_samples > 0:
            raise ArgumentError("`n_samples` must be positive")                                   # :118
        nfft = _nextpow2(n_samples) if nfft is None else int(nfft)
        if not nfft >= n_samples:
            raise ArgumentError("Must have `nfft >= n_samples`")                                  # :119
        ntapers = int(2 * nw - 1) if ntapers is None else int(ntapers)
        if taper_weights is None:
            taper_weights = np.full(max(ntapers, 0), 1 / ntapers if ntapers else 0.0)
        taper_weights = np.asarray(taper_weights, dtype=np.float64)
        self.freq = util.rfftfreq(nfft, fs) if onesided else util.fftfreq(nfft, fs)
        if window is None:
            r = fs / taper_weights                                                                # :127
            window = windows.dpss(n_samples, nw, ntapers)                                         # :128
        else:
            window = np.asarray(window.cpu() if hasattr(window, "cpu") else window, dtype=np.float64)
            if window.ndim != 2:
                raise DimensionMismatch("Must have `size(window) == (n_samples, ntapers)`")
            if window.shape[1] != len(taper_weights):                                            # the broadcast of :130 throws
                raise DimensionMismatch(f"arrays could not be broadcast to a common size; got {window.shape[1]} tapers and {len(taper_weights)} weights")
            r = fs * np.sum(np.abs(window) ** 2, axis=0) / taper_weights                         # :130
        if not ntapers > 0:
            raise ArgumentError("`ntapers` must be positive")                                     # :23
        if not fs > 0:
            raise ArgumentError("`fs` must be positive")                                          # :24
        if window.shape != (n_samples, ntapers):
            raise DimensionMismatch(f"Must have `size(window) == (n_samples, ntapers)`; got {window.shape}")   # :34-36
        r = np.asarray(r, dtype=np.float64)
        if r.shape != (ntapers,):
            raise DimensionMismatch("Must have `size(r) == (ntapers,)`")                          # :37-39
        self.T, self.n_samples, self.fs, self.nfft, self.ntapers = T, n_samples, fs, nfft, ntapers
        self.window, self.onesided, self.r = window, onesided, r
        self._h = C.c_void_p()
        wcol = np.asfortranarray(window)                                                         # (n, ntapers) column-major
        _lib.check(_lib.lib().mdsp_mt_plan_create(C.byref(self._h), n_samples, nfft, wcol.ctypes.data_as(C.c_void_p), ntapers,
                                                  np.ascontiguousarray(r).ctypes.data_as(C.c_void_p), int(onesided), _dev.md_dtype(T), engine))
        no, nt, eng = C.c_int64(), C.c_int64(), C.c_int()
        _lib.check(_lib.lib().mdsp_mt_plan_info(self._h, C.byref(no), C.byref(nt), C.byref(eng)))
        self.nout, self.engine = no.value, eng.value

    def __del__(self):
        try:
            if self._h:
                _lib.lib().mdsp_mt_plan_destroy(self._h)
        except Exception:
            pass


def dpss_config(T, n_samples, *, nw=4, ntapers=None, fs=1, keep_only_large_evals=False, weight_by_evals=False, **kw) -> MTConfig:
    """``DSP.Periodograms.dpss_config`` (multitaper.jl:52-78)."""
    ntapers = int(2 * nw - 1) if ntapers is None else int(ntapers)
    window = windows.dpss(n_samples, nw, ntapers)
    evals = None
    if keep_only_large_evals:
        evals = windows.dpsseig(window, nw)
        keep = evals > 0.9
        window, evals = window[:, keep], evals[keep]
        ntapers = window.shape[1]
    if weight_by_evals:
        if evals is None:
            evals = windows.dpsseig(window, nw)
        weights = evals / evals.sum()
    else:
        weights = np.full(ntapers, 1 / ntapers)
    return MTConfig(T, n_samples, window=window, nw=nw, ntapers=ntapers, taper_weights=weights, fs=fs, **kw)


def _psd(config: MTConfig, signal, noverlap: int):
    """(nch, K, nout) device tensor of multitaper PSDs of the frames of each channel of ``signal``."""
    cols, shape = _dev.to_columns(signal, config.T)
    nch, length = cols.shape
    K = frame_count(length, config.n_samples, noverlap)
    out = _dev.torch.zeros((nch, K, config.nout), dtype=_dev.torch_dtype(util.fftabs2type(config.T)), device=cols.device)   # output .= 0, :239
    if K and nch:
        _lib.check(_lib.lib().mdsp_mt_psd_exec(config._h, _dev.ptr(cols), length, noverlap, nch, length, _dev.ptr(out), config.nout,
                                               K * config.nout, _dev.stream_ptr()))
    return out, shape


def _signal_T(signal):
    dt = _dev.np_dtype_of(signal)
    return dt if dt.kind in "fc" else util.fftintype(dt)


def mt_pgram(s, config: MTConfig | None = None, *, onesided: bool | None = None, nfft: int | None = None, fs=1, nw=4,
             ntapers: int | None = None, window=None, engine: int = _lib.ENGINE_AUTO) -> Periodogram:
    """``mt_pgram(s; onesided, nfft=nextfastfft(n), fs, nw, ntapers=ceil(2nw)-1, window)`` /
    ``mt_pgram(signal, config::MTConfig)`` (multitaper.jl:177-245)."""
    if len(s.shape) != 1:
        raise ArgumentError("mt_pgram expects a vector")
    if config is None:   # the keyword form builds its config on every call: keep the most recent ones (tapers in HBM, transforms)
        n = int(s.shape[0])
        nfft_ = util.nextfastfft(n) if nfft is None else nfft
        ntap = math.ceil(2 * nw) - 1 if ntapers is None else ntapers
        key = ("mt", _plancache.ctx_key(), np.dtype(_signal_T(s)).str, n, float(fs), nfft_, _plancache.window_key(window), float(nw), ntap,
               onesided, engine)
        config = _plancache.plans.get(key, lambda: MTConfig(_signal_T(s), n, fs=fs, nfft=nfft_, window=window, nw=nw, ntapers=ntap,
                                                           onesided=onesided, engine=engine))
    if int(s.shape[0]) != config.n_samples:
        raise DimensionMismatch(f"Expected `signal` to be of length `config.n_samples`; got {int(s.shape[0])} and {config.n_samples}")   # :230-233
    out, _ = _psd(config, s, 0)
    res = out[0, 0]
    return Periodogram(res if _dev.is_device_array(s) else res.cpu().numpy(), config.freq)


def mt_pgram_(output, s, config: MTConfig | None = None, **kw) -> Periodogram:
    """``mt_pgram!(output, s, config)`` (multitaper.jl:225-245)."""
    if config is not None and len(output) != len(config.freq):
        raise DimensionMismatch("Expected `output` to be of length `length(config.freq)`")        # :226-229
    p = mt_pgram(s, config, **kw)
    if len(output) != len(p.freq):
        raise DimensionMismatch("Expected `output` to be of length `length(config.freq)`")
    _assign(output, p.power)
    return Periodogram(output, p.freq)


def _assign(out, res):
    if isinstance(out, np.ndarray):
        out[...] = res if isinstance(res, np.ndarray) else res.cpu().numpy()
    else:
        out.copy_(res if not isinstance(res, np.ndarray) else _dev.torch.from_numpy(res))


class MTSpectrogramConfig:
    """``MTSpectrogramConfig{T}(n_samples, samples_per_window, n_overlap_samples; fs=1, kwargs...)`` /
    ``MTSpectrogramConfig(n_samples, mt_config, n_overlap_samples)`` (multitaper.jl:248-287)."""

    def __init__(self, *args, **kw):
        if len(args) == 3 and isinstance(args[1], MTConfig):
            n_samples, mt_config, n_overlap = args
        elif len(args) == 4:
            T, n_samples, spw, n_overlap = args
            mt_config = MTConfig(T, spw, **kw)
        else:
            raise TypeError("MTSpectrogramConfig(T, n_samples, samples_per_window, n_overlap; ...) or (n_samples, mt_config, n_overlap)")
        spw = mt_config.n_samples
        if spw <= n_overlap:
            raise ArgumentError(f"Need `samples_per_window > n_overlap_samples`; got {spw} and {n_overlap}.")   # :264-266
        hop = spw - n_overlap
        length = 0 if n_samples < spw else (n_samples - spw) // hop + 1
        self.time = (spw / 2 + hop * np.arange(length)) / mt_config.fs                          # :270
        self.n_samples, self.n_overlap_samples, self.mt_config = int(n_samples), int(n_overlap), mt_config


def mt_spectrogram(signal, *args, **kw) -> Spectrogram:
    """``mt_spectrogram(signal, n=length>>3, n_overlap=n>>1; fs, onesided, kwargs...)``,
    ``mt_spectrogram(signal, config::MTSpectrogramConfig)``, ``mt_spectrogram(signal, mt_config::MTConfig, n_overlap=n>>1)``
    (multitaper.jl:336-392)."""
    if len(signal.shape) != 1:
        raise ArgumentError("mt_spectrogram expects a vector")
    length = int(signal.shape[0])
    if args and isinstance(args[0], MTSpectrogramConfig):
        config = args[0]
    elif args and isinstance(args[0], MTConfig):
        mc = args[0]
        config = MTSpectrogramConfig(length, mc, args[1] if len(args) > 1 else mc.n_samples >> 1)
    else:
        n = args[0] if len(args) > 0 else length >> 3
        n_overlap = args[1] if len(args) > 1 else n >> 1
        try:
            key = ("mtspec", _plancache.ctx_key(), np.dtype(_signal_T(signal)).str, length, int(n), int(n_overlap),
                   tuple(sorted((k, _plancache.window_key(v) if k == "window" else v) for k, v in kw.items())))
            config = _plancache.plans.get(key, lambda: MTSpectrogramConfig(_signal_T(signal), length, n, n_overlap, **kw))
        except TypeError:       # an unhashable keyword value: no caching
            config = MTSpectrogramConfig(_signal_T(signal), length, n, n_overlap, **kw)
    if length != config.n_samples:
        raise DimensionMismatch(f"Expected `signal` to be of length `config.n_samples`; got {length} and {config.n_samples}")   # :318-321
    mc = config.mt_config
    out, _ = _psd(mc, signal, config.n_overlap_samples)
    assert out.shape[1] == len(config.time)                                                     # :324
    res = out[0].t()                                                                            # (nfreq, ntime)
    return Spectrogram(res if _dev.is_device_array(signal) else res.cpu().numpy(), mc.freq, config.time)


def mt_spectrogram_(destination, signal, *args, **kw) -> Spectrogram:
    """``mt_spectrogram!(destination, signal, ...)`` (multitaper.jl:305-330)."""
    if args and isinstance(args[0], MTSpectrogramConfig):
        cfg = args[0]
        if tuple(destination.shape) != (len(cfg.mt_config.freq), len(cfg.time)):
            raise DimensionMismatch("Expected `destination` to be of size `(length(config.mt_config.freq), length(config.time))`")   # :314-317
    sp = mt_spectrogram(signal, *args, **kw)
    if tuple(destination.shape) != tuple(sp.power.shape):
        raise DimensionMismatch("Expected `destination` to be of size `(length(config.mt_config.freq), length(config.time))`")
    _assign(destination, sp.power)
    return Spectrogram(destination, sp.freq, sp.time)


@dataclass
class CrossPowerSpectra:
    """multitaper.jl:409-415."""
    power: object
    freq: np.ndarray


@dataclass
class Coherence:
    """multitaper.jl:734-747."""
    coherence: object
    freq: np.ndarray


def coherence(c: Coherence):
    return c.coherence


class MTCrossSpectraConfig:
    """``MTCrossSpectraConfig{T}(n_channels, n_samples; fs=1, demean=false, freq_range=nothing, kwargs...)`` /
    ``MTCrossSpectraConfig(n_channels, mt_config; demean, freq_range)`` (multitaper.jl:424-516)."""

    def __init__(self, *args, demean: bool = False, freq_range=None, **kw):
        if len(args) == 2 and isinstance(args[1], MTConfig):
            n_channels, mt_config = args
        elif len(args) == 3:
            T, n_channels, n_samples = args
            mt_config = MTConfig(T, n_samples, **kw)
        else:
            raise TypeError("MTCrossSpectraConfig(T, n_channels, n_samples; ...) or MTCrossSpectraConfig(n_channels, mt_config; ...)")
        if mt_config.T.kind == "c" or not mt_config.onesided:                                   # check_onesided_real :417-422
            raise ArgumentError("Only real data is supported (with the default choice of `onesided=true`) for this operation.")
        self.n_channels, self.mt_config, self.demean, self.freq_range = int(n_channels), mt_config, bool(demean), freq_range
        self.normalization_weights = 2 / mt_config.r                                             # :499
        if freq_range is not None:
            mask = (freq_range[0] < mt_config.freq) & (mt_config.freq < freq_range[-1])         # :503
            self.freq_inds = np.flatnonzero(mask).astype(np.int64)
            self.freq = mt_config.freq[mask]
        else:
            self.freq_inds = np.arange(len(mt_config.freq), dtype=np.int64)
            self.freq = mt_config.freq


def _cross(signal, config: MTCrossSpectraConfig):
    """(device tensor (nfi, nch, nch) = power[l, m, fi] transposed to C order, like-host flag)."""
    mc = config.mt_config
    if tuple(signal.shape) != (config.n_channels, mc.n_samples):
        raise DimensionMismatch(f"Size of `signal` does not match `(config.n_channels, config.mt_config.n_samples)`; got {tuple(signal.shape)}")   # :557-560
    t = _dev.as_device(signal, mc.T).contiguous()              # (nch, n_samples): one channel per contiguous row
    nch = config.n_channels
    cdt = _dev.torch_dtype(util.fftouttype(mc.T))
    xmt = _dev.torch.empty((nch, mc.ntapers, mc.nout), dtype=cdt, device=t.device)               # x_mt[f, taper, ch], f fastest
    _lib.check(_lib.lib().mdsp_mt_spectra_exec(mc._h, _dev.ptr(t), nch, mc.n_samples, int(config.demean), _dev.ptr(xmt), _dev.stream_ptr()))
    nfi = len(config.freq_inds)
    out = _dev.torch.empty((nfi, nch, nch), dtype=cdt, device=t.device)                          # out[l, m, fi], l fastest
    fi = np.ascontiguousarray(config.freq_inds, dtype=np.int64)
    _lib.check(_lib.lib().mdsp_mt_cross_spectra(mc._h, _dev.ptr(xmt), nch, fi.ctypes.data_as(C.c_void_p), nfi, _dev.ptr(out), _dev.stream_ptr()))
    return out


def _cs_config(signal, config, kw):
    if config is not None:
        return config
    dt = _dev.np_dtype_of(signal)
    if dt.kind == "c":
        raise ArgumentError("Only real data is supported (with the default choice of `onesided=true`) for this operation.")
    T = dt if dt.kind == "f" else np.dtype(np.float64)
    nch, ns = int(signal.shape[0]), int(signal.shape[1])
    try:                        # keyword form: keep the most recent configs (tapers, transforms, work buffers) alive
        key = ("mtcs", _plancache.ctx_key(), np.dtype(T).str, nch, ns,
               tuple(sorted((k, _plancache.window_key(v) if k == "window" else (tuple(v) if isinstance(v, (list, tuple)) else v)) for k, v in kw.items())))
        return _plancache.plans.get(key, lambda: MTCrossSpectraConfig(T, nch, ns, **kw))
    except TypeError:           # an unhashable keyword value: no caching
        return MTCrossSpectraConfig(T, nch, ns, **kw)


def mt_cross_power_spectra(signal, config: MTCrossSpectraConfig | None = None, **kw) -> CrossPowerSpectra:
    """``mt_cross_power_spectra(signal::AbstractMatrix; fs=1, kwargs...)`` / ``(signal, config)`` (multitaper.jl:640-650):
    ``signal`` is (n_channels, n_samples); ``power`` is (n_channels, n_channels, n_frequencies)."""
    config = _cs_config(signal, config, kw)
    out = _cross(signal, config)
    res = out.permute(2, 1, 0)                                  # [l, m, fi]
    return CrossPowerSpectra(res if _dev.is_device_array(signal) else res.cpu().numpy(), config.freq)


def mt_cross_power_spectra_(output, signal, config: MTCrossSpectraConfig | None = None, **kw) -> CrossPowerSpectra:
    """``mt_cross_power_spectra!`` (multitaper.jl:544-585)."""
    config = _cs_config(signal, config, kw)
    if tuple(output.shape) != (config.n_channels, config.n_channels, len(config.freq_inds)):
        raise DimensionMismatch("Size of `output` does not match `(config.n_channels, config.n_channels, length(config.freq_inds))`")   # :561-564
    cs = mt_cross_power_spectra(signal, config)
    _assign(output, cs.power)
    return CrossPowerSpectra(output, cs.freq)


class MTCoherenceConfig:
    """multitaper.jl:656-691: wraps an ``MTCrossSpectraConfig``."""

    def __init__(self, *args, **kw):
        if len(args) == 1 and isinstance(args[0], MTCrossSpectraConfig):
            self.cs_config = args[0]
        else:
            self.cs_config = MTCrossSpectraConfig(*args, **kw)


def mt_coherence(signal, config: MTCoherenceConfig | None = None, **kw) -> Coherence:
    """``mt_coherence(signal::AbstractMatrix; fs=1, freq_range=nothing, demean=false, kwargs...)`` / ``(signal, config)``
    (multitaper.jl:765-817)."""
    cs_config = config.cs_config if config is not None else _cs_config(signal, None, kw)
    mc = cs_config.mt_config
    if tuple(signal.shape) != (cs_config.n_channels, mc.n_samples):
        raise DimensionMismatch("Size of `signal` does not match `(config.cs_config.n_channels, config.cs_config.mt_config.n_samples)`")   # :770-773
    cs = _cross(signal, cs_config)
    nfi, nch = cs.shape[0], cs.shape[1]
    rdt = util.fftabs2type(mc.T)
    out = _dev.torch.empty((nfi, nch, nch), dtype=_dev.torch_dtype(rdt), device=cs.device)
    _lib.check(_lib.lib().mdsp_coherence_from_cs(_dev.ptr(cs), nch, nfi, _dev.md_dtype(rdt), _dev.ptr(out), _dev.stream_ptr()))
    res = out.permute(2, 1, 0)
    return Coherence(res if _dev.is_device_array(signal) else res.cpu().numpy(), cs_config.freq)


def mt_coherence_(output, signal, config: MTCoherenceConfig | None = None, **kw) -> Coherence:
    """``mt_coherence!`` (multitaper.jl:765-790)."""
    cs_config = config.cs_config if config is not None else _cs_config(signal, None, kw)
    if tuple(output.shape) != (cs_config.n_channels, cs_config.n_channels, len(cs_config.freq)):
        raise DimensionMismatch("Size of `output` does not match `(config.cs_config.n_channels, config.cs_config.n_channels, length(config.cs_config.freq))`")   # :774-777
    c = mt_coherence(signal, MTCoherenceConfig(cs_config))
    _assign(output, c.coherence)
    return Coherence(output, c.freq)
