"""Spectral estimation: host side of DSP.jl ``src/periodograms.jl`` over libmi355dsp.

``arraysplit`` / ``periodogram`` / ``WelchConfig`` / ``welch_pgram`` / ``spectrogram`` / ``stft`` keep the reference's
signatures, defaults, checks and normalisation; framing, windowing, the transforms and ``fft2pow!`` run on the GPU.
"""
from __future__ import annotations

import ctypes as C
import functools
import warnings
from dataclasses import dataclass

import numpy as np

from . import _dev, _lib, _plancache, util
from ._lib import ArgumentError, DimensionMismatch, DomainError


@dataclass
class Periodogram:
    """periodograms.jl:270-273."""
    power: object
    freq: np.ndarray


@dataclass
class Spectrogram:
    """periodograms.jl:773-777."""
    power: object
    freq: np.ndarray
    time: np.ndarray


def power(p):
    """periodograms.jl:310."""
    return p.power


def freq(p):
    """periodograms.jl:329."""
    return p.freq


def time(p):
    """periodograms.jl:793."""
    return p.time


def _is_twosided_freq(f: np.ndarray) -> bool:
    """``fftfreq`` layout (a ``Frequencies`` object with negative bins) as opposed to ``rfftfreq`` / already shifted."""
    f = np.asarray(f)
    return f.size > 1 and bool(np.any(np.diff(f) < 0))


def fftshift(p):
    """``fftshift(p::Periodogram)`` / ``fftshift(p::Spectrogram)`` (periodograms.jl:331-333, :778-780): centre the zero
    frequency of a two-sided estimate; one-sided (and already shifted) ones are returned as they are."""
    if not _is_twosided_freq(p.freq):
        return p
    xp = _dev.torch if _dev.is_device_array(p.power) else np
    power = xp.fft.fftshift(p.power, 0)        # frequency is the first axis of both layouts
    f = np.fft.fftshift(p.freq)
    return Spectrogram(power, f, p.time) if isinstance(p, Spectrogram) else Periodogram(power, f)


def compute_window(window, n: int):
    """periodograms.jl:248-257 -> (Float64 window or None, norm2)."""
    if window is None:
        return None, float(n)
    if callable(window):
        try:
            return _window_of(window, int(n))          # memoised: a 65536-point hanning costs a millisecond of host time per call
        except TypeError:                              # unhashable callable
            win = np.ascontiguousarray(window(n), dtype=np.float64)
            return win, float(np.sum(win * win))
    win = np.asarray(window.cpu() if hasattr(window, "cpu") else window)
    if len(win) != n:
        raise DimensionMismatch("length of window must match input")
    if win.dtype.kind == "c":
        raise _lib.UnsupportedError("complex windows are not accelerated")
    win = np.ascontiguousarray(win, dtype=np.float64)
    return win, float(np.sum(win * win))


@functools.lru_cache(maxsize=16)
def _window_of(window, n: int):
    win = np.ascontiguousarray(window(n), dtype=np.float64)
    win.setflags(write=False)                          # shared between calls
    return win, float(np.sum(win * win))


def _winptr(win):
    return None if win is None else win.ctypes.data_as(C.POINTER(C.c_double))


def _check_split(n, noverlap, nfft):
    if not (0 <= noverlap < n):
        raise DomainError(f"noverlap must be between zero and n (noverlap={noverlap}, n={n})")     # :44
    if not nfft >= n:
        raise DomainError(f"nfft must be >= n (nfft={nfft}, n={n})")                                # :45


def frame_count(length: int, n: int, noverlap: int) -> int:
    """periodograms.jl:49-50."""
    return int(_lib.lib().mdsp_frame_count(int(length), int(n), int(noverlap)))


def arraysplit(s, n: int, noverlap: int, nfft: int | None = None, window=None):
    """``arraysplit(s, n, noverlap, nfft=n, window=nothing)`` (periodograms.jl:134): all frames at once as a
    (k, nfft) array -- row i is what iterating the reference's ``ArraySplit`` yields at step i."""
    nfft = n if nfft is None else nfft
    _check_split(n, noverlap, nfft)
    win, _ = compute_window(window, n)
    S = util.fftintype(_dev.np_dtype_of(s))
    cols, _ = _dev.to_columns(s, S)
    if cols.shape[0] != 1:
        raise ArgumentError("arraysplit expects a vector")
    k = frame_count(cols.shape[1], n, noverlap)
    out = _dev.empty_columns(k, nfft, S)
    if k:
        _lib.check(_lib.lib().mdsp_frames(_dev.ptr(cols), cols.shape[1], _dev.md_dtype(S), n, noverlap, nfft, _winptr(win), 0, k,
                                          _dev.ptr(out), _dev.stream_ptr()))
    return out if _dev.is_device_array(s) else out.cpu().numpy()


class _StftPlan:
    def __init__(self, n, noverlap, nfft, win, r, onesided, psd_only, dtype, engine):
        self._h = C.c_void_p()
        _lib.check(_lib.lib().mdsp_stft_plan_create(C.byref(self._h), n, noverlap, nfft, _winptr(win), float(r), int(onesided), int(psd_only),
                                                    _dev.md_dtype(dtype), engine))
        no, eng = C.c_int64(), C.c_int()
        _lib.check(_lib.lib().mdsp_stft_plan_info(self._h, C.byref(no), C.byref(eng)))
        self.nout, self.engine = no.value, eng.value

    def __del__(self):
        try:
            if self._h:
                _lib.lib().mdsp_stft_plan_destroy(self._h)
        except Exception:
            pass


def stft(s, n: int | None = None, noverlap: int | None = None, psdonly: bool = False, *, onesided: bool | None = None,
         nfft: int | None = None, fs=1, window=None, engine: int = _lib.ENGINE_AUTO):
    """``stft(s, n, noverlap[, PSDOnly()]; onesided, nfft, fs, window)`` (periodograms.jl:872-897).

    Returns the (nout, k) matrix (column k = segment k).  A 2-d ``s`` of shape (len, channels) is processed as
    independent channels and returns (nout, k, channels).
    """
    sdt = _dev.np_dtype_of(s)
    cplx = sdt.kind == "c"
    length = int(s.shape[0])
    n = length >> 3 if n is None else int(n)
    noverlap = n >> 1 if noverlap is None else int(noverlap)
    onesided = (not cplx) if onesided is None else bool(onesided)
    nfft = util.nextfastfft(n) if nfft is None else int(nfft)
    if onesided and cplx:
        raise ArgumentError("cannot compute one-sided FFT of a complex signal")      # :876
    win, norm2 = compute_window(window, n)                                            # :878
    _check_split(n, noverlap, nfft)                                                   # ArraySplit checks, :44-45
    S = util.fftintype(sdt)
    T = util.fftabs2type(S) if psdonly else util.fftouttype(S)
    k = frame_count(length, n, noverlap)
    plan = _plancache.plans.get(("stft", _plancache.ctx_key(), n, noverlap, nfft, _plancache.window_key(window), float(fs), onesided, bool(psdonly), np.dtype(S).str, engine),
                                lambda: _StftPlan(n, noverlap, nfft, win, fs * norm2, onesided, psdonly, S, engine))
    hcols = _dev.host_columns(s, S)
    if hcols is not None and k > 0:    # large host array: mdsp_stft_exec_host (chunked H2D || kernel || D2H, the output never lives on the device whole)
        nch = hcols.shape[0]
        outh = np.empty((nch, k, plan.nout), dtype=T)
        _lib.check(_lib.lib().mdsp_stft_exec_host(plan._h, hcols.ctypes.data_as(C.c_void_p), length, nch, length, outh.ctypes.data_as(C.c_void_p),
                                                  plan.nout, k * plan.nout, 0))
        res = outh.transpose(2, 1, 0)
        return res[:, :, 0] if s.ndim == 1 else res
    cols, shape = _dev.to_columns(s, S)
    nch = cols.shape[0]
    out = _dev.torch.zeros((nch, k, plan.nout), dtype=_dev.torch_dtype(T), device=cols.device)     # zeros(...), :881
    if k and nch:
        _lib.check(_lib.lib().mdsp_stft_exec(plan._h, _dev.ptr(cols), length, nch, length, _dev.ptr(out), plan.nout, k * plan.nout,
                                             _dev.stream_ptr()))
    res = out.permute(2, 1, 0)            # (nout, k, nch): a view, column-major like the reference's matrix
    if len(shape) == 1:
        res = res[:, :, 0]
    return res if _dev.is_device_array(s) else res.cpu().numpy()


def spectrogram(s, n: int | None = None, noverlap: int | None = None, *, onesided: bool | None = None, nfft: int | None = None,
                fs=1, window=None, engine: int = _lib.ENGINE_AUTO) -> Spectrogram:
    """``spectrogram(s, n, noverlap; onesided, nfft, fs, window)`` (periodograms.jl:828-837)."""
    cplx = _dev.np_dtype_of(s).kind == "c"
    n = int(s.shape[0]) >> 3 if n is None else int(n)
    noverlap = n >> 1 if noverlap is None else int(noverlap)
    onesided = (not cplx) if onesided is None else bool(onesided)
    nfft = util.nextfastfft(n) if nfft is None else int(nfft)
    out = stft(s, n, noverlap, True, onesided=onesided, nfft=nfft, fs=fs, window=window, engine=engine)
    k = out.shape[1]
    t = (n / 2 + np.arange(k) * (n - noverlap)) / fs                                   # :835
    return Spectrogram(out, util.rfftfreq(nfft, fs) if onesided else util.fftfreq(nfft, fs), t)


def periodogram(s, *, onesided: bool | None = None, nfft: int | None = None, fs=1, window=None,
                engine: int = _lib.ENGINE_AUTO) -> Periodogram:
    """``periodogram(s; onesided, nfft, fs, window)`` (periodograms.jl:393-417): the single-segment PSD."""
    sdt = _dev.np_dtype_of(s)
    cplx = sdt.kind == "c"
    length = int(s.shape[0])
    onesided = (not cplx) if onesided is None else bool(onesided)
    nfft = util.nextfastfft(length) if nfft is None else int(nfft)
    if onesided and cplx:
        raise ArgumentError("cannot compute one-sided FFT of a complex signal")       # :396
    if not nfft >= length:
        raise DomainError(f"nfft must be >= n = length(s) (nfft={nfft}, n={length})")  # :397
    out = stft(s, length, 0, True, onesided=onesided, nfft=nfft, fs=fs, window=window, engine=engine)
    return Periodogram(out[:, 0], util.rfftfreq(nfft, fs) if onesided else util.fftfreq(nfft, fs))


_MISSING = object()


class WelchConfig:
    """``WelchConfig(nsamples, eltype; n, noverlap, onesided, nfft, fs, window)`` / ``WelchConfig(data; ...)``
    (periodograms.jl:516-587).  Owns the device plan (window, tables, rocFFT plan / work buffers): re-using a
    config re-uses them, and gives bit-identical results call after call (test/periodograms.jl:222-224)."""

    def __init__(self, nsamples, eltype=None, *, n: int | None = None, noverlap: int | None = None, onesided: bool | None = None,
                 nfft: int | None = None, fs=1, window=_MISSING, engine: int = _lib.ENGINE_AUTO):
        if eltype is None:                       # WelchConfig(data; kw...)  (:578-580)
            data = nsamples
            nsamples, eltype = int(data.shape[-1] if len(data.shape) else 0), _dev.np_dtype_of(data)
        T = np.dtype(eltype)
        cplx = T.kind == "c"
        n = int(nsamples) >> 3 if n is None else int(n)
        noverlap = n >> 1 if noverlap is None else int(noverlap)
        onesided = (not cplx) if onesided is None else bool(onesided)
        nfft = util.nextfastfft(n) if nfft is None else int(nfft)
        if window is _MISSING:                   # welch_config_default_window (:582-587)
            warnings.warn("Omitting `window` is deprecated; specify `window=None` for the old behaviour or "
                          "`window=hanning` for the future default.", DeprecationWarning, stacklevel=2)
            window = None
        if onesided and cplx:
            raise ArgumentError("cannot compute one-sided FFT of a complex signal")   # :564
        if not nfft >= n:
            raise DomainError(f"nfft must be >= n (nfft={nfft}, n={n})")               # :565
        win, norm2 = compute_window(window, n)                                          # :567
        self.nsamples, self.noverlap, self.onesided, self.nfft, self.fs = n, noverlap, onesided, nfft, fs
        self.window = win
        self.r = fs * norm2                                                             # :568
        self.intype = util.fftintype(T)                                                 # eltype(inbuf) = float(T)
        self.freq = util.rfftfreq(nfft, fs) if onesided else util.fftfreq(nfft, fs)     # :573
        self._h = C.c_void_p()
        if not (0 <= noverlap < n):
            raise DomainError(f"noverlap must be between zero and n (noverlap={noverlap}, n={n})")   # raised by arraysplit in the reference
        _lib.check(_lib.lib().mdsp_welch_plan_create(C.byref(self._h), n, noverlap, nfft, _winptr(win), float(self.r), int(onesided),
                                                     _dev.md_dtype(self.intype), engine))
        no, eng = C.c_int64(), C.c_int()
        _lib.check(_lib.lib().mdsp_welch_plan_info(self._h, C.byref(no), C.byref(eng)))
        self.nout, self.engine = no.value, eng.value

    # -- streaming form: welch_pgram of a stream handed over slice by slice (mdsp_welch_reset / _accumulate / _finalize) --------------
    def reset(self):
        _lib.check(_lib.lib().mdsp_welch_reset(self._h))
        self._acc_dev = None
        return self

    def accumulate(self, cols):
        """Add the frames of ``cols`` ((nch, len) device columns of ``intype``: whole frames; consecutive slices of one stream overlap by
        n - hop samples) to the plan's Float64 sums."""
        nch, length = cols.shape
        self._acc_dev = cols.device
        _lib.check(_lib.lib().mdsp_welch_accumulate(self._h, _dev.ptr(cols), length, nch, length, _dev.stream_ptr()))
        return self

    def frames_accumulated(self) -> int:
        k = C.c_int64()
        _lib.check(_lib.lib().mdsp_welch_frames_accumulated(self._h, C.byref(k)))
        return k.value

    def accumulator(self):
        """The plan's Float64 accumulator as a device tensor VIEW (what a collective sums over ranks)."""
        p, cnt = C.c_void_p(), C.c_int64()
        _lib.check(_lib.lib().mdsp_welch_accumulator(self._h, C.byref(p), C.byref(cnt)))
        return _dev.tensor_view(p.value, cnt.value, np.float64, self._acc_dev)

    def finalize(self, frames_total: int = 0, nch: int = 1):
        """(nch, nout) PSDs from the accumulated sums; ``frames_total`` = K of the whole stream (0: the frames accumulated here)."""
        T = util.fftabs2type(self.intype)
        out = _dev.empty_columns(nch, self.nout, T)
        _lib.check(_lib.lib().mdsp_welch_finalize(self._h, int(frames_total), _dev.ptr(out), self.nout, _dev.stream_ptr()))
        return out

    def exec_host(self, s: np.ndarray):
        """``welch_pgram`` of HOST columns ((nch, len) C-contiguous numpy of ``intype``) through the pinned, chunked H2D || kernel
        pipeline of ``mdsp_welch_exec_host``; returns a (nch, nout) numpy array."""
        nch, length = s.shape
        out = np.empty((nch, self.nout), dtype=util.fftabs2type(self.intype))
        _lib.check(_lib.lib().mdsp_welch_exec_host(self._h, s.ctypes.data_as(C.c_void_p), length, nch, length, out.ctypes.data_as(C.c_void_p),
                                                   self.nout, 0))
        return out

    def __del__(self):
        try:
            if self._h:
                _lib.lib().mdsp_welch_plan_destroy(self._h)
        except Exception:
            pass


def _welch_exec(cols, config: WelchConfig):
    nch, length = cols.shape
    T = util.fftabs2type(config.intype)
    out = _dev.empty_columns(nch, config.nout, T)
    _lib.check(_lib.lib().mdsp_welch_exec(config._h, _dev.ptr(cols), length, nch, length, _dev.ptr(out), config.nout, _dev.stream_ptr()))
    return out


def welch_pgram(s, n=None, noverlap=None, *, config: WelchConfig | None = None, **kw) -> Periodogram:
    """``welch_pgram(s, n, noverlap; kw...)`` / ``welch_pgram(s, config)`` (periodograms.jl:647-649, :702-705).

    ``s`` may be (len,) or (len, channels); channels are independent and give a (nout, channels) PSD.
    """
    if isinstance(n, WelchConfig):
        config, n = n, None
    sdt = _dev.np_dtype_of(s)
    if config is None:
        length = int(s.shape[0])
        n = length >> 3 if n is None else int(n)
        noverlap = n >> 1 if noverlap is None else int(noverlap)
        if "window" in kw:      # without it the constructor warns (deprecated default) on every call, like the reference
            key = ("welch", _plancache.ctx_key(), np.dtype(sdt).str, n, noverlap, kw.get("nfft"), kw.get("onesided"), float(kw.get("fs", 1)),
                   _plancache.window_key(kw["window"]), kw.get("engine", _lib.ENGINE_AUTO))
            config = _plancache.plans.get(key, lambda: WelchConfig(length, sdt, n=n, noverlap=noverlap, **kw))
        else:
            config = WelchConfig(length, sdt, n=n, noverlap=noverlap, **kw)
    if util.fftintype(sdt) != config.intype:
        raise ArgumentError(f"float(eltype(s)) = {util.fftintype(sdt)} doesn't match the eltype of the input buffer: {config.intype}.")
    hcols = _dev.host_columns(s, config.intype)
    if hcols is not None:              # large host array: mdsp_welch_exec_host (pinned, chunked H2D || kernel), no torch copy
        out = config.exec_host(hcols)
        return Periodogram(out[0] if s.ndim == 1 else out.T, config.freq)
    cols, shape = _dev.to_columns(s, config.intype)
    out = _welch_exec(cols, config)
    res = out.t() if len(shape) > 1 else out[0]
    return Periodogram(res if _dev.is_device_array(s) else res.cpu().numpy(), config.freq)


def welch_pgram_(out, s, n=None, noverlap=None, *, config: WelchConfig | None = None, **kw) -> Periodogram:
    """``welch_pgram!(out, s, ...)`` (periodograms.jl:683-686, :734-744) with the reference's checks."""
    if isinstance(n, WelchConfig):
        config, n = n, None
    sdt = _dev.np_dtype_of(s)
    if config is None:
        length = int(s.shape[0])
        n = length >> 3 if n is None else int(n)
        noverlap = n >> 1 if noverlap is None else int(noverlap)
        config = WelchConfig(length, sdt, n=n, noverlap=noverlap, **kw)
    if out.shape[0] != len(config.freq):
        raise DimensionMismatch(f"Expected `output` to be of length `length(config.freq)`; got `length(output)` = {out.shape[0]} "
                                f"and `length(config.freq)` = {len(config.freq)}")
    odt = _dev.np_dtype_of(out)
    if odt != util.fftabs2type(sdt):
        raise ArgumentError(f"Eltype of output ({odt}) doesn't match the expected type: {util.fftabs2type(sdt)}.")
    if util.fftintype(sdt) != config.intype:
        raise ArgumentError(f"float(eltype(s)) = {sdt} doesn't match the eltype of the input buffer: {config.intype}.")
    res = welch_pgram(s, config=config).power
    if isinstance(out, np.ndarray):
        out[...] = res if isinstance(res, np.ndarray) else res.cpu().numpy()
    else:
        out.copy_(res if not isinstance(res, np.ndarray) else _dev.torch.from_numpy(res))
    return Periodogram(out, config.freq)
