"""Oracle for ``src/multitaper.jl``: multitaper periodogram / spectrogram / cross power spectra / coherence.

Test infrastructure only (see package docstring).  Each function cites the reference lines it restates.
"""
from __future__ import annotations

import math

import numpy as np

from . import windows
from .periodograms import fft2pow
from .util import fftabs2type, fftfreq, fftintype, fftouttype, nextfastfft, rfftfreq


def nextpow2(n: int) -> int:
    return 1 << max(0, (int(n) - 1).bit_length())


class MTConfig:
    """multitaper.jl:5-49 (struct + checks) and :112-135 (keyword constructor)."""

    def __init__(self, T, n_samples: int, fs=1, nfft: int | None = None, window=None, nw=4, ntapers: int | None = None,
                 taper_weights=None, onesided: bool | None = None):
        T = np.dtype(T)
        if onesided is None:
            onesided = T.kind != "c"
        if onesided and T.kind == "c":
            raise ValueError("ArgumentError: cannot compute one-sided FFT of a complex signal")        # :115-117
        if nfft is None:
            nfft = nextpow2(n_samples)
        if ntapers is None:
            ntapers = int(2 * nw - 1)
        if not n_samples > 0:
            raise ValueError("ArgumentError: `n_samples` must be positive")
        if not nfft >= n_samples:
            raise ValueError("ArgumentError: Must have `nfft >= n_samples`")
        if taper_weights is None:
            taper_weights = np.full(ntapers, 1 / ntapers)
        taper_weights = np.asarray(taper_weights, dtype=np.float64)
        self.freq = rfftfreq(nfft, fs) if onesided else fftfreq(nfft, fs)
        if window is None:
            self.r = fs / taper_weights                                                                 # :127
            window = windows.dpss(n_samples, nw, ntapers)
        else:
            window = np.asarray(window, dtype=np.float64)
            if window.ndim != 2 or window.shape[1] != len(taper_weights):
                raise IndexError("DimensionMismatch: arrays could not be broadcast to a common size")
            self.r = fs * np.sum(np.abs(window) ** 2, axis=0) / taper_weights                         # :130
        if not ntapers > 0:
            raise ValueError("ArgumentError: `ntapers` must be positive")
        if not fs > 0:
            raise ValueError("ArgumentError: `fs` must be positive")
        if window.shape != (n_samples, ntapers):
            raise IndexError("DimensionMismatch: Must have `size(window) == (n_samples, ntapers)`")   # :34-36
        if self.r.shape != (ntapers,):
            raise IndexError("DimensionMismatch: Must have `size(r) == (ntapers,)`")
        self.T, self.n_samples, self.fs, self.nfft, self.ntapers = T, n_samples, fs, nfft, ntapers
        self.window, self.onesided = window, onesided

    def fft_tapered(self, signal, taper_index: int) -> np.ndarray:
        """mt_fft_tapered! :143-153 (taper in Float64, stored into the zero-padded buffer of eltype T, then r/fft)."""
        buf = np.zeros(self.nfft, dtype=self.T)
        buf[:self.n_samples] = (self.window[:, taper_index] * np.asarray(signal)).astype(self.T)
        wide = np.complex128 if self.T.kind == "c" else np.float64
        X = np.fft.rfft(buf.astype(wide)) if self.onesided else np.fft.fft(buf.astype(wide))
        return X.astype(fftouttype(self.T))


def dpss_config(T, n_samples, nw=4, ntapers=None, fs=1, keep_only_large_evals=False, weight_by_evals=False, **kw) -> MTConfig:
    """multitaper.jl:52-78."""
    if ntapers is None:
        ntapers = int(2 * nw - 1)
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


def mt_pgram(signal, config: MTConfig | None = None, **kw) -> tuple[np.ndarray, np.ndarray]:
    """mt_pgram / mt_pgram! multitaper.jl:177-245 -> (power, freq)."""
    signal = np.asarray(signal)
    if config is None:
        T = fftintype(signal.dtype) if signal.dtype.kind != "c" else signal.dtype
        kw.setdefault("nfft", nextfastfft(len(signal)))                                                # :178
        if "ntapers" not in kw:
            kw["ntapers"] = math.ceil(2 * kw.get("nw", 4)) - 1
        config = MTConfig(signal.dtype if signal.dtype.kind in "fc" else T, len(signal), **kw)
    if len(signal) != config.n_samples:
        raise IndexError("DimensionMismatch: Expected `signal` to be of length `config.n_samples`")
    out = np.zeros(len(config.freq), dtype=fftabs2type(config.T))
    for k in range(config.ntapers):
        out += fft2pow(config.fft_tapered(signal, k), config.nfft, config.r[k], config.onesided, out.dtype)   # :241-242
    return out, config.freq


class MTSpectrogramConfig:
    """multitaper.jl:248-287."""

    def __init__(self, n_samples: int, mt_config: MTConfig, n_overlap_samples: int):
        spw = mt_config.n_samples
        if spw <= n_overlap_samples:
            raise ValueError("ArgumentError: Need `samples_per_window > n_overlap_samples`")
        hop = spw - n_overlap_samples
        length = 0 if n_samples < spw else (n_samples - spw) // hop + 1
        self.time = (spw / 2 + hop * np.arange(length)) / mt_config.fs                                  # :270
        self.n_samples, self.n_overlap_samples, self.mt_config = n_samples, n_overlap_samples, mt_config


def mt_spectrogram(signal, n: int | None = None, n_overlap: int | None = None, config: MTSpectrogramConfig | None = None, **kw):
    """mt_spectrogram multitaper.jl:305-392 -> (power (nfreq, ntime), freq, time)."""
    signal = np.asarray(signal)
    if config is None:
        if n is None:
            n = len(signal) >> 3
        if n_overlap is None:
            n_overlap = n >> 1
        T = signal.dtype if signal.dtype.kind in "fc" else fftintype(signal.dtype)
        config = MTSpectrogramConfig(len(signal), MTConfig(T, n, **kw), n_overlap)
    if len(signal) != config.n_samples:
        raise IndexError("DimensionMismatch: Expected `signal` to be of length `config.n_samples`")
    mc = config.mt_config
    hop = mc.n_samples - config.n_overlap_samples
    out = np.zeros((len(mc.freq), len(config.time)), dtype=fftabs2type(mc.T))
    for ti in range(len(config.time)):
        out[:, ti], _ = mt_pgram(signal[ti * hop: ti * hop + mc.n_samples], mc)                         # :325-327
    return out, mc.freq, config.time


class MTCrossSpectraConfig:
    """multitaper.jl:424-516."""

    def __init__(self, n_channels: int, mt_config: MTConfig, demean: bool = False, freq_range=None):
        if mt_config.T.kind == "c" or not mt_config.onesided:
            raise ValueError("ArgumentError: Only real data is supported (with the default choice of `onesided=true`)")  # :417-422
        self.n_channels, self.mt_config, self.demean, self.freq_range = n_channels, mt_config, demean, freq_range
        self.normalization_weights = 2 / mt_config.r                                                     # :499
        if freq_range is not None:
            mask = (freq_range[0] < mt_config.freq) & (mt_config.freq < freq_range[-1])                 # :503
            self.freq_inds = np.flatnonzero(mask)
            self.freq = mt_config.freq[mask]
        else:
            self.freq_inds = np.arange(len(mt_config.freq))
            self.freq = mt_config.freq


def mt_cross_power_spectra(signal, config: MTCrossSpectraConfig | None = None, fs=1, **kw):
    """mt_cross_power_spectra! multitaper.jl:551-585 + cs_inner! :602-616 -> (power (nch, nch, nfreq), freq).
    ``signal`` is (n_channels, n_samples)."""
    signal = np.asarray(signal)
    if config is None:
        if signal.dtype.kind == "c":
            raise ValueError("ArgumentError: Only real data is supported")
        T = signal.dtype if signal.dtype.kind == "f" else np.dtype(np.float64)
        demean = kw.pop("demean", False)
        freq_range = kw.pop("freq_range", None)
        config = MTCrossSpectraConfig(signal.shape[0], MTConfig(T, signal.shape[1], fs=fs, **kw), demean=demean, freq_range=freq_range)
    mc = config.mt_config
    if signal.shape != (config.n_channels, mc.n_samples):
        raise IndexError("DimensionMismatch: Size of `signal` does not match `(config.n_channels, config.mt_config.n_samples)`")
    sig = signal.astype(mc.T)
    if config.demean:
        sig = sig - sig.mean(axis=1, keepdims=True)                                                      # :566-570
    x_mt = np.empty((len(mc.freq), mc.ntapers, config.n_channels), dtype=fftouttype(mc.T))
    for k in range(config.n_channels):
        for t in range(mc.ntapers):
            x_mt[:, t, k] = mc.fft_tapered(sig[k], t)
    x_mt[0] /= math.sqrt(2)                                                                              # :577
    if mc.nfft % 2 == 0:
        x_mt[-1] /= math.sqrt(2)
    xs = x_mt[config.freq_inds]                                                                          # (nf, ntapers, nch)
    w = config.normalization_weights.astype(x_mt.real.dtype)
    out = np.einsum("k,fkl,fkm->lmf", w, xs, np.conj(xs)).astype(fftouttype(mc.T))                       # :610-614
    return out, config.freq


def coherence_from_cs(cs: np.ndarray) -> np.ndarray:
    """coherence_from_cs! multitaper.jl:704-723."""
    nch, _, nf = cs.shape
    out = np.zeros((nch, nch, nf), dtype=cs.real.dtype)
    for c2 in range(nch):
        for c1 in range(c2 + 1, nch):
            out[c1, c2] = np.abs(cs[c1, c2]) / np.sqrt((cs[c1, c1] * cs[c2, c2]).real)
    out = out + out.transpose(1, 0, 2)
    for i in range(nch):
        out[i, i] = 1
    return out


def mt_coherence(signal, config: MTCrossSpectraConfig | None = None, **kw):
    """mt_coherence! multitaper.jl:765-783 -> (coherence (nch, nch, nfreq), freq)."""
    cs, f = mt_cross_power_spectra(signal, config, **kw)
    return coherence_from_cs(cs), f
