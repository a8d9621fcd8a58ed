"""Oracle for ``src/periodograms.jl``: framing, periodogram, Welch, spectrogram, STFT.

Test infrastructure only (see package docstring).  ``dtype=np.float64`` on the entry points evaluates
the same algorithm in double precision regardless of the input element type (used as the Float32
parity reference).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import scipy.fft as sfft

from . import util, windows


# ---------------------------------------------------------------------------------------------
# framing, periodograms.jl:32-137
# ---------------------------------------------------------------------------------------------
def frame_count(length: int, n: int, noverlap: int) -> int:
    """periodograms.jl:49-50 (the trailing partial frame is dropped)."""
    return (length - n) // (n - noverlap) + 1 if length >= n else 0


def check_split_args(n: int, noverlap: int, nfft: int):
    """periodograms.jl:44-45."""
    if not (0 <= noverlap < n):
        raise ValueError("DomainError: noverlap must be between zero and n")
    if not nfft >= n:
        raise ValueError("DomainError: nfft must be >= n")


def arraysplit(s, n: int, noverlap: int, nfft: int | None = None, window=None, dtype=None) -> np.ndarray:
    """All frames of ``ArraySplit`` as a (k, nfft) array (periodograms.jl:57-69).

    Frame i (1-based) = s[(i-1)(n-noverlap)+1 ... +n] (.* window), zero tail up to nfft.  The product
    with the (Float64) window is formed in the promoted type and then rounded to the buffer eltype
    ``fftintype(eltype(s))``, as the reference's ``x.buf[i] = x.s[offset+i] * window[i]`` does.
    """
    s = np.asarray(s)
    if nfft is None:
        nfft = n
    check_split_args(n, noverlap, nfft)
    S = util.fftintype(s.dtype) if dtype is None else np.dtype(dtype)
    k = frame_count(len(s), n, noverlap)
    hop = n - noverlap
    out = np.zeros((k, nfft), dtype=S)
    if k == 0:
        return out
    idx = (np.arange(k) * hop)[:, None] + np.arange(n)[None, :]
    fr = s[idx]
    if window is not None:
        w = np.asarray(window)
        if len(w) != n:
            raise ValueError("DimensionMismatch: length of window must match input")
        fr = fr * w[None, :]       # numpy promotes f32*f64 -> f64, c64*f64 -> c128 like Julia
    out[:, :n] = fr.astype(S)
    return out


# ---------------------------------------------------------------------------------------------
# fft2pow! / fft2oneortwosided!, periodograms.jl:142-172, :234-244
# ---------------------------------------------------------------------------------------------
def fft2pow_weights(nfft: int, nspec: int, r: float, onesided: bool, T) -> np.ndarray:
    """Per-bin multiplier of ``fft2pow!`` for an input spectrum of length ``nspec`` (one column)."""
    T = np.dtype(T)
    m1 = T.type(1 / r)
    m2 = T.type(2 / r)
    if onesided:
        w = np.full(nspec, m2, dtype=T)
        w[0] = m1
        w[-1] = m1 if nfft % 2 == 0 else m2
        return w
    return np.full(nspec, m1, dtype=T)


def fft2pow(spec: np.ndarray, nfft: int, r: float, onesided: bool, T) -> np.ndarray:
    """One application of ``fft2pow!`` onto a zero ``out`` (periodograms.jl:142-172).  ``spec`` is
    (..., nspec); returns (..., nout) with nout = nspec (one-sided, or complex two-sided) or nfft
    (real input converted to two-sided, :158-168)."""
    T = np.dtype(T)
    nspec = spec.shape[-1]
    p = (spec.real.astype(T) ** 2 + spec.imag.astype(T) ** 2)
    if onesided or nspec == nfft:
        return p * fft2pow_weights(nfft, nspec, r, onesided, T)
    m1 = T.type(1 / r)
    out = np.zeros(spec.shape[:-1] + (nfft,), dtype=T)
    out[..., :nspec] = p * m1
    # out[nfft-i+2] (1-based) for i = 2..n-1, plus the odd-nfft duplicate of bin n
    hi = nspec - 1 if nfft % 2 == 0 else nspec
    for i in range(2, hi + 1):
        out[..., nfft - i + 1] = p[..., i - 1] * m1
    return out


def fft2oneortwosided(spec: np.ndarray, nfft: int, onesided: bool) -> np.ndarray:
    """periodograms.jl:234-244."""
    nspec = spec.shape[-1]
    if onesided or nspec == nfft:
        return spec.copy()
    out = np.zeros(spec.shape[:-1] + (nfft,), dtype=spec.dtype)
    out[..., :nspec] = spec
    for i in range(2, nspec - (1 if nfft % 2 == 0 else 0) + 1):
        out[..., nfft - i + 1] = np.conj(spec[..., i - 1])
    return out


def compute_window(window, n: int):
    """periodograms.jl:248-257 -> (win or None, norm2)."""
    if window is None:
        return None, float(n)
    if callable(window):
        win = np.asarray(window(n), dtype=np.float64)
        return win, float(np.sum(win * win))
    win = np.asarray(window)
    if len(win) != n:
        raise ValueError("DimensionMismatch: length of window must match input")
    return win, float(np.sum(np.abs(win) ** 2))


def _forward(frames: np.ndarray) -> np.ndarray:
    """``forward_plan`` (periodograms.jl:511-514): rfft for real buffers, fft for complex."""
    if frames.dtype.kind == "c":
        return sfft.fft(frames, axis=-1)
    return sfft.rfft(frames, axis=-1)


# ---------------------------------------------------------------------------------------------
# periodogram, periodograms.jl:393-417
# ---------------------------------------------------------------------------------------------
@dataclass
class Periodogram:
    power: np.ndarray
    freq: np.ndarray


@dataclass
class Spectrogram:
    power: np.ndarray
    freq: np.ndarray
    time: np.ndarray


def periodogram(s, onesided=None, nfft=None, fs=1.0, window=None, dtype=None) -> Periodogram:
    s = np.asarray(s)
    cplx = s.dtype.kind == "c"
    if onesided is None:
        onesided = not cplx
    if nfft is None:
        nfft = util.nextfastfft(len(s))
    if onesided and cplx:
        raise ValueError("ArgumentError: cannot compute one-sided FFT of a complex signal")
    if not nfft >= len(s):
        raise ValueError("DomainError: nfft must be >= n = length(s)")
    win, norm2 = compute_window(window, len(s))
    S = util.fftintype(s.dtype) if dtype is None else np.result_type(dtype, s.dtype if cplx else dtype)
    inp = np.zeros(nfft, dtype=S)
    inp[:len(s)] = (s * win if win is not None else s).astype(S)
    spec = _forward(inp)
    T = util.fftabs2type(S)
    return Periodogram(fft2pow(spec, nfft, fs * norm2, onesided, T),
                       util.rfftfreq(nfft, fs) if onesided else util.fftfreq(nfft, fs))


# ---------------------------------------------------------------------------------------------
# Welch, periodograms.jl:516-759
# ---------------------------------------------------------------------------------------------
@dataclass
class WelchConfig:
    nsamples: int
    noverlap: int
    onesided: bool
    nfft: int
    fs: float
    freq: np.ndarray
    window: np.ndarray | None
    r: float
    intype: np.dtype

    @staticmethod
    def create(nsamples: int, T, n=None, noverlap=None, onesided=None, nfft=None, fs=1.0, window=None):
        """periodograms.jl:560-576."""
        T = np.dtype(T)
        cplx = T.kind == "c"
        if n is None:
            n = nsamples >> 3
        if noverlap is None:
            noverlap = n >> 1
        if onesided is None:
            onesided = not cplx
        if nfft is None:
            nfft = util.nextfastfft(n)
        if onesided and cplx:
            raise ValueError("ArgumentError: cannot compute one-sided FFT of a complex signal")
        if not nfft >= n:
            raise ValueError("DomainError: nfft must be >= n")
        win, norm2 = compute_window(window, n)
        r = fs * norm2
        freq = util.rfftfreq(nfft, fs) if onesided else util.fftfreq(nfft, fs)
        intype = np.result_type(T, np.float32) if T.kind in "fc" else np.dtype(np.float64)  # float(T)
        return WelchConfig(n, noverlap, onesided, nfft, fs, freq, win, r, intype)


def welch_pgram(s, n=None, noverlap=None, config: WelchConfig | None = None, dtype=None,
                sequential: bool = False, **kw) -> Periodogram:
    """periodograms.jl:647-649, :702-705, :746-759.

    ``sequential=True`` reproduces the reference's frame-by-frame accumulation in the output eltype
    (``muladd(abs2(X), m, out)`` per frame); the default sums frame powers in Float64 before the
    final rounding, which is the form GPU results are compared against.
    """
    s = np.asarray(s)
    if config is None:
        if n is None:
            n = len(s) >> 3
        if noverlap is None:
            noverlap = n >> 1
        config = WelchConfig.create(len(s), s.dtype, n=n, noverlap=noverlap, **kw)
    frames = arraysplit(s, config.nsamples, config.noverlap, config.nfft, config.window,
                        dtype=(None if dtype is None else np.result_type(dtype, s.dtype) if s.dtype.kind == "c" else dtype))
    k = frames.shape[0]
    T = util.fftabs2type(frames.dtype)
    nout = config.nfft // 2 + 1 if config.onesided else config.nfft
    out = np.zeros(nout, dtype=T)
    r = k * config.r                                               # :751
    if k:
        if sequential:
            for i in range(k):
                out = out + fft2pow(_forward(frames[i]), config.nfft, r, config.onesided, T)
        else:
            CH = 4096
            acc = np.zeros(nout, dtype=np.float64)
            for c0 in range(0, k, CH):
                spec = _forward(frames[c0:c0 + CH])
                p = fft2pow(spec, config.nfft, 1.0, config.onesided, np.float64)
                acc += p.sum(axis=0)
            out = (acc / r).astype(T)
    return Periodogram(out, config.freq)


# ---------------------------------------------------------------------------------------------
# STFT / spectrogram, periodograms.jl:828-897
# ---------------------------------------------------------------------------------------------
def stft(s, n=None, noverlap=None, psdonly: bool = False, onesided=None, nfft=None, fs=1.0, window=None,
         dtype=None) -> np.ndarray:
    """periodograms.jl:872-897.  Returns the (nout, k) matrix (column k = frame k), like the reference."""
    s = np.asarray(s)
    cplx = s.dtype.kind == "c"
    if n is None:
        n = len(s) >> 3
    if noverlap is None:
        noverlap = n >> 1
    if onesided is None:
        onesided = not cplx
    if nfft is None:
        nfft = util.nextfastfft(n)
    if onesided and cplx:
        raise ValueError("ArgumentError: cannot compute one-sided FFT of a complex signal")
    win, norm2 = compute_window(window, n)
    fdt = None if dtype is None else (np.result_type(dtype, s.dtype) if cplx else np.dtype(dtype))
    frames = arraysplit(s, n, noverlap, nfft, win, dtype=fdt)
    spec = _forward(frames)
    r = fs * norm2
    if psdonly:
        out = fft2pow(spec, nfft, r, onesided, util.fftabs2type(frames.dtype))
    else:
        out = fft2oneortwosided(spec, nfft, onesided)
    return np.ascontiguousarray(out.T)


def spectrogram(s, n=None, noverlap=None, onesided=None, nfft=None, fs=1.0, window=None, dtype=None) -> Spectrogram:
    """periodograms.jl:828-837."""
    s = np.asarray(s)
    cplx = s.dtype.kind == "c"
    if n is None:
        n = len(s) >> 3
    if noverlap is None:
        noverlap = n >> 1
    if onesided is None:
        onesided = not cplx
    if nfft is None:
        nfft = util.nextfastfft(n)
    out = stft(s, n, noverlap, True, onesided=onesided, nfft=nfft, fs=fs, window=window, dtype=dtype)
    k = out.shape[1]
    t = (n / 2 + np.arange(k) * (n - noverlap)) / fs                # :835
    return Spectrogram(out, util.rfftfreq(nfft, fs) if onesided else util.fftfreq(nfft, fs), t)


__all__ = ["arraysplit", "frame_count", "fft2pow", "fft2oneortwosided", "compute_window", "periodogram",
           "WelchConfig", "welch_pgram", "stft", "spectrogram", "Periodogram", "Spectrogram", "windows"]
