"""Oracle helpers.  Follows DSP.jl ``src/util.jl`` (test infrastructure only, see package docstring)."""
from __future__ import annotations

import numpy as np

FAST_FFT_SIZES = (2, 3, 5, 7)  # util.jl:107


def nextprod(factors, n: int) -> int:
    """Smallest product of powers of ``factors`` that is >= n (Julia Base.nextprod)."""
    n = int(n)
    if n <= 1:
        return 1
    best = None
    fs = sorted(set(int(f) for f in factors))

    def rec(idx, cur):
        nonlocal best
        if cur >= n:
            if best is None or cur < best:
                best = cur
            return
        if best is not None and cur >= best:
            return
        if idx == len(fs):
            return
        f = fs[idx]
        c = cur
        while True:
            rec(idx + 1, c)
            if c >= n:
                break
            c *= f

    rec(0, 1)
    return best


def nextfastfft(n: int) -> int:
    """util.jl:134  nextfastfft(n) = nextprod((2,3,5,7), n)."""
    return nextprod(FAST_FFT_SIZES, n)


# --- element-type rules, util.jl:92-104 -------------------------------------------------------
_FFTW_REAL = (np.dtype(np.float32), np.dtype(np.float64))
_FFTW_CPLX = (np.dtype(np.complex64), np.dtype(np.complex128))


def fftintype(dt) -> np.dtype:
    """util.jl:93-95."""
    dt = np.dtype(dt)
    if dt in _FFTW_REAL or dt in _FFTW_CPLX:
        return dt
    if dt.kind == "c":
        return np.dtype(np.complex128)
    return np.dtype(np.float64)


def fftouttype(dt) -> np.dtype:
    """util.jl:98-100."""
    dt = np.dtype(dt)
    if dt in _FFTW_CPLX:
        return dt
    if dt == np.float32:
        return np.dtype(np.complex64)
    return np.dtype(np.complex128)


def fftabs2type(dt) -> np.dtype:
    """util.jl:103-105."""
    dt = np.dtype(dt)
    if dt in (np.dtype(np.float32), np.dtype(np.complex64)):
        return np.dtype(np.float32)
    return np.dtype(np.float64)


def rfftfreq(n: int, fs=1.0) -> np.ndarray:
    """AbstractFFTs.rfftfreq(n, fs): (0:n>>1) * fs/n."""
    return np.arange(n // 2 + 1, dtype=np.float64) * (fs / n)


def fftfreq(n: int, fs=1.0) -> np.ndarray:
    """AbstractFFTs.fftfreq(n, fs): [0..(n-1)>>1, -(n>>1)..-1] * fs/n."""
    npos = (n - 1) // 2 + 1
    k = np.concatenate([np.arange(npos), np.arange(-(n // 2), 0)])
    return k.astype(np.float64) * (fs / n)


# --- polyphase inner products, util.jl:225-283 -------------------------------------------------
def unsafe_dot_mat(a: np.ndarray, col: int, b: np.ndarray, b_last: int):
    """util.jl:225-238.  ``col`` and ``b_last`` are 1-based; ``b_last`` is the LAST element of b used."""
    alen = a.shape[0]
    base = b_last - alen  # 0-based start index of the window
    return np.dot(a[:, col - 1], b[base:base + alen])


def unsafe_dot_mat_hist(a: np.ndarray, col: int, hist: np.ndarray, c: np.ndarray, c_last: int):
    """util.jl:240-255: window straddles ``hist`` (length alen-1) and the first ``c_last`` of ``c``."""
    alen = a.shape[0]
    if len(hist) != alen - 1:
        raise ValueError("length(b) must equal size(a, 1) - 1")
    if not c_last < alen:
        raise ValueError("cLastIdx must be < length(a)")
    w = np.concatenate([hist[c_last - 1:], c[:c_last]])
    return np.dot(a[:, col - 1], w)


def unsafe_dot_vec(a: np.ndarray, b: np.ndarray, b_last: int):
    """util.jl:257-270."""
    alen = len(a)
    base = b_last - alen
    return np.dot(a, b[base:base + alen])


def unsafe_dot_vec_hist(a: np.ndarray, hist: np.ndarray, c: np.ndarray, c_last: int):
    """util.jl:272-283."""
    w = np.concatenate([hist[c_last - 1:], c[:c_last]])
    return np.dot(a, w)


def shiftin(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """util.jl:299-314  shiftin!(a, b): shift b into the end of a (returns the new a)."""
    alen, blen = len(a), len(b)
    if blen >= alen:
        return np.array(b[blen - alen:], dtype=a.dtype, copy=True)
    out = np.empty_like(a)
    out[:alen - blen] = a[blen:]
    out[alen - blen:] = b
    return out


def hilbert(x):
    """util.jl:31-87: analytic signal along the first dimension (rfft, double the positive bins, zero the negative
    ones, unnormalised inverse, divide by N)."""
    x = np.asarray(x)
    T = fftintype(x.dtype)
    xs = x.astype(T)
    N = xs.shape[0]
    X = np.zeros(xs.shape, dtype=np.complex128)
    X[: (N >> 1) + 1] = np.fft.rfft(xs.astype(np.float64), axis=0)
    X[1: N // 2 + (N & 1)] *= 2
    return (np.fft.ifft(X, axis=0)).astype(fftouttype(T))
