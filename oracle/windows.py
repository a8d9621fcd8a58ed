"""Oracle window generators.  Follows DSP.jl ``src/windows.jl`` (test infrastructure only).

Windows are always Float64 vectors in the reference (``windows.jl:104`` ``zeros(n+padding)``).
"""
from __future__ import annotations

import numpy as np


def _cospi(y: np.ndarray) -> np.ndarray:
    """cos(pi*y) with exact argument reduction (Julia ``cospi``), y any real array."""
    y = np.abs(np.asarray(y, dtype=np.float64))
    y = np.mod(y, 2.0)                     # exact
    y = np.where(y > 1.0, 2.0 - y, y)      # cos(pi*(2-y)) = cos(pi*y); now y in [0,1]
    neg = y > 0.5
    y = np.where(neg, 1.0 - y, y)          # cos(pi*(1-y)) = -cos(pi*y); now y in [0,0.5]
    use_sin = y > 0.25
    r = np.where(use_sin, 0.5 - y, y)      # cos(pi*y) = sin(pi*(0.5-y))
    v = np.where(use_sin, np.sin(np.pi * r), np.cos(np.pi * r))
    return np.where(neg, -v, v)


def _sample_points(n: int) -> np.ndarray:
    """``range(-0.5, 0.5; length=n)`` (windows.jl:117): correctly rounded -0.5 + i/(n-1)."""
    i = np.arange(n, dtype=np.float64)
    return (2.0 * i - (n - 1)) / (2.0 * (n - 1))


def makewindow(winfunc, n: int, padding: int = 0, zerophase: bool = False) -> np.ndarray:
    """windows.jl:97-121."""
    if n < 0:
        raise ValueError("`n` must be nonnegative")
    if padding < 0:
        raise ValueError("`padding` must be nonnegative")
    win = np.zeros(n + padding, dtype=np.float64)
    if n == 1:
        win[0] = winfunc(np.array([0.0]))[0]
    elif zerophase:
        h = n // 2
        win[:h + 1] = winfunc(np.arange(h + 1, dtype=np.float64) / n)
        if h > 0:
            win[len(win) - h:] = winfunc(-(np.arange(h, 0, -1, dtype=np.float64)) / n)
    elif n > 1:
        win[:n] = winfunc(_sample_points(n))
    return win


def rect(n, padding=0, zerophase=False):
    """windows.jl:142-144."""
    return makewindow(lambda x: np.ones_like(x), n, padding, zerophase)


def hanning(n, padding=0, zerophase=False):
    """windows.jl:181-183: 0.5*(1+cospi(2x))."""
    return makewindow(lambda x: 0.5 * (1.0 + _cospi(2.0 * x)), n, padding, zerophase)


hann = hanning


def hamming(n, padding=0, zerophase=False):
    """windows.jl:206-208: muladd(0.46, cospi(2x), 0.54)."""
    return makewindow(lambda x: 0.46 * _cospi(2.0 * x) + 0.54, n, padding, zerophase)


def bartlett(n, padding=0, zerophase=False):
    """windows.jl:380-382: 1 - abs(2x)."""
    return makewindow(lambda x: 1.0 - np.abs(2.0 * x), n, padding, zerophase)


def kaiser(n, alpha, padding=0, zerophase=False):
    """windows.jl:600-605: besseli0(pi*alpha*sqrt(1-(2x)^2)) / besseli0(pi*alpha)."""
    pf = 1.0 / np.i0(np.pi * alpha)
    return makewindow(lambda x: pf * np.i0(np.pi * alpha * np.sqrt(np.maximum(0.0, 1.0 - (2.0 * x) ** 2))),
                      n, padding, zerophase)


def dpss(n: int, nw: float, ntapers: int | None = None, padding: int = 0, zerophase: bool = False) -> np.ndarray:
    """windows.jl:668-726: the first ``ntapers`` Slepian tapers as an (n, ntapers) matrix -- eigenvectors of the
    symmetric tridiagonal matrix of Gruenbacher & Hummels (largest eigenvalues first); sign convention: the first
    non-zero element of the skew-symmetric (even-numbered, 1-based) tapers is positive.  The symmetric tapers keep
    LAPACK's sign in the reference; here they are normalised to a positive sum, which is what LAPACK returns for the
    reference's golden vector (``dpss128,4.txt``) and what MATLAB's convention prescribes."""
    import math
    from scipy.linalg import eigh_tridiagonal
    if ntapers is None:
        ntapers = math.ceil(2 * nw) - 1
    if n % 2 == 1 and zerophase:
        raise ValueError("ArgumentError: `dpss` does not currently support odd-length zerophase windows")
    if zerophase:
        n += 1
    if not 0 < ntapers <= n:
        raise ValueError("DomainError: ntapers must be in the interval (0, n]")
    if not 0 <= nw < n / 2:
        raise ValueError("DomainError: nw must be in the interval [0, n/2)")
    v = float(_cospi(np.array([2 * nw / n]))[0])
    i = np.arange(n, dtype=np.float64)
    dv = v * ((n - 1) / 2 - i) ** 2
    k = np.arange(1, n, dtype=np.float64)
    ev = 0.5 * (k * n - k * k)
    _, vec = eigh_tridiagonal(dv, ev, select="i", select_range=(n - ntapers, n - 1))
    rv = vec[:, ::-1].copy()
    for c in range(rv.shape[1]):
        if c % 2 == 1:                      # 1-based even tapers: first non-zero element positive (:698-707)
            nzv = rv[np.flatnonzero(rv[:, c])[0], c]
            if nzv < 0:
                rv[:, c] = -rv[:, c]
        elif rv[:, c].sum() < 0:            # symmetric tapers: positive mean
            rv[:, c] = -rv[:, c]
    if zerophase:
        rv = rv[:-1, :]
    if padding > 0:
        rv = np.vstack([rv, np.zeros((padding, ntapers))])
    if zerophase:
        rv = np.fft.ifftshift(rv, axes=0)
    return rv


def dpsseig(A: np.ndarray, nw: float) -> np.ndarray:
    """windows.jl:739-776: concentration ratios of the tapers (Percival & Walden ex. 8.1)."""
    from .util import nextfastfft
    n = A.shape[0]
    if not 0 <= nw < n / 2:
        raise ValueError("DomainError: nw must be in the interval [0, n/2)")
    w = nw / n
    seq = np.empty(n)
    seq[0] = 1.0
    ii = np.arange(1, n)
    seq[1:] = 2 * np.sinc(2 * w * ii)
    nfft = nextfastfft(2 * n - 1)
    q = np.empty(A.shape[1])
    for c in range(A.shape[1]):
        t = np.zeros(nfft)
        t[:n] = A[:, c]
        ac = np.fft.irfft(np.abs(np.fft.rfft(t)) ** 2, nfft) * nfft      # brfft is unnormalised
        q[c] = 2 * w * float(np.dot(seq, ac[:n])) / nfft
    return q
