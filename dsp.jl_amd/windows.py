"""Host-side window generators (DSP.jl ``src/windows.jl``): always Float64 vectors, evaluated once per plan."""
from __future__ import annotations

import functools

import numpy as np

from ._lib import ArgumentError, DomainError


def _cospi(x):
    """cos(pi x) with the argument reduced exactly before the trig call (Julia's ``cospi``)."""
    x = np.abs(np.asarray(x, dtype=np.float64)) % 2.0
    x = np.where(x > 1.0, 2.0 - x, x)
    flip = x > 0.5
    x = np.where(flip, 1.0 - x, x)
    c = np.where(x > 0.25, np.sin(np.pi * (0.5 - x)), np.cos(np.pi * x))
    return np.where(flip, -c, c)


def makewindow(winfunc, n: int, padding: int = 0, zerophase: bool = False) -> np.ndarray:
    """windows.jl:97-121: sample ``winfunc`` on range(-0.5, 0.5; length=n) (or its zero-phase rotation)."""
    if n < 0:
        raise ArgumentError("`n` must be nonnegative")
    if padding < 0:
        raise ArgumentError("`padding` must be nonnegative")
    win = np.zeros(n + padding)
    if n == 1:
        win[0] = winfunc(np.zeros(1))[0]
    elif zerophase:
        half = n // 2
        win[: half + 1] = winfunc(np.arange(half + 1) / n)
        if half:
            win[-half:] = winfunc(np.arange(-half, 0) / n)
    elif n > 1:
        k = np.arange(n)
        win[:n] = winfunc((2.0 * k - (n - 1)) / (2.0 * (n - 1)))   # correctly rounded -0.5 + k/(n-1)
    return win


def rect(n, padding=0, zerophase=False):
    """windows.jl:142."""
    return makewindow(lambda x: np.ones_like(x), n, padding, zerophase)


def hanning(n, padding=0, zerophase=False):
    """windows.jl:181-183."""
    return makewindow(lambda x: 0.5 * (1.0 + _cospi(2.0 * x)), n, padding, zerophase)


hann = hanning


def hamming(n, padding=0, zerophase=False):
    """windows.jl:206-208."""
    return makewindow(lambda x: 0.46 * _cospi(2.0 * x) + 0.54, n, padding, zerophase)


def cosine(n, padding=0, zerophase=False):
    """windows.jl:289: cospi(x)."""
    return makewindow(lambda x: _cospi(x), n, padding, zerophase)


def bartlett(n, padding=0, zerophase=False):
    """windows.jl:380-382."""
    return makewindow(lambda x: 1.0 - np.abs(2.0 * x), n, padding, zerophase)


def blackman(n, padding=0, zerophase=False):
    """windows.jl:455: 0.42 + 0.5 cospi(2x) + 0.08 cospi(4x)."""
    return makewindow(lambda x: 0.5 * _cospi(2.0 * x) + (0.08 * _cospi(4.0 * x) + 0.42), n, padding, zerophase)


def kaiser(n, alpha, padding=0, zerophase=False):
    """windows.jl:600-605."""
    scale = 1.0 / np.i0(np.pi * alpha)
    return makewindow(lambda x: scale * np.i0(np.pi * alpha * np.sqrt(np.clip(1.0 - 4.0 * x * x, 0.0, None))), n, padding, zerophase)


def tukey(n, alpha, padding=0, zerophase=False):
    """windows.jl:245-263."""
    if not 0 <= alpha <= 1:
        raise DomainError("α must be in the range [0, 1].")
    if abs(alpha) <= np.finfo(np.float64).eps:
        return rect(n, padding, zerophase)

    def f(x):
        lo = 0.5 * (1 + _cospi(2 / alpha * (x + (1 - alpha) / 2)))
        hi = 0.5 * (1 + _cospi(2 / alpha * (x - (1 - alpha) / 2)))
        return np.where(x <= -(1 - alpha) / 2, lo, np.where(x <= (1 - alpha) / 2, 1.0, hi))
    return makewindow(f, n, padding, zerophase)


def lanczos(n, padding=0, zerophase=False):
    """windows.jl:314-316."""
    return makewindow(lambda x: np.sinc(2 * x), n, padding, zerophase)


def triang(n, padding=0, zerophase=False):
    """windows.jl:350-357."""
    m = n + 1 if zerophase else n
    scale = 2 * (m - 1) / m if m % 2 == 0 else 2 * (m - 1) / (m + 1)
    return makewindow(lambda x: -scale * np.abs(x) + 1, n, padding, zerophase)


def gaussian(n, sigma, padding=0, zerophase=False):
    """windows.jl:405-408."""
    if not sigma > 0:
        raise DomainError("σ must be positive")
    return makewindow(lambda x: np.exp(-0.5 * (x / sigma) ** 2), n, padding, zerophase)


def bartlett_hann(n, padding=0, zerophase=False):
    """windows.jl:429-434."""
    return makewindow(lambda x: 0.38 * _cospi(2 * x) + (-0.48 * np.abs(x) + 0.62), n, padding, zerophase)


def blackmanharris(n, term=4, padding=0, zerophase=False):
    """windows.jl:503-517."""
    if term == 4:
        a0, a1, a2, a3 = 0.35875, 0.48829, 0.14128, 0.01168
        return makewindow(lambda x: a1 * _cospi(2 * x) + (a2 * _cospi(4 * x) + (a3 * _cospi(6 * x) + a0)), n, padding, zerophase)
    if term == 3:
        a0, a1, a2 = 0.42323, 0.49755, 0.07922
        return makewindow(lambda x: a1 * _cospi(2 * x) + (a2 * _cospi(4 * x) + a0), n, padding, zerophase)
    raise ArgumentError("`term` must be either 3 or 4")


def nuttall(n, term=4, padding=0, zerophase=False):
    """windows.jl:556-570."""
    if term == 4:
        a0, a1, a2, a3 = 0.3635819, 0.4891775, 0.1365995, 0.0106411
        return makewindow(lambda x: a1 * _cospi(2 * x) + (a2 * _cospi(4 * x) + (a3 * _cospi(6 * x) + a0)), n, padding, zerophase)
    if term == 3:
        a0, a1, a2 = 0.4243801, 0.4973406, 0.0782793
        return makewindow(lambda x: a1 * _cospi(2 * x) + (a2 * _cospi(4 * x) + a0), n, padding, zerophase)
    raise ArgumentError("`term` must be either 3 or 4")


def flattop(n, padding=0, zerophase=False):
    """windows.jl:640-645."""
    a0, a1, a2, a3, a4 = 0.21557895, 0.41663158, 0.277263158, 0.083578947, 0.006947368
    return makewindow(lambda x: a1 * _cospi(2 * x) + (a2 * _cospi(4 * x) + (a3 * _cospi(6 * x) + (a4 * _cospi(8 * x) + a0))), n, padding, zerophase)


def dpss(n: int, nw: float, ntapers: int | None = None, padding: int = 0, zerophase: bool = False) -> np.ndarray:
    """Memoised front of ``_dpss`` (the tridiagonal eigenproblem costs tens of milliseconds at n = 2^14, and ``mt_pgram(s)`` /
    ``mt_spectrogram(s, n)`` ask for the same tapers on every call); returns a fresh copy."""
    try:
        return _dpss_cached(int(n), float(nw), None if ntapers is None else int(ntapers), int(padding), bool(zerophase)).copy()
    except TypeError:
        return _dpss(n, nw, ntapers, padding, zerophase)


@functools.lru_cache(maxsize=16)
def _dpss_cached(n, nw, ntapers, padding, zerophase):
    return _dpss(n, nw, ntapers, padding, zerophase)


def _dpss(n: int, nw: float, ntapers: int | None = None, padding: int = 0, zerophase: bool = False) -> np.ndarray:
    """``dpss(n, nw, ntapers=ceil(2nw)-1; padding, zerophase)`` (windows.jl:668-726): Slepian tapers, (n, ntapers).

    Host-side table generation like every window here (the reference solves the same symmetric tridiagonal
    eigenproblem with LAPACK); the first non-zero element of the skew-symmetric tapers is positive (:696-707), the
    symmetric ones have a positive mean."""
    import math
    from scipy.linalg import eigh_tridiagonal
    if ntapers is None:
        ntapers = math.ceil(2 * nw) - 1
    if n % 2 == 1 and zerophase:
        raise ArgumentError("`dpss` does not currently support odd-length zerophase windows")
    if zerophase:
        n += 1
    if not 0 < ntapers <= n:
        raise DomainError("ntapers must be in the interval (0, n]")
    if not 0 <= nw < n / 2:
        raise DomainError("nw must be in the interval [0, n/2)")
    v = float(_cospi(np.array([2 * nw / n]))[0])
    i = np.arange(n, dtype=np.float64)
    dv = v * ((n - 1) / 2 - i) ** 2
    k = np.arange(1, n, dtype=np.float64)
    ev = 0.5 * (k * n - k * k)
    _, vec = eigh_tridiagonal(dv, ev, select="i", select_range=(n - ntapers, n - 1))
    rv = np.ascontiguousarray(vec[:, ::-1])
    for c in range(rv.shape[1]):
        if c % 2 == 1:
            if rv[np.flatnonzero(rv[:, c])[0], c] < 0:
                rv[:, c] = -rv[:, c]
        elif rv[:, c].sum() < 0:
            rv[:, c] = -rv[:, c]
    if zerophase:
        rv = rv[:-1, :]
    if padding > 0:
        rv = np.vstack([rv, np.zeros((padding, ntapers))])
    if zerophase:
        rv = np.fft.ifftshift(rv, axes=0)
    return rv


def dpsseig(A: np.ndarray, nw: float) -> np.ndarray:
    """``dpsseig(A, nw)`` (windows.jl:739-776): concentration ratios of the tapers in ``A`` (host arithmetic)."""
    from .util import nextfastfft
    A = np.asarray(A, dtype=np.float64)
    n = A.shape[0]
    if not 0 <= nw < n / 2:
        raise DomainError("nw must be in the interval [0, n/2)")
    w = nw / n
    seq = np.empty(n)
    seq[0] = 1.0
    seq[1:] = 2 * np.sinc(2 * w * np.arange(1, n))
    nfft = nextfastfft(2 * n - 1)
    q = np.empty(A.shape[1])
    for c in range(A.shape[1]):
        t = np.zeros(nfft)
        t[:n] = A[:, c]
        ac = np.fft.irfft(np.abs(np.fft.rfft(t)) ** 2, nfft) * nfft
        q[c] = 2 * w * float(np.dot(seq, ac[:n])) / nfft
    return q
