"""Oracle FIR tap generators (host-only in the reference).  Follows ``src/Filters/design.jl``.

Only the pieces the hot path needs as *inputs*: Kaiser order estimate, windowed-sinc low-pass,
and the default resampling filter.  Test infrastructure only.
"""
from __future__ import annotations

import math
from fractions import Fraction

import numpy as np

from . import windows


def kaiserord(transitionwidth: float, attenuation: float = 60.0):
    """design.jl:547-559 -> (n, alpha)."""
    n = math.ceil((attenuation - 7.95) / (math.pi * 2.285 * transitionwidth)) + 1
    if attenuation > 50:
        beta = 0.1102 * (attenuation - 8.7)
    elif attenuation >= 21:
        beta = 0.5842 * (attenuation - 21) ** 0.4 + 0.07886 * (attenuation - 21)
    else:
        beta = 0.0
    return n, beta / math.pi


def normalize_freq(w: float, fs: float) -> float:
    """design.jl:235-240."""
    if w <= 0:
        raise ValueError("frequencies must be positive")
    f = 2 * w / fs
    if f >= 1:
        raise ValueError("frequencies must be less than the Nyquist frequency")
    return f


def firprototype_lowpass(n: int, w: float, fs: float = 2.0) -> np.ndarray:
    """design.jl:598-602: w*sinc(w*(k-(n+1)/2)), k = 1..n."""
    wn = normalize_freq(w, fs)
    k = np.arange(1, n + 1, dtype=np.float64)
    return wn * np.sinc(wn * (k - (n + 1) / 2))


def digitalfilter_lowpass_firwindow(w: float, window: np.ndarray, fs: float = 2.0, scale: bool = True) -> np.ndarray:
    """design.jl:669-674 with ``scalefactor(::Lowpass) = sum(coefs)`` (design.jl:642)."""
    coefs = firprototype_lowpass(len(window), w, fs)
    out = coefs * np.asarray(window, dtype=np.float64)
    if scale:
        out = out * (1.0 / np.sum(out))
    return out


def _resample_filter(f_nyq: float, nphi: int, rel_bw: float, attenuation: float) -> np.ndarray:
    """design.jl:701-720."""
    cutoff = f_nyq * rel_bw
    trans_width = cutoff * 0.2
    hlen, alpha = kaiserord(trans_width, attenuation)
    hlen = nphi * math.ceil(hlen / nphi)
    if hlen % 2 == 0:
        hlen += 1
    h = digitalfilter_lowpass_firwindow(cutoff, windows.kaiser(hlen, alpha))
    return h * nphi


def resample_filter(rate, *args) -> np.ndarray:
    """design.jl:683-699.  ``rate`` Fraction/int -> rational form; float -> arbitrary form."""
    if isinstance(rate, float):
        nphi = int(args[0]) if len(args) > 0 else 32
        rel_bw = args[1] if len(args) > 1 else 1.0
        att = args[2] if len(args) > 2 else 60
        f_nyq = 1.0 / nphi if rate >= 1.0 else rate / nphi
        return _resample_filter(f_nyq, nphi, rel_bw, att)
    r = Fraction(rate)
    rel_bw = args[0] if len(args) > 0 else 1.0
    att = args[1] if len(args) > 1 else 60
    nphi, dec = r.numerator, r.denominator
    f_nyq = min(1 / nphi, 1 / dec)
    return _resample_filter(f_nyq, nphi, rel_bw, att)
