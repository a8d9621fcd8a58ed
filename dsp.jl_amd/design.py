"""Host-side FIR tap generators needed as *inputs* of the hot path (DSP.jl ``src/Filters/design.jl``)."""
from __future__ import annotations

import functools
import math
from fractions import Fraction

import numpy as np

from . import windows
from ._lib import DomainError


def kaiserord(transitionwidth, attenuation=60):
    """design.jl:547-559 -> (n, alpha)."""
    n = int(math.ceil((attenuation - 7.95) / (math.pi * 2.285 * transitionwidth))) + 1
    if attenuation > 50:
        beta = 0.1102 * (attenuation - 8.7)
    elif attenuation >= 21:
        beta = 0.5842 * (attenuation - 21) ** 0.4 + 0.07886 * (attenuation - 21)
    else:
        beta = 0.0
    return n, beta / math.pi


def lowpass_firwindow(w, window, fs=2, scale=True) -> np.ndarray:
    """``digitalfilter(Lowpass(w), FIRWindow(window; scale); fs)`` (design.jl:235-240, :598-602, :642, :669-674)."""
    if w <= 0:
        raise DomainError("frequencies must be positive")
    f = 2 * w / fs
    if f >= 1:
        raise DomainError("frequencies must be less than the Nyquist frequency")
    window = np.asarray(window, dtype=np.float64)
    n = len(window)
    k = np.arange(1, n + 1)
    taps = f * np.sinc(f * (k - (n + 1) / 2)) * window
    return taps / taps.sum() if scale else taps


def resample_filter(rate, *args) -> np.ndarray:
    """design.jl:683-720.  Rational/integer ``rate`` -> (rel_bw=1.0, attenuation=60); float -> (Nphi=32, rel_bw, att).
    The design (a Kaiser window of a few thousand points) is memoised per argument tuple: ``resample(x, rate)`` redesigns the same
    filter on every call."""
    try:
        return _resample_filter_cached(rate if isinstance(rate, float) else Fraction(rate), tuple(args)).copy()
    except TypeError:           # unhashable argument: design without the memo
        return _resample_filter(rate, *args)


@functools.lru_cache(maxsize=32)
def _resample_filter_cached(rate, args):
    return _resample_filter(rate, *args)


def _resample_filter(rate, *args) -> np.ndarray:
    if isinstance(rate, float):
        nphi = int(args[0]) if len(args) > 0 else 32
        rel_bw = args[1] if len(args) > 1 else 1.0
        att = args[2] if len(args) > 2 else 60
        f_nyq = 1.0 / nphi if rate >= 1.0 else rate / nphi
    else:
        r = Fraction(rate)
        nphi = r.numerator
        rel_bw = args[0] if len(args) > 0 else 1.0
        att = args[1] if len(args) > 1 else 60
        f_nyq = min(1 / nphi, 1 / r.denominator)
    cutoff = f_nyq * rel_bw
    hlen, alpha = kaiserord(cutoff * 0.2, att)
    hlen = nphi * int(math.ceil(hlen / nphi))
    if hlen % 2 == 0:
        hlen += 1
    return lowpass_firwindow(cutoff, windows.kaiser(hlen, alpha)) * nphi
