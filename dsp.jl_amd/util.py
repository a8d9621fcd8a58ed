"""Host-side helpers of the hot path: element-type rules and FFT-size selection (DSP.jl ``src/util.jl``).

Index arithmetic is delegated to the C ABI (``mdsp_nextfastfft`` etc.) so that Julia and Python hosts share one
bit-exact implementation.
"""
from __future__ import annotations

import numpy as np

from . import _lib

_F32, _F64, _C32, _C64 = (np.dtype(t) for t in (np.float32, np.float64, np.complex64, np.complex128))


def nextfastfft(n):
    """``nextfastfft(n)`` / ``nextfastfft.(ns)`` (util.jl:134-135)."""
    if isinstance(n, (tuple, list)):
        return tuple(int(_lib.lib().mdsp_nextfastfft(int(k))) for k in n)
    return int(_lib.lib().mdsp_nextfastfft(int(n)))


def fftintype(dt) -> np.dtype:
    """util.jl:93-95."""
    dt = np.dtype(dt)
    if dt in (_F32, _F64, _C32, _C64):
        return dt
    return _C64 if dt.kind == "c" else _F64


def fftouttype(dt) -> np.dtype:
    """util.jl:98-100."""
    dt = np.dtype(dt)
    if dt in (_C32, _C64):
        return dt
    return _C32 if dt == _F32 else _C64


def fftabs2type(dt) -> np.dtype:
    """util.jl:103-105."""
    dt = np.dtype(dt)
    return _F32 if dt in (_F32, _C32) else _F64


def rfftfreq(n: int, fs=1.0) -> np.ndarray:
    """AbstractFFTs.rfftfreq."""
    return np.arange(n // 2 + 1) * (fs / n)


def fftfreq(n: int, fs=1.0) -> np.ndarray:
    """AbstractFFTs.fftfreq."""
    k = np.arange(n)
    k[k > (n - 1) // 2] -= n
    return k * (fs / n)
