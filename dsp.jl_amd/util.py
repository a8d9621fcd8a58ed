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


def promote_type(*dts) -> np.dtype:
    """Julia's ``promote_type`` for the element types of this path (Bool, Int*, UInt*, Float16/32/64 and their Complex forms).

    It is NOT ``np.result_type``: an integer never widens a float (``promote_type(Float32, Int64) == Float32``, numpy says float64), so
    ``filt(b::Vector{Float32}, 1, x::Vector{Float32})`` stays Float32 as in the reference (dspbase.jl:26-31, 775-777).  Mixed signedness
    follows Julia: the unsigned type wins at equal size, the larger type otherwise."""
    dts = [np.dtype(d) for d in dts]
    if not dts:
        raise TypeError("promote_type needs at least one type")
    cplx = any(d.kind == "c" for d in dts)
    reals = [np.dtype(np.float32) if d == _C32 else np.dtype(np.float64) if d == _C64 else d for d in dts]
    floats = [d for d in reals if d.kind == "f"]
    if floats:
        r = max(floats, key=lambda d: d.itemsize)
    else:
        ints = [d for d in reals if d.kind in "iu"]
        if not ints:
            r = np.dtype(np.bool_)
        else:
            size = max(d.itemsize for d in ints)
            unsigned_wins = any(d.kind == "u" and d.itemsize == size for d in ints)
            r = np.dtype(("u" if unsigned_wins else "i") + str(size))
    if not cplx:
        return r
    if r == np.dtype(np.float32) or r == np.dtype(np.float16):
        return _C32
    return _C64            # Complex{Int} has no numpy twin: the device computes those in ComplexF64 and rounds (dspbase._cast_result)
