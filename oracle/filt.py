"""Oracle for the FIR front of ``src/Filters/filt.jl`` (:426-555): tdfilt / fftfilt / filt(b, x).

Test infrastructure only (see package docstring).
"""
from __future__ import annotations

import numpy as np
import scipy.fft as sfft

from .dspbase import SMALL_FILT_CUTOFF, filt_ba, optimalfftfiltlength


def fftfilt_block_table(nb: int, nx: int, nfft: int):
    """Per-block index arithmetic of ``_fftfilt!`` (filt.jl:490, :504-517), all 1-based as in the reference.

    Returns (L, [ (off, npadbefore, xstart, n, nout) ... ]).
    """
    L = min(nx, nfft - (nb - 1))
    rows = []
    if nx == 0 or L <= 0:
        return L, rows
    off = 1
    while off <= nx:
        npadbefore = max(0, nb - off)
        xstart = off - nb + npadbefore + 1
        n = min(nfft - npadbefore, nx - xstart + 1)
        nout = min(L, nx - off + 1)
        rows.append((off, npadbefore, xstart, n, nout))
        off += L
    return L, rows


def _fftfilt(b, x, nfft: int, W=None):
    """filt.jl:479-521.  ``x`` is (nx,) or (nx, cols...); filtering along the first axis."""
    b = np.asarray(b)
    x = np.asarray(x)
    if W is None:
        W = np.result_type(b.dtype, x.dtype)
        if np.dtype(W).kind != "f":
            W = np.float64
    W = np.dtype(W)
    nb = len(b)
    nx = x.shape[0]
    out = np.empty(x.shape, dtype=W)
    x2 = x.reshape(nx, -1)
    o2 = out.reshape(nx, -1)
    L, rows = fftfilt_block_table(nb, nx, nfft)
    tmp1 = np.zeros(nfft, dtype=W)
    tmp1[:nb] = (b.astype(W) / W.type(nfft))          # filt.jl:499
    filterft = sfft.rfft(tmp1)                         # :501
    for c in range(x2.shape[1]):
        for (off, npadbefore, xstart, n, nout) in rows:
            tmp1 = np.zeros(nfft, dtype=W)                             # :509
            tmp1[npadbefore:npadbefore + n] = x2[xstart - 1:xstart - 1 + n, c]   # :510
            tmp2 = sfft.rfft(tmp1) * filterft                           # :512-513
            td = (sfft.irfft(tmp2, nfft) * nfft).astype(W)              # :514 (brfft: unnormalised)
            o2[off - 1:off - 1 + nout, c] = td[nb - 1:nb - 1 + nout]    # :517
    return out


def fftfilt(b, x, nfft: int | None = None):
    """filt.jl:458-461."""
    b = np.asarray(b)
    x = np.asarray(x)
    if nfft is None:
        nfft = optimalfftfiltlength(len(b), x.size)     # note: length(x), not size(x,1)  (:459)
    return _fftfilt(b, x, nfft)


def tdfilt(h, x):
    """filt.jl:431-433."""
    return filt_ba(np.asarray(h), np.ones(1, dtype=np.asarray(h).dtype), x)


def filt(b, x):
    """filt.jl:445-446, :525-555: FFT path for real taps longer than SMALL_FILT_CUTOFF, else time domain."""
    b = np.asarray(b)
    x = np.asarray(x)
    T = np.result_type(b.dtype, x.dtype)
    real = b.dtype.kind in "fiu" and x.dtype.kind in "fiu"
    if real and len(b) > SMALL_FILT_CUTOFF:
        nfft = optimalfftfiltlength(len(b), x.shape[0])   # :545-546 uses size(x, 1)
        W = T if T.kind == "f" else np.dtype(np.float64)
        return _fftfilt(b, x, nfft, W).astype(T) if T.kind == "f" else _fftfilt(b, x, nfft, W)
    return tdfilt(b, x).astype(T)


# ---------------------------------------------------------------------------------------------
# Stateful FIR (DF2TFilter{PolynomialRatio} with a = [1]) and FIR filtfilt
# ---------------------------------------------------------------------------------------------
class DF2TFilterFIR:
    """filt.jl:122-181 restricted to FIR coefficients (length(a) == 1): the state is the TDF-II register file
    ``si`` of length nb-1 per column, advanced by ``_filt_fir!`` (dspbase.jl:95-105)."""

    def __init__(self, b, dtype=None, coldims=()):
        self.b = np.asarray(b)
        dt = np.result_type(self.b.dtype, dtype) if dtype is not None else self.b.dtype
        self.state = np.zeros((len(self.b) - 1,) + tuple(coldims), dtype=dt)

    def filt(self, x):
        x = np.asarray(x)
        if x.shape[1:] != self.state.shape[1:]:
            raise ValueError("ArgumentError: state size must match x")              # filt.jl:158
        T = np.result_type(self.b.dtype, x.dtype, self.state.dtype)
        out = np.empty(x.shape, dtype=T)
        b = self.b.astype(T) if T.kind != "c" else self.b
        n = self.state.shape[0] + 1
        x2 = x.reshape(x.shape[0], -1)
        o2 = out.reshape(x.shape[0], -1)
        s2 = self.state.reshape(self.state.shape[0], -1)
        if n == 1:
            o2[...] = x2 * b[0]                                                        # mul!(out, x, b[1]), :163
            return out
        for c in range(x2.shape[1]):
            si = s2[:, c].astype(T)
            for i in range(x2.shape[0]):                                               # _filt_fir! dspbase.jl:95-105
                xi = x2[i, c]
                o2[i, c] = xi * b[0] + si[0]
                for j in range(n - 2):
                    si[j] = xi * b[j + 1] + si[j + 1]
                si[n - 2] = b[n - 1] * xi
            s2[:, c] = si
        return out


def extrapolate_signal(sig, pad_length: int):
    """filt.jl:243-257: odd-symmetric extension by ``pad_length`` samples at both ends."""
    sig = np.asarray(sig)
    n = len(sig)
    i = np.arange(1, pad_length + 1)
    head = 2 * sig[0] - sig[1 + pad_length - i]          # out[i] = 2 sig[1] - sig[2 + pad - i]  (1-based)
    tail = 2 * sig[n - 1] - sig[n - 1 - i]               # out[n + pad + i] = 2 sig[n] - sig[n - i]
    return np.concatenate([head, sig, tail])


def filtfilt(b, x):
    """FIR ``filtfilt(b, x)`` filt.jl:301-325: filter with conv(b, reverse(b)) after odd extension by nb-1 samples."""
    b = np.asarray(b)
    x = np.asarray(x)
    nb = len(b)
    T = np.result_type(b.dtype, x.dtype)
    newb = filt_ba(b, np.ones(1, dtype=b.dtype), b[::-1].copy())                       # filt!(newb, b, newb), :309-310
    newb = np.concatenate([newb, np.zeros(nb - 1, dtype=newb.dtype)])
    for i in range(1, nb):
        newb[nb - 1 + i] = newb[nb - 1 - i]                                            # :312-314
    x2 = x.reshape(x.shape[0], -1)
    ext = np.empty((x.shape[0] + 2 * (nb - 1), x2.shape[1]), dtype=T)
    for c in range(x2.shape[1]):
        ext[:, c] = extrapolate_signal(x2[:, c], nb - 1)                               # :317-319
    y = filt(newb, ext)                                                                 # filt!(extrapolated, newb, extrapolated), :322
    return y[2 * nb - 2:].reshape((x.shape[0],) + x.shape[1:])                          # :325


def filt_stepstate(b, a):
    """filt.jl:370-398."""
    b = np.asarray(b, dtype=np.float64)
    a = np.asarray(a, dtype=np.float64)
    scale = a[0]
    if scale != 1:
        a, b = a / scale, b / scale
    sz = max(len(a), len(b))
    b = np.concatenate([b, np.zeros(sz - len(b))])
    a = np.concatenate([a, np.zeros(sz - len(a))])
    if sz == 1:
        return np.zeros(0), b, a
    A = np.hstack([-a[1:, None], np.eye(sz - 1, sz - 2)])
    B = a[1:] * (-b[0]) + b[1:]
    si = np.linalg.solve(np.eye(sz - 1) - A, B) * scale
    return si, b, a


def iir_filtfilt(b, a, x):
    """filt.jl:260-282 (the reference's own cross-check of the FIR ``filtfilt``, test/filt.jl:334-339)."""
    from .dspbase import _filt_iir_state
    b = np.asarray(b, dtype=np.float64)
    a = np.asarray(a, dtype=np.float64)
    x = np.asarray(x)
    pad = min(3 * (max(len(a), len(b)) - 1), x.shape[0] - 1)
    zi, bn, an = filt_stepstate(b, a)
    T = np.result_type(bn.dtype, x.dtype)
    x2 = x.reshape(x.shape[0], -1)
    out = np.empty(x2.shape, dtype=T)
    for c in range(x2.shape[1]):
        e = extrapolate_signal(x2[:, c].astype(T), pad)
        e = _filt_iir_state(bn, an, e, zi * e[0])
        e = e[::-1].copy()
        e = _filt_iir_state(bn, an, e, zi * e[0])
        for j in range(x2.shape[0]):
            out[j, c] = e[len(e) - pad - 1 - j]
    return out.reshape(x.shape)
