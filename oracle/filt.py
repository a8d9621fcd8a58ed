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
