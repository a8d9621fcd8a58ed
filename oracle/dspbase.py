"""Oracle for ``src/dspbase.jl``: time-domain filt, conv algorithm selection, 1-D overlap-save.

Test infrastructure only (see package docstring).  FFTs use scipy's pocketfft in the element type the
reference would use (Float32 stays Float32), standing in for FFTW.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.fft as sfft

from .util import nextfastfft

SMALL_FILT_CUTOFF = 66  # dspbase.jl:3
FFT_TYPES = (np.dtype(np.float32), np.dtype(np.float64), np.dtype(np.complex64), np.dtype(np.complex128))  # :674


def _promote(*dts):
    return np.result_type(*dts)


# ---------------------------------------------------------------------------------------------
# filt(b, a, x) -- transposed direct form II, dspbase.jl:14-105
# ---------------------------------------------------------------------------------------------
def filt_ba(b, a, x):
    """dspbase.jl:14-66.  ``b``/``a`` scalars or vectors; ``x`` (nx,) or (nx, cols...) column-major."""
    b = np.atleast_1d(np.asarray(b))
    a = np.atleast_1d(np.asarray(a))
    x = np.asarray(x)
    if b.size == 0:
        raise ValueError("filter vector b must be non-empty")          # :28
    if a.size == 0:
        raise ValueError("filter vector a must be non-empty")          # :29
    if a[0] == 0:
        raise ValueError("filter vector a[1] must be nonzero")         # :30
    T = _promote(b.dtype, a.dtype, x.dtype)
    out = np.empty(x.shape, dtype=T)
    if x.shape[0] == 0:
        return out                                                      # :39
    sz = max(len(a), len(b))
    if sz == 1:
        return (x * (b[0] / a[0])).astype(np.result_type(T, (b[0] / a[0]).dtype))  # :40
    if a[0] != 1:                                                       # :43-47
        norml = a[0]
        a = a / norml
        b = b / norml
        T = _promote(b.dtype, a.dtype, x.dtype)
        out = np.empty(x.shape, dtype=T)
    x2 = x.reshape(x.shape[0], -1)
    o2 = out.reshape(x.shape[0], -1)
    if len(a) == 1:
        for c in range(x2.shape[1]):
            o2[:, c] = _filt_fir(b.astype(T), x2[:, c].astype(T))
    else:
        for c in range(x2.shape[1]):
            o2[:, c] = _filt_iir(b.astype(T), a.astype(T), x2[:, c].astype(T))
    return out


def _filt_fir(b, x):
    """dspbase.jl:95-105.  out[i] = b1*x[i] + (b2*x[i-1] + (... + bn*x[i-n+1])), innermost first
    (the order the TDF-II state recursion accumulates in)."""
    nb, nx = len(b), len(x)
    xp = np.concatenate([np.zeros(nb - 1, dtype=x.dtype), x])
    acc = b[nb - 1] * xp[0:nx]
    for j in range(nb - 2, -1, -1):
        acc = xp[nb - 1 - j: nb - 1 - j + nx] * b[j] + acc
    return acc


def _filt_iir(b, a, x):
    """dspbase.jl:69-92 (serial; small inputs only)."""
    nb, na = len(b), len(a)
    silen = max(nb, na) - 1
    si = np.zeros(silen, dtype=x.dtype)
    out = np.empty_like(x)
    bp = np.concatenate([b, np.zeros(silen + 1 - nb, dtype=b.dtype)])
    ap = np.concatenate([a, np.zeros(silen + 1 - na, dtype=a.dtype)])
    for i in range(len(x)):
        xi = x[i]
        val = xi * bp[0] + si[0]
        out[i] = val
        for j in range(silen - 1):
            si[j] = val * (-ap[j + 1]) + (xi * bp[j + 1] + si[j + 1])
        si[silen - 1] = xi * bp[silen] - ap[silen] * val
    return out


def _filt_iir_state(b, a, x, si0):
    """dspbase.jl:69-92 with an initial state (used by iir_filtfilt); b, a zero-padded to equal length."""
    silen = len(b) - 1
    si = np.array(si0, dtype=x.dtype, copy=True)
    out = np.empty_like(x)
    for i in range(len(x)):
        xi = x[i]
        val = xi * b[0] + (si[0] if silen else 0)
        out[i] = val
        for j in range(silen - 1):
            si[j] = val * (-a[j + 1]) + (xi * b[j + 1] + si[j + 1])
        if silen:
            si[silen - 1] = xi * b[silen] - a[silen] * val
    return out


# ---------------------------------------------------------------------------------------------
# FFT length selection, dspbase.jl:262-291
# ---------------------------------------------------------------------------------------------
def os_fft_complexity(nfft, nb):
    """dspbase.jl:262."""
    return (nfft * math.log2(nfft) + nfft) / (nfft - nb + 1)


def optimalfftfiltlength(nb: int, nx: int) -> int:
    """dspbase.jl:268-291."""
    nfull = nb + nx - 1
    first_pow2 = math.ceil(math.log2(nb))
    max_pow2 = math.ceil(math.log2(nfull))
    prev = os_fft_complexity(2 ** first_pow2, nb)
    pow2 = first_pow2 + 1
    while pow2 <= max_pow2:
        new = os_fft_complexity(2 ** pow2, nb)
        if new > prev:
            break
        prev = new
        pow2 += 1
    nfft = 2 ** max_pow2 if pow2 > max_pow2 else 2 ** (pow2 - 1)
    if nfft > nfull:
        nfft = nextfastfft(nfull)
    return nfft


# ---------------------------------------------------------------------------------------------
# conv kernels (1-D), dspbase.jl:490-660
# ---------------------------------------------------------------------------------------------
def _conv_td(u, v, out_len, T):
    """dspbase.jl:646-660 (direct)."""
    out = np.zeros(out_len, dtype=T)
    if len(u) and len(v):
        out[:len(u) + len(v) - 1] = np.convolve(u.astype(T), v.astype(T))
    return out


def _conv_kern_fft(u, v, out_len, T):
    """dspbase.jl:611-644 (one FFT of nextfastfft(outsize))."""
    outsize = len(u) + len(v) - 1
    nfft = nextfastfft(outsize)
    out = np.zeros(out_len, dtype=T)
    if np.dtype(T).kind == "c":
        up = np.zeros(nfft, dtype=T); up[:len(u)] = u
        vp = np.zeros(nfft, dtype=T); vp[:len(v)] = v
        raw = sfft.ifft(sfft.fft(up) * sfft.fft(vp))
    else:
        up = np.zeros(nfft, dtype=T); up[:len(u)] = u
        vp = np.zeros(nfft, dtype=T); vp[:len(v)] = v
        raw = sfft.irfft(sfft.rfft(up) * sfft.rfft(vp), nfft)
    out[:outsize] = raw[:outsize]
    return out


def os_block_table(su: int, sv: int, nfft: int, sout: int):
    """Block geometry of ``unsafe_conv_kern_os!`` for N=1 (dspbase.jl:495-528, :447-482, :586-605).

    Returns a list of dicts, one per block, with 1-based inclusive ranges exactly as the reference
    computes them: ``data`` (range of u copied), ``dest`` (first index in tdbuff), ``out`` (range of
    out written), ``valid`` (range of tdbuff copied out) and ``edge`` (bool).
    """
    ideal_save = nfft - sv + 1
    sout_deficit = max(0, ideal_save - sout)
    save = ideal_save - sout_deficit
    nblocks = -(-sout // save)
    first_center = -(-(sv - 1) // save) + 1
    last_center = su // save
    if last_center > 1:
        edge_idx = list(range(1, first_center)) + list(range(last_center + 1, nblocks + 1))
        center_idx = list(range(first_center, last_center + 1))
    else:
        edge_idx = list(range(1, nblocks + 1))
        center_idx = []
    blocks = []
    for k in edge_idx:
        data_offset = save * (k - 1)
        pad_before = max(0, sv - data_offset - 1)
        data_ideal_stop = data_offset + save
        pad_after = max(0, data_ideal_stop - su)
        d0 = 1 + data_offset - sv + pad_before + 1
        d1 = 1 + data_ideal_stop - pad_after - 1
        out_stop = min(1 + data_offset + save - 1, sout)
        u_deficit = max(0, pad_after - sv + 1)
        blocks.append(dict(k=k, edge=True, data=(d0, d1), dest=pad_before + 1,
                           out=(1 + data_offset, out_stop), valid=(sv, nfft - u_deficit - sout_deficit)))
    for k in center_idx:
        data_offset = save * (k - 1)
        data_stop = data_offset + save
        blocks.append(dict(k=k, edge=False, data=(1 + data_offset - sv + 1, 1 + data_stop - 1), dest=1,
                           out=(data_offset + 1, data_stop), valid=(sv, nfft)))
    return blocks, save, nblocks


def unsafe_conv_kern_os(u, v, nfft: int, out_len: int, T):
    """dspbase.jl:490-609 restricted to N=1 (the hot-path case).  ``u`` is the longer input."""
    T = np.dtype(T)
    su, sv = len(u), len(v)
    out = np.zeros(out_len, dtype=T)
    sout = out_len
    blocks, save, nblocks = os_block_table(su, sv, nfft, sout)
    cplx = T.kind == "c"
    vp = np.zeros(nfft, dtype=T)
    vp[:sv] = v
    # filter spectrum, normalised once (dspbase.jl:514-516)
    filter_fd = (sfft.fft(vp) if cplx else sfft.rfft(vp)) * T.type(1 / nfft).real
    filter_fd = filter_fd.astype(np.result_type(T, np.complex64))
    uu = np.asarray(u, dtype=T)
    for blk in blocks:
        td = np.zeros(nfft, dtype=T)
        d0, d1 = blk["data"]
        n = d1 - d0 + 1
        if n > 0:
            td[blk["dest"] - 1: blk["dest"] - 1 + n] = uu[d0 - 1:d1]
        if cplx:   # os_conv_block!, dspbase.jl:348-356 (unnormalised backward transform)
            td = (sfft.ifft(sfft.fft(td) * filter_fd) * nfft).astype(T)
        else:      # dspbase.jl:337-345 (plan_brfft is unnormalised)
            td = (sfft.irfft(sfft.rfft(td) * filter_fd, nfft) * nfft).astype(T)
        o0, o1 = blk["out"]
        v0, v1 = blk["valid"]
        m = min(o1 - o0 + 1, v1 - v0 + 1)
        if m > 0:
            out[o0 - 1:o0 - 1 + m] = td[v0 - 1:v0 - 1 + m]
    return out


def conv_select_algorithm(nu: int, nv: int, T, algorithm: str = "auto") -> str:
    """Algorithm choice of ``conv!`` (dspbase.jl:720-743) for 1-D inputs."""
    T = np.dtype(T)
    if algorithm == "auto":
        algorithm = "fast" if T in FFT_TYPES else "direct"
    if algorithm == "fast":
        algorithm = "direct" if nu * nv < 2 ** 16 else "fft"
    if algorithm == "direct" or nu == 0 or nv == 0:
        return "direct"
    if algorithm == "fft":
        nl, ns = max(nu, nv), min(nu, nv)
        os_nfft = optimalfftfiltlength(ns, nl)
        algorithm = "fft_overlapsave" if os_nfft < nu + nv - 1 else "fft_simple"
    if algorithm not in ("fft_overlapsave", "fft_simple"):
        raise ValueError("algorithm must be :auto, :fast, :direct, :fft, :fft_simple, or :fft_overlapsave")
    return algorithm


def conv(u, v, algorithm: str = "auto", out_len: int | None = None):
    """``conv`` / ``conv!`` for vectors (dspbase.jl:709-792).  ``out_len`` > nu+nv-1 zero-fills the tail."""
    u = np.asarray(u)
    v = np.asarray(v)
    T = np.result_type(u.dtype, v.dtype)
    nu, nv = len(u), len(v)
    full = max(nu + nv - 1, 0)
    if out_len is None:
        out_len = full
    alg = conv_select_algorithm(nu, nv, T, algorithm)
    if alg == "direct":
        return _conv_td(u, v, out_len, T)
    Tf = T if T in FFT_TYPES else np.result_type(T, np.float64)
    if alg == "fft_simple":
        res = _conv_kern_fft(u.astype(Tf), v.astype(Tf), out_len, Tf)
    else:
        big, small = (u, v) if nu >= nv else (v, u)
        nfft = optimalfftfiltlength(len(small), len(big))
        res = unsafe_conv_kern_os(big.astype(Tf), small.astype(Tf), nfft, out_len, Tf)
        res[full:] = 0
    return res


# ---------------------------------------------------------------------------------------------
# conv for arrays, dspbase.jl:611-660, :709-818
# ---------------------------------------------------------------------------------------------
def _conv_td_nd(u, v, T):
    """_conv_td! (dspbase.jl:646-660): out[n + m] += u[m] v[n] over all index pairs."""
    so = tuple(a + b - 1 for a, b in zip(u.shape, v.shape))
    out = np.zeros(so, dtype=T)
    small, big = (u, v) if u.size <= v.size else (v, u)
    for m in np.ndindex(*small.shape):
        sl = tuple(slice(i, i + n) for i, n in zip(m, big.shape))
        out[sl] += small[m] * big.astype(T)
    return out


def _conv_kern_fft_nd(u, v, T):
    """_conv_kern_fft! (dspbase.jl:611-644): one N-d (r)fft of nextfastfft(outsize) per dimension."""
    so = tuple(a + b - 1 for a, b in zip(u.shape, v.shape))
    nffts = tuple(nextfastfft(n) for n in so)
    up = np.zeros(nffts, dtype=T); up[tuple(slice(0, n) for n in u.shape)] = u
    vp = np.zeros(nffts, dtype=T); vp[tuple(slice(0, n) for n in v.shape)] = v
    if np.dtype(T).kind == "c":
        raw = sfft.ifftn(sfft.fftn(up) * sfft.fftn(vp))
    else:
        # Julia's rfft halves the FIRST dimension; numpy's rfftn the last -- the transform is the same set of sums
        raw = sfft.irfftn(sfft.rfftn(up) * sfft.rfftn(vp), nffts)
    return raw[tuple(slice(0, n) for n in so)].astype(T)


def conv_nd(u, v, algorithm: str = "auto"):
    """``conv(u, v; algorithm)`` for arrays (dspbase.jl:709-792), trailing-singleton promotion of :784-792 included.
    :fft_overlapsave is evaluated by the single-transform kernel (same sums; the block geometry is a CPU memory device)."""
    u = np.asarray(u)
    v = np.asarray(v)
    nd = max(u.ndim, v.ndim)
    u = u.reshape(u.shape + (1,) * (nd - u.ndim))
    v = v.reshape(v.shape + (1,) * (nd - v.ndim))
    T = np.result_type(u.dtype, v.dtype)
    so = tuple(max(a + b - 1, 0) for a, b in zip(u.shape, v.shape))
    if algorithm not in ("auto", "fast", "direct", "fft", "fft_simple", "fft_overlapsave"):
        raise ValueError("algorithm must be :auto, :fast, :direct, :fft, :fft_simple, or :fft_overlapsave")
    if u.size == 0 or v.size == 0:
        return np.zeros(so, dtype=T)
    alg = algorithm
    if alg == "auto":
        alg = "fast" if T in FFT_TYPES else "direct"
    if alg == "fast":
        alg = "direct" if u.size * v.size < 2 ** 16 else "fft"
    if alg == "direct":
        return _conv_td_nd(u, v, T)
    Tf = T if T in FFT_TYPES else np.result_type(T, np.float64)
    res = _conv_kern_fft_nd(u.astype(Tf), v.astype(Tf), Tf)
    return np.round(res).astype(T) if T.kind in "iu" else res


def conv_separable(u, v, A):
    """``conv(u, transpose(v), A)`` (dspbase.jl:801-818): fft sizes m, n exactly (no nextfastfft)."""
    u = np.asarray(u); v = np.asarray(v); A = np.asarray(A)
    T = np.result_type(u.dtype, v.dtype, A.dtype)
    m, n = len(u) + A.shape[0] - 1, len(v) + A.shape[1] - 1
    B = np.zeros((m, n), dtype=np.result_type(T, np.float64)); B[:A.shape[0], :A.shape[1]] = A
    uf = sfft.fft(np.concatenate([u, np.zeros(m - len(u))]))
    vf = sfft.fft(np.concatenate([v, np.zeros(n - len(v))]))
    Cm = sfft.ifft2(sfft.fft2(B) * (uf[:, None] * vf[None, :]))
    return Cm if T.kind == "c" else Cm.real


def xcorr(u, v=None, padmode: str = "none", scaling: str = "none"):
    """dspbase.jl:867-898."""
    u = np.asarray(u)
    v = u if v is None else np.asarray(v)
    su, sv = len(u), len(v)
    if scaling == "biased" and su != sv:
        raise ValueError("scaling only valid for vectors of same length")
    if padmode == "longest":
        if su < sv:
            u = np.concatenate([u, np.zeros(sv - su, dtype=u.dtype)])
        elif sv < su:
            v = np.concatenate([v, np.zeros(su - sv, dtype=v.dtype)])
    elif padmode != "none":
        raise ValueError("padmode keyword argument must be either :none or :longest")
    res = conv(u, np.conj(v)[::-1])
    if scaling == "biased":
        res = res / su
    return res
