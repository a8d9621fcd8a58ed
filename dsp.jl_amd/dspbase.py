"""``filt(b, a, x)``, ``conv``, ``xcorr`` -- the host side of DSP.jl ``src/dspbase.jl`` over libmi355dsp.

Argument checks, promotion rules and algorithm selection follow the reference line by line; the arithmetic runs
in HIP kernels (time-domain FIR: ``mdsp_tdfir_exec``; FFT convolution: ``mdsp_ols_*``).  Mutating ``f!`` methods are
spelled ``f_`` here.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _dev, _lib, _plancache, util
from ._lib import ArgumentError, UnsupportedError

SMALL_FILT_CUTOFF = 66   # dspbase.jl:3
# Integer operands whose sums stay below this are convolved by Float64 FFTs and rounded: the transform's relative error (~1e-15 log2 n)
# times the bound is orders of magnitude below 1/2, so the integers are exact (the reference's :direct sum is O(nu nv)).
_INT_FFT_BOUND = 2.0 ** 36

_FFT_TYPES = tuple(np.dtype(t) for t in (np.float32, np.float64, np.complex64, np.complex128))   # dspbase.jl:674


def optimalfftfiltlength(nb: int, nx: int) -> int:
    """dspbase.jl:268-291 (evaluated by the C ABI so every host shares one implementation)."""
    return int(_lib.lib().mdsp_optimal_fft_len(int(nb), int(nx)))


def os_fft_complexity(nfft, nb):
    """dspbase.jl:262."""
    return (nfft * np.log2(nfft) + nfft) / (nfft - nb + 1)


def _host_vec(v) -> np.ndarray:
    if _dev.is_device_array(v) or hasattr(v, "cpu"):
        v = v.cpu().numpy()
    return np.atleast_1d(np.asarray(v))


def _compute_dtype(T: np.dtype) -> np.dtype:
    """Device arithmetic type for a result eltype T: FFT types as they are, everything else in Float64/ComplexF64."""
    T = np.dtype(T)
    if T in _FFT_TYPES:
        return T
    return np.dtype(np.complex128) if T.kind == "c" else np.dtype(np.float64)


def _cast_result(t, T: np.dtype):
    """Round a device result back to an integer eltype when the reference would have stayed in integers."""
    T = np.dtype(T)
    if T.kind in "iu":
        return t.round().to(_dev.torch_dtype(T))
    if T.kind == "b":
        return t.round().to(_dev.torch_dtype(np.int64))
    return t


# ---------------------------------------------------------------------------------------------------------
# tdfir plumbing shared with Filters.tdfilt
# ---------------------------------------------------------------------------------------------------------
def _tdfir(b: np.ndarray, x, T: np.dtype):
    """Zero-state FIR along the first axis of x with real taps b, result eltype T."""
    if b.dtype.kind == "c":
        raise UnsupportedError("complex FIR taps are not accelerated (use DSP.jl on the CPU)")
    W = _compute_dtype(T)
    cols, shape = _dev.to_columns(x, W)
    ncols, nx = cols.shape
    out = _dev.empty_columns(ncols, nx, W)
    if nx and ncols:
        taps = np.ascontiguousarray(b, dtype=np.float32 if W in (np.dtype(np.float32), np.dtype(np.complex64)) else np.float64)
        _lib.check(_lib.lib().mdsp_tdfir_exec(taps.ctypes.data_as(C.c_void_p), len(taps), _dev.md_dtype(W), _dev.ptr(cols), nx, ncols, nx,
                                              _dev.ptr(out), nx, _dev.stream_ptr()))
    return _dev.from_columns(_cast_result(out, T), shape, x)


def filt(b, a, x):
    """``filt(b, a, x)`` (dspbase.jl:14-66) for FIR filters (scalar / length-1 ``a``).

    IIR (``length(a) > 1``) is a serial recursion per column and is out of scope for the device path.
    """
    bv, av = _host_vec(b), _host_vec(a)
    xdt = _dev.np_dtype_of(x)
    if bv.size == 0:
        raise ArgumentError("filter vector b must be non-empty")
    if av.size == 0:
        raise ArgumentError("filter vector a must be non-empty")
    if av[0] == 0:
        raise ArgumentError("filter vector a[1] must be nonzero")
    if av.size > 1:
        raise UnsupportedError("IIR filt(b, a, x) is a serial recursion; only FIR (scalar a) runs on the device")
    T = util.promote_type(bv.dtype, av.dtype, xdt)   # eltype of `out`, fixed BEFORE the normalisation (dspbase.jl:14-15): Float32 taps and
    if av[0] != 1:                                   # signal with an integer `a` stay Float32.  Coefficient normalisation, :43-47
        bv = bv / av[0]
    return _tdfir(bv, x, T)


def filt_(out, b, a, x):
    """``filt!(out, b, a, x)`` (dspbase.jl:26-66): size check then write into ``out``."""
    if tuple(out.shape) != tuple(x.shape):
        raise ArgumentError(f"output size {tuple(out.shape)} must match input size {tuple(x.shape)}")
    res = filt(b, a, x)
    if isinstance(out, np.ndarray):
        out[...] = res if isinstance(res, np.ndarray) else res.cpu().numpy()
    else:
        out.copy_(res if not isinstance(res, np.ndarray) else _dev.torch.from_numpy(res))
    return out


# ---------------------------------------------------------------------------------------------------------
# conv
# ---------------------------------------------------------------------------------------------------------
class OlsPlan:
    """Owner of an ``mdsp_ols_plan`` (filter spectrum, rocFFT plans / tables, work buffers in HBM)."""

    def __init__(self, taps: np.ndarray, nfft: int, nx_hint: int, mode: int, engine: int = _lib.ENGINE_AUTO, cached: bool = False):
        """``cached``: borrow the plan from the library's own LRU (``mdsp_ols_plan_cached``: keyed by device, thread, stream and the
        contents of ``taps``) instead of owning a fresh one -- the per-call path of ``filt(b, x)`` / ``conv(u, v)``.  A borrowed plan
        is for immediate use; it is never destroyed from here."""
        self.dtype = taps.dtype
        self._h = C.c_void_p()
        self._owned = not cached
        taps = np.ascontiguousarray(taps)
        if cached:
            _dev.device()
            _lib.check(_lib.lib().mdsp_ols_plan_cached(C.byref(self._h), taps.ctypes.data_as(C.c_void_p), len(taps), int(nfft), int(nx_hint),
                                                       _dev.md_dtype(taps.dtype), mode, engine, _dev.stream_ptr()))
        else:
            _lib.check(_lib.lib().mdsp_ols_plan_create(C.byref(self._h), taps.ctypes.data_as(C.c_void_p), len(taps), int(nfft), int(nx_hint),
                                                       _dev.md_dtype(taps.dtype), mode, engine))
        nf, L, eng = C.c_int64(), C.c_int64(), C.c_int()
        _lib.check(_lib.lib().mdsp_ols_plan_info(self._h, C.byref(nf), C.byref(L), C.byref(eng)))
        self.nb, self.nfft, self.block_len, self.engine = len(taps), nf.value, L.value, eng.value

    def exec(self, cols, nout: int):
        ncols, nx = cols.shape
        out = _dev.empty_columns(ncols, nout, self.dtype)
        _lib.check(_lib.lib().mdsp_ols_exec(self._h, _dev.ptr(cols), nx, ncols, nx, _dev.ptr(out), nout, nout, _dev.stream_ptr()))
        return out

    def exec_host(self, cols: np.ndarray, nout: int) -> np.ndarray:
        """Filter HOST columns ((ncols, nx) C-contiguous numpy of the plan's dtype) through the pinned, chunked
        H2D || kernel || D2H pipeline of ``mdsp_ols_exec_host``; bit-identical to ``exec`` on a device copy."""
        ncols, nx = cols.shape
        out = np.empty((ncols, nout), dtype=self.dtype)
        _lib.check(_lib.lib().mdsp_ols_exec_host(self._h, cols.ctypes.data_as(C.c_void_p), nx, ncols, nx, out.ctypes.data_as(C.c_void_p), nout, nout, 0))
        return out

    def segment(self, col, first: int, count: int):
        """Blocks [first, first+count) of one column as the reference's ``tmp1`` contents, shape (count, nfft)."""
        seg = _dev.empty_columns(count, self.nfft, self.dtype)
        _lib.check(_lib.lib().mdsp_ols_segment(self._h, _dev.ptr(col), col.numel(), first, count, _dev.ptr(seg), _dev.stream_ptr()))
        return seg

    def __del__(self):
        try:
            if self._h and self._owned:
                _lib.lib().mdsp_ols_plan_destroy(self._h)
        except Exception:
            pass


def conv_select_algorithm(nu: int, nv: int, T, algorithm: str = "auto") -> str:
    """Algorithm choice of ``conv!`` (dspbase.jl:720-743, 1-D)."""
    T = np.dtype(T)
    if algorithm == "auto":
        algorithm = "fast" if T in _FFT_TYPES else "direct"
    if algorithm == "fast":
        algorithm = "direct" if nu * nv < 2 ** 16 else "fft"
    if algorithm == "direct" or nu == 0 or nv == 0:
        return "direct"
    if algorithm == "fft":
        os_nfft = optimalfftfiltlength(min(nu, nv), max(nu, nv))
        algorithm = "fft_overlapsave" if os_nfft < nu + nv - 1 else "fft_simple"
    if algorithm not in ("fft_overlapsave", "fft_simple"):
        raise ArgumentError("algorithm must be :auto, :fast, :direct, :fft, :fft_simple, or :fft_overlapsave")
    return algorithm


def conv(u, v, algorithm: str = "auto", out_len: int | None = None, engine: int = _lib.ENGINE_AUTO):
    """``conv(u, v; algorithm)`` (dspbase.jl:709-792): vectors here, arrays in ``_conv_nd``, ``conv(u, v, A)`` with a matrix
    ``A`` is the separable form (:801-818).  ``out_len`` > nu+nv-1 zero-fills the tail like ``conv!`` into an oversized
    ``out`` (:733-735)."""
    if not isinstance(algorithm, str):                  # conv(u, transpose(v), A): the third positional argument is the matrix
        return conv_separable(u, v, algorithm)
    udt, vdt = _dev.np_dtype_of(u), _dev.np_dtype_of(v)
    if len(u.shape) != 1 or len(v.shape) != 1:
        if out_len is not None:
            raise ArgumentError("out_len is a vector argument")
        return _conv_nd(u, v, algorithm)
    T = util.promote_type(udt, vdt)
    nu, nv = int(u.shape[0]), int(v.shape[0])
    full = max(nu + nv - 1, 0)
    n_out = full if out_len is None else int(out_len)
    if n_out < full:
        raise ArgumentError("output too small for the convolution result")
    alg = conv_select_algorithm(nu, nv, T, algorithm)
    W = _compute_dtype(T)
    like = u
    if nu == 0 or nv == 0:
        z = _dev.torch.zeros(n_out, dtype=_dev.torch_dtype(T), device=_dev.device())
        return z if _dev.is_device_array(like) else z.cpu().numpy()
    big, small = (u, v) if nu >= nv else (v, u)
    if T.kind in "iu":
        # the reference stays in integer arithmetic (exact); the device computes in Float64, exact below 2^53
        bound = float(abs(_host_vec(small).astype(np.float64)).max()) * float(_dev.to_columns(big, np.float64)[0].abs().max()) * min(nu, nv)
        if bound >= 2.0 ** 53:
            raise UnsupportedError("integer convolution would not be exact in Float64; use DSP.jl on the CPU")
        if alg == "direct" and algorithm == "auto" and nu * nv >= 2 ** 16 and bound < _INT_FFT_BOUND:
            alg = conv_select_algorithm(nu, nv, np.float64, "fft")   # same integers, O(n log n): the FFT's error stays far below 1/2
    small_h = _host_vec(small).astype(W)
    if alg == "direct" and W.kind != "c":
        # direct sum: zero-state FIR over u extended by nv-1 zeros (dspbase.jl:646-660)
        cols, _ = _dev.to_columns(big, W)
        ext = _dev.torch.zeros((1, full), dtype=cols.dtype, device=cols.device)
        ext[:, :cols.shape[1]] = cols
        res = _dev.empty_columns(1, full, W)
        taps = np.ascontiguousarray(small_h)
        _lib.check(_lib.lib().mdsp_tdfir_exec(taps.ctypes.data_as(C.c_void_p), len(taps), _dev.md_dtype(W), _dev.ptr(ext), full, 1, full,
                                              _dev.ptr(res), full, _dev.stream_ptr()))
    else:
        if alg == "fft_simple":
            nfft = int(_lib.lib().mdsp_nextfastfft(full))           # _conv_kern_fft!: one transform of nextfastfft(outsize)
        else:
            nfft = optimalfftfiltlength(len(small_h), max(nu, nv))
        cols, _ = _dev.to_columns(big, W)
        plan = OlsPlan(small_h, nfft, max(nu, nv), _lib.OLS_CONV, engine, cached=True)     # the library's own plan LRU (mdsp_ols_plan_cached); any kernel length
        res = plan.exec(cols, full)
    res = _cast_result(res, T)[0]
    if n_out > full:
        pad = _dev.torch.zeros(n_out, dtype=res.dtype, device=res.device)
        pad[:full] = res
        res = pad
    return res if _dev.is_device_array(like) else res.cpu().numpy()


def _conv_nd(u, v, algorithm: str = "auto"):
    """``conv(u, v; algorithm)`` for arrays (dspbase.jl:709-792).  Operands of different rank are promoted with trailing
    singleton dimensions (:784-792).  :direct (chosen by the reference for integer eltypes and for length(u) length(v) <
    2^16) is the convolution sum on the device; every FFT algorithm is one N-d transform of the padded output
    (``_conv_kern_fft!``, :611-644 -- :fft_overlapsave computes the same sums block-wise)."""
    udt, vdt = _dev.np_dtype_of(u), _dev.np_dtype_of(v)
    T = util.promote_type(udt, vdt)
    nd = max(len(u.shape), len(v.shape))
    su = tuple(int(k) for k in u.shape) + (1,) * (nd - len(u.shape))
    sv = tuple(int(k) for k in v.shape) + (1,) * (nd - len(v.shape))
    so = tuple(max(a + b - 1, 0) for a, b in zip(su, sv))
    like = u
    if algorithm not in ("auto", "fast", "direct", "fft", "fft_simple", "fft_overlapsave"):
        raise ArgumentError("algorithm must be :auto, :fast, :direct, :fft, :fft_simple, or :fft_overlapsave")
    nu, nv = int(np.prod(su)), int(np.prod(sv))
    if nu == 0 or nv == 0:
        z = _dev.torch.zeros(so, dtype=_dev.torch_dtype(T), device=_dev.device())
        return z if _dev.is_device_array(like) else z.cpu().numpy()
    alg = algorithm
    if alg == "auto":
        alg = "fast" if T in _FFT_TYPES else "direct"
    if alg == "fast":
        alg = "direct" if nu * nv < 2 ** 16 else "fft"
    W = _compute_dtype(T)
    ud = _dev.as_device(u, W).reshape(su).contiguous()
    vd = _dev.as_device(v, W).reshape(sv).contiguous()
    if T.kind in "iu":
        bound = float(ud.abs().max()) * float(vd.abs().max()) * min(nu, nv)
        if bound >= 2.0 ** 53:
            raise UnsupportedError("integer convolution would not be exact in Float64; use DSP.jl on the CPU")
        if alg == "direct" and algorithm == "auto" and nu * nv >= 2 ** 16 and bound < _INT_FFT_BOUND and sum(1 for a, b in zip(su, sv) if a > 1 or b > 1) <= 3:
            alg = "fft"                                               # exact after rounding, O(n log n) instead of O(nu nv)
    out = _dev.torch.empty(so, dtype=ud.dtype, device=ud.device)
    # C-ordered (row-major) arrays are column-major arrays with the dimensions reversed; convolution treats every
    # dimension alike, so only the size vectors are reversed
    csu = (C.c_int64 * nd)(*reversed(su))
    csv = (C.c_int64 * nd)(*reversed(sv))
    fn = _lib.lib().mdsp_convnd_direct if alg == "direct" else _lib.lib().mdsp_convnd_fft
    _lib.check(fn(_dev.ptr(ud), csu, _dev.ptr(vd), csv, nd, _dev.md_dtype(W), _dev.ptr(out), _dev.stream_ptr()))
    res = out
    if T.kind in "iu":
        res = _dev.torch.round(res).to(_dev.torch_dtype(T))
    elif res.dtype != _dev.torch_dtype(T):
        res = res.to(_dev.torch_dtype(T))
    return res if _dev.is_device_array(like) else res.cpu().numpy()


def conv_separable(u, v, A):
    """``conv(u, transpose(v), A)`` (dspbase.jl:801-818): 2-D convolution of the matrix ``A`` with the separable kernel
    ``u * transpose(v)`` -- evaluated as the 2-D convolution with that rank-one kernel."""
    if len(u.shape) != 1 or len(v.shape) != 1 or len(A.shape) != 2:
        raise ArgumentError("conv(u, v', A) takes two vectors and a matrix")
    T = util.promote_type(_dev.np_dtype_of(u), _dev.np_dtype_of(v), _dev.np_dtype_of(A))
    W = _compute_dtype(T) if T in _FFT_TYPES else np.dtype(np.float64)
    ud, vd = _dev.as_device(u, W), _dev.as_device(v, W)
    kern = ud.reshape(-1, 1) * vd.reshape(1, -1)
    res = _conv_nd(_dev.as_device(A, W), kern, "fft_simple")
    res = res if res.dtype == _dev.torch_dtype(T) else (_dev.torch.round(res) if T.kind in "iu" else res).to(_dev.torch_dtype(T))
    return res if _dev.is_device_array(A) else res.cpu().numpy()


def conv_(out, u, v, algorithm: str = "auto"):
    """``conv!(out, u, v; algorithm)``: ``out`` at least nu+nv-1 long; the excess is zeroed."""
    res = conv(u, v, algorithm, out_len=int(out.shape[0]))
    if isinstance(out, np.ndarray):
        out[...] = res if isinstance(res, np.ndarray) else res.cpu().numpy()
    else:
        out.copy_(res if not isinstance(res, np.ndarray) else _dev.torch.from_numpy(res))
    return out


def xcorr(u, v=None, padmode: str = "none", scaling: str = "none"):
    """dspbase.jl:867-898 (vectors)."""
    v = u if v is None else v
    if len(u.shape) != 1 or len(v.shape) != 1:
        raise TypeError("xcorr only supports vectors")       # MethodError in the reference (test/dsp.jl:347)
    su, sv = int(u.shape[0]), int(v.shape[0])
    if scaling == "biased" and su != sv:
        raise _lib.DimensionMismatch("scaling only valid for vectors of same length")
    if padmode not in ("none", "longest"):
        raise ArgumentError("padmode keyword argument must be either :none or :longest")
    # device in -> device out, like every other entry (round 4 pulled both operands to the host): padding, conjugation and reversal happen where the
    # operand lives; conv() keeps the longer operand where it is and takes the shorter one as the filter
    tt = _dev.torch

    def prep(a, n, rev):
        if _dev.is_device_array(a):
            a = tt.cat([a, tt.zeros(n - int(a.shape[0]), dtype=a.dtype, device=a.device)]) if n > int(a.shape[0]) else a
            return tt.conj(a).flip(0).resolve_conj().contiguous() if rev else a
        a = np.asarray(a)
        a = np.concatenate([a, np.zeros(n - len(a), dtype=a.dtype)]) if n > len(a) else a
        return np.conj(a)[::-1].copy() if rev else a

    n_u, n_v = (max(su, sv),) * 2 if padmode == "longest" else (su, sv)
    res = conv(prep(u, n_u, False), prep(v, n_v, True))
    return res / su if scaling == "biased" else res


# ---------------------------------------------------------------------------------------------------------
# hilbert (src/util.jl:31-87)
# ---------------------------------------------------------------------------------------------------------
def hilbert(x):
    """``hilbert(x)``: analytic representation ``x + j H{x}`` of a real signal along the first dimension."""
    xdt = _dev.np_dtype_of(x)
    if xdt.kind == "c":
        raise TypeError("hilbert is defined for real signals")             # MethodError in the reference
    from . import util
    S = util.fftintype(xdt)
    cols, shape = _dev.to_columns(x, S)
    ncols, n = cols.shape
    out = _dev.empty_columns(ncols, n, util.fftouttype(S))
    if n and ncols:
        _lib.check(_lib.lib().mdsp_hilbert(_dev.ptr(cols), n, ncols, n, _dev.md_dtype(S), _dev.ptr(out), n, _dev.stream_ptr()))
    return _dev.from_columns(out, shape, x)
