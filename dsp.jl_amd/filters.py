"""FIR application and polyphase resampling: host side of DSP.jl ``src/Filters/filt.jl:426-555`` and
``src/Filters/stream_filt.jl`` over libmi355dsp.
"""
from __future__ import annotations

import ctypes as C
import math
from fractions import Fraction

import numpy as np

from . import _dev, _lib, _plancache, design, dspbase, util
from ._lib import ArgumentError, DomainError, UnsupportedError
from .dspbase import SMALL_FILT_CUTOFF, OlsPlan, _cast_result, _compute_dtype, _host_vec, optimalfftfiltlength

_REAL_KINDS = "fiub"


# ---------------------------------------------------------------------------------------------------------
# tdfilt / fftfilt / filt(b, x)   (Filters/filt.jl:426-555)
# ---------------------------------------------------------------------------------------------------------
def tdfilt(h, x):
    """filt.jl:431-433: naive time-domain FIR = ``filt(h, one(H), x)``."""
    hv = _host_vec(h)
    return dspbase.filt(hv, np.ones(1, dtype=hv.dtype), x)


def tdfilt_(out, h, x):
    """filt.jl:441-443."""
    hv = _host_vec(h)
    return dspbase.filt_(out, hv, np.ones(1, dtype=hv.dtype), x)


def _fftfilt(b: np.ndarray, x, nfft: int, engine: int = _lib.ENGINE_AUTO):
    """filt.jl:479-521 on the device: one overlap-save plan, all columns in one launch sequence."""
    xdt = _dev.np_dtype_of(x)
    if b.dtype.kind not in _REAL_KINDS or xdt.kind not in _REAL_KINDS:
        raise TypeError("fftfilt is defined for real taps and real signals only")      # MethodError in the reference
    W = util.promote_type(b.dtype, xdt)
    Wc = _compute_dtype(W)
    hcols = _dev.host_columns(x, Wc) if W == Wc else None
    if hcols is not None:              # large host array: the library's pinned, chunked H2D || kernel || D2H pipeline, no torch copy
        if nfft < len(b):
            raise ArgumentError("nfft must be at least length(b)")
        taps = b.astype(Wc)
        _dev.device()
        plan = OlsPlan(taps, nfft, hcols.shape[1], _lib.OLS_FILT, engine, cached=True)
        out = plan.exec_host(hcols, hcols.shape[1])
        return out[0] if x.ndim == 1 else out.T
    cols, shape = _dev.to_columns(x, Wc)
    ncols, nx = cols.shape
    if nx == 0 or ncols == 0:
        return _dev.from_columns(_dev.empty_columns(ncols, nx, Wc), shape, x)
    if nfft < len(b):
        raise ArgumentError("nfft must be at least length(b)")
    taps = b.astype(Wc)
    plan = OlsPlan(taps, nfft, nx, _lib.OLS_FILT, engine, cached=True)     # the library's own plan LRU (mdsp_ols_plan_cached); any filter length
    out = plan.exec(cols, nx)
    return _dev.from_columns(_cast_result(out, W) if W.kind in "iu" else out, shape, x)


# the longest filter the fused engine's partitioned kernels take (four partitions of half an in-LDS transform: DESIGN.md 4.4), by element size;
# longer ones run in blocks of 2^20 points on the multi-pass engine inside the same plan (mdsp_ols_plan_geometry)
FUSED_MAX_TAPS = {4: 16384, 8: 8192}


def fftfilt(b, x, nfft: int | None = None, engine: int = _lib.ENGINE_AUTO):
    """``fftfilt(b, x[, nfft])`` (filt.jl:458-461); default nfft = optimalfftfiltlength(length(b), length(x))."""
    bv = _host_vec(b)
    if nfft is None:
        nfft = optimalfftfiltlength(len(bv), int(np.prod(x.shape)))
    return _fftfilt(bv, x, int(nfft), engine)


def fftfilt_(out, b, x, nfft: int | None = None):
    """``fftfilt!(out, b, x[, nfft])`` (filt.jl:468-476)."""
    if tuple(out.shape) != tuple(x.shape):
        raise ArgumentError("out and x must be the same size")
    return _assign(out, fftfilt(b, x, nfft))


def filt(b, x, engine: int = _lib.ENGINE_AUTO):
    """``filt(b, x)`` (filt.jl:445-446, :525-555): FFT overlap-save when both are real and
    length(b) > SMALL_FILT_CUTOFF (66), time domain otherwise."""
    bv = _host_vec(b)
    xdt = _dev.np_dtype_of(x)
    if bv.dtype.kind in _REAL_KINDS and xdt.kind in _REAL_KINDS and len(bv) > SMALL_FILT_CUTOFF:
        return _fftfilt(bv, x, optimalfftfiltlength(len(bv), int(x.shape[0])), engine)
    return tdfilt(bv, x)


def filt_(out, b, x):
    """``filt!(out, b, x)`` (filt.jl:530-533)."""
    if tuple(out.shape) != tuple(x.shape):
        raise ArgumentError("out must be the same size as x")
    return _assign(out, filt(b, x))


def _assign(out, res):
    if isinstance(out, np.ndarray):
        out[...] = res if isinstance(res, np.ndarray) else res.cpu().numpy()
    else:
        out.copy_(res if not isinstance(res, np.ndarray) else _dev.torch.from_numpy(res))
    return out


# ---------------------------------------------------------------------------------------------------------
# FIRFilter (stream_filt.jl:137-210) -- the stateful polyphase filter
# ---------------------------------------------------------------------------------------------------------
def outputlength(inputlength: int, ratio, initial_phi: int) -> int:
    """stream_filt.jl:317-322."""
    r = Fraction(ratio)
    return int(_lib.lib().mdsp_outputlength(int(inputlength), r.numerator, r.denominator, int(initial_phi)))


def inputlength(outputlength_: int, ratio, initial_phi: int, round_up: bool = False) -> int:
    """stream_filt.jl:358-364 (RoundDown default, RoundUp with ``round_up=True``)."""
    r = Fraction(ratio)
    return int(_lib.lib().mdsp_inputlength(int(outputlength_), r.numerator, r.denominator, int(initial_phi), int(round_up)))


class FIRFilter:
    """``FIRFilter(h, ratio=1)``: single-rate / interpolating / decimating / rational polyphase FIR with persistent
    state (``phi_idx``, ``input_deficit``, ``history``), exactly the reference's (stream_filt.jl:8-79, :137-178).

    The kernel object lives in libmi355dsp (``mdsp_fir`` / ``mdsp_firarb``); it is created at the first ``filt`` call,
    when the signal eltype and channel count are known.

    ``FIRFilter(h, rate::float, Nphi=32)`` is the arbitrary-rate resampler (FIRArbitrary, stream_filt.jl:92-156): state
    ``phi_accumulator`` (Float64), ``phi_idx``, ``alpha``, ``input_deficit``, ``x_idx``.
    """

    KINDS = ("FIRStandard", "FIRInterpolator", "FIRDecimator", "FIRRational", "FIRArbitrary")

    def __init__(self, h, ratio=1, Nphi: int = 32, *, exact: bool = False):
        self.exact = bool(exact)     # exact=True: the generic kernel only -- every output reads exactly its own window (mdsp_fir_set_exact)
        self.h = _host_vec(h)
        if self.h.dtype.kind == "c":
            raise UnsupportedError("complex FIR taps are not accelerated")
        if self.h.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            self.h = self.h.astype(np.float64)
        self._handle = None
        self._xdtype = None
        self._nch = None
        self._history_host = None     # pending history to push into a (re)created handle
        if isinstance(ratio, (float, np.floating)):                 # FIRFilter(h, rate::AbstractFloat, Nphi) :150-156
            rate = float(ratio)
            if not rate > 0.0:
                raise DomainError("rate must be greater than 0")
            self.kind = 4
            self.rate = rate
            self.ratio = rate
            self.hLen = len(self.h)
            self.Nphi = int(Nphi)
            if self.Nphi < 1:
                raise ArgumentError("Nphi must be positive")
            self.tapsPerphi = -(-self.hLen // self.Nphi)
            self.historyLen = self.tapsPerphi - 1
            self.delta = self.Nphi / rate                           # :114
            self.phi_accumulator = 0.0
            self.phi_idx = 1
            self.alpha = 0.0
            self.input_deficit = 1
            self.x_idx = 1
            return
        self.ratio = Fraction(ratio)
        if self.ratio <= 0:
            raise DomainError("resampling ratio must be positive")
        L, M = self.ratio.numerator, self.ratio.denominator
        self.kind = 0 if (L == 1 and M == 1) else (1 if M == 1 else (2 if L == 1 else 3))
        self.hLen = len(self.h)
        self.Nphi = L
        self.tapsPerphi = -(-self.hLen // L)
        self.historyLen = self.tapsPerphi - 1
        self.phi_idx = 1
        self.input_deficit = 1

    @classmethod
    def from_ratio(cls, ratio, *args):
        """``FIRFilter(ratio, args...)`` (stream_filt.jl:207-210) / ``FIRFilter(rate::AbstractFloat, Nphi=32, args...)``
        (:159-162): taps from ``resample_filter``."""
        if isinstance(ratio, (float, np.floating)):
            nphi = int(args[0]) if args else 32
            return cls(design.resample_filter(float(ratio), nphi, *args[1:]), float(ratio), nphi)
        return cls(design.resample_filter(Fraction(ratio), *args), Fraction(ratio))

    def _destroy_handle(self):
        if self._handle:
            (_lib.lib().mdsp_firarb_destroy if self.kind == 4 else _lib.lib().mdsp_fir_destroy)(self._handle)
        self._handle = None

    # -- reference state API ------------------------------------------------------------------------------
    @property
    def kernel(self) -> str:
        return self.KINDS[self.kind]

    def reset(self):
        """``reset!`` (stream_filt.jl:247-276)."""
        self.phi_idx = 1
        self.input_deficit = 1
        self._history_host = None
        if self.kind == 4:                                           # :260-267
            self.phi_accumulator = 0.0
            self.alpha = 0.0
            self.x_idx = 1
            if self._handle:
                _lib.check(_lib.lib().mdsp_firarb_reset(self._handle))
            return self
        if self._handle:
            _lib.check(_lib.lib().mdsp_fir_reset(self._handle))
        return self

    def timedelay(self) -> float:
        """stream_filt.jl:400-403."""
        if self.kind in (1, 3, 4):
            return (self.hLen - 1) / (2 * self.Nphi)
        return (self.hLen - 1) / 2

    def setphase(self, phi: float):
        """``setphase!`` (stream_filt.jl:216-241); ``round`` is round-half-even as in Julia."""
        if not phi >= 0:
            raise DomainError("phi must be >= 0")
        if self.kind == 0:
            raise TypeError("setphase! has no method for FIRStandard")
        if self.kind == 4:                                           # :231-239
            frac, throwaway = math.modf(phi)
            self.input_deficit += round(throwaway)
            self.phi_accumulator = frac * self.Nphi
            self.phi_idx = 1 + math.floor(self.phi_accumulator)
            self.alpha = math.modf(self.phi_accumulator)[0]
            return
        if self.kind == 2:
            self.input_deficit += round(phi)
        else:
            throwaway, idx = divmod(round(phi * self.Nphi), self.Nphi)
            self.input_deficit += throwaway
            self.phi_idx = idx + 1

    def outputlength(self, inputlength_: int) -> int:
        """stream_filt.jl:324-342."""
        if self.kind == 0:
            return int(inputlength_)
        if self.kind == 4:                                           # :340-342 (no fused multiply-add: Python floats)
            return math.ceil((inputlength_ - self.input_deficit + 1) * self.rate - self.phi_accumulator / self.delta)
        return outputlength(inputlength_ - self.input_deficit + 1, self.ratio, 1 if self.kind == 2 else self.phi_idx)

    def inputlength(self, outputlength_: int, round_up: bool = False) -> int:
        """stream_filt.jl:366-389."""
        if self.kind == 0:
            return int(outputlength_)
        if self.kind == 4:                                           # :385-389
            d = 1 if round_up else 0
            return math.floor((outputlength_ - d + self.phi_accumulator / self.delta) / self.rate) + d + self.input_deficit - 1
        return inputlength(outputlength_, self.ratio, 1 if self.kind == 2 else self.phi_idx, round_up) + self.input_deficit - 1

    @property
    def history(self) -> np.ndarray:
        """The reference's ``history`` vector(s): shape (historyLen,) or (historyLen, nch)."""
        if self._handle is None:
            return np.zeros(self.historyLen) if self._history_host is None else self._history_host
        buf = np.zeros((self._nch, max(self.historyLen, 0)), dtype=self._xdtype)
        if self.historyLen > 0:
            if self.kind == 4:
                _lib.check(_lib.lib().mdsp_firarb_get_state(self._handle, None, None, None, None, None, buf.ctypes.data_as(C.c_void_p)))
            else:
                _lib.check(_lib.lib().mdsp_fir_get_state(self._handle, None, None, buf.ctypes.data_as(C.c_void_p)))
        return buf[0].copy() if self._nch == 1 else buf.T.copy()

    # -- device handle -------------------------------------------------------------------------------------
    def _ensure(self, xdtype: np.dtype, nch: int):
        if self._handle is not None and (self._xdtype != xdtype or self._nch != nch):
            hist = self.history
            self._destroy_handle()
            self._history_host = hist.astype(xdtype) if (hist.ndim == 1 and nch == 1) or (hist.ndim == 2 and hist.shape[1] == nch) else None
        if self._handle is None:
            h = C.c_void_p()
            taps = np.ascontiguousarray(self.h)
            od = C.c_int()
            if self.kind == 4:
                _lib.check(_lib.lib().mdsp_firarb_create(C.byref(h), taps.ctypes.data_as(C.c_void_p), len(taps), self.rate, self.Nphi,
                                                         _dev.md_dtype(taps.dtype), _dev.md_dtype(xdtype), nch))
                _lib.check(_lib.lib().mdsp_firarb_info(h, None, None, None, C.byref(od), None))
            else:
                _lib.check(_lib.lib().mdsp_fir_create(C.byref(h), taps.ctypes.data_as(C.c_void_p), len(taps), self.ratio.numerator,
                                                      self.ratio.denominator, _dev.md_dtype(taps.dtype), _dev.md_dtype(xdtype), nch))
                _lib.check(_lib.lib().mdsp_fir_info(h, None, None, None, None, None, C.byref(od)))
                if self.exact:
                    _lib.check(_lib.lib().mdsp_fir_set_exact(h, 1))
            self._handle, self._xdtype, self._nch = h, np.dtype(xdtype), nch
            self._outdtype = {_lib.F32: np.float32, _lib.F64: np.float64, _lib.C32: np.complex64, _lib.C64: np.complex128}[od.value]
            if self._history_host is not None and self.historyLen > 0:
                hh = np.ascontiguousarray(np.atleast_2d(self._history_host.T if self._history_host.ndim == 2 else self._history_host), dtype=xdtype)
                if self.kind == 4:
                    _lib.check(_lib.lib().mdsp_firarb_set_state(h, self.phi_accumulator, self.input_deficit, hh.ctypes.data_as(C.c_void_p)))
                else:
                    _lib.check(_lib.lib().mdsp_fir_set_state(h, self.phi_idx, self.input_deficit, hh.ctypes.data_as(C.c_void_p)))
            self._history_host = None

    def filt(self, x):
        """``filt(self, x)`` (stream_filt.jl:627-637): filter the next chunk, carrying state.  ``x`` is (n,) or
        (n, channels); all channels share (phi_idx, input_deficit) and keep their own history."""
        xdt = _dev.np_dtype_of(x)
        W = _compute_dtype(xdt)
        hcols = _dev.host_columns(x, W) if self.kind != 4 else None
        if hcols is not None:              # large host array: mdsp_fir_exec_host (time chunks through the stateful filter, H2D || kernel || D2H)
            nch, xlen = hcols.shape
            self._ensure(W, max(nch, 1))
            _lib.check(_lib.lib().mdsp_fir_set_state(self._handle, self.phi_idx, self.input_deficit, None))
            ycap = max(self.outputlength(xlen), 0) if xlen >= self.input_deficit or self.kind == 0 else 0
            outh = np.empty((nch, ycap), dtype=self._outdtype)
            nw = C.c_int64(0)
            _lib.check(_lib.lib().mdsp_fir_exec_host(self._handle, hcols.ctypes.data_as(C.c_void_p), xlen, xlen, outh.ctypes.data_as(C.c_void_p), ycap,
                                                     max(ycap, 1), C.byref(nw), 0))
            phi, dfc = C.c_int64(), C.c_int64()
            _lib.check(_lib.lib().mdsp_fir_get_state(self._handle, C.byref(phi), C.byref(dfc), None))
            self.phi_idx, self.input_deficit = phi.value, dfc.value
            if nw.value != ycap:
                raise AssertionError("Length of resampled output different from expectation.")     # stream_filt.jl:634
            return outh[0] if x.ndim == 1 else outh.T
        cols, shape = _dev.to_columns(x, W)
        nch, xlen = cols.shape
        self._ensure(W, max(nch, 1))
        if self.kind == 4:
            return self._filt_arbitrary(cols, shape, x, nch, xlen)
        _lib.check(_lib.lib().mdsp_fir_set_state(self._handle, self.phi_idx, self.input_deficit, None))
        ycap = max(self.outputlength(xlen), 0) if xlen >= self.input_deficit or self.kind == 0 else 0
        out = _dev.empty_columns(nch, ycap, self._outdtype)
        nw = C.c_int64(0)
        _lib.check(_lib.lib().mdsp_fir_exec(self._handle, _dev.ptr(cols), xlen, xlen, _dev.ptr(out), ycap, max(ycap, 1), C.byref(nw),
                                            _dev.stream_ptr()))
        phi, dfc = C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().mdsp_fir_get_state(self._handle, C.byref(phi), C.byref(dfc), None))
        self.phi_idx, self.input_deficit = phi.value, dfc.value
        if nw.value != ycap:
            raise AssertionError("Length of resampled output different from expectation.")     # stream_filt.jl:634
        return _dev.from_columns(out, shape, x)

    def _filt_arbitrary(self, cols, shape, x, nch, xlen):
        """``filt(self::FIRFilter{FIRArbitrary}, x)`` (stream_filt.jl:627-637): the buffer holds outputlength + 1 samples
        (allocate_output :639-655) and is resized to the number actually written."""
        L = _lib.lib()
        _lib.check(L.mdsp_firarb_set_state(self._handle, self.phi_accumulator, self.input_deficit, None))
        ycap = max(self.outputlength(xlen), 0) + 1
        out = _dev.empty_columns(nch, ycap, self._outdtype)
        nw = C.c_int64(0)
        _lib.check(L.mdsp_firarb_exec(self._handle, _dev.ptr(cols), xlen, xlen, _dev.ptr(out), ycap, ycap, C.byref(nw), _dev.stream_ptr()))
        acc, al = C.c_double(), C.c_double()
        phi, dfc, xi = C.c_int64(), C.c_int64(), C.c_int64()
        _lib.check(L.mdsp_firarb_get_state(self._handle, C.byref(acc), C.byref(al), C.byref(phi), C.byref(dfc), C.byref(xi), None))
        self.phi_accumulator, self.alpha, self.phi_idx, self.input_deficit, self.x_idx = acc.value, al.value, phi.value, dfc.value, xi.value
        return _dev.from_columns(out[:, :nw.value], (nw.value,) + tuple(shape[1:]), x)

    def __del__(self):
        try:
            self._destroy_handle()
        except Exception:
            pass


def filt_stateless(h, x, ratio=1, Nphi: int = 32):
    """``filt(h, x, ratio)`` (stream_filt.jl:663-666) / ``filt(h, x, rate::AbstractFloat, Nphi=32)`` (:669-672)."""
    return FIRFilter(h, ratio, Nphi).filt(x)


def _undelay(sf: FIRFilter):
    """``undelay!`` (stream_filt.jl:706-714)."""
    if sf.kind != 0:
        sf.setphase(sf.timedelay())


def resample(x, rate, h=None, dims: int | None = None, Nphi: int = 32):
    """``resample(x, rate[, h]; dims)`` (stream_filt.jl:688-775): integer / rational rates (polyphase FIRRational &
    co.) and floating-point rates (``FIRArbitrary`` with ``Nphi`` phases, :692-694, :752-755).

    Vector ``x``: delay-compensated polyphase resampling to ceil(length(x)*rate) samples.  Array ``x`` with
    ``dims``: every slice along ``dims`` is resampled independently (``mapslices``) -- on the device all slices are
    channels of one launch, each starting from the same reset + undelay!-ed state (:768-774).
    """
    # resample() always starts from a reset, undelay!-ed filter (:696, :768-774), so the filter object -- polyphase banks in HBM,
    # history buffers -- is reused across calls with the same taps and rate (LRU, _plancache) instead of being rebuilt each time
    if isinstance(rate, (float, np.floating)):
        rate = float(rate)
        if h is None:
            h = design.resample_filter(rate, Nphi)
        key = ("resample", _plancache.ctx_key(), _plancache.array_key(_host_vec(h)), rate, int(Nphi))
        sf = _plancache.plans.get(key, lambda: FIRFilter(h, rate, Nphi))
    else:
        rate = Fraction(rate)
        if h is None:
            h = design.resample_filter(rate)
        key = ("resample", _plancache.ctx_key(), _plancache.array_key(_host_vec(h)), rate, 0)
        sf = _plancache.plans.get(key, lambda: FIRFilter(h, rate))
    sf.reset()
    _undelay(sf)
    nd = len(x.shape)
    if nd == 1:
        n = int(x.shape[0])
        moved, axis = x, None
    else:
        if dims is None:
            raise TypeError("resample of an array needs the `dims` keyword")
        axis = dims % nd
        moved = (x.movedim(axis, 0) if hasattr(x, "movedim") else np.moveaxis(np.asarray(x), axis, 0))
        n = int(moved.shape[0])
    out_len = math.ceil(n * rate)                                    # :698 / :762 (exact rational, or Float64 product)
    npad = sf.inputlength(out_len, round_up=True)                    # :699 / :763
    if npad < n:
        raise ArgumentError("padded length shorter than the input")   # copyto! would throw in the reference
    xdt = _dev.np_dtype_of(moved)
    cols, shape = _dev.to_columns(moved, _compute_dtype(xdt))
    padded = _dev.torch.zeros((cols.shape[0], npad), dtype=cols.dtype, device=cols.device)
    padded[:, :n] = cols
    y = sf.filt(padded.t())                                          # (npad, nch) view -> filt -> (nout, nch)
    if y.shape[0] < out_len:
        raise AssertionError("Resample output shorter than expected.")   # :722
    y = y[:out_len]
    y = y.reshape((out_len,) + tuple(shape[1:]))
    if axis is not None:
        y = y.movedim(0, axis)
    if nd == 1:
        y = y.reshape(out_len)
    return y if _dev.is_device_array(x) else y.cpu().numpy()


# ---------------------------------------------------------------------------------------------------------
# Stateful FIR filtering (DF2TFilter) and zero-phase FIR filtering (filtfilt)   Filters/filt.jl:122-181, :296-338
# ---------------------------------------------------------------------------------------------------------
class DF2TFilter:
    """``DF2TFilter(PolynomialRatio(b, [1]), [T], coldims=())`` for FIR coefficients: a filter object carrying the
    TDF-II state ``state`` of shape ``(length(b) - 1, coldims...)`` between calls (filt.jl:122-181).

    ``DF2TFilter(b)`` / ``DF2TFilter(b, a)`` with scalar or length-1 ``a``; an IIR ``a`` is a serial recursion and is
    left to DSP.jl's CPU code (``UnsupportedError``)."""

    def __init__(self, b, a=1.0, dtype=None, coldims=()):
        bv, av = _host_vec(b), _host_vec(a)
        if bv.size == 0 or av.size == 0:
            raise ArgumentError("filter coefficients must be non-empty")
        if av.size > 1:
            raise UnsupportedError("IIR DF2TFilter is a serial recursion; only FIR coefficients run on the device")
        if av[0] == 0:
            raise ArgumentError("filter vector a[1] must be nonzero")
        if bv.dtype.kind == "c":
            raise UnsupportedError("complex FIR taps are not accelerated")
        if av[0] != 1:
            bv = bv / av[0]                                             # PolynomialRatio normalises by a[1]
        self.b = bv if bv.dtype in (np.dtype(np.float32), np.dtype(np.float64)) else bv.astype(np.float64)
        T = util.promote_type(self.b.dtype, dtype) if dtype is not None else self.b.dtype   # zeros(promote_type(T, V), ...), :150
        self._T = _compute_dtype(T)
        self.state = _dev.torch.zeros((len(self.b) - 1,) + tuple(coldims), dtype=_dev.torch_dtype(self._T), device=_dev.device())

    def filt(self, x):
        """``filt(f, x)`` = ``filt!(similar(x, promote_type(...)), f, x)`` (filt.jl:153-181)."""
        if tuple(x.shape[1:]) != tuple(self.state.shape[1:]):
            raise ArgumentError("state size must match x")              # :158
        xdt = _dev.np_dtype_of(x)
        W = _compute_dtype(util.promote_type(self.b.dtype, xdt, self._T))
        cols, shape = _dev.to_columns(x, W)
        ncols, nx = cols.shape
        nb = len(self.b)
        if nb == 1:                                                     # mul!(out, x, b[1]), :163
            return _dev.from_columns(cols * float(self.b[0]), shape, x)
        if self._T != W:                                                # a wider signal eltype widens the state for good
            self.state = self.state.to(_dev.torch_dtype(W))
            self._T = W
        out = _dev.empty_columns(ncols, nx, W)
        if nx and ncols:
            si = self.state.reshape(nb - 1, -1).t().contiguous()        # (ncols, nb-1): one register file per column
            taps = np.ascontiguousarray(self.b, dtype=np.float32 if W in (np.dtype(np.float32), np.dtype(np.complex64)) else np.float64)
            _lib.check(_lib.lib().mdsp_tdfir_state_exec(taps.ctypes.data_as(C.c_void_p), nb, _dev.md_dtype(W), _dev.ptr(cols), nx, ncols, nx,
                                                        _dev.ptr(out), nx, _dev.ptr(si), _dev.stream_ptr()))
            self.state = si.t().reshape(self.state.shape).contiguous()
        return _dev.from_columns(out, shape, x)


def filtfilt(b, *args):
    """``filtfilt(b, x)`` / ``filtfilt(b, a, x)`` with scalar or length-1 ``a`` (filt.jl:301-338): zero-phase FIR
    filtering -- odd-symmetric extension by ``length(b)-1`` samples, one pass with ``conv(b, reverse(b))``, trim."""
    if len(args) == 2:
        a, x = args
        av = _host_vec(a)
        if av.size != 1:
            raise UnsupportedError("IIR filtfilt is a serial recursion; only FIR coefficients run on the device")
        bv = _host_vec(b)
        if av[0] != 1:
            bv = bv / av[0]                                             # :331-333
    elif len(args) == 1:
        x = args[0]
        bv = _host_vec(b)
    else:
        raise TypeError("filtfilt(b, x) or filtfilt(b, a, x)")
    if bv.dtype.kind == "c":
        raise UnsupportedError("complex FIR taps are not accelerated")
    nb = len(bv)
    n = int(x.shape[0])
    if nb - 1 > n - 1:
        raise ArgumentError("the signal must be longer than the filter order")   # sig[2 + pad_length - i] is a BoundsError in the reference
    # newb = conv(b, reverse(b)), built as the reference builds it: causal half by filt!, mirrored (:309-314)
    bw = bv.astype(np.float64) if bv.dtype.kind != "f" else bv
    rev = bw[::-1].copy()
    half = np.array([np.dot(bw[:k + 1][::-1], rev[:k + 1]) for k in range(nb)], dtype=bw.dtype)
    newb = np.concatenate([half, half[nb - 2::-1] if nb > 1 else half[:0]])
    xdt = _dev.np_dtype_of(x)
    W = _compute_dtype(util.promote_type(bw.dtype, xdt))
    cols, shape = _dev.to_columns(x, W)
    ncols = cols.shape[0]
    ext = _dev.empty_columns(ncols, n + 2 * (nb - 1), W)
    if ncols and n:
        _lib.check(_lib.lib().mdsp_extrapolate(_dev.ptr(cols), n, ncols, n, _dev.md_dtype(W), nb - 1, _dev.ptr(ext), n + 2 * (nb - 1),
                                               _dev.stream_ptr()))
    y = filt(newb, ext.t())                                             # filt!(extrapolated, newb, extrapolated), :322
    y = y[2 * nb - 2:]                                                  # drop garbage at start, :325
    res = y.reshape((n,) + tuple(shape[1:]))
    return res if _dev.is_device_array(x) else res.cpu().numpy()
