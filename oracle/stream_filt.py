"""Oracle for ``src/Filters/stream_filt.jl``: stateful polyphase FIRFilter and ``resample``.

Test infrastructure only (see package docstring).  State (``phi_idx``, ``input_deficit``, ``history``)
is kept 1-based exactly as the reference keeps it, so it can be compared bit-for-bit with the
device-side state after every chunk.
"""
from __future__ import annotations

import math
from fractions import Fraction

import numpy as np

from . import design
from .util import shiftin, unsafe_dot_mat, unsafe_dot_mat_hist, unsafe_dot_vec, unsafe_dot_vec_hist


def taps2pfb(h: np.ndarray, nphi: int) -> np.ndarray:
    """stream_filt.jl:294-307: (tapsPerPhi x Nphi), each column a flipped phase filter."""
    h = np.asarray(h)
    hlen = len(h)
    tpp = math.ceil(hlen / nphi)
    pfb = np.zeros((tpp, nphi), dtype=h.dtype)
    hidx = 0
    for row in range(tpp - 1, -1, -1):
        for col in range(nphi):
            pfb[row, col] = h[hidx] if hidx < hlen else 0
            hidx += 1
    return pfb


def outputlength_ratio(inputlength: int, ratio: Fraction, initial_phi: int) -> int:
    """stream_filt.jl:317-322 (float division then ceil, as the reference)."""
    out_len = ((inputlength * ratio.numerator) - initial_phi + 1) / ratio.denominator
    return math.ceil(out_len)


def inputlength_ratio(outputlength: int, ratio: Fraction, initial_phi: int, roundup: bool = False) -> int:
    """stream_filt.jl:358-364."""
    d = ratio.denominator if roundup else 1
    in_len = (outputlength * ratio.denominator + initial_phi - d) / ratio.numerator
    return math.ceil(in_len) if roundup else math.floor(in_len)


class FIRFilter:
    """stream_filt.jl:137-210 with kernels FIRStandard/FIRInterpolator/FIRDecimator/FIRRational (:8-79)."""

    def __init__(self, h, ratio=1, nphi: int = 32):
        h = np.asarray(h)
        if isinstance(ratio, (float, np.floating)):
            self._init_arbitrary(h, float(ratio), int(nphi))
            return
        ratio = Fraction(ratio)
        self.h = h
        self.ratio = ratio
        L, M = ratio.numerator, ratio.denominator
        self.hlen = len(h)
        self.phi_idx = 1
        self.input_deficit = 1
        if ratio == 1:
            self.kind = "standard"
            self.hrev = h[::-1].copy()
            self.history_len = self.hlen - 1
            self.nphi = 1
        elif M == 1:
            self.kind = "interpolator"
            self.pfb = taps2pfb(h, L)
            self.taps_per_phi, self.nphi = self.pfb.shape
            self.history_len = self.taps_per_phi - 1
        elif L == 1:
            self.kind = "decimator"
            self.hrev = h[::-1].copy()
            self.history_len = self.hlen - 1
            self.nphi = 1
        else:
            self.kind = "rational"
            self.pfb = taps2pfb(h, L)
            self.taps_per_phi, self.nphi = self.pfb.shape
            self.phi_step = M % L                                   # :73
            self.history_len = self.taps_per_phi - 1
        self.history = np.zeros(self.history_len, dtype=np.float64)  # :175

    def _init_arbitrary(self, h, rate: float, nphi: int):
        """FIRArbitrary(h, rate, Nphi) stream_filt.jl:106-134 and FIRFilter(h, rate::AbstractFloat, Nphi) :150-156."""
        if not rate > 0.0:
            raise ValueError("DomainError: rate must be greater than 0")      # :151
        self.kind = "arbitrary"
        self.h = h
        self.rate = rate
        self.ratio = rate
        self.hlen = len(h)
        dh = np.concatenate([np.diff(h), np.zeros(1, dtype=h.dtype)]).astype(h.dtype)   # :107 (in eltype(h))
        self.pfb = taps2pfb(h, nphi)                  # :108
        self.dpfb = taps2pfb(dh, nphi)                # :109
        self.taps_per_phi, self.nphi = self.pfb.shape
        self.phi_acc = 0.0
        self.phi_idx = 1
        self.alpha = 0.0
        self.delta = nphi / rate                      # :114
        self.input_deficit = 1
        self.x_idx = 1
        self.history_len = self.taps_per_phi - 1      # :153
        self.history = np.zeros(self.history_len, dtype=np.float64)

    def _arb_update(self):
        """update! stream_filt.jl:567-577."""
        self.phi_acc += self.delta
        if self.phi_acc >= self.nphi:
            dx, self.phi_acc = divmod(self.phi_acc, float(self.nphi))   # both operands positive: divrem == divmod
            self.x_idx += int(dx)
        foffset = math.floor(self.phi_acc)
        self.alpha = self.phi_acc - foffset           # modf: exact fractional part
        self.phi_idx = 1 + int(foffset)

    # -- state -------------------------------------------------------------------------------
    def reset(self):
        """stream_filt.jl:247-276."""
        self.history = np.zeros(self.history_len, dtype=self.history.dtype)
        self.phi_idx = 1
        self.input_deficit = 1
        if self.kind == "arbitrary":                  # :260-267
            self.phi_acc = 0.0
            self.alpha = 0.0
            self.x_idx = 1
        return self

    def timedelay(self) -> float:
        """stream_filt.jl:400-403."""
        if self.kind in ("rational", "interpolator", "arbitrary"):
            return (self.hlen - 1) / (2 * self.nphi)
        return (self.hlen - 1) / 2

    def setphase(self, phi: float):
        """stream_filt.jl:216-241 (``round`` is round-half-even in both Julia and Python)."""
        if not phi >= 0:
            raise ValueError("DomainError: phi must be >= 0")
        if self.kind == "arbitrary":                  # :231-239
            frac, throwaway = math.modf(phi)
            self.input_deficit += round(throwaway)
            self.phi_acc = frac * self.nphi
            self.phi_idx = 1 + math.floor(self.phi_acc)
            self.alpha = math.modf(self.phi_acc)[0]
            return
        if self.kind == "decimator":
            self.input_deficit += round(phi)
        elif self.kind in ("interpolator", "rational"):
            q = round(phi * self.nphi)
            throwaway, idx = divmod(q, self.nphi)
            self.input_deficit += throwaway
            self.phi_idx = idx + 1
        else:
            raise TypeError("setphase! is not defined for FIRStandard")

    # -- lengths -----------------------------------------------------------------------------
    def outputlength(self, inputlength: int) -> int:
        """stream_filt.jl:324-342."""
        if self.kind == "standard":
            return inputlength
        if self.kind == "arbitrary":                  # :340-342
            return math.ceil((inputlength - self.input_deficit + 1) * self.rate - self.phi_acc / self.delta)
        if self.kind == "interpolator":
            return outputlength_ratio(inputlength - self.input_deficit + 1, Fraction(self.ratio.numerator), self.phi_idx)
        if self.kind == "decimator":
            return outputlength_ratio(inputlength - self.input_deficit + 1, Fraction(1, self.ratio.denominator), 1)
        return outputlength_ratio(inputlength - self.input_deficit + 1, self.ratio, self.phi_idx)

    def inputlength(self, outputlength: int, roundup: bool = False) -> int:
        """stream_filt.jl:366-389."""
        if self.kind == "standard":
            return outputlength
        if self.kind == "arbitrary":                  # :385-389
            d = 1 if roundup else 0
            n = math.floor((outputlength - d + self.phi_acc / self.delta) / self.rate) + d
            return n + self.input_deficit - 1
        if self.kind == "interpolator":
            n = inputlength_ratio(outputlength, Fraction(self.ratio.numerator), self.phi_idx, roundup)
        elif self.kind == "decimator":
            n = inputlength_ratio(outputlength, Fraction(1, self.ratio.denominator), 1, roundup)
        else:
            n = inputlength_ratio(outputlength, self.ratio, self.phi_idx, roundup)
        return n + self.input_deficit - 1

    # -- filtering ---------------------------------------------------------------------------
    def _out_dtype(self, x):
        return np.result_type(self.h.dtype, x.dtype)

    def filt(self, x) -> np.ndarray:
        """stream_filt.jl:627-637 (+ per-kernel ``filt!`` :409-558)."""
        x = np.asarray(x)
        if x.dtype.kind in "iu":
            pass
        hist = self.history.astype(x.dtype)          # ``history::Vector{Tx} = self.history``
        xlen = len(x)
        T = self._out_dtype(x)
        if self.kind == "standard":
            buf = np.empty(xlen, dtype=T)
            for i in range(1, min(self.hlen - 1, xlen) + 1):
                buf[i - 1] = unsafe_dot_vec_hist(self.hrev, hist, x, i)
            for i in range(self.hlen, xlen + 1):
                buf[i - 1] = unsafe_dot_vec(self.hrev, x, i)
            self.history = shiftin(hist, x)
            return buf
        if xlen < self.input_deficit:                 # :483-487 etc.
            self.history = shiftin(hist, x)
            self.input_deficit -= xlen
            return np.empty(0, dtype=T)
        if self.kind == "arbitrary":                  # filt! :579-625, filt :627-637 (resize! to samplesWritten)
            self.x_idx = self.input_deficit
            out = []
            wide = np.complex128 if np.dtype(T).kind == "c" else np.float64
            while self.x_idx <= xlen:
                if self.x_idx < self.taps_per_phi:
                    ylo = unsafe_dot_mat_hist(self.pfb, self.phi_idx, hist, x, self.x_idx)
                    yup = unsafe_dot_mat_hist(self.dpfb, self.phi_idx, hist, x, self.x_idx)
                else:
                    ylo = unsafe_dot_mat(self.pfb, self.phi_idx, x, self.x_idx)
                    yup = unsafe_dot_mat(self.dpfb, self.phi_idx, x, self.x_idx)
                out.append(wide(yup) * self.alpha + wide(ylo))      # muladd(yUpper, alpha::Float64, yLower) :616
                self._arb_update()
            self.input_deficit = self.x_idx - xlen
            self.history = shiftin(hist, x)
            return np.asarray(out, dtype=wide).astype(T) if out else np.empty(0, dtype=T)
        out_len = self.outputlength(xlen)
        buf = np.empty(out_len, dtype=T)
        buf_idx = 0
        input_idx = self.input_deficit
        L, M = self.ratio.numerator, self.ratio.denominator
        if self.kind == "interpolator":               # :435-469
            while input_idx <= xlen:
                if input_idx < self.taps_per_phi:
                    acc = unsafe_dot_mat_hist(self.pfb, self.phi_idx, hist, x, input_idx)
                else:
                    acc = unsafe_dot_mat(self.pfb, self.phi_idx, x, input_idx)
                buf[buf_idx] = acc
                buf_idx += 1
                if self.phi_idx == self.nphi:
                    self.phi_idx, input_idx = 1, input_idx + 1
                else:
                    self.phi_idx += 1
            self.input_deficit = 1
        elif self.kind == "rational":                 # :476-515
            while input_idx <= xlen:
                if input_idx < self.taps_per_phi:
                    acc = unsafe_dot_mat_hist(self.pfb, self.phi_idx, hist, x, input_idx)
                else:
                    acc = unsafe_dot_mat(self.pfb, self.phi_idx, x, input_idx)
                buf[buf_idx] = acc
                buf_idx += 1
                input_idx += (self.phi_idx + M - 1) // L
                phi = self.phi_idx + self.phi_step
                self.phi_idx = phi - L if phi > L else phi
            self.input_deficit = input_idx - xlen
        else:                                          # decimator :522-558
            while input_idx <= xlen:
                if input_idx < self.hlen:
                    acc = unsafe_dot_vec_hist(self.hrev, hist, x, input_idx)
                else:
                    acc = unsafe_dot_vec(self.hrev, x, input_idx)
                buf[buf_idx] = acc
                buf_idx += 1
                input_idx += M
            self.input_deficit = input_idx - xlen
        self.history = shiftin(hist, x)
        if buf_idx != out_len:
            raise AssertionError("Length of resampled output different from expectation.")   # :634
        return buf


def filt_stateless(h, x, ratio=1, nphi: int = 32):
    """stream_filt.jl:663-672."""
    return FIRFilter(h, ratio, nphi).filt(x)


def arb_trajectory(phi_acc: float, input_deficit: int, delta: float, nphi: int, xlen: int):
    """Index trajectory of filt!(buffer, ::FIRFilter{FIRArbitrary}, x) (stream_filt.jl:593-622) without the dot
    products: returns (x_idx[], phi_acc[]) of every output plus the final (phi_acc, input_deficit)."""
    xs, accs = [], []
    if xlen < input_deficit:
        return np.zeros(0, np.int64), np.zeros(0), phi_acc, input_deficit - xlen
    x_idx = input_deficit
    fn = float(nphi)
    while x_idx <= xlen:
        xs.append(x_idx)
        accs.append(phi_acc)
        phi_acc += delta
        if phi_acc >= fn:
            dx, phi_acc = divmod(phi_acc, fn)
            x_idx += int(dx)
    return np.asarray(xs, np.int64), np.asarray(accs), phi_acc, x_idx - xlen


def resample_arbitrary(x, rate: float, h=None, nphi: int = 32):
    """stream_filt.jl:692-704, 717-725 for an AbstractFloat rate."""
    x = np.asarray(x)
    rate = float(rate)
    if h is None:
        h = design.resample_filter(rate, nphi)
    sf = FIRFilter(h, rate, nphi)
    sf.setphase(sf.timedelay())                                      # undelay!
    out_len = math.ceil(len(x) * rate)
    npad = sf.inputlength(out_len, roundup=True)
    xp = np.zeros(npad, dtype=x.dtype)
    xp[:min(len(x), npad)] = x[:npad]
    y = sf.filt(xp)
    if not len(y) >= out_len:
        raise AssertionError("Resample output shorter than expected.")
    return y[:out_len]


def resample(x, rate, h=None, nphi: int = 32):
    """stream_filt.jl:688-725 for vectors; ``rate`` int / Fraction (rational kernels) or float (FIRArbitrary)."""
    x = np.asarray(x)
    if isinstance(rate, (float, np.floating)):
        return resample_arbitrary(x, float(rate), h, nphi)
    rate = Fraction(rate)
    if h is None:
        h = design.resample_filter(rate)
    sf = FIRFilter(h, rate)
    if sf.kind != "standard":
        sf.setphase(sf.timedelay())                                  # undelay! :706-714
    out_len = math.ceil(len(x) * rate)                               # :698 (exact rational arithmetic)
    npad = sf.inputlength(out_len, roundup=True)
    xp = np.zeros(npad, dtype=x.dtype)                               # _zeropad :699
    xp[:min(len(x), npad)] = x[:npad]
    y = sf.filt(xp)
    if not len(y) >= out_len:
        raise AssertionError("Resample output shorter than expected.")  # :722
    return y[:out_len]


def resample_dims(x, rate, h=None, dims: int = 0):
    """stream_filt.jl:747-775: ``mapslices`` over ``dims`` (0-based axis here)."""
    x = np.asarray(x)
    return np.apply_along_axis(lambda v: resample(v, rate, h), dims, x)


def polyphase_closed_form(phi_idx0: int, input_deficit0: int, L: int, M: int, m: np.ndarray):
    """SURVEY 3.5 closed form of the recurrence at stream_filt.jl:506-508, for 0-based output index m:
    p = (phi0-1) + m*M;  phi_m = p mod L + 1;  inputIdx_m = inputDeficit0 + p div L  (all 1-based)."""
    p = (phi_idx0 - 1) + np.asarray(m, dtype=np.int64) * M
    return p % L + 1, input_deficit0 + p // L
