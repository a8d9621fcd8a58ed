"""The algebra behind the matrix-core polyphase kernel, replayed in numpy on the host: rows of RB rounds, blocks of 16 output columns,
H[k][j] = pfb[phase_j][k - delta_j], X[row][k] = z[row Mr + c_block + k], Y = X H -- with the geometry the library itself chooses
(mdsp_fir_mm_geometry, no device needed) -- against the oracle's FIRFilter (stream_filt.jl:476-515 restated).  The GPU tests compare the
kernel with the oracle; this one pins the index arithmetic the kernel implements (dsp.jl_amd/csrc/fir.hip, polyphase_mfma_kernel)."""
import ctypes as C
from fractions import Fraction
from math import gcd

import numpy as np
import pytest

from dsp_jl_amd import _lib
from oracle import stream_filt as osf


def matrix_form(h, L, M, x, phi_idx=1, deficit=1, tdt=None, xdt=None):
    """Outputs of FIRFilter(h, L//M) on x (zero history, state (phi_idx, deficit)) computed the way the kernel does."""
    g0 = gcd(L, M); L //= g0; M //= g0
    out = (C.c_int64 * 12)()
    _lib.check(_lib.lib().mdsp_fir_mm_geometry(L, M, len(h), _lib.F64 if tdt is None else tdt, _lib.F64 if xdt is None else xdt, out))
    ok, RB, Lr, Mr, NB, NG, steps = list(out)[:7]
    assert ok
    tp = -(-len(h) // L)
    hp = np.concatenate([np.asarray(h, np.float64), np.zeros(tp * L - len(h))])
    pfbT = np.empty((tp, L))                      # pfbT[i][c] = h[(tp - 1 - i) L + c]: row i multiplies the i-th oldest sample of a window
    for i in range(tp):
        pfbT[i] = hp[(tp - 1 - i) * L:(tp - i) * L]
    hl = tp - 1
    phi0m1, d0 = (0 if L == 1 else phi_idx - 1), deficit
    if len(x) < d0:
        return np.zeros(0)
    nout = -(-((len(x) - d0 + 1) * L - phi0m1) // M)            # outputlength (stream_filt.jl:324-338) in closed form
    nrows = -(-nout // Lr)
    K = 4 * steps
    z = np.concatenate([np.zeros(hl), np.asarray(x, np.float64), np.zeros(nrows * Mr + K + Mr + 8)])   # [history ; x ; zero fill]
    cbase = d0 - 1
    y = np.zeros(nrows * Lr)
    for b in range(NB):
        cols = np.arange(16 * b, min(16 * b + 16, Lr))
        p = phi0m1 + cols * M
        c, phase = p // L, p % L
        c0 = (phi0m1 + 16 * b * M) // L
        delta = c - c0
        assert delta.min() >= 0 and tp + delta.max() <= K                           # every tap of every column has a k-step
        H = np.zeros((K, len(cols)))
        for jj, (dl, ph) in enumerate(zip(delta, phase)):
            H[dl:dl + tp, jj] = pfbT[:, ph]
        rows = np.arange(nrows)
        X = z[(rows * Mr + cbase + c0)[:, None] + np.arange(K)[None, :]]
        y.reshape(nrows, Lr)[:, cols] = X @ H
    return y[:nout]


@pytest.mark.parametrize("L,M,ntaps", [(160, 147, 5120), (147, 160, 5881), (2, 1, 49), (1, 2, 31), (3, 2, 73), (7, 4, 120), (1, 1, 33), (23, 17, 300),
                                       (250, 249, 4000), (1, 8, 293), (5, 3, 101), (16, 9, 129)])
def test_matrix_form_equals_the_oracle(L, M, ntaps):
    rng = np.random.default_rng(L * 1000 + M)
    h = rng.standard_normal(ntaps)
    x = rng.standard_normal(3000)
    ref = osf.FIRFilter(h, Fraction(L, M)).filt(x)
    got = matrix_form(h, L, M, x)
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref)))


@pytest.mark.parametrize("L,M,ntaps,types", [(1, 16, 583, "c64"), (160, 441, 16001, "c32"), (1, 32, 1100, "f64"), (1, 32, 1100, "c64")])
def test_matrix_form_of_the_last_resort_tiles(L, M, ntaps, types):
    # round 4: the geometries tried when nothing else fits the LDS (rows of fewer rounds; 40 k-steps in Float32) obey the same algebra
    tdt, xdt = {"c64": (_lib.F64, _lib.C64), "c32": (_lib.F32, _lib.C32), "f64": (_lib.F64, _lib.F64)}[types]
    rng = np.random.default_rng(L * 1000 + M)
    h = rng.standard_normal(ntaps)
    x = rng.standard_normal(20000)
    ref = osf.FIRFilter(h, Fraction(L, M)).filt(x)
    got = matrix_form(h, L, M, x, tdt=tdt, xdt=xdt)
    assert got.shape == ref.shape
    assert np.max(np.abs(got - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref)))


def test_matrix_form_from_a_mid_stream_state():
    # a filter that has already consumed a chunk: phase and input deficit of the second chunk come from the first (the history is then
    # the tail of the first chunk; the matrix form reads [history ; x], so feed it the concatenation and compare the tail of the outputs)
    rng = np.random.default_rng(7)
    L, M = 160, 147
    h = rng.standard_normal(5120)
    x = rng.standard_normal(5000)
    f = osf.FIRFilter(h, Fraction(L, M))
    y1 = f.filt(x[:1777]); y2 = f.filt(x[1777:])
    full = matrix_form(h, L, M, x)
    assert len(full) == len(y1) + len(y2)
    assert np.max(np.abs(full - np.concatenate([y1, y2]))) <= 1e-12 * np.max(np.abs(full))


def test_matrix_form_random_shapes():
    rng = np.random.default_rng(2026)
    done = 0
    out = (C.c_int64 * 12)()
    while done < 40:
        L, M = int(rng.integers(1, 300)), int(rng.integers(1, 200))
        tp = int(rng.integers(1, 90))
        g0 = gcd(L, M)
        ntaps = max(1, tp * (L // g0) - int(rng.integers(0, L // g0)))
        _lib.check(_lib.lib().mdsp_fir_mm_geometry(L, M, ntaps, _lib.F64, _lib.F64, out))
        if not out[0]:
            continue
        h = rng.standard_normal(ntaps)
        x = rng.standard_normal(int(rng.integers(1, 2500)))
        ref = osf.FIRFilter(h, Fraction(L, M)).filt(x)
        got = matrix_form(h, L, M, x)
        assert got.shape == ref.shape, (L, M, ntaps, len(x))
        if len(ref):
            assert np.max(np.abs(got - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref))), (L, M, ntaps, len(x))
        done += 1


def test_padded_run_addressing_model():
    """Round 3, padded runs (fir.hip, polyphase_mfma_kernel<..., RP = true>): a tile is staged as one run of 256-dword DMA granules with PAD dwords
    behind every granule (dword d of the tile at d + PAD (d >> 8)); a lane reads its window positions through ONE of two base pointers -- the
    window's start, or PAD further on from the step at which the window crosses a granule boundary -- with immediate offsets 4 t DW.  Replayed
    here for random geometries: every read lands on the sample the plain run would have read, no read lands on a pad, the crossing step is the
    same for every 16-row chunk of a wave (16 rows are a whole number of granules), and the 16 rows of an A operand spread over the banks."""
    rng = np.random.default_rng(20260926)
    PAD = 4
    for _ in range(400):
        DW = int(rng.choice([1, 2, 4]))                       # dwords per sample: Float32, Float64 / ComplexF32, ComplexF64
        Mr = int(rng.integers(1, 64)) * (16 // DW if DW < 16 else 1)
        if (Mr * DW) % 16:
            continue                                          # the form applies where 16 rows are a whole number of granules
        T = int(rng.choice([4, 8, 12, 16, 20, 24, 32]))
        if T * DW > 64:
            continue                                          # ... and a window (4 T DW dwords) is at most one granule long
        rows = 64
        ndw = (rows * Mr + Mr + 4 * T + 4) * DW
        lds = np.full(ndw + PAD * (ndw // 256 + 2), -1, dtype=np.int64)
        for d in range(ndw):                                  # DMA: granule i of the source lands at (256 + PAD) i
            lds[d + PAD * (d >> 8)] = d
        c0 = int(rng.integers(0, Mr))
        crossing = {}
        for row in range(rows):
            for lk in range(4):
                d0 = (row * Mr + c0 + lk) * DW                 # first window position of this lane, dwords from the tile start
                base = d0 + PAD * (d0 >> 8)
                th = 256 - (d0 & 255)
                for t in range(T):
                    hi = 4 * t * DW >= th
                    for part in range(DW):
                        got = lds[base + (PAD if hi else 0) + 4 * t * DW + part]
                        assert got == d0 + 4 * t * DW + part, (DW, Mr, T, row, lk, t)
                crossing.setdefault((row % 16, lk), set()).add(th)
        assert all(len(v) == 1 for v in crossing.values())    # one threshold per lane, whatever the chunk
        # bank spread of the 16 rows of an A operand (4-byte banks; k = 0 lanes): never worse than the plain run, and usable where that one is hostile
        def ways(pad):
            cnt = {}
            for i in range(16):
                d = i * Mr * DW
                cnt[(d + pad * (d >> 8)) % 32] = cnt.get((d + pad * (d >> 8)) % 32, 0) + 1
            return max(cnt.values())
        if 48 <= Mr * DW < 1008:
            assert ways(PAD) <= 5                   # (strides 16, 32, 1008, 1024 stay 8- / 15-way: the library keeps the row-staged form where the
                                                    #  padded run spreads the rows more than twice as badly, fir_mm_geo)
        if Mr * DW == 160:
            assert ways(PAD) == 4 and ways(0) == 16


def test_padded_run_addressing_model_long_windows():
    """Round 5 prep (MDSP_FIR_MM_RPX=1, unmeasured): the padded run for windows longer than a granule.  Register forms of up to 4 T DW = 512 dwords
    choose among THREE base pointers (the window start, one pad on from the step that crosses the first granule boundary, two pads on from 256 dwords
    later), still with immediate offsets; the fetched-tap form (T = 0, windows of any length) computes the pads in front of a position per k-step:
    ((d0 & 255) + 4 t DW) >> 8.  Every read lands on the sample the plain run would have read."""
    rng = np.random.default_rng(20260927)
    PAD = 4
    cases = 0
    for _ in range(600):
        DW = int(rng.choice([1, 2, 4]))
        Mr = int(rng.integers(1, 64)) * (16 // DW if DW < 16 else 1)
        if (Mr * DW) % 16:
            continue
        fetched = bool(rng.integers(0, 2))
        T = int(rng.choice([104, 208, 392])) if fetched else int(rng.choice([20, 24, 32, 40, 48, 64, 80, 96]))
        if not fetched and 4 * T * DW > 512:
            continue
        rows = 32
        ndw = (rows * Mr + Mr + 4 * T + 4) * DW
        lds = np.full(ndw + PAD * (ndw // 256 + 2), -1, dtype=np.int64)
        d = np.arange(ndw)
        lds[d + PAD * (d >> 8)] = d
        c0 = int(rng.integers(0, Mr))
        for row in range(rows):
            for lk in range(4):
                d0 = (row * Mr + c0 + lk) * DW
                base = d0 + PAD * (d0 >> 8)
                th = 256 - (d0 & 255)
                off0 = d0 & 255
                for t in range(T):
                    if fetched:
                        pads = (off0 + 4 * t * DW) >> 8
                    else:
                        pads = 2 if 4 * t * DW >= th + 256 else 1 if 4 * t * DW >= th else 0
                    for part in range(DW):
                        assert lds[base + PAD * pads + 4 * t * DW + part] == d0 + 4 * t * DW + part, (DW, Mr, T, fetched, row, lk, t)
        cases += 1
    assert cases > 100
    # what it buys where the round-4 counters put the conflicts: 1//16 Float32 (rows 240 dwords apart) eight-way -> three-way, 1//8 (88) four-way -> two-way
    def ways(stride, pad):
        cnt = {}
        for i in range(16):
            dd = i * stride
            for k in range(4):
                b = (dd + pad * (dd >> 8) + k) % 32
                cnt[b] = cnt.get(b, 0) + 1
        return max(cnt.values())
    assert ways(240, 0) == 8 and ways(240, PAD) == 3
    assert ways(88, 0) == 4 and ways(88, PAD) <= 3

