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


def matrix_form(h, L, M, x, phi_idx=1, deficit=1):
    """Outputs of FIRFilter(h, L//M) on x (zero history, state (phi_idx, deficit)) computed the way the kernel does."""
    g0 = gcd(L, M); L //= g0; M //= g0
    out = (C.c_int64 * 12)()
    _lib.check(_lib.lib().mdsp_fir_mm_geometry(L, M, len(h), _lib.F64, _lib.F64, out))
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
