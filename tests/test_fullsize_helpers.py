"""CPU checks of tests/fullsize.py: the piecewise oracle evaluations used by the BASELINE-size GPU tests equal the
one-shot oracle wherever both can run."""
from fractions import Fraction

import numpy as np
import pytest

import fullsize as fz
from oracle import design as odes
from oracle import periodograms as opg
from oracle import stream_filt as osf
from oracle import windows as ow


def test_welch_chunked_equals_one_shot():
    rng = np.random.default_rng(1776)
    s = rng.standard_normal(70_000).astype(np.float32)
    for n, nov, cf in ((512, 256, 7), (512, 384, 64), (500, 123, 1)):
        ref = opg.welch_pgram(s, n, nov, window=ow.hanning, dtype=np.float64).power
        got, K = fz.oracle_welch_chunked(lambda lo, hi: s[lo:hi], len(s), n, nov, ow.hanning, chunk_frames=cf)
        assert K == opg.frame_count(len(s), n, nov)
        assert np.allclose(got, ref, rtol=1e-12, atol=0)


def test_stft_columns_equal_one_shot():
    rng = np.random.default_rng(7)
    s = (rng.standard_normal(9000) + 1j * rng.standard_normal(9000)).astype(np.complex64)
    full = opg.stft(s, 256, 192, window=ow.hanning, dtype=np.float64)
    for f0, cnt in ((0, 3), (17, 5), (full.shape[1] - 2, 2)):
        got = fz.oracle_stft_columns(lambda lo, hi: s[lo:hi], 256, 192, f0, cnt, ow.hanning)
        assert np.array_equal(got, full[:, f0:f0 + cnt])
    fullp = opg.stft(s, 256, 192, psdonly=True, window=ow.hanning, fs=2.0, dtype=np.float64)
    got = fz.oracle_stft_columns(lambda lo, hi: s[lo:hi], 256, 192, 9, 4, ow.hanning, psdonly=True, fs=2.0)
    assert np.array_equal(got, fullp[:, 9:13])


@pytest.mark.parametrize("ratio,hlen", [(Fraction(160, 147), 5120), (Fraction(3, 2), 97), (Fraction(2, 3), 64), (Fraction(7, 5), None)])
def test_resample_window_equals_one_shot(ratio, hlen):
    rng = np.random.default_rng(3)
    x = rng.standard_normal(6000)
    h = odes.resample_filter(ratio)
    if hlen is not None:
        h = np.resize(h, hlen) if len(h) >= hlen else np.concatenate([h, np.zeros(hlen - len(h))])
    ref = osf.resample(x, ratio, h)
    nout = fz.resample_output_length(len(x), ratio)
    assert len(ref) == nout
    for m0, cnt in ((0, 50), (1, 333), (1234, 600), (nout - 600, 600), (nout - 1, 1)):
        got = fz.oracle_resample_window(lambda lo, hi: x[lo:hi], len(x), ratio, h, m0, cnt)
        assert np.array_equal(got, ref[m0:m0 + cnt]), (m0, cnt)


def test_filt_chunked_equals_one_shot():
    from oracle import dspbase as odsp
    rng = np.random.default_rng(11)
    x = rng.standard_normal(50_000).astype(np.float32)
    b = rng.standard_normal(256)
    ref = odsp.filt_ba(b, 1.0, x.astype(np.float64))
    for chunk in (1 << 12, 10_007, 1 << 16):
        got = np.empty_like(ref)
        for lo, hi, y in fz.oracle_filt_chunked(lambda lo, hi: x[lo:hi], len(x), b, chunk=chunk):
            assert len(y) == hi - lo
            got[lo:hi] = y
        assert np.allclose(got, ref, rtol=0, atol=1e-11 * np.max(np.abs(ref)))
