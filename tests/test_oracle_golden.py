"""Pin the CPU oracle against the reference's own golden vectors and known answers (CPU-only).

Each test names the DSP.jl test (file:line) whose assertion it replays; ``≈`` is Julia's norm-wise
``isapprox`` with rtol = sqrt(eps) (see conftest.isapprox); ``==`` cases are exact.
"""
import math
from fractions import Fraction

import numpy as np
import pytest

from conftest import isapprox, relerr
from oracle import design, dspbase, filt, periodograms as pg, stream_filt as sf, util, windows


# ------------------------------------------------------------------------------------ helpers
def test_nextfastfft_literals():
    # test/util.jl:56-59
    assert util.nextfastfft(64) == 64
    assert util.nextfastfft(65) == 70
    assert util.nextfastfft(127) == 128
    assert [util.nextfastfft(n) for n in (1, 2, 11, 13, 211)] == [1, 2, 12, 14, 216]


def test_optimalfftfiltlength_literals():
    assert dspbase.optimalfftfiltlength(1, 3) == 1          # test/dsp.jl:39
    assert dspbase.optimalfftfiltlength(256, 2 ** 30) == 2048   # BASELINE config 2
    assert dspbase.optimalfftfiltlength(127, 10 ** 6) == 1024   # BASELINE config 1
    assert abs(dspbase.os_fft_complexity(2048, 256) - 13.71) < 0.01


def test_hanning128(golden):
    # test/windows.jl:55-59
    assert isapprox(windows.hanning(128), golden["hanning128"])
    assert np.max(np.abs(windows.hanning(128) - golden["hanning128"])) < 5e-16


def test_firwindow_taps(golden):
    # test/filter_design.jl:988-1060: digitalfilter(Lowpass(0.25), FIRWindow(hamming(N); scale); fs=1)
    for n in (128, 129):
        for scaled, key in ((False, f"digitalfilter_hamming_{n}_lowpass_fc0p25_fs1p0"),
                            (True, f"digitalfilter_hamming_{n}_lowpass_scaled_fc0p25_fs1p0")):
            h = design.digitalfilter_lowpass_firwindow(0.25, windows.hamming(n), fs=1.0, scale=scaled)
            assert isapprox(h, golden[key]), (n, scaled)


def test_kaiserord_against_known():
    # test/filter_design.jl:965-981 compares with scipy.signal.kaiserord; spot values
    n, alpha = design.kaiserord(0.1, 60)
    assert n == 74 and abs(alpha * math.pi - 5.65326) < 1e-5


# ------------------------------------------------------------------------------ filt / conv
def test_filt_exact_integers():
    # test/dsp.jl:10-21
    b = np.array([1., 2., 3., 4.])
    x = np.array([1., 1., 0., 1., 1., 0., 0., 0.])
    assert np.array_equal(dspbase.filt_ba(b, 1.0, x), [1., 3., 5., 8., 7., 5., 7., 4.])
    assert np.array_equal(dspbase.filt_ba(b, [1., -0.5], x),
                          [1., 3.5, 6.75, 11.375, 12.6875, 11.34375, 12.671875, 10.3359375])
    assert np.array_equal(dspbase.filt_ba(b, 1.0, np.arange(1.0, 11.0)), [1., 4., 10., 20., 30., 40., 50., 60., 70., 80.])
    X = np.stack([x, np.arange(1.0, 9.0)], axis=1)
    both = dspbase.filt_ba(b, 1.0, X)
    assert np.array_equal(both[:, 0], dspbase.filt_ba(b, 1.0, x))
    assert np.array_equal(both[:, 1], dspbase.filt_ba(b, 1.0, np.arange(1.0, 9.0)))


def test_conv_small_known():
    # test/dsp.jl:53-70
    a = np.array([1, 2, 1, 2]); b = np.array([1, 2, 3])
    assert np.array_equal(dspbase.conv(a, b), [1, 4, 8, 10, 7, 6])
    fa = a.astype(float); fb = b.astype(float)
    assert isapprox(dspbase.conv(fa, fb), [1., 4, 8, 10, 7, 6])
    got = dspbase.conv(fa + 1j, fb + 0j)
    assert isapprox(got, np.array([1, 4, 8, 10, 7, 6]) + 1j * np.array([1, 3, 6, 6, 5, 3]))
    assert np.array_equal(dspbase.conv(np.array([314159265]), np.array([314159265])), [314159265 ** 2])  # issue #410


def test_conv_algorithms_agree():
    # test/dsp.jl:72-76 and :98-121
    rng = np.random.default_rng(1776)
    u, v = rng.random(190), rng.random(200)
    ref = dspbase.conv(u, v, "direct")
    for alg in ("fft_simple", "fft_overlapsave", "fft", "fast", "auto"):
        assert isapprox(dspbase.conv(u, v, alg), ref), alg
    with pytest.raises(ValueError):
        dspbase.conv(u, v, "quantum")
    for M in (10, 200):
        for N in (10, 200):
            for cplx in (False, True):
                u = rng.random(M) + (1j * rng.random(M) if cplx else 0)
                v = rng.random(N) + (1j * rng.random(N) if cplx else 0)
                d = dspbase.conv(u, v, "direct")
                assert isapprox(dspbase.conv(u, v, "fft_simple"), d)
                assert isapprox(dspbase.conv(u, v, "fft_overlapsave"), d)
                for alg in ("direct", "fft_simple", "fft_overlapsave"):
                    out = dspbase.conv(u, v, alg, out_len=M + N + 10)
                    assert np.array_equal(out[:M + N - 1], dspbase.conv(u, v, alg))
                    assert np.all(out[M + N - 1:] == 0)


@pytest.mark.parametrize("dt", [np.float32, np.float64, np.complex128])
@pytest.mark.parametrize("nsmall", [12, 128])
def test_overlap_save_kernel(dt, nsmall):
    # test/dsp.jl:289-297 (N=1)
    rng = np.random.default_rng(7)
    nlarge = 128
    u = rng.random(nlarge).astype(dt); v = rng.random(nsmall).astype(dt)
    if np.dtype(dt).kind == "c":
        u = u + 1j * rng.random(nlarge); v = v + 1j * rng.random(nsmall)
    nfft = dspbase.optimalfftfiltlength(nsmall, nlarge)
    n = nlarge + nsmall - 1
    assert isapprox(dspbase.unsafe_conv_kern_os(u, v, nfft, n, dt), dspbase._conv_kern_fft(u, v, n, dt))


@pytest.mark.parametrize("nlarge,nsmall,nfft", [(128, 12, 256), (128, 13, 32), (128, 12, 32), (25, 4, 16)])
def test_overlap_save_adversarial(nlarge, nsmall, nfft):
    # test/dsp.jl:304-313
    rng = np.random.default_rng(8)
    u, v = rng.random(nlarge), rng.random(nsmall)
    n = nlarge + nsmall - 1
    assert isapprox(dspbase.unsafe_conv_kern_os(u, v, nfft, n, np.float64), np.convolve(u, v))
    if (nlarge, nsmall, nfft) == (25, 4, 16):
        blocks, save, nblocks = dspbase.os_block_table(25, 4, 16, 28)
        assert nblocks == 3 and all(b["edge"] for b in blocks)      # "three blocks need to be padded"


def test_os_geometry_config2():
    # 2^30 (+255) samples, nb=256 -> nfft 2048, save 1793, 598 853 blocks; first/last are edge blocks
    su, sv, nfft = 2 ** 30, 256, 2048
    ideal = nfft - sv + 1
    assert ideal == 1793
    assert -(-(su + sv - 1) // ideal) == 598853
    assert -(-su // ideal) == 598853
    L, rows = filt.fftfilt_block_table(256, 10 ** 4, 2048)
    assert L == 1793 and rows[0] == (1, 255, 1, 1793, 1793) and rows[1][0] == 1794


@pytest.mark.parametrize("xlen", [2 ** 7 - 1, 2 ** 10 - 1, 2 ** 13 - 1])
@pytest.mark.parametrize("blen", [1, 3, 31, 127])
def test_fftfilt_matches_tdfilt(xlen, blen):
    # test/filt.jl:312-331
    rng = np.random.default_rng(xlen * 131 + blen)
    b = rng.standard_normal(blen)
    for x in (rng.random(xlen), rng.random((xlen, 2))):
        filtres = dspbase.filt_ba(b, [1.0], x)
        assert isapprox(filt.fftfilt(b, x), filtres)
        assert isapprox(filt.filt(b, x), filtres)
        assert isapprox(filt.tdfilt(b, x), filtres)


# ------------------------------------------------------------------------------- periodograms
def test_spectrogram_matlab(golden):
    # test/periodograms.jl:25-36
    spec = pg.spectrogram(golden["spectrogram_x"], 256, 128, fs=10)
    assert isapprox(spec.power, golden["spectrogram_p"])
    assert isapprox(spec.freq, golden["spectrogram_f"])
    assert isapprox(spec.time, golden["spectrogram_t"])
    assert relerr(spec.power, golden["spectrogram_p"]) < 1e-13


def test_stft_matlab(golden):
    # test/periodograms.jl:332-344
    S = pg.stft(golden["stft_x"], 400, 240, nfft=512, fs=16000, window=windows.hanning)
    ref = golden["stft_S_real"] + 1j * golden["stft_S_imag"]
    assert S.shape == (257, 29)
    assert isapprox(S, ref)


DATA07 = np.arange(8)


def test_pwelch_0_7_twosided():
    # test/periodograms.jl:92-135
    data0 = np.array([98.0, 13.656854249492380, 4.0, 2.343145750507620, 2.0, 2.343145750507620, 4.0, 13.656854249492380])
    assert isapprox(pg.periodogram(DATA07, onesided=False).power, data0)
    assert isapprox(pg.welch_pgram(DATA07, 8, 0, onesided=False, window=None).power, data0)
    assert isapprox(pg.spectrogram(DATA07, 8, 0, onesided=False).power[:, 0], data0)
    z = DATA07 + 1j * DATA07
    assert isapprox(pg.periodogram(z, onesided=False).power, data0 * 2)
    assert isapprox(pg.welch_pgram(z, 8, 0, onesided=False, window=None).power, data0 * 2)
    assert isapprox(pg.spectrogram(z, 8, 0, onesided=False).power[:, 0], data0 * 2)
    for (n, nov, exp) in ((2, 0, [34.5, 0.5]), (3, 0, [25.5, 1.0, 1.0]), (3, 1, [35.0, 1.0, 1.0]), (4, 1, [45, 2, 1, 2])):
        assert isapprox(pg.welch_pgram(DATA07, n, nov, onesided=False, window=None).power, np.array(exp, float))
        assert isapprox(pg.welch_pgram(DATA07, n, nov, onesided=False, window=None, sequential=True).power, np.array(exp, float))
        assert isapprox(pg.spectrogram(DATA07, n, nov, onesided=False).power.mean(axis=1), np.array(exp, float))


def test_pwelch_0_7_windows():
    # test/periodograms.jl:143-169
    cases = ((windows.hamming, [65.461623986801527, 20.556791795515764, 0.369313143650544, 0.022167446610882,
                                0.025502985564107, 0.022167446610882, 0.369313143650544, 20.556791795515764]),
             (windows.bartlett, [62.999999999999993, 21.981076052592442, 0.285714285714286, 0.161781090264695,
                                 0.142857142857143, 0.161781090264695, 0.285714285714286, 21.981076052592442]))
    for w, exp in cases:
        exp = np.array(exp)
        for win in (w, w(8)):
            assert isapprox(pg.periodogram(DATA07, window=win, onesided=False).power, exp)
            assert isapprox(pg.welch_pgram(DATA07, 8, 0, window=win, onesided=False).power, exp)
            assert isapprox(pg.spectrogram(DATA07, 8, 0, window=win, onesided=False).power[:, 0], exp)


def test_padded_periodogram():
    # test/periodograms.jl:171-220
    exp = np.array([98, 174.463067389405, 121.968086934209, 65.4971744936088, 27.3137084989848, 12.1737815028909,
                    10.3755170959439, 10.4034038628775, 8, 5.25810953219633, 4.47015397150535, 4.89522578856669,
                    4.68629150101524, 3.69370284475603, 3.1862419983415, 3.61553458569862, 2])
    assert isapprox(pg.periodogram(DATA07, nfft=32).power, exp)
    assert isapprox(pg.welch_pgram(DATA07, 8, 0, nfft=32, window=None).power, exp)
    assert isapprox(pg.spectrogram(DATA07, 8, 0, nfft=32).power[:, 0], exp)
    exph = np.array([65.4616239868015, 122.101693164395, 98.8444689598445, 69.020252632913, 41.1135835910315,
                     20.5496474310966, 8.43291449161938, 2.78001620362588, 0.738626287301088, 0.174995741770789,
                     0.0501563022944516, 0.0327357460012861, 0.0443348932217643, 0.0553999745503552,
                     0.0561319901616643, 0.0526025934871384, 0.0255029855641069])
    assert isapprox(pg.periodogram(DATA07, window=windows.hamming, nfft=32).power, exph)
    assert isapprox(pg.welch_pgram(DATA07, 8, 0, window=windows.hamming, nfft=32).power, exph)
    cfg = pg.WelchConfig.create(8, DATA07.dtype, n=8, noverlap=0, window=windows.hamming, nfft=32)
    a = pg.welch_pgram(DATA07, config=cfg).power
    assert np.array_equal(a, pg.welch_pgram(DATA07, config=cfg).power)        # :222-224 (== on reuse)


def test_fft2oneortwosided():
    # test/periodograms.jl:346-379 semantics: two-sided completion of a real spectrum equals the full fft
    rng = np.random.default_rng(3)
    for nfft in (10, 12, 13):
        x = rng.standard_normal(nfft)
        full = np.fft.fft(x)
        assert isapprox(pg.fft2oneortwosided(np.fft.rfft(x)[None, :], nfft, False)[0], full)
        assert np.array_equal(pg.fft2oneortwosided(np.fft.rfft(x)[None, :], nfft, True)[0], np.fft.rfft(x))
        p2 = pg.fft2pow(np.fft.rfft(x)[None, :], nfft, 1.0, False, np.float64)[0]
        assert isapprox(p2, np.abs(full) ** 2)


def test_arraysplit():
    # test/periodograms.jl:393-402 and the docstring examples periodograms.jl:97-111
    fr = pg.arraysplit(np.ones(1000), 100, 10)
    assert fr.shape == (11, 100)
    assert np.array_equal(pg.arraysplit(np.array([0.1, 0.2, 0.3, 0.4, 0.5]), 3, 1), [[0.1, 0.2, 0.3], [0.3, 0.4, 0.5]])
    got = pg.arraysplit(np.array([0.1, 0.2, 0.3, 0.4, 0.5]), 3, 2, 8)
    assert got.shape == (3, 8) and np.array_equal(got[1, :3], [0.2, 0.3, 0.4]) and np.all(got[:, 3:] == 0)
    assert np.array_equal(pg.arraysplit(np.array([0.1, 0.2, 0.3, 0.4, 0.5]), 3, 1, 3, np.array([1, 2, 1])),
                          np.array([[0.1, 0.4, 0.3], [0.3, 0.8, 0.5]]))
    assert pg.frame_count(2 ** 30, 4096, 2048) == 524287        # BASELINE config 3
    assert pg.frame_count(2 ** 26, 1024, 768) == 262141          # BASELINE config 4
    with pytest.raises(ValueError):
        pg.arraysplit(np.ones(10), 4, 4)
    with pytest.raises(ValueError):
        pg.arraysplit(np.ones(10), 4, 1, 3)


def test_welch_sequential_vs_f64_sum():
    rng = np.random.default_rng(11)
    x = rng.standard_normal(20000).astype(np.float32)
    a = pg.welch_pgram(x, 256, 128, window=windows.hanning, sequential=True).power
    b = pg.welch_pgram(x, 256, 128, window=windows.hanning).power
    c = pg.welch_pgram(x, 256, 128, window=windows.hanning, dtype=np.float64).power
    assert a.dtype == np.float32 and b.dtype == np.float32 and c.dtype == np.float64
    assert relerr(a, c) < 2e-6 and relerr(b, c) < 2e-6


# --------------------------------------------------------------------------------- resampling
@pytest.mark.parametrize("rate", [Fraction(1, 2), Fraction(2, 1), Fraction(3, 2), Fraction(2, 3)])
def test_resample_matlab(golden, rate):
    # test/resample.jl:8-24
    x = golden["resample_x"]
    h = golden[f"resample_taps_{rate.numerator}_{rate.denominator}"]
    y = golden[f"resample_y_{rate.numerator}_{rate.denominator}"]
    got = sf.resample(x, rate, h)
    assert got.shape == y.shape
    assert isapprox(got, y)
    assert isapprox(sf.resample(x, rate), y, rtol=1e-3)     # default taps


def test_resample_exact_small():
    # test/filt_stream.jl:366-367
    h = np.array([0, 0, 1, 0, 0, 0.0])
    assert np.array_equal(sf.resample(np.array([1, 2]), 3, h), [1, 0, 0, 2, 0, 0])
    assert np.array_equal(sf.resample(np.array([1, 2]), Fraction(3, 2), h), [1, 0, 0])


def test_resample_dims(golden):
    # test/resample.jl:35-45
    x = golden["resample_x"]; h = golden["resample_taps_1_2"]; y = golden["resample_y_1_2"]
    X = np.stack([x, math.e * x], axis=1)
    got = sf.resample_dims(X, Fraction(1, 2), h, dims=0)
    assert isapprox(got, np.stack([y, math.e * y], axis=1))
    assert isapprox(sf.resample_dims(X.T, Fraction(1, 2), h, dims=1), np.stack([y, math.e * y], axis=1).T)


def _naivefilt(h, x, L=1, M=1):
    # test/filt_stream.jl:4-17: zero-stuff, filter, decimate
    xz = np.zeros(len(x) * L, dtype=np.result_type(x.dtype, h.dtype))
    xz[::L] = x
    y = np.convolve(xz, h)[:len(xz)]
    return y[::M]


@pytest.mark.parametrize("L", [1, 5, 14, 23])
@pytest.mark.parametrize("M", [1, 9, 17, 21])
def test_firfilter_vs_naive_and_streaming(L, M):
    # test/filt_stream.jl:231-281, :338-364
    rng = np.random.default_rng(L * 100 + M)
    for Th in (np.float32, np.float64):
        for Tx in (np.float32, np.float64, np.complex64, np.complex128):
            h = rng.random(int(rng.integers(16, 129))).astype(Th)
            xlen = int(rng.integers(200, 301))
            x = rng.random(xlen).astype(Tx)
            if np.dtype(Tx).kind == "c":
                x = (x + 1j * rng.random(xlen)).astype(Tx)
            ratio = Fraction(L, M)
            naive = _naivefilt(h, x, ratio.numerator, ratio.denominator)
            tol = 1e-4 if (Th == np.float32 or np.dtype(Tx).itemsize in (4, 8) and np.dtype(Tx) in (np.dtype(np.float32), np.dtype(np.complex64))) else 1e-8
            stateless = sf.filt_stateless(h, x, ratio)
            assert stateless.shape == naive.shape
            assert relerr(stateless, naive) < tol
            # two chunks
            f = sf.FIRFilter(h, ratio)
            cut = xlen // 3
            y2 = np.concatenate([f.filt(x[:cut]), f.filt(x[cut:])])
            assert y2.shape == naive.shape and relerr(y2, naive) < tol
            # inputlength(filt, len) == xLen   (filt_stream.jl:262)
            g = sf.FIRFilter(h, ratio)
            assert g.outputlength(g.inputlength(len(naive))) <= len(naive)
    # one sample at a time (F64 only, it is slow)
    h = rng.random(40); x = rng.random(120)
    ratio = Fraction(L, M)
    f = sf.FIRFilter(h, ratio)
    pieces = [f.filt(x[i:i + 1]) for i in range(len(x))]
    y1 = np.concatenate(pieces)
    naive = _naivefilt(h, x, ratio.numerator, ratio.denominator)
    assert y1.shape == naive.shape and relerr(y1, naive) < 1e-12


def test_polyphase_closed_form_matches_recurrence():
    # closed form == the serial recurrence of stream_filt.jl:506-508
    for (L, M, phi0, d0) in ((160, 147, 1, 1), (160, 147, 77, 17), (3, 2, 2, 1), (5, 9, 4, 3), (23, 21, 23, 2)):
        phi, idx = phi0, d0
        ms = np.arange(500)
        cphi, cidx = sf.polyphase_closed_form(phi0, d0, L, M, ms)
        for m in ms:
            assert (phi, idx) == (cphi[m], cidx[m])
            idx += (phi + M - 1) // L
            phi += M % L
            if phi > L:
                phi -= L


def test_inputlength_outputlength_bracketing():
    # test/resample.jl:154-166
    rng = np.random.default_rng(5)
    for _ in range(1000):
        Mr = Fraction(int(rng.integers(1, 11)), int(rng.integers(1, 11)))
        H = sf.FIRFilter(np.zeros(int(rng.integers(1, 101))), Mr)
        if Mr != 1:
            H.setphase(10 * rng.random())
        yL = int(rng.integers(1, 101))
        assert H.outputlength(H.inputlength(yL)) <= yL < H.outputlength(H.inputlength(yL) + 1)
        assert H.outputlength(H.inputlength(yL, True) - 1) < yL <= H.outputlength(H.inputlength(yL, True))


def test_config5_lengths():
    # 2^28 inputs at 160//147 with a 5120-tap filter -> 292 174 646 outputs per channel
    assert math.ceil(2 ** 28 * Fraction(160, 147)) == 292174646
    pfb = sf.taps2pfb(np.arange(1, 10), 4)
    assert np.array_equal(pfb, [[9, 0, 0, 0], [5, 6, 7, 8], [1, 2, 3, 4]])     # stream_filt.jl:286-292


# ---- multitaper (src/multitaper.jl) and dpss (windows.jl:668-776): oracle pinned to the reference's MATLAB / MNE goldens ----
def test_dpss_matlab_golden(golden):
    from oracle import windows as ow
    d1 = ow.dpss(128, 4)
    assert isapprox(d1, golden["dpss128_4"])                                     # test/windows.jl:34-36
    lam = [0.9999999997159923, 0.9999999731146645, 0.9999988168667646, 0.9999680890685374, 0.9994167543397652,
           0.9925560207018469, 0.9368556668429153]
    assert isapprox(ow.dpsseig(d1, 4), np.array(lam))                            # test/windows.jl:39-42
    z = ow.dpss(8, 2, 1, zerophase=True)[:, 0]
    assert np.array_equal(z, np.fft.ifftshift(ow.dpss(9, 2, 1)[:8, 0]))          # test/windows.jl:171
    with pytest.raises(ValueError):
        ow.dpss(9, 2, 1, zerophase=True)                                         # test/windows.jl:173


def test_mt_pgram_matlab_goldens(golden):
    from oracle import multitaper as omt, windows as ow
    s = golden["stft_x"]
    p, _ = omt.mt_pgram(s, fs=16000)
    assert isapprox(p, golden["mt_pgram"])                                       # test/periodograms.jl:385
    p, _ = omt.mt_pgram(s, fs=16000, window=ow.dpss(len(s), 4))
    assert isapprox(p, golden["mt_pgram"])                                       # :386
    x = golden["pmtm_x"]
    p, f = omt.mt_pgram(x, fs=1000, nw=4, nfft=omt.nextpow2(len(x)))
    assert isapprox(f, golden["pmtm_fx"]) and isapprox(p, golden["pmtm_pxx"])    # :416-418


def test_mt_spectrogram_first_column_is_mt_pgram(golden):
    from oracle import multitaper as omt
    x0 = golden["spectrogram_x"]
    P, f, t = omt.mt_spectrogram(x0, 256, 128, fs=10)
    assert np.array_equal(f, golden["spectrogram_f"]) or isapprox(f, golden["spectrogram_f"])
    assert isapprox(t, golden["spectrogram_t"])                                  # test/periodograms.jl:39-41
    assert isapprox(P[:, 0], omt.mt_pgram(x0[:256], fs=10)[0])                   # :42
    for n_samples in range(20, 101, 20):                                         # test/multitaper.jl:16-19
        for spw in range(20, 101, 20):
            for ov in range(0, spw, 20):
                cfg = omt.MTSpectrogramConfig(n_samples, omt.MTConfig(np.float64, spw), ov)
                from oracle.periodograms import arraysplit
                assert len(cfg.time) == len(arraysplit(np.arange(1, n_samples + 1), spw, ov))


def test_mt_cross_spectra_and_coherence_mne_goldens(golden):
    from oracle import multitaper as omt
    fs, n = 1000.0, 1024
    t = np.arange(n) / fs
    sig = np.vstack([np.sin(np.pi * 2 * 12.0 * t), np.sin(np.pi * (2 * 12.0 * t + 1))])
    mc = omt.dpss_config(np.float64, n, fs=fs, keep_only_large_evals=True, weight_by_evals=True)
    cs, fr = omt.mt_cross_power_spectra(sig, omt.MTCrossSpectraConfig(2, mc, demean=True))
    ref = (golden["csd_array_multitaper_values_re"] + 1j * golden["csd_array_multitaper_values_im"]).reshape((2, 2, 512), order="F")
    assert isapprox(fr[1:], golden["csd_array_multitaper_frequencies"])          # test/multitaper.jl:281
    assert isapprox(cs[:, :, 1:], ref)                                           # :282
    noisy = np.vstack([sig[0], sig[0] + 3 * golden["noise"]])
    coh, f = omt.mt_coherence(noisy, omt.MTCrossSpectraConfig(2, mc, demean=True, freq_range=(10, 15)))
    assert all(10 <= x <= 15 for x in f)
    assert abs(coh.mean(axis=2)[1, 0] - 0.982356762670818) < 1e-12               # :270-272 (MNE reference value)
    x = sig[0] + 3 * golden["noise"]                                             # :308-316: cross spectra of one channel == mt_pgram
    cs1, f1 = omt.mt_cross_power_spectra(x[None, :], fs=fs)
    pg, fg = omt.mt_pgram(x, fs=fs)
    assert isapprox(f1, fg) and isapprox(cs1[0, 0].real, pg)
    with pytest.raises(ValueError):
        omt.mt_cross_power_spectra(x[None, :].astype(complex), fs=fs)            # :333


def test_oracle_array_convolution_known_answers():
    """oracle.dspbase.conv_nd / conv_separable against the literal tables of test/dsp.jl:130-268."""
    import conv_cases as cc
    from oracle import dspbase as odsp
    for alg in ("auto", "direct", "fft_simple", "fft"):
        assert np.array_equal(odsp.conv_nd(cc.A2, cc.B2, alg), cc.EXP2) and np.array_equal(odsp.conv_nd(cc.B2, cc.A2, alg), cc.EXP2)
        assert np.array_equal(odsp.conv_nd(cc.A3, cc.B3, alg), cc.EXP3)
        fa, fb = cc.A2.astype(np.float64), cc.B2.astype(np.float64)
        assert np.allclose(odsp.conv_nd(fa, fb, alg), cc.EXP2, rtol=1e-13, atol=1e-13)
        got = odsp.conv_nd(fa + 1j, fb + 0j, alg)
        assert np.allclose(got.real, cc.EXP2, atol=1e-12) and np.allclose(got.imag, cc.IM_EXP2, atol=1e-12)
    assert np.allclose(odsp.conv_separable(cc.SEP_U.astype(float), cc.SEP_V.astype(float), cc.SEP_A.astype(float)), cc.SEP_EXP, rtol=1e-12)
    a, b = cc.promoted_case()
    exp = np.stack([odsp.conv_nd(a[:, :, 0], b) * n for n in range(1, 7)], axis=2)
    assert np.array_equal(odsp.conv_nd(a, b), exp) and np.array_equal(odsp.conv_nd(b, a), exp)
    ones6 = np.ones((2,) * 6)
    assert np.array_equal(odsp.conv_nd(ones6, np.ones((1,) * 6)), ones6)                      # test/dsp.jl:256-259
    rng = np.random.default_rng(0)
    u, v = rng.standard_normal((9, 14, 5)), rng.standard_normal((4, 3, 6))
    assert np.allclose(odsp.conv_nd(u, v, "direct"), odsp.conv_nd(u, v, "fft_simple"), rtol=1e-12, atol=1e-12)   # test/dsp.jl:169-171


def test_oracle_xcorr_known_answers():
    """oracle.dspbase.xcorr against test/dsp.jl:317-360."""
    import conv_cases as cc
    from oracle import dspbase as odsp
    for u, v, kw, exp in cc.XCORR:
        assert np.allclose(odsp.xcorr(np.asarray(u), np.asarray(v), **kw), exp, atol=1e-13), (u, v, kw)
    with pytest.raises(ValueError):
        odsp.xcorr(np.array([1]), np.array([2]), padmode="bug")
    with pytest.raises(ValueError):
        odsp.xcorr(np.array([1]), np.array([2, 3]), scaling="biased")
