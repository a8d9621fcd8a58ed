"""GPU parity tests of the run-time-schedule spectral kernel (dsp.jl_amd/csrc/gx_kernels.h, round 6): welch_pgram / stft / spectrogram / periodogram at
7-smooth transform sizes WITHOUT a compile-time schedule -- what nextfastfft (util.jl:134) returns for most frame lengths, and every default call
(n = length(s) >> 3, periodograms.jl:560, :647, :828, :872) on a 65 537 .. 400 000-sample signal.  Sizes cover: one workgroup per transform (R0 = 1; one
and two LDS buffers, 256 and 512 threads, every radix 2 .. 16 in first / middle / last position), the fused column step (nfft = R0 x S: 8400, 12500,
16384, 20000, 40000 -- VERDICT r5 item 1 -- and an odd size; in Float32 on the compile-time row kernels of csrc/spectral_ctcols.hip where R0 x S is one of
theirs), zero padding, odd frame counts, several channels.

    Float32 / ComplexF32:  norm-wise <= 5e-6 and element-wise |err| <= 8 log2(nfft) ulp of the largest bin
    Float64 / ComplexF64:  norm-wise <= 1e-12
against the Float64 oracle (oracle/periodograms.py); frame counts, axes and shapes bit-exact."""
import math

import numpy as np
import pytest

from conftest import relerr, ulps_of_max

pytestmark = pytest.mark.gpu

TOL64 = 1e-12
TOL32 = 5e-6
DTYPES = [(np.float32, TOL32), (np.float64, TOL64), (np.complex64, TOL32), (np.complex128, TOL64)]


@pytest.fixture(scope="module")
def d():
    import dsp_jl_amd as dd
    from dsp_jl_amd import _lib
    if _lib.device_count() < 1:
        pytest.fail("GPU tests need a HIP device")
    _lib.check(_lib.lib().mdsp_init(0))
    return dd


def _signal(rng, length, dt):
    s = rng.standard_normal(length) + 0.5 * np.sin(2 * np.pi * 0.1234 * np.arange(length))
    if np.dtype(dt).kind == "c":
        return (s + 1j * rng.standard_normal(length)).astype(dt)
    return s.astype(dt)


def _ulp_bound(nfft):
    return 8.0 * math.log2(nfft)


# n, noverlap, nfft, window, frames
WELCH_CASES = (
    (1125, 562, 1125, "hanning", 9),         # 9 5 5 5, 256 threads, two buffers; odd frame count
    (2187, 0, 2187, None, 4),                # 3^7: radices 3 and 9
    (4802, 2401, 4802, "hamming", 6),        # 7 7 7 14
    (6561, 3000, 6561, "hanning", 5),        # 3^8
    (7000, 3500, 7168, "hanning", 5),        # zero-padded to 2^10 7
    (8400, 4200, 8400, "hanning", 7),        # 2 x 4200: the column step (8400, 12500, 16384, 20000, 40000)
    (12500, 6250, 12500, "hanning", 8),
    (16384, 8192, 16384, "hanning", 5),
    (20000, 10000, 20000, "hamming", 6),
    (40000, 20000, 40000, "hanning", 5),
    (9261, 4630, 9261, None, 4),             # odd (21^3 = 3 x 3087)
    (19000, 9500, 19683, "hanning", 3),      # 3^9 = 9 x 2187: no split with R0 <= 8 exists (the R0 <= 32 fallback)
    # Float32 / ComplexF32: R0 x a row size with a COMPILE-TIME schedule (csrc/spectral_ctcols.hip) -- 16384 = 2 x 8192, 12500 = 2 x 6250 and 20000 = 4 x 5000 above,
    (11520, 5760, 11520, "hanning", 5),      # 3 x 3840
    (9000, 4000, 9216, "hamming", 4),        # 2 x 4608 = 9 16 32, zero-padded
    (13500, 6750, 13500, None, 3),           # 2 x 6750 = 15 18 25
    (65536, 32768, 65536, "hanning", 3),     # 4 x 16384 (Float64: the multi-pass engine)
    # ... one workgroup per transform up to 16384 points in the lean forms (csrc/spectral_ctbig.hip: Float32 sums, window beside the samples, derived twiddles)
    (15625, 7000, 15625, None, 3),           # 25 25 5 5 at 1024 threads
    (10240, 5120, 10240, "hamming", 4),      # 32 16 20
    # ... and as the rows of nfft = R0 x S (csrc/spectral_ctcols_big.hip)
    (25000, 12500, 25000, "hanning", 4),     # 2 x 12500
    (32768, 16384, 32768, "hanning", 3),     # 2 x 16384 (Float64: the multi-pass engine)
    (100000, 50000, 100000, "hanning", 3),   # 8 x 12500
    (16800, 8400, 16800, "hamming", 4),      # 2 x 8400: Float32 csrc/spectral_ctcols_big.hip, Float64 csrc/spectral_ctcols_f64.hip (rows of 16-byte elements to 9600 points)
    (28800, 14000, 28800, None, 3),          # 3 x 9600
    (96000, 48000, 96000, "hanning", 3),     # Float32: 6 x 16000, Float64: 10 x 9600, both in two kernels
    # ... and in two kernels where R0 > 8 (csrc/spectral_ctrows.hip: column kernel into a work buffer, the single-workgroup kernel over the rows)
    (200000, 100000, 200000, "hanning", 3),  # 16 x 12500
    (150000, 70000, 150000, None, 4),        # 10 x 15000, odd frame count
)


@pytest.mark.parametrize("dt,tol", DTYPES)
def test_gx_welch_vs_oracle(d, dt, tol):
    from oracle import periodograms as opg, windows as ow
    rng = np.random.default_rng(61)
    cplx = np.dtype(dt).kind == "c"
    f32 = dt in (np.float32, np.complex64)
    for (n, nov, nfft, wname, K) in WELCH_CASES:
        win = getattr(ow, wname) if wname else None
        dwin = getattr(d, wname) if wname else None
        length = (K - 1) * (n - nov) + n + 11
        s = _signal(rng, length, dt)
        for onesided in ((False,) if cplx else (True, False)):
            cfg = d.WelchConfig(length, dt, n=n, noverlap=nov, nfft=nfft, window=dwin, onesided=onesided, fs=2.5)   # AUTO
            assert cfg.engine == d.ENGINE_FUSED, nfft
            got = d.welch_pgram(s, cfg)
            ref = opg.welch_pgram(s, n, nov, nfft=nfft, window=win, onesided=onesided, fs=2.5, dtype=np.float64)
            assert got.power.dtype == (np.float32 if f32 else np.float64)
            assert got.power.shape == ref.power.shape and np.array_equal(got.freq, ref.freq)
            e = relerr(got.power, ref.power)
            assert e < tol, (n, nov, nfft, onesided, e)
            if f32:
                assert ulps_of_max(got.power, ref.power) < _ulp_bound(nfft), (n, nfft, ulps_of_max(got.power, ref.power))
            # deterministic: a re-used config is bit-identical call after call (test/periodograms.jl:222-224)
            assert np.array_equal(np.asarray(d.welch_pgram(s, cfg).power), np.asarray(got.power))


def test_gx_welch_long_streams_flush_and_channels(d):
    """Three channels in one call, each long enough that every workgroup folds its LDS sums into its Float64 partial row several times (64 units
    between flushes; 256 CUs x 4 workgroups / 3 channels = 341 groups per channel -> more than 21 824 frame pairs per channel)."""
    from oracle import periodograms as opg, windows as ow
    rng = np.random.default_rng(62)
    n, K = 1125, 2 * 341 * 64 * 2 + 5
    hop = n - n // 2
    length = (K - 1) * hop + n
    s = np.stack([_signal(rng, length, np.float32) for _ in range(3)], axis=1)   # (len, channels)
    got = np.asarray(d.welch_pgram(s, n, n // 2, window=d.hanning).power)
    assert got.shape == (n // 2 + 1, 3)
    for c in range(3):
        ref = opg.welch_pgram(s[:, c], n, n // 2, window=ow.hanning, dtype=np.float64).power
        assert relerr(got[:, c], ref) < TOL32, (c, relerr(got[:, c], ref))
        assert ulps_of_max(got[:, c], ref) < _ulp_bound(n)


def test_lean_schedules_flush_their_float32_sums(d):
    """nfft 12500 (one workgroup per CU, Float32 sums flushed to the Float64 partials every 64 frame pairs) and 25000 = 2 x 12500 (the same rows behind the column
    step): four channels, each long enough that every workgroup flushes more than once (256 / 4 = 64 workgroups per channel -> more than 4096 frame pairs per
    channel), against the Float64 oracle applied chunk by chunk (tests/fullsize.py)."""
    from oracle import windows as ow
    from fullsize import oracle_welch_chunked
    rng = np.random.default_rng(66)
    for n, K in ((12500, 2 * 64 * 64 * 2 + 3), (25000, 2 * 32 * 64 * 2 + 3)):
        hop = n - n // 2
        length = (K - 1) * hop + n
        s = np.stack([_signal(rng, length, np.float32) for _ in range(4)], axis=1)
        got = np.asarray(d.welch_pgram(s, n, n // 2, window=d.hanning).power)
        assert got.shape == (n // 2 + 1, 4)
        for c in (0, 3):
            ref, frames = oracle_welch_chunked(lambda lo, hi: s[lo:hi, c], length, n, n // 2, ow.hanning, chunk_frames=1024)
            assert frames == K
            assert relerr(got[:, c], ref) < TOL32, (n, c, relerr(got[:, c], ref))
            assert ulps_of_max(got[:, c], ref) < _ulp_bound(n)


def test_two_kernel_rows_form_in_chunks_and_channels(d):
    """nfft = 150000 = 10 x 15000 in two kernels with a work buffer of 16 MiB (MDSP_BIG_CHUNK_MIB): two channels, 61 frames = 31 frame pairs in chunks of six -- the
    row kernel's partial sums persist from chunk to chunk."""
    from oracle import periodograms as opg, windows as ow
    from dsp_jl_amd import _lib
    rng = np.random.default_rng(67)
    n, K = 150000, 61
    hop = n - n // 2
    length = (K - 1) * hop + n + 17
    s = np.stack([_signal(rng, length, np.float32) for _ in range(2)], axis=1)
    _lib.set_tunable("MDSP_BIG_CHUNK_MIB", 16)
    try:
        got = np.asarray(d.welch_pgram(s, n, n // 2, window=d.hamming).power)
    finally:
        _lib.set_tunable("MDSP_BIG_CHUNK_MIB", None)
    assert got.shape == (n // 2 + 1, 2)
    for c in range(2):
        ref = opg.welch_pgram(s[:, c], n, n // 2, window=ow.hamming, dtype=np.float64).power
        assert relerr(got[:, c], ref) < TOL32, (c, relerr(got[:, c], ref))
        assert ulps_of_max(got[:, c], ref) < _ulp_bound(n)


@pytest.mark.parametrize("dt,tol", DTYPES)
def test_gx_stft_spectrogram_periodogram_vs_oracle(d, dt, tol):
    from oracle import periodograms as opg, windows as ow
    rng = np.random.default_rng(63)
    cplx = np.dtype(dt).kind == "c"
    f32 = dt in (np.float32, np.complex64)
    # ... and the column modes of the single-workgroup compile-time schedules (csrc/spectral_ctbig_cols.hip): real signals to 9600 points, complex ones to 16384
    extra = ((3087, 1500, 3087, "hanning", 5), (9600, 4800, 9600, "hamming", 4), (9000, 2000, 9216, "hanning", 3), (10240, 5120, 10240, "hamming", 3), (15625, 7000, 15625, None, 3))
    for (n, nov, nfft, wname, K) in WELCH_CASES[:3] + WELCH_CASES[5:8] + WELCH_CASES[10:11] + extra:
        win = getattr(ow, wname) if wname else None
        dwin = getattr(d, wname) if wname else None
        length = (K - 1) * (n - nov) + n + 5
        s = _signal(rng, length, dt)
        for onesided in ((False,) if cplx else (True, False)):
            got = d.stft(s, n, nov, nfft=nfft, window=dwin, onesided=onesided)
            ref = opg.stft(s, n, nov, nfft=nfft, window=win, onesided=onesided, dtype=np.float64)
            assert got.shape == ref.shape == ((nfft // 2 + 1) if onesided else nfft, K)
            assert relerr(got, ref) < tol, (n, nfft, onesided, relerr(got, ref))
            if f32:
                assert ulps_of_max(got, ref, axis=0) < _ulp_bound(nfft)
            sp = d.spectrogram(s, n, nov, nfft=nfft, window=dwin, onesided=onesided, fs=3.0)
            rs = opg.spectrogram(s, n, nov, nfft=nfft, window=win, onesided=onesided, fs=3.0, dtype=np.float64)
            assert sp.power.shape == rs.power.shape and relerr(sp.power, rs.power) < tol, (n, nfft, onesided)
            assert np.array_equal(sp.time, rs.time) and np.array_equal(sp.freq, rs.freq)
        x = s[:n]
        pg = d.periodogram(x, nfft=nfft, window=dwin, fs=2.0)
        rp = opg.periodogram(x, nfft=nfft, window=win, fs=2.0, dtype=np.float64)
        assert pg.power.shape == rp.power.shape and relerr(pg.power, rp.power) < tol, (n, nfft)


def test_float64_real_columns_on_one_buffer(d):
    """Float64 real-signal STFT / spectrogram at 6000 and 8000 points: the compile-time schedules of csrc/ct_sched.h on ONE LDS buffer of 16-byte elements (the last pass
    leaves the natural-order spectrum over its operands, the two frames of a transform are untangled from there) -- two buffers end at 5000 points."""
    from oracle import periodograms as opg, windows as ow
    rng = np.random.default_rng(69)
    for n, nov, K in ((6000, 3000, 5), (8000, 2000, 4)):
        s = _signal(rng, (K - 1) * (n - nov) + n + 3, np.float64)
        for onesided in (True, False):
            got = d.stft(s, n, nov, window=d.hanning, onesided=onesided)
            ref = opg.stft(s, n, nov, window=ow.hanning, onesided=onesided, dtype=np.float64)
            assert got.shape == ref.shape and relerr(got, ref) < TOL64, (n, onesided, relerr(got, ref))
        sp = d.spectrogram(s, n, nov, window=d.hanning, fs=3.0)
        rs = opg.spectrogram(s, n, nov, window=ow.hanning, fs=3.0, dtype=np.float64)
        assert relerr(sp.power, rs.power) < TOL64


def test_multitaper_on_the_compile_time_columns(d):
    """mt_pgram of Float32 / ComplexF32 signals of 3087 and 6912 samples (nfft = nextfastfft(n) = n: single-workgroup compile-time columns,
    csrc/spectral_ctbig_cols.hip): one launch per taper, each with its own window, accumulated into the same PSD (multitaper.jl:240-243)."""
    from oracle import multitaper as omt
    rng = np.random.default_rng(68)
    for n in (3087, 6912):
        for dt in (np.float32, np.complex64):
            x = _signal(rng, n, dt)
            got = d.mt_pgram(x, nw=4, ntapers=5, fs=2.0)
            ref_power, ref_freq = omt.mt_pgram(x.astype(np.float64 if dt == np.float32 else np.complex128), nw=4, ntapers=5, fs=2.0)
            assert np.array_equal(got.freq, ref_freq)
            assert relerr(got.power, ref_power) < TOL32, (n, dt, relerr(got.power, ref_power))


def test_gx_default_calls_on_mid_size_signals(d):
    """welch_pgram(s) / spectrogram(s) / periodogram(s) with nothing but the signal at the lengths VERDICT r5 names: 10^5 samples -> n = nfft = 12500;
    2^17 -> 16384; 160000 -> 20000; 320000 -> 40000."""
    from oracle import periodograms as opg, util as outil
    rng = np.random.default_rng(64)
    for length in (100000, 1 << 17, 160000, 320000):
        s = _signal(rng, length, np.float32)
        n = length >> 3
        assert outil.nextfastfft(n) in (12500, 16384, 20000, 40000)
        with pytest.warns(DeprecationWarning):
            got = d.welch_pgram(s)
        ref = opg.welch_pgram(s, n, n >> 1, nfft=outil.nextfastfft(n), window=None, dtype=np.float64)
        assert got.power.shape == ref.power.shape and relerr(got.power, ref.power) < TOL32
        assert ulps_of_max(got.power, ref.power) < _ulp_bound(outil.nextfastfft(n))
