"""GPU parity tests of the multi-pass spectral engine (dsp.jl_amd/csrc/bigfft.hip): welch_pgram / spectrogram / stft / periodogram with the
transform sizes the reference's DEFAULT arguments produce -- n = length(s) >> 3, nfft = nextfastfft(n) (periodograms.jl:560, :647, :828, :872) --
which no single workgroup holds.  Every result is compared with the Float64 oracle:

    Float32 / ComplexF32:  norm-wise <= 5e-6, and element-wise |err| <= 8 log2(nfft) x 2^-24 x max|ref|   (a Float32 FFT's rounding grows ~ log2 N)
    Float64 / ComplexF64:  norm-wise <= 1e-12

Frame boundaries, frame counts, frequency / time axes: bit-exact.  The parity cases ask for the engine by name (engine=ENGINE_FUSED: every size it
plans, also those AUTO leaves to rocFFT because rocFFT measured faster) and assert that they got it; what AUTO takes is asserted separately."""
import math
import os

import numpy as np
import pytest

from conftest import relerr, ulps_of_max

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


def _signal(rng, length, dt):
    s = rng.standard_normal(length) + 0.5 * np.sin(2 * np.pi * 0.1234 * np.arange(length))
    if np.dtype(dt).kind == "c":
        return (s + 1j * rng.standard_normal(length)).astype(dt)
    return s.astype(dt)


def _ulp_bound(nfft):
    return 8.0 * math.log2(nfft)


CASES = (  # n, noverlap, nfft, window, frames
    (16384, 8192, 16384, "hanning", 7),       # two passes of 128; odd frame count: the last transform carries one real frame
    (65536, 32768, 65536, "hanning", 4),      # 256 x 256
    (100000, 50000, 125000, "hamming", 3),    # zero-padded to 2^3 5^6 = 250 x 500 (what welch_pgram(randn(10^6)) asks for is n = nfft = 125000)
    (12500, 0, 12500, None, 9),               # 100 x 125, no window, no overlap
    (30375, 30000, 30375, "hanning", 6),      # odd transform (3^5 5^3 = 135 x 225): partial tiles in every pass, hop of 375 samples
    (262144, 131072, 524288, "hanning", 3),   # zero-padded; Welch: the rows form (64 x 8192, Float64 128 x 4096: column pass + single-workgroup Welch kernel over
                                              # the rows), columns: three passes (64 x 64 x 128)
    (200000, 100000, 262144, "hamming", 5),   # Welch rows form with 32 (Float64: 64) rows, odd frame count
)


@pytest.mark.parametrize("dt,tol", [(np.float32, TOL32), (np.float64, TOL64), (np.complex64, TOL32), (np.complex128, TOL64)])
def test_large_nfft_welch_vs_oracle(d, dt, tol):
    from oracle import periodograms as opg, windows as ow
    rng = np.random.default_rng(51)
    cplx = np.dtype(dt).kind == "c"
    f32 = dt in (np.float32, np.complex64)
    for (n, nov, nfft, wname, K) in CASES:
        win = getattr(ow, wname) if wname else None
        dwin = getattr(d, wname) if wname else None
        length = (K - 1) * (n - nov) + n + 17
        s = _signal(rng, length, dt)
        for onesided in ((False,) if cplx else (True, False)):
            cfg = d.WelchConfig(length, dt, n=n, noverlap=nov, nfft=nfft, window=dwin, onesided=onesided, fs=2.5, engine=d.ENGINE_FUSED)
            assert cfg.engine == d.ENGINE_FUSED
            # what AUTO takes (round 6): never the rocFFT pipeline for a 7-smooth size -- the run-time-schedule kernel (csrc/gx_kernels.h) up to 8 x 8192
            # points, the multi-pass engine beyond and for the powers of two from 32768
            auto = d.WelchConfig(length, dt, n=n, noverlap=nov, nfft=nfft, window=dwin, onesided=onesided, fs=2.5)
            assert auto.engine == d.ENGINE_FUSED, nfft
            got = d.welch_pgram(s, cfg)
            ref = opg.welch_pgram(s, n, nov, nfft=nfft, window=win, onesided=onesided, fs=2.5, dtype=np.float64)
            assert got.power.dtype == (np.float32 if f32 else np.float64)
            assert got.power.shape == ref.power.shape and np.array_equal(got.freq, ref.freq)
            e = relerr(got.power, ref.power)
            assert e < tol, (n, nov, nfft, onesided, e)
            if f32:
                assert ulps_of_max(got.power, ref.power) < _ulp_bound(nfft), (n, nfft, ulps_of_max(got.power, ref.power))
            # the rocFFT pipeline (north_star's literal form) agrees with the same oracle, and the two engines with each other
            roc = d.welch_pgram(s, n, nov, nfft=nfft, window=dwin, onesided=onesided, fs=2.5, engine=d.ENGINE_ROCFFT)
            assert relerr(roc.power, ref.power) < tol
            # a re-used config is bit-identical call after call (test/periodograms.jl:222-224)
            assert np.array_equal(np.asarray(d.welch_pgram(s, cfg).power), np.asarray(got.power))


@pytest.mark.parametrize("dt,tol", [(np.float32, TOL32), (np.float64, TOL64), (np.complex64, TOL32), (np.complex128, TOL64)])
def test_large_nfft_stft_spectrogram_periodogram_vs_oracle(d, dt, tol):
    from oracle import periodograms as opg, windows as ow
    rng = np.random.default_rng(53)
    cplx = np.dtype(dt).kind == "c"
    f32 = dt in (np.float32, np.complex64)
    for (n, nov, nfft, wname, K) in CASES[:5]:
        win = getattr(ow, wname) if wname else None
        dwin = getattr(d, wname) if wname else None
        length = (K - 1) * (n - nov) + n + 5
        s = _signal(rng, length, dt)
        for onesided in ((False,) if cplx else (True, False)):
            got = d.stft(s, n, nov, nfft=nfft, window=dwin, onesided=onesided, engine=d.ENGINE_FUSED)
            ref = opg.stft(s, n, nov, nfft=nfft, window=win, onesided=onesided, dtype=np.float64)
            assert got.shape == ref.shape == ((nfft // 2 + 1) if onesided else nfft, K)
            assert relerr(got, ref) < tol, (n, nfft, onesided, relerr(got, ref))
            if f32:
                assert ulps_of_max(got, ref, axis=0) < _ulp_bound(nfft)
            sp = d.spectrogram(s, n, nov, nfft=nfft, window=dwin, onesided=onesided, fs=3.0, engine=d.ENGINE_FUSED)
            rs = opg.spectrogram(s, n, nov, nfft=nfft, window=win, onesided=onesided, fs=3.0, dtype=np.float64)
            assert sp.power.shape == rs.power.shape and relerr(sp.power, rs.power) < tol
            assert np.array_equal(sp.time, rs.time) and np.array_equal(sp.freq, rs.freq)
        # periodogram: ONE frame (a real signal's transform carries a single frame)
        x = s[:n]
        pg = d.periodogram(x, nfft=nfft, window=dwin, fs=2.0, engine=d.ENGINE_FUSED)
        rp = opg.periodogram(x, nfft=nfft, window=win, fs=2.0, dtype=np.float64)
        assert pg.power.shape == rp.power.shape and relerr(pg.power, rp.power) < tol, (n, nfft)


def test_default_arguments_take_the_multipass_engine(d, torch):
    """welch_pgram(s) / spectrogram(s) / stft(s) / periodogram(s) with NOTHING but the signal: n = length >> 3, noverlap = n >> 1,
    nfft = nextfastfft(n) (periodograms.jl:560, :647, :828, :872); 10^6 samples -> n = nfft = 125000, 15 frames."""
    from oracle import periodograms as opg, util as outil
    rng = np.random.default_rng(55)
    for length in (10 ** 6, (1 << 20) + 12345):
        s = _signal(rng, length, np.float32)
        n = length >> 3
        nfft = outil.nextfastfft(n)
        K = opg.frame_count(length, n, n >> 1)
        assert K == (15 if n % 2 == 0 else 14)     # hop = n - (n >> 1): an odd n loses the fifteenth frame
        cfg = d.WelchConfig(length, np.float32, window=None)
        assert (cfg.nsamples, cfg.noverlap, cfg.nfft, cfg.engine) == (n, n >> 1, nfft, d.ENGINE_FUSED)
        got = d.welch_pgram(s, window=None)
        ref = opg.welch_pgram(s, n, n >> 1, window=None, dtype=np.float64)
        assert got.power.shape == (nfft // 2 + 1,) and np.array_equal(got.freq, ref.freq)
        assert relerr(got.power, ref.power) < TOL32 and ulps_of_max(got.power, ref.power) < _ulp_bound(nfft)
        sp = d.spectrogram(s)
        rs = opg.spectrogram(s, n, n >> 1, dtype=np.float64)
        assert sp.power.shape == rs.power.shape == (nfft // 2 + 1, K) and relerr(sp.power, rs.power) < TOL32
        assert np.array_equal(sp.time, rs.time)
        st = d.stft(s)
        rt = opg.stft(s, n, n >> 1, dtype=np.float64)
        assert relerr(st, rt) < TOL32 and ulps_of_max(st, rt, axis=0) < _ulp_bound(nfft)
        # device-resident input -> device-resident output, same numbers
        sd = torch.from_numpy(s).cuda()
        assert np.array_equal(d.welch_pgram(sd, window=None).power.cpu().numpy(), np.asarray(got.power))
    x = _signal(rng, 200000, np.float64)        # periodogram(s): one transform of nextfastfft(200000) = 200000 points
    pg = d.periodogram(x)
    rp = opg.periodogram(x, dtype=np.float64)
    assert relerr(pg.power, rp.power) < TOL64


@pytest.mark.parametrize("n,nov,nfft", [(20000, 10000, None), (250000, 125000, 262144)])   # (the second: 16 rows of 16384 points in two kernels, csrc/spectral_ctrows.hip -- up to round 5 the multi-pass engine's rows form, 32 rows of 8192 points)
def test_multichannel_streaming_and_multitaper_on_large_transforms(d, torch, n, nov, nfft):
    from oracle import periodograms as opg, windows as ow
    rng = np.random.default_rng(57)
    S = rng.standard_normal((n * 4 + 100, 3)).astype(np.float32)
    cfg = d.WelchConfig(S.shape[0], np.float32, n=n, noverlap=nov, nfft=nfft, window=d.hanning, engine=d.ENGINE_FUSED)
    assert cfg.engine == d.ENGINE_FUSED
    P = np.asarray(d.welch_pgram(S, cfg).power)
    for c in range(3):
        ref = opg.welch_pgram(S[:, c], n, nov, nfft=nfft, window=ow.hanning, dtype=np.float64)
        assert relerr(P[:, c], ref.power) < TOL32
        assert np.array_equal(P[:, c], np.asarray(d.welch_pgram(S[:, c].copy(), cfg).power))
    # the streaming form (reset / accumulate / finalize): two slices of whole frames == the one-shot call
    cols = torch.from_numpy(np.ascontiguousarray(S.T)).cuda()
    hop, K = n - nov, opg.frame_count(S.shape[0], n, nov)
    k1 = 3
    cfg.reset()
    cfg.accumulate(cols[:, : (k1 - 1) * hop + n].contiguous())
    cfg.accumulate(cols[:, k1 * hop: (K - 1) * hop + n].contiguous())
    assert cfg.frames_accumulated() == K
    out = cfg.finalize(nch=3).cpu().numpy()
    assert relerr(out.T, P) < 1e-6
    if nfft is not None:
        return
    # multitaper PSD: one pass set per taper, accumulated into the same output (mt_pgram!, multitaper.jl:240-243)
    x = rng.standard_normal(20000)
    got = d.mt_pgram(x, nw=4, ntapers=5, engine=d.ENGINE_FUSED)
    from oracle import multitaper as omt
    ref_power, ref_freq = omt.mt_pgram(x, nw=4, ntapers=5)
    assert relerr(got.power, ref_power) < TOL64 and np.array_equal(got.freq, ref_freq)


def test_welch_default_2p27_vs_oracle(d, torch):
    """The acceptance case of the round: welch_pgram(s) with DEFAULT n on a 2^27-sample Float32 device stream -- n = nfft = 2^24, 15 frames of
    16 Mi points, 50 % overlap; three passes of 256.  The oracle transforms every frame in Float64 on the host."""
    from oracle import periodograms as opg
    lg = int(os.environ.get("MDSP_TEST_BIG_LOG2", 27))
    length = 1 << lg
    g = torch.Generator(device="cuda"); g.manual_seed(1776)
    s = torch.randn(length, generator=g, device="cuda", dtype=torch.float32)
    s += (0.5 * torch.sin(2 * math.pi * 0.1234 * torch.arange(length, device="cuda", dtype=torch.float64))).to(torch.float32)
    n = length >> 3
    P = d.welch_pgram(s, window=d.hanning).power
    assert P.shape == (n // 2 + 1,) and P.dtype == torch.float32
    ref = opg.welch_pgram(s.cpu().numpy(), n, n >> 1, window=__import__("oracle.windows", fromlist=["hanning"]).hanning, dtype=np.float64)
    got = P.double().cpu().numpy()
    assert relerr(got, ref.power) < TOL32
    assert ulps_of_max(got, ref.power) < _ulp_bound(n)
    nz = ref.power > 1e-3 * np.median(ref.power)
    assert np.max(np.abs(got[nz] - ref.power[nz]) / ref.power[nz]) < 1e-4     # bin-wise: no bin is off by more than Float32 rounding noise
