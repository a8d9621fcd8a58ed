"""GPU parity at BASELINE sizes against the CPU ORACLE (not against another GPU engine).

    config 2   filt(b, x), 256 taps, nfft 2048, 2^30 Float32 samples                      EVERY output vs the Float64 oracle (round 6)
    config 3   welch_pgram nfft = 4096, hanning, 50 % overlap, 2^30 Float32 samples      whole PSD vs the Float64 oracle
    config 4   stft / spectrogram nfft = 1024, hop = 256, 8 channels x 2^26 ComplexF32   oracle columns + every column's energy
    config 5   resample 160//147, 5120 taps, 4 channels x 2^28 Float32                   oracle output windows at depth

The oracle is applied piecewise (tests/fullsize.py, pinned to the one-shot oracle by tests/test_fullsize_helpers.py).
Tolerances as in test_gpu_parity.py: Float32 FFT paths <= 5e-6, Float32 polyphase <= 2e-6 (norm-wise, vs Float64 oracle).
MDSP_TEST_STREAM / MDSP_TEST_C4_LOG2 / MDSP_TEST_C5_LOG2 shrink the streams for debugging only.
"""
import math
import os
from fractions import Fraction

import numpy as np
import pytest

import fullsize as fz
from conftest import relerr, ulps_of_max

pytestmark = pytest.mark.gpu

TOL32 = 5e-6
TOLFIR = 2e-6
# element-wise bounds, in Float32 unit roundoffs (2^-24) of the largest magnitude of the column / window (conftest.ulps_of_max):
#   FFT columns      <= ULP_FFT * log2(nfft)            (one transform: every output is a sum of nfft terms through log2(nfft) butterfly layers)
#   |X|^2 columns    <= 2 ULP_FFT * log2(nfft)
#   polyphase output <= ULP_FIR * sqrt(taps per phase)  (one dot product of tapsPerPhi products, FMA-accumulated in Float32)
# measured on MI355X (gpurun_out/s4, s7: config 4 STFT columns 2.8 = 0.28 log2(1024), spectrogram 4.9, config 5 windows 5.3 = 0.94 sqrt(32)): the
# bounds below leave a factor of two to four.
ULP_FFT = 1.0
ULP_FIR = 2.0


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


def test_config2_filt_2p30_whole_stream_vs_oracle(d, torch):
    """Filters/filt.jl:479-521 on the full stream (config 2 used to be compared with the oracle on nine 600-sample windows only): all 2^30
    outputs of the fused overlap-save kernel against the Float64 oracle, evaluated piecewise (tests/fullsize.py oracle_filt_chunked: the oracle's own
    fftfilt on slices with their nb - 1 samples of history).  Per chunk: norm-wise error, the largest element-wise error in Float32 unit roundoffs of the
    chunk's largest output, and a Float64 block-sum checksum; over the stream: the same three."""
    n = int(os.environ.get("MDSP_TEST_STREAM", 2 ** 30))
    g = torch.Generator(device="cuda"); g.manual_seed(1776)
    x = torch.randn(n, generator=g, device="cuda", dtype=torch.float32)
    from test_gpu_parity import _lowpass_taps
    b = _lowpass_taps(256, np.float32)
    y = d.fftfilt(b, x, 2048, engine=d.ENGINE_FUSED)
    assert y.shape == (n,) and y.dtype == torch.float32
    err2 = ref2 = 0.0
    worst_u = 0.0
    sum_got = sum_ref = 0.0
    for lo, hi, ref in fz.oracle_filt_chunked(lambda lo, hi: x[lo:hi].cpu().numpy(), n, b, chunk=1 << 24):
        got = y[lo:hi].cpu().numpy().astype(np.float64)
        e = got - ref
        err2 += float(e @ e)
        ref2 += float(ref @ ref)
        assert np.sqrt(float(e @ e) / float(ref @ ref)) < TOL32, (lo, hi)
        u = ulps_of_max(got, ref)
        worst_u = max(worst_u, u)
        assert u < 2 * ULP_FFT * 11, (lo, hi, u)     # two 2048-point transforms and a spectrum product per output: 2 log2(nfft)
        sum_got += float(got.sum())
        sum_ref += float(ref.sum())
    assert np.sqrt(err2 / ref2) < TOL32
    # the checksum of all 2^30 outputs, in Float64: rounding errors of random sign do not add up (a block of 1793 missing outputs would show as ~40 sigma
    # against the 3 sigma this admits)
    assert abs(sum_got - sum_ref) <= 1e-4 * np.sqrt(ref2), (sum_got, sum_ref)
    print("config 2, whole stream: norm-wise", np.sqrt(err2 / ref2), "worst element", worst_u, "unit roundoffs of the chunk maximum")


def test_config3_welch_2p30_vs_oracle(d, torch):
    """periodograms.jl:746-759 on the full stream: K = 524287 frames; each of the 512 transform slots of the fused kernel
    folds its Float32 pair accumulators into Float64 four times (FLUSH = 128 units) -- the whole 2049-bin PSD is compared
    with the Float64 oracle, which sums every frame."""
    from oracle import windows as ow
    n = int(os.environ.get("MDSP_TEST_STREAM", 2 ** 30))
    g = torch.Generator(device="cuda"); g.manual_seed(1776)
    s = torch.randn(n, generator=g, device="cuda", dtype=torch.float32)
    t = torch.arange(n, device="cuda", dtype=torch.float64)
    s += (0.5 * torch.sin(2 * math.pi * 0.1234 * t)).to(torch.float32)
    del t
    P = d.welch_pgram(s, 4096, 2048, window=d.hanning, engine=d.ENGINE_FUSED).power
    assert P.shape == (2049,) and P.dtype == torch.float32
    ref, K = fz.oracle_welch_chunked(lambda lo, hi: s[lo:hi].cpu().numpy(), n, 4096, 2048, ow.hanning, chunk_frames=1 << 14)
    assert K == (n - 4096) // 2048 + 1
    got = P.double().cpu().numpy()
    assert relerr(got, ref) < TOL32
    assert np.max(np.abs(got - ref) / ref) < 2e-5            # bin-wise too: no bin is off by more than Float32 rounding noise
    # the rocFFT engine (north_star's literal pipeline) against the same oracle
    P2 = d.welch_pgram(s, 4096, 2048, window=d.hanning, engine=d.ENGINE_ROCFFT).power.double().cpu().numpy()
    assert relerr(P2, ref) < TOL32


def test_config4_stft_8x2p26_vs_oracle(d, torch):
    """periodograms.jl:872-897 at one GPU's share of config 4: 8 channels x 2^26 ComplexF32 -> 8 x (1024 x 262141)
    ComplexF32 (16 GiB; channel 2 starts at byte offset 2^32, channel 7 ends past 2^34).  Oracle columns at frames
    {0, 1, K/2, K/2+1, K-2, K-1} of channels {0, 1, 2, 7}; every column of every channel through Parseval
    (sum |S|^2 = nfft sum |w s|^2, evaluated independently with torch); then the same for spectrogram (PSD)."""
    from oracle import windows as ow
    lg = int(os.environ.get("MDSP_TEST_C4_LOG2", 26))
    nch, n = 8, 2 ** lg
    g = torch.Generator(device="cuda"); g.manual_seed(1776)
    cols = torch.view_as_complex(torch.randn((nch, n, 2), generator=g, device="cuda", dtype=torch.float32) * math.sqrt(0.5))
    s = cols.t()                                     # (n, nch): Julia's column-major (n, nch) matrix, no copy
    K = d.frame_count(n, 1024, 768)
    assert lg != 26 or K == 262141
    w = torch.from_numpy(ow.hanning(1024)).cuda()
    w2 = (w * w).float()
    frames_e = []                                    # per-channel frame energies sum |w s|^2, straight from the samples
    for c in range(nch):
        a2 = (cols[c].real ** 2 + cols[c].imag ** 2)
        frames_e.append((a2.unfold(0, 1024, 256) * w2).sum(dim=1).double())
        del a2
    spots = [0, K // 2, K - 2]
    chans = [0, 1, 2, 7]
    worst_ulps = {}

    S = d.stft(s, 1024, 768, window=d.hanning, engine=d.ENGINE_FUSED)
    assert S.shape == (1024, K, nch) and S.dtype == torch.complex64
    for c in chans:
        get = lambda lo, hi, c=c: cols[c, lo:hi].cpu().numpy()
        for f0 in spots:
            ref = fz.oracle_stft_columns(get, 1024, 768, f0, 2, ow.hanning)
            assert relerr(S[:, f0:f0 + 2, c].cpu().numpy(), ref) < TOL32, (c, f0)
            u = ulps_of_max(S[:, f0:f0 + 2, c].cpu().numpy(), ref, axis=0)          # every bin, against its own column's largest magnitude
            worst_ulps["stft"] = max(worst_ulps.get("stft", 0.0), u)
            assert u < ULP_FFT * 10, (c, f0, u)
    for c in range(nch):
        e = (S[:, :, c].real.double() ** 2 + S[:, :, c].imag.double() ** 2).sum(dim=0)
        assert float(((e / 1024 - frames_e[c]).abs() / frames_e[c]).max()) < 2e-5, c
        del e
    del S

    sp = d.spectrogram(s, 1024, 768, window=d.hanning, fs=2.0, engine=d.ENGINE_FUSED)
    P = sp.power
    assert P.shape == (1024, K, nch) and P.dtype == torch.float32
    r = 2.0 * float((w * w).sum())
    for c in chans:
        get = lambda lo, hi, c=c: cols[c, lo:hi].cpu().numpy()
        for f0 in spots:
            ref = fz.oracle_stft_columns(get, 1024, 768, f0, 2, ow.hanning, psdonly=True, fs=2.0)
            assert relerr(P[:, f0:f0 + 2, c].cpu().numpy(), ref) < TOL32, (c, f0)
            u = ulps_of_max(P[:, f0:f0 + 2, c].cpu().numpy(), ref, axis=0)          # |X|^2: twice the relative error of X
            worst_ulps["spectrogram"] = max(worst_ulps.get("spectrogram", 0.0), u)
            assert u < 2 * ULP_FFT * 10, (c, f0, u)
    for c in range(nch):
        e = P[:, :, c].double().sum(dim=0) * r
        assert float(((e / 1024 - frames_e[c]).abs() / frames_e[c]).max()) < 2e-5, c
        del e
    assert np.array_equal(sp.time[:3], (512 + np.arange(3) * 256) / 2.0)
    print("config 4 element-wise error, Float32 unit roundoffs of the column maximum:", worst_ulps)


def test_config5_resample_4x2p28_vs_oracle(d, torch):
    """stream_filt.jl:476-515 / :688-725 at one GPU's share of config 5: 4 channels x 2^28 Float32 -> 292174646 outputs each.
    600-output oracle windows (Float64; the state of the reference's loop at the window start from the oracle's closed form) at
    the start, inside a tile, across tile boundaries of the matrix-core kernel that runs this shape (rows x outputs per row from
    mdsp_fir_mm_geometry: 64 rows x 160 outputs; tiles are grid-strided over workgroups, so seams deep in the stream belong to
    different workgroups than their neighbours) and of the retired register-tap kernel (33 x 160), where the output byte offset
    crosses 2^30 / 2^32 (channel 3), at pseudo-random depths, and at the very end (zero-padded tail, stream_filt.jl:699)."""
    from oracle import design as odes
    lg = int(os.environ.get("MDSP_TEST_C5_LOG2", 28))
    nch, n = 4, 2 ** lg
    ratio = Fraction(160, 147)
    h = odes.resample_filter(ratio)
    h = np.resize(h, 5120).astype(np.float32) if len(h) >= 5120 else np.concatenate([h, np.zeros(5120 - len(h))]).astype(np.float32)
    g = torch.Generator(device="cuda"); g.manual_seed(1776)
    cols = torch.randn((nch, n), generator=g, device="cuda", dtype=torch.float32)
    y = d.resample(cols.t(), ratio, h, dims=0)
    nout = fz.resample_output_length(n, ratio)
    assert lg != 28 or nout == 292174646
    assert y.shape == (nout, nch) and y.dtype == torch.float32
    import ctypes as C
    from dsp_jl_amd import _lib
    geo = (C.c_int64 * 12)()
    _lib.check(_lib.lib().mdsp_fir_mm_geometry(160, 147, len(h), _lib.F32, _lib.F32, geo))
    fits, RB, Lr, Mr, NB, NG, steps, CH = list(geo)[:8]
    assert fits and Lr == 160
    tile = int(Lr * 16 * CH * NG)                  # outputs per tile of polyphase_mfma_kernel (config 5: 64 rows x 160 = 10240)
    assert lg != 28 or tile == 10240
    old_tile = 33 * 160                            # the register-tap kernel's tile (kept: its seams are ordinary positions now)
    rng = np.random.default_rng(1776)
    ntiles = nout // tile
    spots = [0, 1, tile // 2, tile - 300, 2 * tile - 300, 7 * tile - 300, 255 * tile - 300, 256 * tile - 300, 257 * tile - 300, 1023 * tile - 300,
             (ntiles // 2) * tile - 300, (ntiles - 1) * tile - 300, ntiles * tile - 300, old_tile - 300, 7 * old_tile - 300,
             2 ** 28 - 300, 2 ** 30 - 3 * nout - 300, nout - 600]      # 2^30 - 3 nout: channel 3's output crosses byte offset 2^32
    spots += [int(k) * tile - 300 for k in rng.integers(1, max(2, ntiles), size=6)]     # seams at pseudo-random depths
    spots += [int(v) for v in rng.integers(0, nout - 600, size=12)]
    worst = worst_u = 0.0
    for c in (0, 3):
        get = lambda lo, hi, c=c: cols[c, lo:hi].cpu().numpy()
        for m0 in spots:
            m0 = max(0, min(m0, nout - 600))
            ref = fz.oracle_resample_window(get, n, ratio, h.astype(np.float64), m0, 600)
            e = relerr(y[m0:m0 + 600, c].cpu().numpy(), ref)
            worst = max(worst, e)
            assert e < TOLFIR, (c, m0, e)
            u = ulps_of_max(y[m0:m0 + 600, c].cpu().numpy(), ref)                   # every output of the window
            worst_u = max(worst_u, u)
            assert u < ULP_FIR * math.sqrt(32), (c, m0, u)
    print("config 5 element-wise error, Float32 unit roundoffs of the window maximum:", worst_u, "norm-wise", worst)
    # every output is finite and the output power matches the input power scaled by the filter's passband gain
    # (white noise through a unit-passband-gain interpolator: var(y) ~ var(x) * sum(h^2) * L / L^2 ... checked loosely)
    for c in range(nch):
        assert bool(torch.isfinite(y[:, c]).all())
        v = float(y[:, c].double().pow(2).mean())
        vref = float((np.asarray(h, dtype=np.float64) ** 2).sum() / 160.0)
        assert abs(v / vref - 1.0) < 5e-3, (c, v, vref)
