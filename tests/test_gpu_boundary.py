"""GPU tests of the boundary pieces added in round 2 (run with -m gpu on an MI355X):

  * the RCCL communicator behind the C ABI (mdsp_comm_*, mdsp_welch_mean_allreduce, mdsp_welch_allreduce) with one rank -- the box
    has one GPU; the world-2 logic of the same product functions runs under gloo in tests/test_dist_gloo.py;
  * the streaming Welch protocol (reset / accumulate / finalize) against the one-shot call and the oracle;
  * block-range overlap-save (mdsp_ols_exec_range): bit-identical to the whole-signal call;
  * the host-pointer pipelines (mdsp_ols_exec_host / mdsp_welch_exec_host), pageable and page-locked arrays, both engines.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu

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


def _taps(n, dtype):
    from oracle import design, windows
    return design.digitalfilter_lowpass_firwindow(0.25, windows.hamming(n)).astype(dtype)


def test_rccl_communicator_behind_the_c_abi_single_rank(d, torch):
    from dsp_jl_amd import _lib, _dev
    comm = d.Comm.single()                                  # ncclGetUniqueId + ncclCommInitRank(1 rank) inside libmi355dsp
    assert (comm.rank, comm.nranks) == (0, 1)
    r, n = C.c_int(-1), C.c_int(-1)
    _lib.check(_lib.lib().mdsp_comm_info(comm._h, C.byref(r), C.byref(n)))
    assert (r.value, n.value) == (0, 1)
    for dt in (torch.float32, torch.float64):
        t = torch.arange(5000, device="cuda", dtype=dt) * 0.25
        ref = t.clone()
        comm.allreduce_sum(t)                               # a real ncclAllReduce launch on the current stream
        torch.cuda.synchronize()
        assert torch.equal(t, ref)
    # cross-channel Welch mean through ONE C-ABI call == mean of the per-channel PSDs
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    S = torch.randn((6, 50000), generator=g, device="cuda", dtype=torch.float32)
    cfg = d.WelchConfig(50000, np.float32, n=1024, noverlap=512, window=d.hanning)
    mean = d.welch_channel_mean(S, cfg, comm=comm)
    per = d.welch_pgram(S.t(), cfg).power
    assert float((mean.double() - per.double().mean(dim=1)).norm() / mean.double().norm()) < 1e-6
    assert torch.allclose(mean, d.welch_channel_mean(S, cfg), rtol=1e-6, atol=0)   # and == the torch.distributed-transport form on one rank
    half = d.welch_channel_mean(S[:3], cfg, nch_total=6, comm=comm)           # a rank holding 3 of 6 channels contributes sum/6
    assert float((half.double() - per[:, :3].double().sum(dim=1) / 6).norm() / half.double().norm()) < 1e-6
    none = d.welch_channel_mean(S[:0], cfg, nch_total=6, comm=comm)           # a rank without channels contributes zeros
    assert float(none.abs().max()) == 0.0
    # time split on one rank == whole-stream PSD (accumulate -> all-reduce of the Float64 sums -> finalize with the total frame count)
    x = S[0]
    K = d.frame_count(x.numel(), 1024, 512)
    p = d.welch_time_split(x, K, 1024, 512, window=d.hanning, comm=comm)
    assert float((p.double() - per[:, 0].double()).norm() / per[:, 0].double().norm()) < 1e-6
    cfg.reset(); cfg.accumulate(S[:1].contiguous())
    _lib.check(_lib.lib().mdsp_welch_allreduce(cfg._h, comm._h, _dev.stream_ptr()))
    assert cfg.frames_accumulated() == K
    comm.close()
    with pytest.raises(d.ArgumentError):
        d.Comm(b"x" * 5, 0, 1)


@pytest.mark.parametrize("engine", [1, 2], ids=["fused", "rocfft"])
@pytest.mark.parametrize("dt,tol", [(np.float32, TOL32), (np.float64, 1e-12), (np.complex64, TOL32)])
def test_welch_streaming_accumulate_equals_one_shot(d, torch, engine, dt, tol):
    from oracle import periodograms as opg, windows as ow
    rng = np.random.default_rng(11)
    n, nov, L = 512, 384, 200_000
    hop = n - nov
    x = rng.standard_normal((2, L)).astype(dt) if np.dtype(dt).kind != "c" else (rng.standard_normal((2, L)) + 1j * rng.standard_normal((2, L))).astype(dt)
    xd = torch.from_numpy(x).cuda()
    cfg = d.WelchConfig(L, dt, n=n, noverlap=nov, window=d.hanning, engine=engine)
    K = d.frame_count(L, n, nov)
    one = d.welch_pgram(xd.t(), cfg).power                      # (nout, 2)
    cfg.reset()
    cuts = [0, 1, 7, 300, 301, K // 2, K]                      # uneven slices of whole frames, including single-frame slices
    for k0, k1 in zip(cuts[:-1], cuts[1:]):
        cfg.accumulate(xd[:, k0 * hop:(k1 - 1) * hop + n].contiguous())
    assert cfg.frames_accumulated() == K
    st = cfg.finalize(nch=2).t()
    assert float((st.double() - one.double()).norm() / one.double().norm()) < (1e-6 if np.dtype(dt).itemsize in (4, 8) and np.dtype(dt) != np.float64 else 1e-13)
    for c in range(2):
        ref = opg.welch_pgram(x[c], n, nov, window=ow.hanning, dtype=np.float64).power
        assert relerr(st[:, c].cpu().numpy(), ref) < tol
    # finalize with an explicit total: a rank that saw half the frames of a stream twice as long
    assert float((cfg.finalize(2 * K, nch=2).t().double() * 2 - st.double()).norm() / st.double().norm()) < 1e-6
    with pytest.raises(d.DimensionMismatch):
        cfg.accumulate(xd[:1, :n * 4].contiguous())             # channel count changed without a reset


@pytest.mark.parametrize("engine", [1, 2], ids=["fused", "rocfft"])
@pytest.mark.parametrize("dt", [np.float32, np.float64, np.complex64])
def test_ols_block_range_is_bit_identical(d, torch, engine, dt):
    from dsp_jl_amd import _lib, _dev
    from dsp_jl_amd.dspbase import OlsPlan
    rng = np.random.default_rng(3)
    nb, nfft, nx = 97, 512, 100_003
    L = nfft - nb + 1
    rdt = np.float32 if dt in (np.float32, np.complex64) else np.float64
    b = _taps(nb, rdt)
    x = rng.standard_normal(nx).astype(rdt) if np.dtype(dt).kind != "c" else (rng.standard_normal(nx) + 1j * rng.standard_normal(nx)).astype(dt)
    taps = b.astype(dt)
    plan = OlsPlan(taps, nfft, nx, _lib.OLS_FILT, engine)
    xd = torch.from_numpy(x).cuda()
    whole = plan.exec(xd.view(1, -1), nx)[0]
    nblocks = -(-nx // L)
    got = torch.empty_like(whole)
    for g0, cnt in ((0, 2), (2, 40), (42, 2), (44, nblocks)):       # even starts; the last range is clipped to the grid
        g1 = min(nblocks, g0 + cnt)
        lo, hi = max(0, g0 * L - (nb - 1)), min(nx, g1 * L)
        o0, o1 = g0 * L, min(nx, g1 * L)
        xs = xd[lo:hi].clone()                                   # a slice that holds nothing but what the blocks read
        ys = torch.empty(o1 - o0, dtype=xd.dtype, device="cuda")
        _lib.check(_lib.lib().mdsp_ols_exec_range(plan._h, _dev.ptr(xs), lo, hi - lo, nx, _dev.ptr(ys), g0, cnt, nx, _dev.stream_ptr()))
        got[o0:o1] = ys
    assert torch.equal(got, whole)
    if np.dtype(dt).kind != "c":
        with pytest.raises(d.ArgumentError):                     # odd first block: two real blocks share a transform
            _lib.check(_lib.lib().mdsp_ols_exec_range(plan._h, _dev.ptr(xd), 0, nx, nx, _dev.ptr(got), 1, 2, nx, _dev.stream_ptr()))
    with pytest.raises(d.ArgumentError):                         # slice does not cover what the blocks read
        _lib.check(_lib.lib().mdsp_ols_exec_range(plan._h, _dev.ptr(xd), 5000, 100, nx, _dev.ptr(got), 2, 4, nx, _dev.stream_ptr()))


@pytest.mark.parametrize("engine", [1, 2], ids=["fused", "rocfft"])
def test_host_pipeline_overlap_save(d, torch, engine, monkeypatch):
    """mdsp_ols_exec_host: chunked, double-buffered H2D || kernel || D2H; bit-identical to the device-resident call."""
    from dsp_jl_amd import _lib
    from dsp_jl_amd.dspbase import OlsPlan
    from oracle import dspbase as odsp
    rng = np.random.default_rng(5)
    _lib.set_tunable("MDSP_HOST_CHUNK_MIB", 1)                  # many chunks on a modest array: both lanes, drains, the last partial chunk
    try:
        for dt, nx, ncols in ((np.float32, 3_000_017, 1), (np.float64, 700_001, 3)):
            b = _taps(256, dt)
            x = rng.standard_normal((ncols, nx)).astype(dt)
            plan = OlsPlan(b, 2048, nx, _lib.OLS_FILT, engine)
            dev = plan.exec(torch.from_numpy(x).cuda(), nx).cpu().numpy()
            host = plan.exec_host(x, nx)                         # pageable numpy memory: staged through pinned buffers
            assert np.array_equal(host, dev)
            ref = odsp.filt_ba(b.astype(np.float64), 1.0, x[ncols - 1, :50000].astype(np.float64))
            assert relerr(host[ncols - 1, :50000], ref) < (TOL32 if dt == np.float32 else 1e-12)
            # page-locked arrays: no staging copies (MDSP_HOST_PINNED)
            nbytes = x.nbytes
            pin_in, pin_out = C.c_void_p(), C.c_void_p()
            _lib.check(_lib.lib().mdsp_host_alloc(C.byref(pin_in), nbytes)); _lib.check(_lib.lib().mdsp_host_alloc(C.byref(pin_out), nbytes))
            try:
                xin = np.ctypeslib.as_array(C.cast(pin_in, C.POINTER(C.c_byte)), shape=(nbytes,)).view(dt).reshape(ncols, nx)
                yout = np.ctypeslib.as_array(C.cast(pin_out, C.POINTER(C.c_byte)), shape=(nbytes,)).view(dt).reshape(ncols, nx)
                xin[...] = x
                _lib.check(_lib.lib().mdsp_ols_exec_host(plan._h, pin_in, nx, ncols, nx, pin_out, nx, nx, _lib.HOST_PINNED))
                assert np.array_equal(yout, dev)
            finally:
                _lib.lib().mdsp_host_free(pin_in); _lib.lib().mdsp_host_free(pin_out)
        # conv mode: nout = nx + nb - 1 (tail blocks read past the end of x)
        b = _taps(256, np.float32); x = rng.standard_normal(1_000_000).astype(np.float32)
        plan = OlsPlan(b, 2048, len(x), _lib.OLS_CONV, engine)
        nout = len(x) + 255
        dev = plan.exec(torch.from_numpy(x).cuda().view(1, -1), nout).cpu().numpy()
        assert np.array_equal(plan.exec_host(x.reshape(1, -1), nout), dev)
    finally:
        _lib.set_tunable("MDSP_HOST_CHUNK_MIB", None)
    # the numpy API takes this path on its own for large host arrays (>= 32 MiB): results equal the device-array call
    x = rng.standard_normal(9_000_000).astype(np.float32); b = _taps(256, np.float32)
    y_host = d.fftfilt(b, x, 2048, engine=engine)
    assert isinstance(y_host, np.ndarray) and np.array_equal(y_host, d.fftfilt(b, torch.from_numpy(x).cuda(), 2048, engine=engine).cpu().numpy())
    X = np.asfortranarray(rng.standard_normal((5_000_000, 2)).astype(np.float32))   # a Julia-layout (column-major) matrix
    Y = d.fftfilt(b, X, 2048, engine=engine)
    assert Y.shape == X.shape and np.array_equal(Y[:, 1], d.fftfilt(b, np.ascontiguousarray(X[:, 1]), 2048, engine=engine))


@pytest.mark.parametrize("engine", [1, 2], ids=["fused", "rocfft"])
def test_host_pipeline_welch(d, torch, engine):
    from dsp_jl_amd import _lib
    from oracle import periodograms as opg, windows as ow
    rng = np.random.default_rng(9)
    _lib.set_tunable("MDSP_HOST_CHUNK_MIB", 1)
    try:
        for dt, L, nch, n, nov, nfft in ((np.float32, 2_500_000, 1, 4096, 2048, 4096), (np.float32, 400_000, 3, 1000, 300, 1024), (np.float64, 300_000, 2, 512, 384, 512)):
            s = rng.standard_normal((nch, L)).astype(dt)
            cfg = d.WelchConfig(L, dt, n=n, noverlap=nov, nfft=nfft, window=d.hanning, engine=engine)
            host = cfg.exec_host(s)
            dev = d.welch_pgram(torch.from_numpy(s).cuda().t(), cfg).power.t().cpu().numpy()
            assert host.shape == dev.shape == (nch, cfg.nout)
            assert relerr(host, dev) < (1e-6 if dt == np.float32 else 1e-13)
            ref = opg.welch_pgram(s[nch - 1], n, nov, nfft=nfft, window=ow.hanning, dtype=np.float64).power
            assert relerr(host[nch - 1], ref) < (TOL32 if dt == np.float32 else 1e-12)
        short = rng.standard_normal((1, 100)).astype(np.float32)         # shorter than one frame: fill!(out, 0)
        cfg = d.WelchConfig(100, np.float32, n=256, noverlap=128, window=d.hanning, engine=engine)
        assert np.array_equal(cfg.exec_host(short), np.zeros((1, 129), np.float32))
    finally:
        _lib.set_tunable("MDSP_HOST_CHUNK_MIB", None)
    s = rng.standard_normal(9_000_000).astype(np.float32)                # numpy API: large host arrays take the pipeline
    p = d.welch_pgram(s, 4096, 2048, window=d.hanning, engine=engine).power
    assert isinstance(p, np.ndarray) and relerr(p, d.welch_pgram(torch.from_numpy(s).cuda(), 4096, 2048, window=d.hanning, engine=engine).power.cpu().numpy()) < 1e-6


def test_library_plan_cache(d, torch):
    """mdsp_*_plan_cached: the per-call fast path behind the C ABI (what the Julia twin's filt(b, x) / welch_pgram(s, n, noverlap) bind)."""
    from dsp_jl_amd import _lib, _dev
    lib = _lib.lib()
    _lib.check(lib.mdsp_plan_cache_clear())
    st = _dev.stream_ptr()
    rng = np.random.default_rng(0)
    b = rng.standard_normal(100).astype(np.float32)
    h1, h2, h3 = C.c_void_p(), C.c_void_p(), C.c_void_p()
    _lib.check(lib.mdsp_ols_plan_cached(C.byref(h1), b.ctypes.data_as(C.c_void_p), 100, 0, 10 ** 6, _lib.F32, _lib.OLS_FILT, 0, st))
    _lib.check(lib.mdsp_ols_plan_cached(C.byref(h2), b.copy().ctypes.data_as(C.c_void_p), 100, 0, 10 ** 6, _lib.F32, _lib.OLS_FILT, 0, st))
    assert h1.value == h2.value                                  # same contents -> same plan
    b2 = b.copy(); b2[7] += 1
    _lib.check(lib.mdsp_ols_plan_cached(C.byref(h3), b2.ctypes.data_as(C.c_void_p), 100, 0, 10 ** 6, _lib.F32, _lib.OLS_FILT, 0, st))
    assert h3.value != h1.value                                  # one tap differs -> another plan
    w = d.hanning(512)
    wp = w.ctypes.data_as(C.POINTER(C.c_double))
    p1, p2, p3 = C.c_void_p(), C.c_void_p(), C.c_void_p()
    _lib.check(lib.mdsp_welch_plan_cached(C.byref(p1), 512, 256, 512, wp, float((w * w).sum()), 1, _lib.F32, 0, st))
    _lib.check(lib.mdsp_welch_plan_cached(C.byref(p2), 512, 256, 512, wp, float((w * w).sum()), 1, _lib.F32, 0, st))
    _lib.check(lib.mdsp_stft_plan_cached(C.byref(p3), 512, 256, 512, wp, float((w * w).sum()), 1, 1, _lib.F32, 0, st))
    assert p1.value == p2.value and p3.value not in (None, p1.value)
    e, hit, miss = C.c_int64(), C.c_int64(), C.c_int64()
    _lib.check(lib.mdsp_plan_cache_stats(C.byref(e), C.byref(hit), C.byref(miss)))
    assert (e.value, hit.value >= 2, miss.value >= 4) == (4, True, True)
    # a borrowed plan computes what an owned one does
    x = torch.randn(200_000, device="cuda")
    y1 = torch.empty_like(x); y2 = torch.empty_like(x)
    _lib.check(lib.mdsp_ols_exec(h1, x.data_ptr(), x.numel(), 1, x.numel(), y1.data_ptr(), x.numel(), x.numel(), st))
    own = C.c_void_p()
    _lib.check(lib.mdsp_ols_plan_create(C.byref(own), b.ctypes.data_as(C.c_void_p), 100, 0, 10 ** 6, _lib.F32, _lib.OLS_FILT, 0))
    _lib.check(lib.mdsp_ols_exec(own, x.data_ptr(), x.numel(), 1, x.numel(), y2.data_ptr(), x.numel(), x.numel(), st))
    torch.cuda.synchronize()
    assert torch.equal(y1, y2)
    lib.mdsp_ols_plan_destroy(own)
    # eviction: more distinct requests than MDSP_PLAN_CACHE_SIZE keep the cache bounded
    for k in range(20):
        bk = b.copy(); bk[0] = k
        hk = C.c_void_p()
        _lib.check(lib.mdsp_ols_plan_cached(C.byref(hk), bk.ctypes.data_as(C.c_void_p), 100, 0, 10 ** 6, _lib.F32, _lib.OLS_FILT, 0, st))
    _lib.check(lib.mdsp_plan_cache_stats(C.byref(e), None, None))
    assert e.value == 16
    _lib.check(lib.mdsp_plan_cache_clear())
    _lib.check(lib.mdsp_plan_cache_stats(C.byref(e), None, None))
    assert e.value == 0
    # the Python per-call path rides on it: repeated filt(b, x) hits
    xb = rng.standard_normal(50_000).astype(np.float32); bb = rng.standard_normal(200).astype(np.float32)
    r1 = d.filt(bb, xb); r2 = d.filt(bb, xb)
    _lib.check(lib.mdsp_plan_cache_stats(C.byref(e), C.byref(hit), C.byref(miss)))
    assert np.array_equal(r1, r2) and e.value == 1


def test_library_plan_cache_is_partitioned_by_thread_and_class(d, torch):
    """a borrowed plan must survive other threads' misses and the library's own cached objects.  Thread A borrows a plan;
    thread B then requests 3 x MDSP_PLAN_CACHE_SIZE distinct plans, and thread A itself runs 2 x MDSP_PLAN_CACHE_SIZE per-call FIR filters
    with distinct taps (the library caches an internal 'f' object for each) -- A's plan is still the same live handle and still filters."""
    import threading
    from dsp_jl_amd import _lib, _dev
    lib = _lib.lib()
    _lib.check(lib.mdsp_plan_cache_clear())
    st = _dev.stream_ptr()
    rng = np.random.default_rng(11)
    b = rng.standard_normal(100).astype(np.float32)
    hA = C.c_void_p()
    _lib.check(lib.mdsp_ols_plan_cached(C.byref(hA), b.ctypes.data_as(C.c_void_p), 100, 0, 10 ** 6, _lib.F32, _lib.OLS_FILT, 0, st))
    x = torch.randn(100_000, device="cuda")
    y0 = torch.empty_like(x)
    _lib.check(lib.mdsp_ols_exec(hA, x.data_ptr(), x.numel(), 1, x.numel(), y0.data_ptr(), x.numel(), x.numel(), st))
    torch.cuda.synchronize()
    errs = []

    def other():
        try:
            torch.cuda.set_device(0)
            for k in range(48):
                bk = b.copy(); bk[3] = 100 + k
                hk = C.c_void_p()
                _lib.check(lib.mdsp_ols_plan_cached(C.byref(hk), bk.ctypes.data_as(C.c_void_p), 100, 0, 10 ** 6, _lib.F32, _lib.OLS_FILT, 0, st))
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = threading.Thread(target=other); th.start(); th.join()
    assert not errs, errs
    xs = rng.standard_normal(3000)
    for k in range(32):                                   # this thread's own internal entries (one cached FIR object per distinct b)
        bk = rng.standard_normal(12)
        d.filt(bk, 1.0, xs)
    e = C.c_int64()
    _lib.check(lib.mdsp_plan_cache_stats(C.byref(e), None, None))
    assert e.value <= 16 + 16 + 16                        # A's user list (1) + A's internal list (<= 16) + B's user list (16)
    h2 = C.c_void_p()
    _lib.check(lib.mdsp_ols_plan_cached(C.byref(h2), b.ctypes.data_as(C.c_void_p), 100, 0, 10 ** 6, _lib.F32, _lib.OLS_FILT, 0, st))
    assert h2.value == hA.value                           # a hit on the very same object: nobody evicted it
    y1 = torch.empty_like(x)
    _lib.check(lib.mdsp_ols_exec(hA, x.data_ptr(), x.numel(), 1, x.numel(), y1.data_ptr(), x.numel(), x.numel(), st))
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    _lib.check(lib.mdsp_plan_cache_clear())


def test_alternating_stateful_filters_and_repeated_tdfilt(d, torch):
    """two DF2TFilter objects with different taps that alternate (no device-wide synchronisation, each keeps its device taps);
    repeated filt(b, 1, x) reuses its cached filter object and stays a fresh zero-state filter every call."""
    from oracle import dspbase as odsp
    rng = np.random.default_rng(4)
    b1, b2 = rng.standard_normal(9), rng.standard_normal(30)
    f1, f2 = d.DF2TFilter(b1), d.DF2TFilter(b2)
    x = rng.standard_normal(4000)
    y1, y2 = [], []
    for k in range(0, 4000, 500):
        y1.append(f1.filt(x[k:k + 500])); y2.append(f2.filt(x[k:k + 500]))
    assert relerr(np.concatenate(y1), odsp.filt_ba(b1, 1.0, x)) < 1e-12
    assert relerr(np.concatenate(y2), odsp.filt_ba(b2, 1.0, x)) < 1e-12
    for _ in range(3):
        assert relerr(d.filt(b1, 1.0, x), odsp.filt_ba(b1, 1.0, x)) < 1e-12     # state never leaks from one call into the next
        assert relerr(d.filt(b2, 1.0, x[:100]), odsp.filt_ba(b2, 1.0, x[:100])) < 1e-12


@pytest.mark.parametrize("dt,nb,expect", [(np.float32, 1500, (8192, 6693, 1)), (np.float32, 5120, (4096, 2048, 3)), (np.float32, 7000, (4096, 2048, 4)),
                                          (np.float32, 12000, (8192, 4096, 3)), (np.float64, 3000, (4096, 2048, 2)), (np.float64, 1800, (4096, 2297, 1)),
                                          (np.complex64, 5000, (4096, 2048, 3))])
def test_long_filters_are_reblocked_by_the_fused_engine(d, torch, dt, nb, expect):
    """VERDICT r1 'missing 4': optimalfftfiltlength (dspbase.jl:268-291) asks for nfft = 16384 ... 2^20 once the filter has more than ~1100 taps;
    the fused engine evaluates the same convolution with the largest in-LDS block, or a uniformly partitioned filter (2..4 partitions, delay
    line of spectra in registers) -- against the oracle's own overlap-save at the REFERENCE's nfft, filt and conv, several columns."""
    from dsp_jl_amd import _lib
    from dsp_jl_amd.dspbase import OlsPlan
    from oracle import filt as ofilt, dspbase as odsp
    rng = np.random.default_rng(nb)
    cplx = np.dtype(dt).kind == "c"
    rdt = np.float32 if dt in (np.float32, np.complex64) else np.float64
    b = (rng.standard_normal(nb) / np.sqrt(nb)).astype(rdt)      # no taper: the first outputs of a short signal are O(1), not rounding-level
    nx = 400_000 + 137
    x = rng.standard_normal((nx, 2)).astype(rdt)
    if cplx:
        x = (x + 1j * rng.standard_normal((nx, 2))).astype(dt)
    nfft_ref = d.optimalfftfiltlength(nb, nx)
    assert nfft_ref > (4096 if rdt == np.float64 else 8192)
    plan = OlsPlan(b.astype(dt), nfft_ref, nx, _lib.OLS_FILT, d.ENGINE_FUSED)
    assert (plan.nfft, plan.block_len, plan.engine) == (nfft_ref, nfft_ref - nb + 1, d.ENGINE_FUSED)     # plan_info: the reference's geometry
    en, el, ep = C.c_int64(), C.c_int64(), C.c_int()
    _lib.check(_lib.lib().mdsp_ols_plan_geometry(plan._h, C.byref(en), C.byref(el), C.byref(ep)))
    assert (en.value, el.value, ep.value) == expect
    tol = TOL32 if rdt == np.float32 else 1e-12
    xd = torch.from_numpy(x).cuda()
    y = plan.exec(xd.t().contiguous(), nx).t().cpu().numpy()
    for c in range(2):
        xc = x[:, c].astype(np.complex128 if cplx else np.float64)
        ref = odsp.filt_ba(b.astype(np.float64), 1.0, xc) if not cplx else (odsp.filt_ba(b.astype(np.float64), 1.0, xc.real) + 1j * odsp.filt_ba(b.astype(np.float64), 1.0, xc.imag))
        assert relerr(y[:, c], ref) < tol, c
        # the edges carry the zero padding and the clamped last block
        assert relerr(y[:3000, c], ref[:3000]) < 5 * tol and relerr(y[-3000:, c], ref[-3000:]) < 5 * tol
    if not cplx:
        # through the API: filt(b, x) picks the same plan; the rocFFT engine runs the reference's own block size; conv adds the tail
        got = d.filt(b, xd[:, 0])
        assert relerr(got.cpu().numpy(), ofilt.fftfilt(b.astype(np.float64), x[:, 0].astype(np.float64))) < tol
        roc = d.fftfilt(b, xd[:, 0], nfft_ref, engine=d.ENGINE_ROCFFT)
        assert relerr(got.cpu().numpy(), roc.cpu().numpy()) < 2 * tol
        cv = d.conv(xd[:, 1], torch.from_numpy(b).cuda())
        assert cv.shape == (nx + nb - 1,)
        assert relerr(cv.cpu().numpy(), odsp.conv(x[:, 1].astype(np.float64), b.astype(np.float64))) < tol
        # host-pointer pipeline on a re-blocked plan (block ranges; partitioned plans warm each chunk's delay line up on its nb - 1 history)
        _lib.set_tunable("MDSP_HOST_CHUNK_MIB", 1)
        try:
            yh = plan.exec_host(np.ascontiguousarray(x.T), nx)
        finally:
            _lib.set_tunable("MDSP_HOST_CHUNK_MIB", None)
        assert relerr(yh.T, y) < 2e-6 if rdt == np.float32 else relerr(yh.T, y) < 1e-13
    # short signals: fewer blocks than one run, fewer samples than one block
    for n_small in (1, 100, 2049, 5000):
        xs = x[:n_small, 0]
        got = plan.exec(torch.from_numpy(np.ascontiguousarray(xs)).cuda().view(1, -1), n_small)[0].cpu().numpy()
        ref = odsp.filt_ba(b.astype(np.float64), 1.0, xs.real.astype(np.float64)) + (1j * odsp.filt_ba(b.astype(np.float64), 1.0, xs.imag.astype(np.float64)) if cplx else 0)
        assert relerr(got, ref) < 5 * tol, n_small


@pytest.mark.parametrize("dt,nb,variants", [(np.float32, 5120, (0, 1, 2, 3, 6, 7, 8)), (np.float32, 7000, (0, 1, 2, 3, 5, 6, 7, 8)), (np.float32, 12000, (0, 4)),
                                            (np.float64, 3000, (0,)), (np.float64, 5000, (0,)), (np.complex64, 5000, (0, 1, 3, 6, 7, 8)), (np.complex64, 8000, (0, 5))])
def test_partitioned_overlap_save_ring_variants_and_block_ranges(d, torch, dt, nb, variants):
    """Round 3: the partitioned kernel (Float32, 4096 points) carries the previous half window in registers, loads the next block's new half a
    block ahead, multiplies the delay line by half spectra that live in LDS BEFORE the block's own transform and prefetches partition 0 across
    it (upols2_fused_kernel; variants 5 .. 8 are its flavours, 1 the round-2 form with every spectrum from L2, 3 that form with LDS spectra,
    2 / 4 with the half windows in an LDS ring filled by buffer_load ... lds) -- every variant against the Float64 oracle and against each other; and mdsp_ols_exec_range now serves partitioned plans: ranges from slices that hold nothing but
    the nb - 1 samples of history, any first block, equal to the whole-column call up to rounding (the delay line of a range warms up on
    zeros where the whole column has samples under the zero taps, so not bit for bit)."""
    from dsp_jl_amd import _lib, _dev
    from dsp_jl_amd.dspbase import OlsPlan
    from oracle import dspbase as odsp
    rng = np.random.default_rng(nb + 1)
    cplx = np.dtype(dt).kind == "c"
    rdt = np.float32 if dt in (np.float32, np.complex64) else np.float64
    tol = TOL32 if rdt == np.float32 else 1e-12
    b = (rng.standard_normal(nb) / np.sqrt(nb)).astype(rdt)
    for nx in (300_007, 2048 * 37, 2048 * 37 + 1, 4500):          # ragged, a whole number of blocks (the last half lands exactly on nx), short
        x = rng.standard_normal(nx).astype(rdt)
        if cplx:
            x = (x + 1j * rng.standard_normal(nx)).astype(dt)
        xc = x.astype(np.complex128 if cplx else np.float64)
        ref = odsp.filt_ba(b.astype(np.float64), 1.0, xc) if not cplx else (odsp.filt_ba(b.astype(np.float64), 1.0, xc.real) + 1j * odsp.filt_ba(b.astype(np.float64), 1.0, xc.imag))
        nfft_ref = d.optimalfftfiltlength(nb, 400_000)
        xd = torch.from_numpy(x).cuda()
        outs = {}
        for v in variants:
            _lib.set_tunable("MDSP_OLS_VARIANT", v)
            try:
                plan = OlsPlan(b.astype(dt), nfft_ref, nx, _lib.OLS_FILT, d.ENGINE_FUSED)
            finally:
                _lib.set_tunable("MDSP_OLS_VARIANT", None)
            en, el, ep = C.c_int64(), C.c_int64(), C.c_int()
            _lib.check(_lib.lib().mdsp_ols_plan_geometry(plan._h, C.byref(en), C.byref(el), C.byref(ep)))
            assert ep.value > 1
            y = plan.exec(xd.view(1, -1), nx)[0]
            outs[v] = y.cpu().numpy()
            assert relerr(outs[v], ref) < tol, (nx, v)
            assert relerr(outs[v][:3000], ref[:3000]) < 5 * tol and relerr(outs[v][-3000:], ref[-3000:]) < 5 * tol, (nx, v)
            # block ranges (ragged starts, a range that ends inside the grid, one clipped by it)
            L = el.value
            nblocks = -(-nx // L)
            got = torch.full_like(y, float("nan"))
            for g0, cnt in ((0, 3), (3, 1), (4, 17), (21, nblocks)):
                g1 = min(nblocks, g0 + cnt)
                if g0 >= g1:
                    continue
                lo, hi = max(0, g0 * L - (nb - 1)), min(nx, g1 * L)
                o0, o1 = g0 * L, min(nx, g1 * L)
                xs = xd[lo:hi].clone()
                ys = torch.empty(o1 - o0, dtype=xd.dtype, device="cuda")
                _lib.check(_lib.lib().mdsp_ols_exec_range(plan._h, _dev.ptr(xs), lo, hi - lo, nx, _dev.ptr(ys), g0, cnt, nx, _dev.stream_ptr()))
                got[o0:o1] = ys
            torch.cuda.synchronize()
            assert relerr(got.cpu().numpy(), ref) < tol, (nx, v, "ranges")
        for v in variants[1:]:
            assert relerr(outs[v], outs[variants[0]]) < tol


@pytest.mark.parametrize("dt,tol", [(np.float32, TOL32), (np.float64, 1e-12), (np.complex64, TOL32), (np.complex128, 1e-12)])
@pytest.mark.parametrize("nfft", [1000, 1536, 2000, 3000, 625, 120, 2401, 18, 7, 6000])
def test_mixed_radix_sizes_run_fused(d, torch, dt, tol, nfft):
    """VERDICT r1 'missing 3': nextfastfft sizes (2^a 3^b 5^c 7^d; util.jl:107-135) are the DEFAULT nfft of periodogram / welch_pgram / stft
    (periodograms.jl:393, :560, :872).  They run on the fused engine (mixed-radix passes through LDS) -- Welch, raw STFT one- and two-sided,
    spectrogram, periodogram, windows shorter than nfft, odd frame counts, two channels -- against the Float64 oracle."""
    from oracle import periodograms as opg, windows as ow
    if nfft > 4096 and np.dtype(dt).itemsize in (8, 16) and np.dtype(dt) != np.complex64:
        pytest.skip("Float64 mixed-radix transforms stop at 4096 points (two LDS buffers)")
    cplx = np.dtype(dt).kind == "c"
    rng = np.random.default_rng(nfft)
    n = nfft if nfft < 50 else nfft - 3                      # a window shorter than the transform (zero tail), except for the tiny sizes
    nov = n // 3
    hop = n - nov
    L = hop * 41 + n                                          # 42 frames: the last real-signal pair is complete; + one more frame below
    x = rng.standard_normal((L + hop, 2)).astype(np.float32 if dt in (np.float32, np.complex64) else np.float64)
    if cplx:
        x = (x + 1j * rng.standard_normal(x.shape)).astype(dt)
    xd = torch.from_numpy(x).cuda()
    for length in (L, L + hop):                               # even and odd frame counts
        K = d.frame_count(length, n, nov)
        cfg = d.WelchConfig(length, dt, n=n, noverlap=nov, nfft=nfft, window=d.hanning, engine=d.ENGINE_FUSED)
        assert cfg.engine == d.ENGINE_FUSED
        P = d.welch_pgram(xd[:length], cfg).power.cpu().numpy()
        for c in range(2):
            ref = opg.welch_pgram(x[:length, c], n, nov, nfft=nfft, window=ow.hanning, dtype=np.float64).power
            assert relerr(P[:, c], ref) < tol, ("welch", length, c)
        for onesided in ((True, False) if not cplx else (False,)):
            S = d.stft(xd[:length], n, nov, nfft=nfft, onesided=onesided, window=d.hanning, engine=d.ENGINE_FUSED).cpu().numpy()
            assert S.shape == ((nfft // 2 + 1) if onesided else nfft, K, 2)
            ref = opg.stft(x[:length, 1], n, nov, nfft=nfft, onesided=onesided, window=ow.hanning, dtype=np.float64)
            assert relerr(S[:, :, 1], ref) < tol, ("stft", length, onesided)
            sp = d.spectrogram(xd[:length], n, nov, nfft=nfft, onesided=onesided, fs=3.0, window=d.hanning, engine=d.ENGINE_FUSED).power.cpu().numpy()
            refp = opg.stft(x[:length, 0], n, nov, psdonly=True, nfft=nfft, onesided=onesided, fs=3.0, window=ow.hanning, dtype=np.float64)
            assert relerr(sp[:, :, 0], refp) < tol, ("spectrogram", length, onesided)
    # periodogram's default nfft IS nextfastfft(length): the single-frame case
    s1 = x[:nfft, 0]
    pg = d.periodogram(torch.from_numpy(np.ascontiguousarray(s1)).cuda(), window=d.hanning, fs=2.0)
    assert relerr(pg.power.cpu().numpy(), opg.periodogram(s1, window=ow.hanning, fs=2.0, dtype=np.float64).power) < tol
    # the same plan against the rocFFT engine (the path these sizes took before)
    roc = d.stft(xd[:L], n, nov, nfft=nfft, window=d.hanning, engine=d.ENGINE_ROCFFT)
    fus = d.stft(xd[:L], n, nov, nfft=nfft, window=d.hanning, engine=d.ENGINE_FUSED)
    assert float((roc - fus).abs().max() / fus.abs().max()) < 20 * tol


def test_host_pipelines_with_padded_leading_dimensions(d, torch):
    """Columns of a larger host matrix (ld > column length) through the host-pointer entry points: only the columns' own samples travel."""
    from dsp_jl_amd import _lib
    from dsp_jl_amd.dspbase import OlsPlan
    from oracle import dspbase as odsp, periodograms as opg, windows as ow
    lib = _lib.lib()
    rng = np.random.default_rng(21)
    nx, ld, ncols = 300_001, 300_100, 3
    X = rng.standard_normal((ncols, ld)).astype(np.float32)            # rows = Julia columns, ld = 300100
    b = _taps(200, np.float32)
    plan = OlsPlan(b, 1024, nx, _lib.OLS_FILT, d.ENGINE_AUTO)
    Y = np.full((ncols, ld), 7.0, np.float32)
    _lib.set_tunable("MDSP_HOST_CHUNK_MIB", 1)
    try:
        _lib.check(lib.mdsp_ols_exec_host(plan._h, X.ctypes.data_as(C.c_void_p), nx, ncols, ld, Y.ctypes.data_as(C.c_void_p), nx, ld, 0))
        cfg = d.WelchConfig(nx, np.float32, n=1024, noverlap=512, window=d.hanning)
        P = np.full((ncols, cfg.nout + 5), -1.0, np.float32)
        _lib.check(lib.mdsp_welch_exec_host(cfg._h, X.ctypes.data_as(C.c_void_p), nx, ncols, ld, P.ctypes.data_as(C.c_void_p), cfg.nout + 5, 0))
    finally:
        _lib.set_tunable("MDSP_HOST_CHUNK_MIB", None)
    assert np.all(Y[:, nx:] == 7.0) and np.all(P[:, cfg.nout:] == -1.0)          # nothing outside the columns is touched
    for c in range(ncols):
        assert relerr(Y[c, :nx], odsp.filt_ba(b.astype(np.float64), 1.0, X[c, :nx].astype(np.float64))) < TOL32
        assert relerr(P[c, :cfg.nout], opg.welch_pgram(X[c, :nx], 1024, 512, window=ow.hanning, dtype=np.float64).power) < TOL32
    with pytest.raises(d.ArgumentError):
        _lib.check(lib.mdsp_ols_exec_host(plan._h, X.ctypes.data_as(C.c_void_p), nx, ncols, nx - 1, Y.ctypes.data_as(C.c_void_p), nx, ld, 0))
    with pytest.raises(d.DimensionMismatch):
        _lib.check(lib.mdsp_welch_exec_host(cfg._h, X.ctypes.data_as(C.c_void_p), nx, ncols, nx - 1, P.ctypes.data_as(C.c_void_p), cfg.nout, 0))


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f32", "c32", "f64", "c64"])
@pytest.mark.parametrize("L,M,ntaps", [(160, 147, 5120), (160, 147, 5921), (23, 17, 300), (16, 9, 129), (17, 35, 1100), (37, 2, 400), (250, 249, 4000), (14, 9, 64),
                                           (2, 1, 49), (1, 2, 31), (3, 2, 73), (2, 3, 61), (4, 1, 97), (1, 4, 40), (5, 3, 101), (1, 1, 33), (7, 4, 120), (25, 24, 700), (192, 191, 6000), (1, 4, 193), (1, 3, 170), (2, 5, 400), (147, 160, 5881), (160, 441, 16001), (20, 441, 2200), (1, 8, 441), (80, 441, 16001), (1, 1, 300), (441, 160, 16001), (320, 147, 9000), (1000, 999, 30000), (1, 16, 583), (1, 32, 1100)])
def test_polyphase_matrix_core_kernel_equals_register_tap_kernel(d, torch, L, M, ntaps, dt):
    # The matrix-core kernel (v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64: 16 rows x 16 outputs, a k-ordered fmaf chain) and the
    # register-tap / generic kernels sum each output in the same order: bit-identical Float32 outputs (Float64: to rounding) and
    # identical states, for any phase / deficit the stream is cut at, with ragged tails (nout not a multiple of L, L not a multiple
    # of 16), several channels on an odd leading dimension, real and complex signals, and against the oracle.
    from fractions import Fraction
    from dsp_jl_amd import _lib
    from oracle import stream_filt as osf
    lib = _lib.lib()
    tdt, hdt, ldt_h, ldt_x, tol = {"f32": (torch.float32, np.float32, _lib.F32, _lib.F32, 2e-6), "c32": (torch.complex64, np.float32, _lib.F32, _lib.C32, 2e-6),
                                   "f64": (torch.float64, np.float64, _lib.F64, _lib.F64, 1e-13), "c64": (torch.complex128, np.float64, _lib.F64, _lib.C64, 1e-13)}[dt]
    if dt != "f32" and (L, M) in ((160, 147), (250, 249), (192, 191), (160, 441), (80, 441), (441, 160), (1000, 999)) and ntaps != 5120 and (dt, L, M) not in (("c32", 160, 441), ("c64", 160, 147)):
        pytest.skip("the large shapes are run once per dtype")                 # (ComplexF32 160//441: round 4's 40-step form, against the generic kernel;
                                                                               #  ComplexF64 160//147 with resample_filter's 5921 taps: round 5's 14-step form)
    rng = np.random.default_rng(L * 1000 + M)
    h = (rng.standard_normal(ntaps) / np.sqrt(ntaps / L)).astype(hdt)
    nch, n = 3, 200_003
    g = torch.Generator(device="cuda"); g.manual_seed(L + M)
    x = torch.randn((nch, n), generator=g, device="cuda", dtype=tdt)
    stream = torch.cuda.current_stream().cuda_stream
    cuts = (0, 1, 777, 100_000, n)
    outs = {}
    try:
        for sg in (0, 1):
            _lib.set_tunable("MDSP_FIR_MM", sg)
            fh = C.c_void_p()
            _lib.check(lib.mdsp_fir_create(C.byref(fh), h.ctypes.data_as(C.c_void_p), len(h), L, M, ldt_h, ldt_x, nch))
            pieces, states = [], []
            for a, b in zip(cuts[:-1], cuts[1:]):
                ol = C.c_int64(); _lib.check(lib.mdsp_fir_outputlength(fh, b - a, C.byref(ol)))
                ldy = ol.value + 1
                y = torch.full((nch, ldy), complex(float("nan"), float("nan")) if tdt.is_complex else float("nan"), dtype=tdt, device="cuda")
                nw = C.c_int64()
                _lib.check(lib.mdsp_fir_exec(fh, x[:, a:].data_ptr(), b - a, n, y.data_ptr(), ol.value, ldy, C.byref(nw), stream))
                torch.cuda.synchronize()
                assert nw.value == ol.value
                assert torch.isnan(torch.view_as_real(y[:, ol.value:]) if y.is_complex() else y[:, ol.value:]).all()
                assert not torch.isnan(torch.view_as_real(y[:, :ol.value]) if y.is_complex() else y[:, :ol.value]).any()
                pieces.append(y[:, :ol.value].clone())
                phi, dfc = C.c_int64(), C.c_int64()
                _lib.check(lib.mdsp_fir_get_state(fh, C.byref(phi), C.byref(dfc), None))
                states.append((phi.value, dfc.value))
            outs[sg] = (torch.cat(pieces, dim=1), states)
            _lib.check(lib.mdsp_fir_destroy(fh))
    finally:
        _lib.set_tunable("MDSP_FIR_MM", None)
    assert outs[0][1] == outs[1][1]
    if dt in ("f32", "c32"):
        assert torch.equal(outs[0][0], outs[1][0])
    else:
        assert relerr(outs[1][0].cpu().numpy(), outs[0][0].cpu().numpy()) < 1e-14
    m = 20_000
    xr = x[1, :m].cpu().numpy()
    ref = osf.FIRFilter(h.astype(np.float64), Fraction(L, M)).filt(xr.astype(np.complex128 if np.iscomplexobj(xr) else np.float64))
    assert relerr(outs[1][0][1, :len(ref)].cpu().numpy(), ref) < tol


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f32", "f64", "c32", "c64"])
@pytest.mark.parametrize("L,M,ntaps", [(147, 160, 5881), (49, 48, 1813), (147, 80, 5439), (147, 320, 5000), (21, 16, 640), (160, 147, 5120), (1, 8, 293), (2, 1, 75),
                                       (1, 4, 147), (1, 6, 219), (160, 441, 16001), (320, 147, 10240), (441, 160, 16317), (1, 3, 111), (1, 2, 75), (1, 16, 583)])
def test_polyphase_matrix_core_round3_forms_are_bit_identical(d, torch, L, M, ntaps, dt):
    """Round 3's forms of the matrix-core kernel change where samples sit in LDS and when instructions issue, never the arithmetic: padded runs
    (MDSP_FIR_MM_ROWS=2, the default where the row stride is bank-hostile) against the row-staged form (=1) and the plain run (=0); the tile
    choice by cost against round 2's rule (MDSP_FIR_MM_NG=8); memory waves at raised / normal priority; gathered wide stores / element stores; taps
    in registers (fewer chunks per wave, shorter rows, several column blocks per wave) against fetched per tile (MDSP_FIR_MM_T64 / _NBLK = 0) --
    bit for bit, with a stream cut at odd places (history tiles, ragged last tile), and against the Float64 oracle."""
    from fractions import Fraction
    from dsp_jl_amd import _lib
    from oracle import stream_filt as osf
    lib = _lib.lib()
    tdt, hdt, ldt_h, ldt_x, tol = {"f32": (torch.float32, np.float32, _lib.F32, _lib.F32, 2e-6), "c32": (torch.complex64, np.float32, _lib.F32, _lib.C32, 2e-6),
                                   "f64": (torch.float64, np.float64, _lib.F64, _lib.F64, 1e-13), "c64": (torch.complex128, np.float64, _lib.F64, _lib.C64, 1e-13)}[dt]
    out12 = (C.c_int64 * 12)()
    _lib.check(lib.mdsp_fir_mm_geometry(L, M, ntaps, ldt_h, ldt_x, out12))
    if not out12[0]:
        pytest.skip("the shape does not fit the matrix-core kernel in this signal type")
    rng = np.random.default_rng(L * 977 + M)
    h = (rng.standard_normal(ntaps) / np.sqrt(ntaps / L)).astype(hdt)
    nch, n = 2, 150_001
    g = torch.Generator(device="cuda"); g.manual_seed(L * 3 + M)
    x = torch.randn((nch, n), generator=g, device="cuda", dtype=tdt)
    stream = torch.cuda.current_stream().cuda_stream
    cuts = (0, 3, 40_000, n)
    knobs = [{"MDSP_FIR_MM": 1}, {"MDSP_FIR_MM": 1, "MDSP_FIR_MM_ROWS": 1}, {"MDSP_FIR_MM": 1, "MDSP_FIR_MM_ROWS": 0}, {"MDSP_FIR_MM": 1, "MDSP_FIR_MM_ROWS": 2},
             {"MDSP_FIR_MM": 1, "MDSP_FIR_MM_NG": 8}, {"MDSP_FIR_MM": 1, "MDSP_FIR_MM_PRIO": 0}, {"MDSP_FIR_MM": 1, "MDSP_FIR_MM_PRIO": 1},
             {"MDSP_FIR_MM": 1, "MDSP_FIR_MM_VSTORE": 0}, {"MDSP_FIR_MM": 1, "MDSP_FIR_MM_CH": 1},
             {"MDSP_FIR_MM": 1, "MDSP_FIR_MM_T64": 0}, {"MDSP_FIR_MM": 1, "MDSP_FIR_MM_NBLK": 0}, {"MDSP_FIR_MM": 1, "MDSP_FIR_MM_T64": 0, "MDSP_FIR_MM_NBLK": 0},   # (these three: round 2's register limits / fetched taps)
             {"MDSP_FIR_MM": 1, "MDSP_FIR_MM_RPX": 1}, {"MDSP_FIR_MM": 1, "MDSP_FIR_MM_RPX": 1, "MDSP_FIR_MM_ROWS": 2}]   # (round 5 prep: padded runs for windows longer than a granule / fetched taps)
    outs = []
    try:
        for kn in knobs:
            for k, v in kn.items():
                _lib.set_tunable(k, v)
            _lib.check(lib.mdsp_fir_mm_geometry(L, M, ntaps, ldt_h, ldt_x, out12))
            fits = bool(out12[0])
            if fits:
                fh = C.c_void_p()
                _lib.check(lib.mdsp_fir_create(C.byref(fh), h.ctypes.data_as(C.c_void_p), len(h), L, M, ldt_h, ldt_x, nch))
                pieces = []
                for a, b in zip(cuts[:-1], cuts[1:]):
                    ol = C.c_int64(); _lib.check(lib.mdsp_fir_outputlength(fh, b - a, C.byref(ol)))
                    y = torch.empty((nch, ol.value), dtype=tdt, device="cuda")
                    nw = C.c_int64()
                    _lib.check(lib.mdsp_fir_exec(fh, x[:, a:].data_ptr(), b - a, n, y.data_ptr(), ol.value, ol.value, C.byref(nw), stream))
                    torch.cuda.synchronize()
                    pieces.append(y)
                outs.append((kn, torch.cat(pieces, dim=1)))
                _lib.check(lib.mdsp_fir_destroy(fh))
            for k in kn:
                _lib.set_tunable(k, None)
    finally:
        for k in ("MDSP_FIR_MM", "MDSP_FIR_MM_ROWS", "MDSP_FIR_MM_NG", "MDSP_FIR_MM_PRIO", "MDSP_FIR_MM_VSTORE", "MDSP_FIR_MM_CH", "MDSP_FIR_MM_T64", "MDSP_FIR_MM_NBLK", "MDSP_FIR_MM_RPX"):
            _lib.set_tunable(k, None)
    assert len(outs) >= 5
    for kn, y in outs[1:]:
        assert torch.equal(y, outs[0][1]), kn
    m = 15_000
    xr = x[1, :m].cpu().numpy()
    ref = osf.FIRFilter(h.astype(np.float64), Fraction(L, M)).filt(xr.astype(np.complex128 if np.iscomplexobj(xr) else np.float64))
    assert relerr(outs[0][1][1, :len(ref)].cpu().numpy(), ref) < tol


@pytest.mark.gpu
def test_polyphase_kernel_choice(d, torch):
    # BASELINE config 5's shape runs on the matrix-core kernel, Float64 on
    # the matrix-core kernel in Float64, mixed precisions on the generic one -- a silent fallback would show up here, not as a slow benchmark.
    from dsp_jl_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(5)

    def path(h, L, M, x_dtype, nch, xlen):
        fh = C.c_void_p()
        tdt = _lib.F32 if h.dtype == np.float32 else _lib.F64
        _lib.check(lib.mdsp_fir_create(C.byref(fh), h.ctypes.data_as(C.c_void_p), len(h), L, M, tdt, x_dtype, nch))
        p = C.c_int(-1)
        _lib.check(lib.mdsp_fir_kernel_path(fh, xlen, C.byref(p)))
        _lib.check(lib.mdsp_fir_destroy(fh))
        return p.value

    h = rng.standard_normal(5120).astype(np.float32)
    assert path(h, 160, 147, _lib.F32, 4, 2 ** 28) == 2
    assert path(h, 160, 147, _lib.F32, 4, 10_000) == 2                          # short chunks too: it is the faster kernel at every size
    assert path(h.astype(np.float64), 160, 147, _lib.F64, 4, 2 ** 28) == 2      # Float64 on v_mfma_f64_16x16x4_f64
    assert path(h, 160, 147, _lib.C32, 4, 2 ** 28) == 2                         # complex signal: two products against the same taps
    assert path(h.astype(np.float64), 160, 147, _lib.F32, 4, 2 ** 28) == 0      # Float64 taps on a Float32 signal: generic kernel
    assert path(rng.standard_normal(48).astype(np.float32), 2, 1, _lib.F32, 1, 2 ** 26) == 2      # interpolation by 2: a row is 7 rounds
    assert path(rng.standard_normal(64).astype(np.float32), 1, 2, _lib.F32, 1, 2 ** 26) == 2      # decimation by 2: still the matrix cores
    assert path(rng.standard_normal(581).astype(np.float32), 1, 16, _lib.F32, 2, 2 ** 26) == 3      # from M = 8 on (Float64: 4): the decimator kernel (round 5)
    assert path(rng.standard_normal(581), 1, 16, _lib.C64, 2, 2 ** 26) == 3                         # ... in every signal type
    assert path(rng.standard_normal(150), 1, 4, _lib.F64, 2, 2 ** 26) == 3
    assert path(rng.standard_normal(581), 1, 16, _lib.F32, 2, 2 ** 26) == 0                         # (Float64 taps on Float32 samples: generic)
    assert path(rng.standard_normal(900).astype(np.float32), 1, 100, _lib.F32, 1, 2 ** 26) != 3     # M > 64: the older kernels
    assert path(rng.standard_normal(4000).astype(np.float32), 250, 249, _lib.F32, 1, 2 ** 26) == 2  # L > 192 in Float32, two column blocks per wave: their taps live in registers (round 3: 0.70 against 0.92 ms)
    assert path(rng.standard_normal(16317).astype(np.float32), 441, 160, _lib.F32, 1, 2 ** 26) == 1  # three blocks per wave: the register-tap kernel is still the faster one (1.67 against 2.39 ms)
    assert path(rng.standard_normal(4000), 250, 249, _lib.F64, 1, 2 ** 26) == 2                     # ... in Float64 the matrix cores (several column blocks per wave)
    assert path(rng.standard_normal(40000).astype(np.float32), 2000, 1999, _lib.F32, 1, 2 ** 26) == 0  # L > 1024


@pytest.mark.gpu
def test_polyphase_matrix_core_kernel_fuzz(d, torch):
    # Random ratios, tap counts, start phases and chunkings: the matrix-core kernel against the register-tap / generic kernels
    # (bit for bit in Float32) and the filter state after every chunk.
    from math import gcd
    from dsp_jl_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(20260925)
    stream = torch.cuda.current_stream().cuda_stream
    tried = used = 0
    try:
        for it in range(60):
            L = int(rng.integers(1, 500)); M = int(rng.integers(1, 260))
            g0 = gcd(L, M); L //= g0; M //= g0
            tp = int(rng.integers(1, 70))
            ntaps = int(max(1, tp * L - rng.integers(0, L)))
            h = rng.standard_normal(ntaps).astype(np.float32)
            nch = int(rng.integers(1, 4))
            n = int(rng.integers(1, 60_000))
            cplx = bool(rng.integers(0, 2))
            tdt = torch.complex64 if cplx else torch.float32
            x = torch.randn((nch, n), device="cuda", dtype=tdt)
            cuts = sorted({0, n, *[int(c) for c in rng.integers(0, n + 1, size=3)]})
            phi = float(rng.random()) if rng.integers(0, 2) else None
            res = {}
            for mm in (0, 1):
                _lib.set_tunable("MDSP_FIR_MM", mm)
                fh = C.c_void_p()
                _lib.check(lib.mdsp_fir_create(C.byref(fh), h.ctypes.data_as(C.c_void_p), len(h), L, M, _lib.F32, _lib.C32 if cplx else _lib.F32, nch))
                if phi is not None and not (L == 1 and M == 1):
                    _lib.check(lib.mdsp_fir_setphase(fh, phi))
                pth = C.c_int(-1); _lib.check(lib.mdsp_fir_kernel_path(fh, n, C.byref(pth)))
                if mm == 1:
                    tried += 1; used += pth.value == 2
                pieces, states = [], []
                for a, b in zip(cuts[:-1], cuts[1:]):
                    ol = C.c_int64(); _lib.check(lib.mdsp_fir_outputlength(fh, b - a, C.byref(ol)))
                    y = torch.zeros((nch, ol.value + 3), dtype=tdt, device="cuda")
                    nw = C.c_int64()
                    _lib.check(lib.mdsp_fir_exec(fh, x[:, a:].data_ptr(), b - a, n, y.data_ptr(), ol.value, ol.value + 3, C.byref(nw), stream))
                    torch.cuda.synchronize()
                    assert nw.value == ol.value and not y[:, ol.value:].abs().any()
                    pieces.append(y[:, :ol.value].clone())
                    p_, d_ = C.c_int64(), C.c_int64()
                    _lib.check(lib.mdsp_fir_get_state(fh, C.byref(p_), C.byref(d_), None))
                    states.append((p_.value, d_.value))
                res[mm] = (torch.cat(pieces, dim=1) if pieces else None, states)
                _lib.check(lib.mdsp_fir_destroy(fh))
            assert res[0][1] == res[1][1], (L, M, ntaps)
            if res[0][0] is not None:
                assert torch.equal(res[0][0], res[1][0]), (L, M, ntaps, nch, n, cplx, cuts)
    finally:
        _lib.set_tunable("MDSP_FIR_MM", None)
    assert used >= tried // 2          # most random shapes fit the matrix-core kernel


def test_plan_cache_reaps_exited_threads_and_honours_contexts(d, torch):
    """csrc/plancache.hip: the plan cache is partitioned by calling thread, and every Welch plan owns device
    buffers -- so the lists of exited threads must not stay.  A churned pool of short-lived OS threads borrows cached plans (and runs them, so
    that they own their partial sums); afterwards one request from this thread drains the graveyard: the entry count is back to this thread's
    own, free device memory is back where it was.  Explicit contexts: two threads that bind the same context id share one partition (a hit, the
    same handle), and releasing the context frees it."""
    import ctypes as C
    import threading
    from dsp_jl_amd import _lib
    lib = _lib.lib()
    _lib.check(lib.mdsp_plan_cache_clear())
    torch.cuda.synchronize()
    x = torch.randn(1 << 20, device="cuda", dtype=torch.float32)
    psd = torch.empty(4097, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    errs = []

    def borrow(n, run=True):
        h = C.c_void_p()
        _lib.check(lib.mdsp_welch_plan_cached(C.byref(h), n, n // 2, n, None, float(n), 1, _lib.F32, d.ENGINE_FUSED, None))
        if run:
            _lib.check(lib.mdsp_welch_exec(h, x.data_ptr(), x.numel(), 1, x.numel(), psd.data_ptr(), n // 2 + 1, None))
            _lib.check(lib.mdsp_stream_synchronize(None))
        return h.value

    def worker(k):
        try:
            _lib.check(lib.mdsp_init(0))
            for n in (256, 512, 1024, 2048, 4096, 8192):
                borrow(n)
        except Exception as e:   # pragma: no cover
            errs.append(repr(e))

    borrow(4096)                                              # this thread's own entry
    free0 = torch.cuda.mem_get_info()[0]
    for rnd in range(6):                                      # 24 threads, four at a time, all gone afterwards
        ts = [threading.Thread(target=worker, args=(4 * rnd + k,)) for k in range(4)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    assert not errs, errs
    borrow(4096)                                              # any request drains the graveyard
    e, hh, m = C.c_int64(), C.c_int64(), C.c_int64()
    parts, reaped = C.c_int64(), C.c_int64()
    _lib.check(lib.mdsp_plan_cache_stats(C.byref(e), C.byref(hh), C.byref(m)))
    _lib.check(lib.mdsp_plan_cache_partitions(C.byref(parts), C.byref(reaped)))
    assert e.value == 1 and parts.value == 1, (e.value, parts.value)
    assert reaped.value >= 24 * 6 - 16, reaped.value          # (a few may have gone through the global cap instead: also counted)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 64 << 20, (free0, free1)           # 144 plans with their partial sums would be gigabytes
    # explicit contexts: the partition follows the id, not the OS thread
    got = {}

    def in_ctx(name):
        _lib.check(lib.mdsp_init(0))
        _lib.check(lib.mdsp_plan_cache_set_context(7))
        got[name] = borrow(1024, run=False)
        _lib.check(lib.mdsp_plan_cache_set_context(0))

    for name in ("a", "b"):
        t = threading.Thread(target=in_ctx, args=(name,))
        t.start(); t.join()
    assert got["a"] == got["b"]                               # second thread: a hit in the shared partition
    _lib.check(lib.mdsp_plan_cache_stats(C.byref(e), None, None))
    assert e.value == 2                                       # context entries survive their threads' exit
    _lib.check(lib.mdsp_plan_cache_release_context(7))
    _lib.check(lib.mdsp_plan_cache_stats(C.byref(e), None, None))
    assert e.value == 1
    _lib.check(lib.mdsp_plan_cache_clear())


@pytest.mark.parametrize("L,M,ntaps", [(160, 147, 5120), (3, 2, 96), (1, 2, 48), (2, 1, 64)])
def test_polyphase_nonfinite_samples_leave_the_reference_hole(d, torch, L, M, ntaps):
    """stream_filt.jl:496-509: an output is the dot product of ONE column of the polyphase bank with ITS window of tapsPerPhi samples, so a NaN / Inf
    sample makes exactly the outputs whose own window holds it non-finite.  With mdsp_fir_set_exact(f, 1) -- FIRFilter(...; exact=true) -- (the generic kernel: the reference's windows
    and nothing else) the non-finite outputs are EXACTLY the oracle's.  The fast kernels multiply a block's (matrix-core) or a residue pair's
    (register-tap, MDSP_FIR_MM=0) common window by explicit zero taps, so their hole may be wider -- by at most 15 outputs plus the outputs of 64
    more input positions on either side -- and outside that widened hole their outputs are finite and equal the exact run's
    to rounding."""
    import ctypes as C
    from fractions import Fraction
    from dsp_jl_amd import _lib
    from oracle import stream_filt as osf
    lib = _lib.lib()
    rng = np.random.default_rng(77 + L + M)
    h = (rng.standard_normal(ntaps) / np.sqrt(ntaps / L)).astype(np.float32)
    h[np.abs(h) < 1e-3] = 1e-3                                   # no tap is zero: Inf * tap stays Inf, never NaN-by-0
    n = 120_001
    x = rng.standard_normal(n).astype(np.float32)
    planted = {5: np.nan, 40_000: np.inf, 40_003: -np.inf, 77_777: np.nan, n - 2: np.inf}
    for pos, v in planted.items():
        x[pos] = v
    with np.errstate(invalid="ignore", over="ignore"):
        ref = osf.filt_stateless(h.astype(np.float64), x.astype(np.float64), Fraction(L, M))
    bad_ref = ~np.isfinite(ref)
    assert 0 < bad_ref.sum() < len(ref) // 10
    xd = torch.from_numpy(x).cuda()[None, :].contiguous()
    stream = torch.cuda.current_stream().cuda_stream
    outs = {}
    try:
        for mode, knobs in (("exact", {}), ("exact_env", {"MDSP_FIR_EXACT": 1}), ("regtap", {"MDSP_FIR_MM": 0}), ("default", {})):
            for k, v in knobs.items():
                _lib.set_tunable(k, v)
            fh = C.c_void_p()
            _lib.check(lib.mdsp_fir_create(C.byref(fh), h.ctypes.data_as(C.c_void_p), len(h), L, M, _lib.F32, _lib.F32, 1))
            if mode == "exact":     # the flag of the FILTER (round 5; the environment variable of round 4 still works: "exact_env")
                _lib.check(lib.mdsp_fir_set_exact(fh, 1))
            ol = C.c_int64(); _lib.check(lib.mdsp_fir_outputlength(fh, n, C.byref(ol)))
            assert ol.value == len(ref)
            y = torch.zeros((1, ol.value + 1), dtype=torch.float32, device="cuda")
            nw = C.c_int64()
            _lib.check(lib.mdsp_fir_exec(fh, xd.data_ptr(), n, n, y.data_ptr(), ol.value, ol.value + 1, C.byref(nw), stream))
            torch.cuda.synchronize()
            pth = C.c_int(-1); _lib.check(lib.mdsp_fir_kernel_path(fh, n, C.byref(pth)))
            outs[mode] = (y[0, :ol.value].cpu().numpy(), pth.value)
            _lib.check(lib.mdsp_fir_destroy(fh))
            for k in knobs:
                _lib.set_tunable(k, None)
    finally:
        _lib.set_tunable("MDSP_FIR_MM", None)
        _lib.set_tunable("MDSP_FIR_EXACT", None)
    y0, path0 = outs["exact"]
    assert path0 == 0                                              # the generic kernel
    assert outs["exact_env"][1] == 0 and np.array_equal(outs["exact_env"][0], y0, equal_nan=True)
    yp = np.asarray(d.FIRFilter(h, Fraction(L, M), exact=True).filt(x))   # ... and through the host mirror: FIRFilter(h, ratio; exact=true)
    assert np.array_equal(yp, y0, equal_nan=True)
    bad0 = ~np.isfinite(y0)
    assert np.array_equal(bad0, bad_ref), (int(bad0.sum()), int(bad_ref.sum()), np.flatnonzero(bad0 != bad_ref)[:10])   # the reference's hole, exactly
    assert relerr(y0[~bad_ref], ref[~bad_ref]) < 2e-6
    W = 15 + int(np.ceil(64 * L / M))                              # the fast kernels exceed it by at most this many outputs on either side
    idx = np.flatnonzero(bad_ref)
    near = np.zeros(len(ref), dtype=bool)
    for i in idx:
        near[max(0, i - W):i + W + 1] = True
    for mode in ("regtap", "default"):
        y1, path1 = outs[mode]
        bad1 = ~np.isfinite(y1)
        assert np.all(bad1[bad_ref]), mode                         # the hole covers the reference's ...
        assert not np.any(bad1 & ~near), (mode, int(np.sum(bad1 & ~near)), W)
        assert np.all(np.isfinite(y1[~near])) and relerr(y1[~near], y0[~near]) < 2e-6, mode   # untouched outputs: finite, the reference-window kernel's values
        if path1 == 0:
            assert np.array_equal(bad1, bad_ref)
    assert outs["regtap"][1] != 2                                  # MDSP_FIR_MM=0 never takes the matrix-core kernel


@pytest.mark.parametrize("variant", [30, 31, 32, 33, 34, 35, 36, 40, 41, 43])
def test_welch_round3_kernel_vs_oracle_and_round2_kernel(d, torch, variant):
    """welch_half3_kernel (paired samples, role-swapping units, window folded into the first butterfly stage, frame b's first half loaded a
    second time under the unit's have-frame-b predicate): every frame count parity, the odd last frame, one to three frames, several
    channels, a stream long enough for the Float64 fold (FLUSH = 128 units) and for slots that walk several runs -- against the oracle and
    against the round-2 kernel (same frames, same accumulators; only the rounding of the first stage differs)."""
    from dsp_jl_amd import _lib
    from oracle import periodograms as opg, windows as ow
    if variant not in (30, 43) and not _lib.lib().mdsp_debug_knobs():
        pytest.skip("product builds keep the default (43), its fallback (30) and the round-2 form (18); the variants that lost are built with -DMDSP_DEBUG_KNOBS only")
    rng = np.random.default_rng(300 + variant)
    try:
        for length in (4096, 6144, 8191, 8192, 10240, 12288, 100_000, 4096 * 700 + 2048, (1 << 23) + 4097):
            s = (rng.standard_normal(length) + 0.5 * np.sin(2 * np.pi * 0.1234 * np.arange(length))).astype(np.float32)
            _lib.set_tunable("MDSP_WELCH_VARIANT", str(variant))
            cfg = d.WelchConfig(length, np.float32, n=4096, noverlap=2048, window=d.hanning, engine=d.ENGINE_FUSED)
            got = d.welch_pgram(s, cfg).power
            again = d.welch_pgram(s, cfg).power
            _lib.set_tunable("MDSP_WELCH_VARIANT", "18")      # the round-2 default form (identity lanes, pad 5), explicitly
            old = d.welch_pgram(s, d.WelchConfig(length, np.float32, n=4096, noverlap=2048, window=d.hanning, engine=d.ENGINE_FUSED)).power
            assert np.array_equal(got, again)                                           # deterministic: config reuse is bit-identical
            if length <= 4096 * 700 + 2048:
                ref = opg.welch_pgram(s, 4096, 2048, window=ow.hanning, dtype=np.float64).power
                assert relerr(got, ref) < TOL32, (variant, length, relerr(got, ref))
            assert relerr(got, old.astype(np.float64)) < 1e-6, (variant, length, relerr(got, old.astype(np.float64)))
        # several channels in one launch == channel by channel
        _lib.set_tunable("MDSP_WELCH_VARIANT", str(variant))
        S = rng.standard_normal((50_000, 3)).astype(np.float32)
        cfg = d.WelchConfig(50_000, np.float32, n=4096, noverlap=2048, window=d.hamming, engine=d.ENGINE_FUSED)
        P = d.welch_pgram(S, cfg).power
        for c in range(3):
            assert np.array_equal(P[:, c], d.welch_pgram(S[:, c].copy(), cfg).power)
            assert relerr(P[:, c], opg.welch_pgram(S[:, c], 4096, 2048, window=ow.hamming, dtype=np.float64).power) < TOL32
    finally:
        _lib.set_tunable("MDSP_WELCH_VARIANT", None)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_filters_beyond_the_partitioned_range_stay_correct(d, torch, dt):
    """32768 taps: beyond the four partitions of the single-workgroup kernels (16384 Float32 / 8192 Float64 taps).  The SAME plan object takes them
    (no host-side segment sums since round 5): blocks of 2^19 (Float64: 2^18) points -- fewer when one block holds the signal -- on the multi-pass engine
    (bigfft.hip run_ols_rows: column pass, row kernel, column pass back), reported by mdsp_ols_plan_geometry.  Against the oracle (filt.jl:479-521), against an explicit rocFFT-engine plan at the
    reference's own block length (optimalfftfiltlength, dspbase.jl:268-291), several columns, signals shorter than the filter, conv, a block range
    (bit-identical to the whole-column call) and the host-pointer pipeline."""
    import ctypes as C
    from dsp_jl_amd import _lib
    from dsp_jl_amd.dspbase import OlsPlan
    from oracle import filt as ofilt
    lib = _lib.lib()
    rng = np.random.default_rng(32768)
    nb, nx = 32768, 2_700_000 + 11
    b = (rng.standard_normal(nb) / np.sqrt(nb)).astype(dt)
    x = rng.standard_normal(nx).astype(dt)
    xd = torch.from_numpy(x).cuda()
    got = d.filt(b, xd).cpu().numpy()
    ref = ofilt.fftfilt(b.astype(np.float64), x.astype(np.float64))
    tol = TOL32 if dt == np.float32 else 1e-12
    assert relerr(got, ref) < tol
    roc = d.fftfilt(b, xd, d.optimalfftfiltlength(nb, nx), engine=d.ENGINE_ROCFFT).cpu().numpy()
    assert relerr(roc, ref) < tol and relerr(got, roc.astype(np.float64)) < 2 * tol
    plan = OlsPlan(b, d.optimalfftfiltlength(nb, nx), nx, _lib.OLS_FILT, d.ENGINE_AUTO)
    en, el, ep = C.c_int64(), C.c_int64(), C.c_int()
    _lib.check(lib.mdsp_ols_plan_geometry(plan._h, C.byref(en), C.byref(el), C.byref(ep)))
    nbig = 64 * (8192 if dt == np.float32 else 4096)     # the rows form: 64 rows of the longest single-workgroup transform while the filter is under a quarter of that
    assert plan.engine == d.ENGINE_FUSED and (en.value, el.value, ep.value) == (nbig, nbig - nb + 1, 1)
    # blocks [2, 4) of the same grid from a slice of the signal -- a whole pair: two real blocks share a transform -- as the host pipeline and a time-axis
    # split over GPUs issue them: bit-identical to the whole-column call
    L = el.value
    lo, hi = 2 * L - (nb - 1), min(nx, 4 * L)
    ys = torch.full((hi - 2 * L,), float("nan"), dtype=xd.dtype, device="cuda")
    _lib.check(lib.mdsp_ols_exec_range(plan._h, xd[lo:hi].data_ptr(), lo, hi - lo, nx, ys.data_ptr(), 2, 2, nx, torch.cuda.current_stream().cuda_stream))
    assert np.array_equal(ys.cpu().numpy(), got[2 * L:hi])
    # host arrays: the chunked H2D || kernels || D2H pipeline on the same plan geometry
    assert np.array_equal(plan.exec_host(x.reshape(1, -1), nx)[0], got)
    # two columns, and a signal shorter than the filter
    X = rng.standard_normal((40_000, 2)).astype(dt)
    Y = d.filt(b, torch.from_numpy(X).cuda()).cpu().numpy()
    for c in range(2):
        assert relerr(Y[:, c], ofilt.fftfilt(b.astype(np.float64), X[:, c].astype(np.float64))) < tol, c
    from oracle import dspbase as odsp
    cv = d.conv(xd, torch.from_numpy(b).cuda()).cpu().numpy()
    assert cv.shape == (nx + nb - 1,)
    assert relerr(cv, odsp.conv(x.astype(np.float64), b.astype(np.float64))) < tol
    # an explicit FUSED request is the same plan (up to round 4: UnsupportedError)
    pf = OlsPlan(b, d.optimalfftfiltlength(nb, nx), nx, _lib.OLS_FILT, d.ENGINE_FUSED)
    assert np.array_equal(pf.exec(xd.reshape(1, -1), nx).cpu().numpy()[0], got)


@pytest.mark.parametrize("dt", [np.complex64, np.complex128])
def test_long_complex_filters_on_the_multipass_engine(d, torch, dt):
    """conv of complex signals with 20001 complex taps (one block per transform instead of two): against numpy's Float64 transform-domain product,
    and 150000 real taps in Float32 -- the block grows to 2^21 points (256 rows); MDSP_BIG_OLS_ROWS=0 (three passes each way on natural-order spectra, what filters
    beyond 2^19 taps take) gives the same convolution."""
    rng = np.random.default_rng(20001)
    nb, nx = 20001, 1_300_017
    b = ((rng.standard_normal(nb) + 1j * rng.standard_normal(nb)) / np.sqrt(nb)).astype(dt)
    x = (rng.standard_normal(nx) + 1j * rng.standard_normal(nx)).astype(dt)
    nf = 1 << 21
    ref = np.fft.ifft(np.fft.fft(x.astype(np.complex128), nf) * np.fft.fft(b.astype(np.complex128), nf))[:nx + nb - 1]
    got = d.conv(torch.from_numpy(x).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
    assert got.shape == ref.shape
    assert relerr(got, ref) < (TOL32 if dt == np.complex64 else 1e-12)
    if dt == np.complex64:
        import ctypes as C
        from dsp_jl_amd import _lib
        from dsp_jl_amd.dspbase import OlsPlan
        nb = 150_000
        h = (rng.standard_normal(nb) / np.sqrt(nb)).astype(np.float32)
        xr = rng.standard_normal(5_000_000).astype(np.float32)
        y = d.filt(h, torch.from_numpy(xr).cuda()).cpu().numpy()
        nf = 1 << 23
        r = np.fft.irfft(np.fft.rfft(xr.astype(np.float64), nf) * np.fft.rfft(h.astype(np.float64), nf), nf)[:len(xr)]
        assert relerr(y, r) < TOL32
        plan = OlsPlan(h, d.optimalfftfiltlength(nb, len(xr)), len(xr), _lib.OLS_FILT, d.ENGINE_AUTO)
        en = C.c_int64()
        _lib.check(_lib.lib().mdsp_ols_plan_geometry(plan._h, C.byref(en), None, None))
        assert en.value == 1 << 21
        try:
            _lib.set_tunable("MDSP_BIG_OLS_ROWS", 0)
            p3 = OlsPlan(h[:40_000].copy(), d.optimalfftfiltlength(40_000, len(xr)), len(xr), _lib.OLS_FILT, d.ENGINE_FUSED)
            _lib.check(_lib.lib().mdsp_ols_plan_geometry(p3._h, C.byref(en), None, None))
            assert en.value == 1 << 20
            y3 = p3.exec(torch.from_numpy(xr).cuda().reshape(1, -1), len(xr)).cpu().numpy()[0]
        finally:
            _lib.set_tunable("MDSP_BIG_OLS_ROWS", None)
        r3 = np.fft.irfft(np.fft.rfft(xr.astype(np.float64), nf) * np.fft.rfft(h[:40_000].astype(np.float64), nf), nf)[:len(xr)]
        assert relerr(y3, r3) < TOL32


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_long_filters_on_several_columns(d, torch, dt):
    """20001 taps on three columns of 1.2 M samples (the rows form: 64 rows of the longest single-workgroup transform, one column after the other on the same
    engine): every column bit-identical to the same column filtered alone, and against a Float64 transform-domain product."""
    rng = np.random.default_rng(3)
    nb, nx = 20001, 1_200_017
    b = (rng.standard_normal(nb) / np.sqrt(nb)).astype(dt)
    X = rng.standard_normal((nx, 3)).astype(dt)
    Xd = torch.from_numpy(X).cuda()
    Y = d.filt(b, Xd).cpu().numpy()
    assert Y.shape == X.shape
    for c in range(3):
        assert np.array_equal(Y[:, c], d.filt(b, Xd[:, c].contiguous()).cpu().numpy()), c
    nf = 1 << 21
    ref = np.fft.irfft(np.fft.rfft(X[:, 1].astype(np.float64), nf) * np.fft.rfft(b.astype(np.float64), nf), nf)[:nx]
    assert relerr(Y[:, 1], ref) < (TOL32 if dt == np.float32 else 1e-12)


def test_welch_hand_allocated_kernel_several_channels(d, torch):
    """mdsp_welch_w64c_asm with more than one channel per launch (grid (G, nch), rows part[((slot nch + ch) nflush + f)], the two-step row reduction
    per channel): three channels of 2^23 + 4096 + 2048 k samples (even and odd frame counts reach the kernel: it takes over from eight units per CU) --
    equal to the channels taken one by one (to rounding: the units are partitioned over the waves differently) and to the Float64 oracle."""
    from oracle import periodograms as opg, windows as ow
    rng = np.random.default_rng(77)
    for extra in (0, 2048):
        L = (1 << 23) + 4096 + extra
        S = (rng.standard_normal((L, 3)) + 0.25 * np.sin(2 * np.pi * 0.05 * np.arange(L))[:, None]).astype(np.float32)
        cfg = d.WelchConfig(L, np.float32, n=4096, noverlap=2048, window=d.hanning, engine=d.ENGINE_FUSED)
        P = d.welch_pgram(S, cfg).power
        for c in range(3):
            one = d.welch_pgram(S[:, c].copy(), cfg).power
            assert relerr(P[:, c], one.astype(np.float64)) < 1e-6, (extra, c)      # another partition of the units over waves: other Float32 run sums
        assert relerr(P[:, 1], opg.welch_pgram(S[:, 1], 4096, 2048, window=ow.hanning, dtype=np.float64).power) < TOL32


@pytest.mark.parametrize("engine", [1, 2], ids=["fused", "rocfft"])
def test_host_pipeline_stft(d, torch, engine):
    """mdsp_stft_exec_host: stft / spectrogram of host arrays in runs of whole frames on the three-stage pipeline -- the same frames the
    device-resident call transforms, so the columns are bit-identical; strided output matrices (ldo > nout), several channels with a
    leading dimension, page-locked arrays, the numpy API's own use of it, and the oracle."""
    from dsp_jl_amd import _lib
    from dsp_jl_amd.periodograms import _StftPlan, compute_window
    from oracle import periodograms as opg, windows as ow
    rng = np.random.default_rng(21)
    lib = _lib.lib()
    _lib.set_tunable("MDSP_HOST_CHUNK_MIB", 1)
    try:
        for sdt, L, nch, n, nov, nfft, psd in ((np.complex64, 600_000, 2, 1024, 768, 1024, 0), (np.float32, 900_001, 1, 512, 256, 512, 1),
                                              (np.float64, 200_000, 3, 300, 100, 512, 0), (np.float32, 700_000, 2, 1000, 250, 1000, 1)):
            cplx = np.dtype(sdt).kind == "c"
            s = rng.standard_normal((nch, L + 7)).astype(sdt)        # rows longer than the signal: lds > len
            if cplx:
                s = (s + 1j * rng.standard_normal((nch, L + 7))).astype(sdt)
            win, norm2 = compute_window(d.hanning, n)
            plan = _StftPlan(n, nov, nfft, win, 1.0 * norm2, not cplx, psd, sdt, engine)
            K = d.frame_count(L, n, nov)
            single = sdt in (np.float32, np.complex64)
            odt = (np.float32 if single else np.float64) if psd else (np.complex64 if single else np.complex128)
            ldo = plan.nout + 3
            dev_s = torch.from_numpy(s).cuda()
            dev_out = torch.zeros((nch, K, ldo), dtype=getattr(torch, np.dtype(odt).name), device="cuda")
            _lib.check(lib.mdsp_stft_exec(plan._h, dev_s.data_ptr(), L, nch, L + 7, dev_out.data_ptr(), ldo, K * ldo, None))
            torch.cuda.synchronize()
            host_out = np.zeros((nch, K, ldo), dtype=odt)
            _lib.check(lib.mdsp_stft_exec_host(plan._h, s.ctypes.data_as(C.c_void_p), L, nch, L + 7, host_out.ctypes.data_as(C.c_void_p), ldo, K * ldo, 0))
            assert np.array_equal(host_out, dev_out.cpu().numpy()), (sdt, n)
            # page-locked arrays
            pin_in, pin_out = C.c_void_p(), C.c_void_p()
            _lib.check(lib.mdsp_host_alloc(C.byref(pin_in), s.nbytes)); _lib.check(lib.mdsp_host_alloc(C.byref(pin_out), host_out.nbytes))
            try:
                C.memmove(pin_in, s.ctypes.data_as(C.c_void_p), s.nbytes)
                C.memset(pin_out, 0, host_out.nbytes)
                _lib.check(lib.mdsp_stft_exec_host(plan._h, pin_in, L, nch, L + 7, pin_out, ldo, K * ldo, _lib.HOST_PINNED))
                got = np.ctypeslib.as_array(C.cast(pin_out, C.POINTER(C.c_byte)), shape=(host_out.nbytes,)).view(odt).reshape(nch, K, ldo)
                assert np.array_equal(got, host_out)
            finally:
                lib.mdsp_host_free(pin_in); lib.mdsp_host_free(pin_out)
            # the oracle on the last channel (a few columns spread over the chunks)
            sig = s[nch - 1, :L]
            if psd:
                ref = opg.spectrogram(sig.astype(np.float64 if not cplx else np.complex128), n, nov, nfft=nfft, window=ow.hanning, onesided=not cplx).power
            else:
                ref = opg.stft(sig.astype(np.float64 if not cplx else np.complex128), n, nov, nfft=nfft, window=ow.hanning, onesided=not cplx)
            got = host_out[nch - 1, :, :plan.nout].T
            assert relerr(got, ref) < (TOL32 if single else 1e-12)
        # shorter than one frame: nothing is written
        z = np.full((1, 5, 10), 7.0, np.float32)
        win, norm2 = compute_window(d.hanning, 256)
        plan = _StftPlan(256, 128, 256, win, norm2, True, 1, np.float32, engine)
        _lib.check(lib.mdsp_stft_exec_host(plan._h, rng.standard_normal(100).astype(np.float32).ctypes.data_as(C.c_void_p), 100, 1, 100, z.ctypes.data_as(C.c_void_p), 129, 129, 0))
        assert np.all(z == 7.0)
    finally:
        _lib.set_tunable("MDSP_HOST_CHUNK_MIB", None)
    # numpy API: large host arrays take the pipeline on their own
    s = (rng.standard_normal(5_000_000) + 1j * rng.standard_normal(5_000_000)).astype(np.complex64)
    S_host = d.stft(s, 1024, 768, window=d.hanning, engine=engine)
    S_dev = d.stft(torch.from_numpy(s).cuda(), 1024, 768, window=d.hanning, engine=engine)
    assert isinstance(S_host, np.ndarray) and S_host.shape == tuple(S_dev.shape) and np.array_equal(S_host, S_dev.cpu().numpy())
    x = rng.standard_normal(9_000_000).astype(np.float32)
    P = d.spectrogram(x, 512, 256, window=d.hanning, engine=engine)
    Pd = d.spectrogram(torch.from_numpy(x).cuda(), 512, 256, window=d.hanning, engine=engine)
    assert isinstance(P.power, np.ndarray) and np.array_equal(P.power, Pd.power.cpu().numpy()) and np.array_equal(P.time, Pd.time)


def test_host_pipeline_polyphase_filter(d, torch):
    """mdsp_fir_exec_host: a host stream through the stateful polyphase filter in time chunks of all channels -- output and final state are those
    of ONE device-resident mdsp_fir_exec (bit for bit), for rational / interpolating / decimating / single-rate filters, Float32 and Float64,
    with a non-trivial initial state, page-locked arrays, and the numpy API (FIRFilter.filt / resample)."""
    from fractions import Fraction
    from dsp_jl_amd import _lib
    from oracle import stream_filt as osf, design as odes
    rng = np.random.default_rng(33)
    lib = _lib.lib()
    _lib.set_tunable("MDSP_HOST_CHUNK_MIB", 1)
    try:
        for dt, ld, (L, M), nch, n in ((np.float32, _lib.F32, (160, 147), 2, 1_200_003), (np.float64, _lib.F64, (3, 2), 1, 500_000), (np.float32, _lib.F32, (1, 4), 3, 900_000),
                                       (np.float32, _lib.F32, (2, 1), 1, 400_001), (np.float32, _lib.F32, (1, 1), 2, 300_000)):
            h = np.asarray(odes.resample_filter(Fraction(L, M)) if (L, M) != (1, 1) else _taps(48, np.float64), dtype=dt)
            x = rng.standard_normal((nch, n + 5)).astype(dt)          # ldx > xlen
            res = {}
            for mode in ("dev", "host", "pinned"):
                f = C.c_void_p()
                _lib.check(lib.mdsp_fir_create(C.byref(f), h.ctypes.data_as(C.c_void_p), len(h), L, M, ld, ld, nch))
                # a non-trivial state: run a short prefix first (device call), then the stream
                pre = torch.from_numpy(np.ascontiguousarray(x[:, :1000])).cuda()
                ol = C.c_int64(); _lib.check(lib.mdsp_fir_outputlength(f, 1000, C.byref(ol)))
                ypre = torch.empty((nch, ol.value + 1), dtype=pre.dtype, device="cuda")
                nw = C.c_int64()
                _lib.check(lib.mdsp_fir_exec(f, pre.data_ptr(), 1000, 1000, ypre.data_ptr(), ol.value, ol.value + 1, C.byref(nw), None))
                rest = n - 1000
                _lib.check(lib.mdsp_fir_outputlength(f, rest, C.byref(ol)))
                ldy = ol.value + 2
                if mode == "dev":
                    xd = torch.from_numpy(x).cuda()
                    yd = torch.zeros((nch, ldy), dtype=xd.dtype, device="cuda")
                    _lib.check(lib.mdsp_fir_exec(f, xd.data_ptr() + 1000 * x.itemsize, rest, n + 5, yd.data_ptr(), ol.value, ldy, C.byref(nw), None))
                    torch.cuda.synchronize()
                    y = yd.cpu().numpy()
                elif mode == "host":
                    y = np.zeros((nch, ldy), dtype=dt)
                    _lib.check(lib.mdsp_fir_exec_host(f, C.c_void_p(x.ctypes.data + 1000 * x.itemsize), rest, n + 5, y.ctypes.data_as(C.c_void_p), ol.value, ldy, C.byref(nw), 0))
                else:
                    pin_in, pin_out = C.c_void_p(), C.c_void_p()
                    _lib.check(lib.mdsp_host_alloc(C.byref(pin_in), x.nbytes)); _lib.check(lib.mdsp_host_alloc(C.byref(pin_out), nch * ldy * x.itemsize))
                    C.memmove(pin_in, x.ctypes.data_as(C.c_void_p), x.nbytes)
                    C.memset(pin_out, 0, nch * ldy * x.itemsize)
                    _lib.check(lib.mdsp_fir_exec_host(f, C.c_void_p(pin_in.value + 1000 * x.itemsize), rest, n + 5, pin_out, ol.value, ldy, C.byref(nw), _lib.HOST_PINNED))
                    y = np.ctypeslib.as_array(C.cast(pin_out, C.POINTER(C.c_byte)), shape=(nch * ldy * x.itemsize,)).view(dt).reshape(nch, ldy).copy()
                    lib.mdsp_host_free(pin_in); lib.mdsp_host_free(pin_out)
                assert nw.value == ol.value
                phi, dfc = C.c_int64(), C.c_int64()
                hl = C.c_int64(); _lib.check(lib.mdsp_fir_info(f, None, None, None, None, C.byref(hl), None))
                hist = np.zeros((nch, max(hl.value, 1)), dtype=dt)
                _lib.check(lib.mdsp_fir_get_state(f, C.byref(phi), C.byref(dfc), hist.ctypes.data_as(C.c_void_p)))
                res[mode] = (y, phi.value, dfc.value, hist)
                _lib.check(lib.mdsp_fir_destroy(f))
            for mode in ("host", "pinned"):
                assert np.array_equal(res[mode][0], res["dev"][0]), (L, M, mode)
                assert res[mode][1:3] == res["dev"][1:3] and np.array_equal(res[mode][3], res["dev"][3]), (L, M, mode)
            # the oracle on channel 0 (prefix + stream through one oracle filter)
            of = osf.FIRFilter(h.astype(np.float64), Fraction(L, M))
            of.filt(x[0, :1000].astype(np.float64))
            ref = of.filt(x[0, 1000:n].astype(np.float64))
            assert relerr(res["host"][0][0, :len(ref)], ref) < (2e-6 if dt == np.float32 else 1e-12)
    finally:
        _lib.set_tunable("MDSP_HOST_CHUNK_MIB", None)
    # numpy API: a large host array through FIRFilter.filt and resample
    x = rng.standard_normal(9_000_000).astype(np.float32)
    ratio = Fraction(160, 147)
    h = np.asarray(d.resample_filter(ratio), dtype=np.float32)
    y_host = d.FIRFilter(h, ratio).filt(x)
    y_dev = d.FIRFilter(h, ratio).filt(torch.from_numpy(x).cuda())
    assert isinstance(y_host, np.ndarray) and np.array_equal(y_host, y_dev.cpu().numpy())
    r_host = d.resample(x, ratio)
    assert np.array_equal(r_host, d.resample(torch.from_numpy(x).cuda(), ratio).cpu().numpy())


@pytest.mark.parametrize("variant", [36, 37])
def test_overlap_save_lds_dma_staging_is_bit_identical(d, torch, variant):
    from dsp_jl_amd import _lib as _l
    if not _l.lib().mdsp_debug_knobs():
        pytest.skip("variants 36 / 37 lost their A/B and are built with -DMDSP_DEBUG_KNOBS only (round 5)")
    """ols_fused_kernel<..., XDMA>: the next unit's span goes HBM -> LDS by buffer_load ... lds while this unit is transformed.  Only WHERE the
    samples wait changes, so outputs must equal the direct-load kernel bit for bit: signals of every edge shape (shorter than a block, one unit,
    an odd block count, leading zero padding, the clamped tail), several columns (units of different columns alternate in a slot), conv mode
    (blocks past the end of x), block ranges from a slice (mdsp_ols_exec_range), and a long stream whose slots walk many interior units."""
    from dsp_jl_amd import _lib
    from dsp_jl_amd.dspbase import OlsPlan
    from oracle import dspbase as odsp
    rng = np.random.default_rng(360 + variant)
    lib = _lib.lib()
    b = _taps(256, np.float32)
    try:
        for nx, ncols, mode in ((100, 1, _lib.OLS_FILT), (1793, 1, _lib.OLS_FILT), (3586, 1, _lib.OLS_FILT), (3587, 2, _lib.OLS_FILT), (10_000, 3, _lib.OLS_FILT),
                                (1_000_003, 1, _lib.OLS_FILT), (700_001, 2, _lib.OLS_CONV), ((1 << 24) + 12345, 1, _lib.OLS_FILT), (3_000_000, 5, _lib.OLS_FILT)):
            x = torch.from_numpy(rng.standard_normal((ncols, nx)).astype(np.float32)).cuda()
            nout = nx if mode == _lib.OLS_FILT else nx + 255
            _lib.set_tunable("MDSP_OLS_VARIANT", "0")
            ref = OlsPlan(b, 2048, nx, mode, d.ENGINE_FUSED).exec(x, nout)
            _lib.set_tunable("MDSP_OLS_VARIANT", str(variant))
            plan = OlsPlan(b, 2048, nx, mode, d.ENGINE_FUSED)
            got = plan.exec(x, nout)
            torch.cuda.synchronize()
            assert torch.equal(got, ref), (variant, nx, ncols, mode)
            assert torch.equal(plan.exec(x, nout), ref)                       # and again: the staging buffer carries nothing over between launches
            if nx == 1_000_003:
                want = odsp.filt_ba(b.astype(np.float64), 1.0, x[0, :60000].cpu().numpy().astype(np.float64))
                assert relerr(got[0, :60000].cpu().numpy(), want) < TOL32
                # a block range from a slice of the signal (host pipeline / time-axis split): blocks [100, 300) of the same grid
                L, g0, g1 = 1793, 100, 300
                lo, hi = g0 * L - 255, g1 * L
                xs = x[0, lo:hi].contiguous()
                ys = torch.empty(hi - g0 * L, dtype=torch.float32, device="cuda")
                _lib.check(lib.mdsp_ols_exec_range(plan._h, xs.data_ptr(), lo, hi - lo, nx, ys.data_ptr(), g0, g1 - g0, nx, None))
                torch.cuda.synchronize()
                assert torch.equal(ys, ref[0, g0 * L:hi])
    finally:
        _lib.set_tunable("MDSP_OLS_VARIANT", None)


@pytest.mark.parametrize("nfft", [1000, 1200, 1280, 1500, 1536, 1600, 1920, 2000, 2400, 2500, 2560, 3000, 3072, 3200, 3840, 4000, 4800, 5000, 5120, 6000, 6144, 6400, 8000])
def test_compile_time_mixed_radix_schedules(d, torch, nfft):
    """Every size with a compile-time schedule (spectral_gen.h MDSP_GEN_CT_SIZES): Welch, raw STFT and spectrogram of Float32 and ComplexF32
    signals against the Float64 oracle -- full-length and short windows, even and odd frame counts, two channels -- and AUTO takes the fused
    engine for them (they beat the rocFFT pipeline in every mode, profiles/r03d_mixed_ct.json)."""
    from oracle import periodograms as opg, windows as ow
    rng = np.random.default_rng(nfft)
    for dt in (np.float32, np.complex64, np.float64, np.complex128):      # Float64 beyond 3000 points: one LDS buffer for Welch / complex columns (round 4); real columns up to 5000, else rocFFT
        cplx = np.dtype(dt).kind == "c"
        TOL = TOL32 if dt in (np.float32, np.complex64) else 1e-12
        for n in (nfft, nfft - 7):
            nov = n // 2
            hop = n - nov
            L = hop * 20 + n + (hop if n == nfft else 0)                  # 21 / 22 frames
            x = rng.standard_normal((L, 2)).astype(np.float32 if dt in (np.float32, np.complex64) else np.float64)
            if cplx:
                x = (x + 1j * rng.standard_normal(x.shape)).astype(dt)
            xd = torch.from_numpy(x).cuda()
            cfg = d.WelchConfig(L, dt, n=n, noverlap=nov, nfft=nfft, window=d.hanning)
            assert cfg.engine == d.ENGINE_FUSED
            P = d.welch_pgram(xd, cfg).power.cpu().numpy()
            for c in range(2):
                assert relerr(P[:, c], opg.welch_pgram(x[:, c], n, nov, nfft=nfft, window=ow.hanning, dtype=np.float64).power) < TOL, ("welch", dt, n, c)
            for onesided in ((True, False) if not cplx else (False,)):
                S = d.stft(xd, n, nov, nfft=nfft, onesided=onesided, window=d.hanning).cpu().numpy()
                assert relerr(S[:, :, 1], opg.stft(x[:, 1], n, nov, nfft=nfft, onesided=onesided, window=ow.hanning, dtype=np.float64)) < TOL, ("stft", dt, n, onesided)
            sp = d.spectrogram(xd[:, 0].contiguous(), n, nov, nfft=nfft, fs=2.0, window=d.hamming).power.cpu().numpy()
            assert relerr(sp, opg.stft(x[:, 0], n, nov, psdonly=True, nfft=nfft, fs=2.0, window=ow.hamming, dtype=np.float64)) < TOL, ("spectrogram", dt, n)


@pytest.mark.gpu
@pytest.mark.parametrize("Tx", [np.float32, np.float64, np.complex64, np.complex128])
def test_decimator_kernel_vs_oracle_streaming_and_nonfinite(d, torch, Tx):
    """FIRDecimator (stream_filt.jl:43-56, :522-558) on the phase-per-lane decimator kernel: every M up to 64 (powers of two and not), filters shorter
    than M, lengths that are not multiples of M, several channels, tiles with a ragged end; against the Float64 oracle (<= 2e-6 Float32, 1e-12
    Float64 -- the kernel sums phases first, the reference oldest sample first); chunked == one-shot BIT FOR BIT (an output's arithmetic does not
    depend on where its tile starts) with the reference's state after every chunk; NaN / Inf samples leave EXACTLY the reference's hole."""
    import ctypes as C
    from fractions import Fraction
    from dsp_jl_amd import _lib
    from oracle import stream_filt as osf
    lib = _lib.lib()
    rng = np.random.default_rng(2026)
    cplx = np.dtype(Tx).kind == "c"
    dbl = np.dtype(Tx) in (np.dtype(np.float64), np.dtype(np.complex128))
    Th = np.float64 if dbl else np.float32
    tol = 1e-12 if dbl else 2e-6
    lx = {np.float32: _lib.F32, np.float64: _lib.F64, np.complex64: _lib.C32, np.complex128: _lib.C64}[Tx]
    lt = _lib.F64 if dbl else _lib.F32
    tdt = {np.float32: torch.float32, np.float64: torch.float64, np.complex64: torch.complex64, np.complex128: torch.complex128}[Tx]
    stream = torch.cuda.current_stream().cuda_stream
    _lib.set_tunable("MDSP_FIR_DEC", 3)     # the decimator kernel for every M <= 64 (by default only where it measured faster than the matrix-core kernel)
    try:
        _decimator_cases(d, torch, Tx, lib, rng, cplx, dbl, Th, tol, lx, lt, tdt, stream)
    finally:
        _lib.set_tunable("MDSP_FIR_DEC", None)


def _decimator_cases(d, torch, Tx, lib, rng, cplx, dbl, Th, tol, lx, lt, tdt, stream):
    import ctypes as C
    from fractions import Fraction
    from dsp_jl_amd import _lib
    from oracle import stream_filt as osf
    for (M, ntaps, n, nch) in ((2, 48, 100_003, 1), (3, 7, 50_000, 2), (4, 147, 70_001, 1), (5, 3, 20_000, 1), (8, 291, 300_000, 3), (16, 581, 400_017, 2),
                               (17, 600, 123_457, 1), (33, 1000, 200_000, 1), (64, 2309, 500_000, 1), (16, 5000, 150_000, 1), (7, 1, 30_000, 1)):
        h = (rng.standard_normal(ntaps) / np.sqrt(ntaps)).astype(Th)
        x = rng.standard_normal((nch, n))
        x = (x + 1j * rng.standard_normal((nch, n))).astype(Tx) if cplx else x.astype(Tx)
        xd = torch.from_numpy(x).cuda()
        fh = C.c_void_p()
        _lib.check(lib.mdsp_fir_create(C.byref(fh), h.ctypes.data_as(C.c_void_p), len(h), 1, M, lt, lx, nch))
        pth = C.c_int(-1); _lib.check(lib.mdsp_fir_kernel_path(fh, n, C.byref(pth)))
        assert pth.value == 3 or (ntaps == 5000 and dbl), (M, ntaps)   # (5000 Float64 taps at M = 16 do not fit the LDS next to a tile: the older kernels take it)
        ol = C.c_int64(); _lib.check(lib.mdsp_fir_outputlength(fh, n, C.byref(ol)))
        y = torch.zeros((nch, ol.value + 2), dtype=tdt, device="cuda")
        nw = C.c_int64()
        _lib.check(lib.mdsp_fir_exec(fh, xd.data_ptr(), n, n, y.data_ptr(), ol.value, ol.value + 2, C.byref(nw), stream))
        torch.cuda.synchronize()
        assert nw.value == ol.value and not y[:, ol.value:].abs().any()
        one = y[:, :ol.value].cpu().numpy()
        for c in range(nch):
            ref = osf.FIRFilter(h.astype(np.float64), Fraction(1, M)).filt(x[c].astype(np.complex128 if cplx else np.float64))
            assert ref.shape == one[c].shape and relerr(one[c], ref) < tol, (M, ntaps, c, relerr(one[c], ref))
        # streaming: three ragged chunks, the oracle's state after each, outputs bit-identical to the one-shot run
        _lib.check(lib.mdsp_fir_reset(fh))
        o = osf.FIRFilter(h, Fraction(1, M))
        cuts = [0, n // 3 + 1, n // 3 + 2, n]
        pieces = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            olc = C.c_int64(); _lib.check(lib.mdsp_fir_outputlength(fh, b - a, C.byref(olc)))
            yc = torch.zeros((nch, max(1, olc.value)), dtype=tdt, device="cuda")
            _lib.check(lib.mdsp_fir_exec(fh, xd[:, a:].data_ptr(), b - a, n, yc.data_ptr(), olc.value, max(1, olc.value), C.byref(nw), stream))
            torch.cuda.synchronize()
            pieces.append(yc[:, :nw.value].cpu().numpy())
            o.filt(x[0, a:b])
            p_, d_ = C.c_int64(), C.c_int64()
            hist = np.zeros((nch, max(1, ntaps - 1)), dtype=Tx)
            _lib.check(lib.mdsp_fir_get_state(fh, C.byref(p_), C.byref(d_), hist.ctypes.data_as(C.c_void_p)))
            assert (p_.value, d_.value) == (o.phi_idx, o.input_deficit)
            if ntaps > 1:
                assert np.array_equal(hist[0, :ntaps - 1], o.history.astype(Tx))
        assert np.array_equal(np.concatenate(pieces, axis=1), one), (M, ntaps)
        _lib.check(lib.mdsp_fir_destroy(fh))
    # non-finite samples: exactly the reference's hole on the DEFAULT path
    for (M, ntaps) in ((2, 48), (16, 581), (5, 33)):
        n = 60_001
        h = (rng.standard_normal(ntaps) / np.sqrt(ntaps)).astype(Th)
        h[np.abs(h) < 1e-3] = 1e-3
        xr = rng.standard_normal(n)
        for pos, v in {5: np.nan, 20_000: np.inf, 20_003: -np.inf, 41_111: np.nan, n - 2: np.inf}.items():
            xr[pos] = v
        x = xr.astype(Tx)
        with np.errstate(invalid="ignore", over="ignore"):
            ref = osf.filt_stateless(h.astype(np.float64), x.astype(np.complex128 if cplx else np.float64), Fraction(1, M))
        got = np.asarray(d.filt(h, x, Fraction(1, M)))
        bad = ~np.isfinite(ref)
        assert 0 < bad.sum() < len(ref) // 5 and np.array_equal(~np.isfinite(got), bad), (M, ntaps)
        assert relerr(got[~bad], ref[~bad]) < tol
