"""The N > 1 path on real hardware: two ranks, one process per GPU, RCCL over xGMI behind the C ABI (`mdsp_comm_init_rank`,
`mdsp_welch_mean_allreduce`, `mdsp_allreduce_sum`).  Skipped below two visible devices -- the first multi-GPU box that runs
`pytest -m gpu` exercises `ncclCommInitRank` with nranks = 2 as a TEST, not only as a bench line.

    * welch_channel_mean(comm=Comm(...)): channels sharded over the ranks, ONE all-reduce of nout values, against the single-rank result of
      the same library and against the oracle;
    * welch_time_split(comm=...): one stream split along time, Float64 sums all-reduced, finalized with the TOTAL frame count, against
      welch_pgram of the whole stream on one rank.

Rendezvous for the 128-byte RCCL id: torch.distributed `gloo` on 127.0.0.1 (only the id travels on it).  The CPU twin of this test
(tests/test_dist_gloo.py) runs the same product functions with world_size 2 on oracle stand-ins.
"""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data(nch, n):
    rng = np.random.default_rng(1776)
    t = np.arange(n)
    return (rng.standard_normal((nch, n)) + 0.5 * np.sin(2 * np.pi * 0.1234 * t)).astype(np.float32)


NCH, LEN, N, NOV = 5, 300_000, 4096, 2048       # 5 channels over 2 ranks: 3 + 2 (uneven shard)


def _worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(rank)
        import dsp_jl_amd as d
        from dsp_jl_amd import _lib, channels as ch
        _lib.check(_lib.lib().mdsp_init(rank))
        comm = d.Comm.from_torch_distributed()
        assert comm.nranks == world
        S = _data(NCH, LEN)
        mine = d.channel_shard(NCH, rank, world)
        cols = torch.from_numpy(S[mine.start:mine.stop].copy()).cuda()
        cfg = d.WelchConfig(LEN, np.float32, n=N, noverlap=NOV, window=d.hanning, engine=d.ENGINE_FUSED)
        mean = ch.welch_channel_mean(cols, cfg, nch_total=NCH, comm=comm)
        torch.cuda.synchronize()
        # the same reduction with torch.distributed as the transport must agree bit for bit on the summands and to rounding on the sum
        # one stream over two ranks, split along time
        x = S[0]
        K = d.frame_count(LEN, N, NOV)
        fr = ch.frame_shard(K, rank, world)
        lo, hi = ch.frame_span(fr, N, NOV)
        psd = ch.welch_time_split(torch.from_numpy(x[lo:hi].copy()).cuda(), K, N, NOV, comm=comm, window=d.hanning, engine=d.ENGINE_FUSED)
        # the bare collective
        t = torch.full((1000,), float(rank + 1), dtype=torch.float64, device="cuda")
        comm.allreduce_sum(t)
        # the statistic of `bench.py --config resample` (BASELINE config 5 "RCCL avg"): resample this rank's channels, sum the last TAIL outputs over them
        # (mdsp_channel_sum), ONE all-reduce of TAIL floats, 1 / nch_total
        import ctypes as C
        from fractions import Fraction
        lib = _lib.lib()
        TAIL, RLEN = 256, 40_000
        h = np.asarray(d.resample_filter(Fraction(160, 147)), dtype=np.float32)
        nloc = len(mine)
        xr = torch.from_numpy(S[mine.start:mine.stop, :RLEN].copy()).cuda()
        fh = C.c_void_p()
        _lib.check(lib.mdsp_fir_create(C.byref(fh), h.ctypes.data_as(C.c_void_p), len(h), 160, 147, _lib.F32, _lib.F32, nloc))
        ol = C.c_int64(); _lib.check(lib.mdsp_fir_outputlength(fh, RLEN, C.byref(ol)))
        yr = torch.empty((nloc, ol.value), dtype=torch.float32, device="cuda")
        nw = C.c_int64()
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.mdsp_fir_exec(fh, xr.data_ptr(), RLEN, RLEN, yr.data_ptr(), ol.value, ol.value, C.byref(nw), st))
        avg = torch.empty(TAIL, dtype=torch.float32, device="cuda")
        tail = yr[:, ol.value - TAIL:]
        _lib.check(lib.mdsp_channel_sum(tail.data_ptr(), TAIL, nloc, ol.value, _lib.F32, avg.data_ptr(), st))
        _lib.check(lib.mdsp_allreduce_sum(comm._h, avg.data_ptr(), TAIL, _lib.F32, st))
        avg.mul_(1.0 / NCH)
        _lib.check(lib.mdsp_fir_destroy(fh))
        torch.cuda.synchronize()
        q.put((rank, "ok", mean.cpu().numpy(), psd.cpu().numpy(), float(t[0]), list(mine), (lo, hi), avg.cpu().numpy()))
        dist.barrier()
        comm.close()
        dist.destroy_process_group()
    except Exception as e:   # pragma: no cover
        import traceback
        q.put((rank, "error", traceback.format_exc(), None, None, None, None, None))
        raise e


def test_rccl_two_ranks_welch_channel_mean_and_time_split():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible devices (RCCL with nranks = 2)")
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    for r in results:
        assert r[1] == "ok", r[2]
    assert all(p.exitcode == 0 for p in procs)
    # single-rank results of the same library, and the oracle
    import dsp_jl_amd as d
    from dsp_jl_amd import _lib
    from oracle import periodograms as opg, windows as ow
    _lib.check(_lib.lib().mdsp_init(0))
    S = _data(NCH, LEN)
    cfg = d.WelchConfig(LEN, np.float32, n=N, noverlap=NOV, window=d.hanning, engine=d.ENGINE_FUSED)
    per_ch = np.stack([d.welch_pgram(S[c].copy(), cfg).power for c in range(NCH)]).astype(np.float64)
    ref_mean = per_ch.mean(axis=0)
    ora_mean = np.mean([opg.welch_pgram(S[c], N, NOV, window=ow.hanning, dtype=np.float64).power for c in range(NCH)], axis=0)
    whole = d.welch_pgram(S[0].copy(), cfg).power.astype(np.float64)
    covered = sorted(c for r in results for c in r[5])
    assert covered == list(range(NCH))
    # config 5's statistic: the oracle resamples every channel, the mean of the last 256 outputs over all five channels
    from fractions import Fraction
    from oracle import stream_filt as osf
    from oracle import design as odes
    h64 = np.asarray(odes.resample_filter(Fraction(160, 147)), dtype=np.float32).astype(np.float64)
    ora_tail = np.mean([osf.FIRFilter(h64, Fraction(160, 147)).filt(S[c, :40_000].astype(np.float64))[-256:] for c in range(NCH)], axis=0)   # (a fresh filter: no undelay, as the workers)
    for rank, _, mean, psd, summed, _, _, avg in results:
        assert relerr(avg, ora_tail) < 2e-6, rank
    assert np.array_equal(results[0][7], results[1][7])
    for rank, _, mean, psd, summed, _, _, _ in results:
        assert summed == 3.0                                               # 1 + 2 over the two ranks
        assert relerr(mean, ref_mean) < 1e-6, rank                        # Float32 sum of five PSDs in a different order
        assert relerr(mean, ora_mean) < 5e-6, rank
        assert relerr(psd, whole) < 1e-6, rank                             # same frames, same Float64 sums, same normalisation
    assert np.array_equal(results[0][2], results[1][2])                    # every rank holds the same mean after the all-reduce
    assert np.array_equal(results[0][3], results[1][3])
