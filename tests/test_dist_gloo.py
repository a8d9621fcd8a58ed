"""world_size-2 `gloo` test (CPU, no GPU) of the multi-GPU path: channels shard across ranks with no data-path
collective; the only exchange is the all-reduce(sum) of the per-rank PSD sums for the cross-channel Welch mean
(SURVEY section 8e).  On the GPU the per-rank PSDs come from libmi355dsp; here the oracle stands in for them so that
the sharding arithmetic and the collective are exercised end to end."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nch_total, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dsp_jl_amd as d
    from oracle import periodograms as opg, windows as ow
    rng = np.random.default_rng(1776)                      # every rank draws the same full data set, then takes its shard
    S = rng.standard_normal((nch_total, 6000)).astype(np.float32)
    mine = d.channel_shard(nch_total, rank, world)
    local_sum = np.zeros(129, dtype=np.float64)
    for c in mine:
        local_sum += opg.welch_pgram(S[c], 256, 128, window=ow.hanning).power
    t = torch.from_numpy(local_sum)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)               # the single collective of the path
    mean = t.numpy() / nch_total
    out_q.put((rank, list(mine), mean))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nch_total", [5, 8])
def test_channel_sharding_and_allreduce_world2(nch_total):
    from oracle import periodograms as opg, windows as ow
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nch_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    covered = sorted(c for _, chans, _ in results for c in chans)
    assert covered == list(range(nch_total))               # every channel on exactly one rank
    rng = np.random.default_rng(1776)
    S = rng.standard_normal((nch_total, 6000)).astype(np.float32)
    ref = np.mean([opg.welch_pgram(S[c], 256, 128, window=ow.hanning).power.astype(np.float64) for c in range(nch_total)], axis=0)
    for _, _, mean in results:
        assert np.allclose(mean, ref, rtol=1e-12, atol=0)


def test_channel_shard_partition():
    import dsp_jl_amd as d
    for n in (1, 7, 8, 32, 64, 65):
        for w in (1, 2, 4, 8):
            parts = [list(d.channel_shard(n, r, w)) for r in range(w)]
            assert sorted(c for p in parts for c in p) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= -(-n // w)
    assert list(d.channel_shard(64, 3, 8)) == list(range(24, 32))      # config 4: 8 channels per GPU
    assert list(d.channel_shard(32, 7, 8)) == list(range(28, 32))      # config 5: 4 channels per GPU


def _time_worker(rank, world, port, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dsp_jl_amd as d
    from oracle import periodograms as opg, windows as ow, dspbase as odsp
    rng = np.random.default_rng(1776)
    x = rng.standard_normal(50000)
    n, nov, nb = 512, 384, 97
    b = rng.standard_normal(nb)
    K = opg.frame_count(len(x), n, nov)
    frames = d.frame_shard(K, rank, world)
    lo, hi = d.frame_span(frames, n, nov)
    local = opg.welch_pgram(x[lo:hi], n, nov, window=ow.hanning).power.astype(np.float64) * len(frames)
    t = torch.from_numpy(local.copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)               # the single collective of the time-split Welch
    psd = t.numpy() / K
    # filtering: contiguous output ranges, nb - 1 samples of halo, no collective
    per = -(-len(x) // world)
    olo, ohi = rank * per, min(len(x), (rank + 1) * per)
    slo, shi = d.filt_time_split_span(olo, ohi, nb)
    y = odsp.filt_ba(b, 1.0, x[slo:shi])[olo - slo:]
    out_q.put((rank, (lo, hi, len(frames)), psd, (olo, ohi), y))
    dist.barrier()
    dist.destroy_process_group()


def test_time_axis_split_world2():
    """One stream over two ranks (SURVEY 8e "next"): frames / output ranges split along time, halo slices, frame-count-weighted
    all-reduce for the PSD, no collective for filt -- against the whole-stream oracle."""
    from oracle import periodograms as opg, windows as ow, dspbase as odsp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_time_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=180) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(1776)
    x = rng.standard_normal(50000)
    b = rng.standard_normal(97)
    ref = opg.welch_pgram(x, 512, 384, window=ow.hanning).power
    K = opg.frame_count(len(x), 512, 384)
    assert sum(r[1][2] for r in results) == K and results[0][1][1] - results[1][1][0] == 384      # all frames once; n - hop samples shared
    for r in results:
        assert np.allclose(r[2], ref, rtol=1e-12, atol=0)
    yref = odsp.filt_ba(b, 1.0, x)
    assert np.allclose(np.concatenate([r[4] for r in results]), yref, rtol=1e-12, atol=1e-12)
