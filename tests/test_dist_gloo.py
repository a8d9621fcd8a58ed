"""world_size-2 `gloo` tests (CPU, no GPU) of the multi-GPU path: channels shard across ranks with no data-path
collective; the only exchange is the all-reduce(sum) of the per-rank PSD sums for the cross-channel Welch mean
(SURVEY section 8e).  The workers run the PRODUCT functions -- dsp_jl_amd.channels.welch_channel_mean / welch_time_split, their
sharding arithmetic, their collective call, their normalisation -- with world_size 2; only the two device hooks those functions
call (the per-channel PSD provider and the device buffers) are replaced by CPU stand-ins built on the oracle, because there is
no GPU here.  On the GPU box the same functions run with libmi355dsp underneath (tests/test_gpu_parity.py, test_gpu_multigpu.py)."""
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


class _CpuDev:
    """Stand-in for dsp_jl_amd._dev in the CPU workers: CPU tensors instead of HBM buffers."""
    import dsp_jl_amd._dev as _real
    np_dtype_of = staticmethod(_real.np_dtype_of)
    torch_dtype = staticmethod(_real.torch_dtype)
    md_dtype = staticmethod(_real.md_dtype)

    @staticmethod
    def to_columns(x, dtype):
        t = torch.as_tensor(np.asarray(x)).to(_CpuDev.torch_dtype(dtype))
        shape = tuple(t.shape)
        return t.reshape(shape[0], -1).t().contiguous(), shape


class _CpuWelchConfig:
    """Stand-in for WelchConfig's device plan: the oracle's frames and fft2pow, Float64 sums, same reset / accumulate / finalize protocol."""

    def __init__(self, nsamples, eltype, n, noverlap, window=None, nfft=None, onesided=None, fs=1, **kw):
        from oracle import periodograms as opg, util as outil
        self.n, self.noverlap = n, noverlap
        self.nfft = outil.nextfastfft(n) if nfft is None else nfft
        self.onesided = True if onesided is None else onesided
        self.win, norm2 = opg.compute_window(window, n)
        self.r = fs * norm2
        self.intype = outil.fftintype(np.dtype(eltype))
        self.nout = self.nfft // 2 + 1 if self.onesided else self.nfft
        self.reset()

    def reset(self):
        self.acc = torch.zeros(self.nout, dtype=torch.float64)
        self.frames = 0
        return self

    def accumulate(self, cols):
        from oracle import periodograms as opg
        x = cols[0].numpy()
        if len(x) >= self.n:
            fr = opg.arraysplit(x, self.n, self.noverlap, self.nfft, self.win, dtype=np.float64)
            p = opg.fft2pow(opg._forward(fr), self.nfft, 1.0, self.onesided, np.float64)
            self.acc += torch.from_numpy(p.sum(axis=0))
            self.frames += fr.shape[0]
        return self

    def accumulator(self):
        return self.acc

    def finalize(self, frames_total=0, nch=1):
        k = frames_total or self.frames
        return (self.acc / (k * self.r)).reshape(1, -1)


def _install_cpu_hooks():
    """Replace the device hooks of dsp_jl_amd.channels (and nothing else of it) for a GPU-less worker process."""
    import dsp_jl_amd.channels as ch
    from oracle import periodograms as opg, windows as ow

    def cpu_welch_exec(cols, config):          # (nch_local, len) -> (nch_local, nout), like periodograms._welch_exec
        rows = [opg.welch_pgram(c.numpy(), config.n, config.noverlap, window=ow.hanning, dtype=np.float64).power for c in cols]
        return torch.from_numpy(np.stack(rows)) if rows else torch.zeros((0, config.nout), dtype=torch.float64)

    def cpu_channel_sum(psd, nout, T):
        return psd.sum(dim=0) if psd.shape[0] else torch.zeros(nout, dtype=psd.dtype)

    ch._welch_exec = cpu_welch_exec
    ch._local_channel_sum = cpu_channel_sum
    ch._dev = _CpuDev
    ch.WelchConfig = _CpuWelchConfig
    return ch


def _worker(rank, world, port, nch_total, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import dsp_jl_amd as d
    ch = _install_cpu_hooks()
    rng = np.random.default_rng(1776)                      # every rank draws the same full data set, then takes its shard
    S = rng.standard_normal((nch_total, 6000))
    mine = d.channel_shard(nch_total, rank, world)
    cols = torch.from_numpy(S[mine.start:mine.stop].copy())
    cfg = _CpuWelchConfig(6000, np.float64, 256, 128, window=None)
    mean = ch.welch_channel_mean(cols, cfg, nch_total=nch_total)       # PRODUCT function: provider -> local sum -> all_reduce -> 1/nch
    out_q.put((rank, list(mine), mean.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nch_total", [1, 5, 8])   # 1: rank 1 owns no channel and contributes zeros
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
    S = rng.standard_normal((nch_total, 6000))
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
    ch = _install_cpu_hooks()
    rng = np.random.default_rng(1776)
    x = rng.standard_normal(50000)
    n, nov, nb = 512, 384, 97
    b = rng.standard_normal(nb)
    K = opg.frame_count(len(x), n, nov)
    frames = d.frame_shard(K, rank, world)
    lo, hi = d.frame_span(frames, n, nov)
    psd = ch.welch_time_split(x[lo:hi], K, n, nov, window=ow.hanning).numpy()     # PRODUCT function: accumulate -> all_reduce -> finalize(K)
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
