#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Metric (BASELINE.json): "Gsamples/s filt+welch, 1 Gsample Float32 stream".  One step = one pass of the hot path over
one 2^30-sample Float32 stream per GPU, already resident in HBM:
    y = fftfilt(b, x)          256-tap overlap-save FIR           (BASELINE config 2, Filters/filt.jl:479-521)
    P = welch_pgram(x)         nfft = 4096, hanning, 50 % overlap   (BASELINE config 3, periodograms.jl:746-759)
    mean over channels         one RCCL all-reduce of 2049 floats   (only for N > 1: one channel (stream) per GPU)
value = (2^30 samples x N) / step time, i.e. input samples that went through BOTH filt and Welch per second.

The JSON line also carries
    roofline      for the dominant kernel (the fused overlap-save kernel): algorithmic bytes (8 B/sample: 4 read + 4
                  written) / its mean launch duration, measured live with HIP events on the launch stream;
                  `traffic` = measured HBM bytes per launch from profiles/pmc_*.json (rocprofv3 --pmc passes) if present.
    kernels       the same for the Welch kernel (4 B/sample) and the on-box float4 copy yardstick.
    cpu_baseline  the CPU oracle (numpy/scipy restatement of DSP.jl's algorithm, 1 thread) timed on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured float4-copy rate


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2n", type=int, default=30, help="stream length per GPU = 2^log2n samples (BASELINE: 30)")
    ap.add_argument("--engine", choices=["auto", "fused", "rocfft"], default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-log2n", type=int, default=29, help="CPU baseline sample = 2^k samples")
    return ap.parse_args()


def lowpass_taps(n):
    import dsp_jl_amd as d
    return d.design.lowpass_firwindow(0.25, d.hamming(n), fs=1.0).astype("float32")


def cpu_baseline(log2n: int):
    """Oracle timed on the host (1 thread): same workload, bounded sample."""
    import numpy as np
    from oracle import filt as ofilt, periodograms as opg, windows as ow
    try:
        import threadpoolctl
        ctl = threadpoolctl.threadpool_limits(1)
    except Exception:
        ctl = None
    n = 1 << log2n
    seg = min(n, 1 << 24)              # the stream is walked in 2^24-sample segments to bound host memory (~1.5 GB)
    rng = np.random.default_rng(1776)
    b = np.asarray(lowpass_taps(256))
    nb, nfft = 256, 2048
    L = nfft - nb + 1
    import scipy.fft as sfft
    H = sfft.rfft(np.concatenate([b / np.float32(nfft), np.zeros(nfft - nb, np.float32)]))
    t_filt = t_welch = 0.0
    for s0 in range(0, n, seg):
        x = rng.standard_normal(seg, dtype=np.float32)
        # vectorised form of the oracle's block loop (same algorithm: rfft -> *H -> irfft per 2048-point block), batched
        t0 = time.perf_counter()
        xp = np.concatenate([np.zeros(nb - 1, np.float32), x, np.zeros(nfft, np.float32)])
        nblk = -(-seg // L)
        y = np.empty(nblk * L, np.float32)
        CH = 2048
        for k0 in range(0, nblk, CH):
            k1 = min(nblk, k0 + CH)
            idx = (np.arange(k0, k1) * L)[:, None] + np.arange(nfft)[None, :]
            blk = sfft.irfft(sfft.rfft(xp[idx], axis=1, workers=1) * H, nfft, axis=1, workers=1) * np.float32(nfft)
            y[k0 * L:k1 * L] = blk[:, nb - 1:].reshape(-1)
        t_filt += time.perf_counter() - t0
        if s0 == 0:  # spot-check the batched form against the line-faithful oracle on a prefix
            ref = ofilt._fftfilt(b, x[:20000], nfft)
            assert np.allclose(y[:20000], ref, rtol=1e-4, atol=1e-5)
        t0 = time.perf_counter()
        opg.welch_pgram(x, 4096, 2048, window=ow.hanning)
        t_welch += time.perf_counter() - t0
        del xp, y
    if ctl is not None:
        ctl.restore_original_limits() if hasattr(ctl, "restore_original_limits") else None
    return {"value": round(n / (t_filt + t_welch) / 1e9, 5), "unit": "Gsamples/s", "cores": 1, "kind": "port",
            "sample": f"2^{log2n} Float32 samples in 2^24-sample segments: overlap-save filt {t_filt:.2f}s + welch {t_welch:.2f}s, numpy/scipy(pocketfft) "
                      f"restatement of DSP.jl (not DSP.jl/FFTW), host has {os.cpu_count()} cores"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    import numpy as np
    import dsp_jl_amd as d
    from dsp_jl_amd import _lib, _dev
    from dsp_jl_amd.dspbase import OlsPlan
    lib = _lib.lib()
    _lib.check(lib.mdsp_init(local if world > 1 else 0))
    eng = {"auto": d.ENGINE_AUTO, "fused": d.ENGINE_FUSED, "rocfft": d.ENGINE_ROCFFT}[args.engine]

    n = 1 << args.log2n
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device=dev)
    g.manual_seed(1776 + rank)                          # seed 1776 = test/runtests.jl:20; one independent stream per GPU
    x = torch.randn(n, generator=g, device=dev, dtype=torch.float32)
    t = torch.arange(n, device=dev, dtype=torch.float32)
    x += 0.5 * torch.sin((2 * np.pi * 0.1234) * t)      # BASELINE.md config 3 line (phase accuracy is irrelevant here)
    del t
    xc = x.view(1, n)                                   # one column = one channel, contiguous
    y = torch.empty_like(xc)
    taps = lowpass_taps(256)
    plan = OlsPlan(np.asarray(taps), 2048, n, _lib.OLS_FILT, eng)
    cfg = d.WelchConfig(n, np.float32, n=4096, noverlap=2048, window=d.hanning, engine=eng)
    psd = torch.empty((1, cfg.nout), dtype=torch.float32, device=dev)
    mean = torch.empty(cfg.nout, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def ev():
        import ctypes as C
        e = C.c_void_p()
        _lib.check(lib.mdsp_event_create(C.byref(e)))
        return e

    import ctypes as C

    def step(evt):
        a, b_, c = evt
        _lib.check(lib.mdsp_event_record(a, stream))
        _lib.check(lib.mdsp_ols_exec(plan._h, x.data_ptr(), n, 1, n, y.data_ptr(), n, n, stream))
        _lib.check(lib.mdsp_event_record(b_, stream))
        _lib.check(lib.mdsp_welch_exec(cfg._h, x.data_ptr(), n, 1, n, psd.data_ptr(), cfg.nout, stream))
        _lib.check(lib.mdsp_event_record(c, stream))
        _lib.check(lib.mdsp_channel_sum(psd.data_ptr(), cfg.nout, 1, cfg.nout, _lib.F32, mean.data_ptr(), stream))
        if world > 1:
            dist.all_reduce(mean, op=dist.ReduceOp.SUM)          # RCCL over xGMI: 2049 floats
        mean.mul_(1.0 / world)

    evs = [(ev(), ev(), ev()) for _ in range(args.steps)]        # HIP events on the launch stream, created up front
    scratch = (ev(), ev(), ev())
    for _ in range(args.warmup):
        step(scratch)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):                                  # the timed region: exactly K steps
        step(evs[k])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t_ols = t_welch = 0.0
    for a, b_, c in evs:
        ms = C.c_float()
        _lib.check(lib.mdsp_event_elapsed_ms(a, b_, C.byref(ms))); t_ols += ms.value
        _lib.check(lib.mdsp_event_elapsed_ms(b_, c, C.byref(ms))); t_welch += ms.value
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        ols_ms = t_ols / args.steps
        welch_ms = t_welch / args.steps
        # on-box copy yardstick (float4 copy kernel, 2 x 4 GiB moved)
        c0, c1 = ev(), ev()
        _lib.check(lib.mdsp_copy_bench(y.data_ptr(), x.data_ptr(), n * 4, stream))
        _lib.check(lib.mdsp_event_record(c0, stream))
        for _ in range(3):
            _lib.check(lib.mdsp_copy_bench(y.data_ptr(), x.data_ptr(), n * 4, stream))
        _lib.check(lib.mdsp_event_record(c1, stream))
        ms = C.c_float()
        _lib.check(lib.mdsp_event_elapsed_ms(c0, c1, C.byref(ms)))
        copy_gbs = 3 * 2 * n * 4 / (ms.value * 1e-3) / 1e9
        # read-only yardstick (float4 loads summed, nothing written): what a 4 B/sample reader like the Welch kernel could reach
        _lib.check(lib.mdsp_copy_bench_mode(y.data_ptr(), x.data_ptr(), n * 4, 4, 8, stream))
        _lib.check(lib.mdsp_event_record(c0, stream))
        for _ in range(3):
            _lib.check(lib.mdsp_copy_bench_mode(y.data_ptr(), x.data_ptr(), n * 4, 4, 8, stream))
        _lib.check(lib.mdsp_event_record(c1, stream))
        _lib.check(lib.mdsp_event_elapsed_ms(c0, c1, C.byref(ms)))
        read_gbs = 3 * n * 4 / (ms.value * 1e-3) / 1e9
        ols_gbs = 8.0 * n / (ols_ms * 1e-3) / 1e9
        welch_gbs = 4.0 * n / (welch_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        kern_traffic = {}
        if os.path.exists(pmc):
            try:
                kern_traffic = json.load(open(pmc))
                traffic = kern_traffic.get("ols_fused_bytes_per_launch")
            except Exception:
                pass
        dominant_is_ols = ols_ms >= welch_ms
        roof = {"bound": "hbm", "kernel": "ols_fused_kernel (overlap-save filt, 8 B/sample)", "achieved": round(ols_gbs, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ols_gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
                "ms_per_launch": round(ols_ms, 4)}
        welch_roof = {"bound": "hbm", "kernel": "welch_fused_kernel (+finalize; 4 B/sample)", "achieved": round(welch_gbs, 1),
                      "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(welch_gbs / HBM_PEAK_GBS, 4),
                      "traffic": kern_traffic.get("welch_fused_bytes_per_launch"), "ms_per_launch": round(welch_ms, 4)}
        # the Welch kernel's other roof: 5 N log2 N flop per 4096-point transform of 4096 new samples, against the packed-FP32
        # add/multiply rate (butterflies are adds, not FMAs): 256 CUs x 4 SIMDs x 16 lanes x 2 x 2.4 GHz (boost; the kernel is
        # power-throttled to ~1.8 GHz)
        welch_tflops = 5.0 * 4096 * 12 * (n / 4096) / (welch_ms * 1e-3) / 1e12
        welch_roof["valu"] = {"achieved": round(welch_tflops, 1), "peak": 78.6, "unit": "TFLOP/s (packed f32 add/mul)", "frac": round(welch_tflops / 78.6, 4)}
        if not dominant_is_ols:
            roof, welch_roof = welch_roof, roof
        out = {
            "metric": "Gsamples/s filt+welch, 1 Gsample Float32 stream; achieved HBM GB/s vs roofline",
            "value": round(n * world / dt * args.steps / 1e9, 3), "unit": "Gsamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "filt(256-tap overlap-save, nfft=2048) + welch_pgram(nfft=4096, hanning, 50% overlap) per "
                                   f"2^{args.log2n}-sample Float32 stream; one stream (channel) per GPU; RCCL all-reduce of the 2049-bin PSD for N>1",
                       "samples_per_gpu": n, "engine": {1: "fused", 2: "rocfft"}[plan.engine], "stages_ms": {"filt": round(ols_ms, 4), "welch": round(welch_ms, 4)},
                       "stage_Gsamples_per_s": {"filt": round(n / ols_ms / 1e6, 2), "welch": round(n / welch_ms / 1e6, 2)}},
            "roofline": roof,
            "kernels": {"other": welch_roof, "copy_float4_GBps": round(copy_gbs, 1), "read_float4_GBps": round(read_gbs, 1)},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_log2n)
            except Exception as e:  # pragma: no cover
                out["cpu_baseline"] = {"value": None, "unit": "Gsamples/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()                                   # rank 0 may still be printing / measuring its yardsticks: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
