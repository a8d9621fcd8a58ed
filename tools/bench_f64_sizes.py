#!/usr/bin/env python3
"""Round 4: Float64 / ComplexF64 at the nextfastfft sizes beyond 3000 points (compile-time schedules on one LDS buffer) against the rocFFT engine they took
before.  Welch 50 % (8 B/sample), ComplexF64 STFT 75 % (16 + 16 * 4 = 80 B/sample); 2^26 samples.  Writes gpurun_out/f64_sizes.json."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import dsp_jl_amd as d
from dsp_jl_amd import _lib
from dsp_jl_amd.periodograms import _StftPlan, compute_window

lib = _lib.lib()
_lib.check(lib.mdsp_init(0))
stream = torch.cuda.current_stream().cuda_stream
n = 1 << int(os.environ.get("LOG2N", "26"))


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


xr = torch.randn(n, device="cuda", dtype=torch.float64)
xc = torch.complex(xr[: n // 2].clone(), torch.randn(n // 2, device="cuda", dtype=torch.float64))
res = {}
for nfft in [int(v) for v in os.environ.get("SIZES", "3000,3072,3840,4000,4096,5000,6000,6400,8000").split(",")]:
    row = {}
    for eng, name in ((d.ENGINE_FUSED, "fused"), (d.ENGINE_ROCFFT, "rocfft")):
        try:
            cfg = d.WelchConfig(n, np.float64, n=nfft, noverlap=nfft // 2, nfft=nfft, window=d.hanning, engine=eng)
            psd = torch.empty(cfg.nout, dtype=torch.float64, device="cuda")
            ms = timeit(lambda: _lib.check(lib.mdsp_welch_exec(cfg._h, xr.data_ptr(), n, 1, n, psd.data_ptr(), cfg.nout, stream)))
            row[f"welch50_{name}_GBps"] = round(8.0 * n / ms / 1e6, 1)
            m = n // 2
            win, norm2 = compute_window(d.hanning, nfft)
            hop = nfft // 4
            K = d.frame_count(m, nfft, nfft - hop)
            plan = _StftPlan(nfft, nfft - hop, nfft, win, norm2, False, 0, np.complex128, eng)
            out = torch.empty((K, nfft), dtype=torch.complex128, device="cuda")
            ms = timeit(lambda: _lib.check(lib.mdsp_stft_exec(plan._h, xc.data_ptr(), m, 1, m, out.data_ptr(), nfft, K * nfft, stream)))
            row[f"stft75_c128_{name}_GBps"] = round(80.0 * m / ms / 1e6, 1)
            del out, plan, psd, cfg
        except Exception as e:
            row[name] = str(e)[:80]
    res[str(nfft)] = row
    print(nfft, row, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "f64_sizes.json"), "w"), indent=1)
