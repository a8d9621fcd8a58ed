#!/usr/bin/env python3
"""Wall-clock latency of whole API calls (plan creation + launch + synchronise) on device-resident inputs of modest size: what a
caller who does not hold on to plan objects pays per call."""
import json
import os
import sys
import time
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import dsp_jl_amd as d

x = torch.randn(1 << 20, device="cuda", dtype=torch.float32)
b = np.random.default_rng(0).standard_normal(256).astype(np.float32)
h = d.resample_filter(Fraction(3, 2)).astype(np.float32)
calls = {
    "filt(b[256], x[2^20])": lambda: d.filt(b, x),
    "filt(b[16], x[2^20]) (time domain)": lambda: d.filt(b[:16], x),
    "welch_pgram(x[2^20], 4096, 2048)": lambda: d.welch_pgram(x, 4096, 2048, window=d.hanning),
    "spectrogram(x[2^20], 1024, 768)": lambda: d.spectrogram(x, 1024, 768, window=d.hanning),
    "resample(x[2^20], 3//2, h)": lambda: d.resample(x, Fraction(3, 2), h),
    "resample(x[2^20], 1.2345)": lambda: d.resample(x, 1.2345),
    "conv(x[2^20], b[256])": lambda: d.conv(x, torch.from_numpy(b).cuda()),
    "hilbert(x[2^20])": lambda: d.hilbert(x),
}
res = {}
for name, fn in calls.items():
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    res[name] = {"median_us": round(1e6 * float(np.median(ts)), 1), "min_us": round(1e6 * min(ts), 1)}
    print(name, res[name], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "api_latency.json"), "w"), indent=1)
