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
xc = torch.randn(1 << 20, device="cuda", dtype=torch.complex64)
img = torch.randn((512, 512), device="cuda", dtype=torch.float32)
ker = torch.randn((9, 9), device="cuda", dtype=torch.float32)
df = d.DF2TFilter(b[:64])
ff = d.FIRFilter(h, Fraction(3, 2))
cfg = d.WelchConfig(1 << 20, np.float32, n=4096, noverlap=2048, window=d.hanning)
calls = {
    "filt(b[256], x[2^20])": lambda: d.filt(b, x),
    "filt(b[16], x[2^20]) (time domain)": lambda: d.filt(b[:16], x),
    "welch_pgram(x[2^20], 4096, 2048)": lambda: d.welch_pgram(x, 4096, 2048, window=d.hanning),
    "spectrogram(x[2^20], 1024, 768)": lambda: d.spectrogram(x, 1024, 768, window=d.hanning),
    "resample(x[2^20], 3//2, h)": lambda: d.resample(x, Fraction(3, 2), h),
    "resample(x[2^20], 1.2345)": lambda: d.resample(x, 1.2345),
    "conv(x[2^20], b[256])": lambda: d.conv(x, torch.from_numpy(b).cuda()),
    "hilbert(x[2^20])": lambda: d.hilbert(x),
    "periodogram(x[2^16])": lambda: d.periodogram(x[:1 << 16], window=d.hanning),
    "stft(xc[2^20], 1024, 768)": lambda: d.stft(xc, 1024, 768, window=d.hanning),
    "xcorr(x[2^16], b[256])": lambda: d.xcorr(x[:1 << 16], torch.from_numpy(b).cuda()),
    "filtfilt(b[256], x[2^20])": lambda: d.filtfilt(b, x),
    "mt_pgram(x[2^14])": lambda: d.mt_pgram(x[:1 << 14]),
    "mt_spectrogram(x[2^20], 1024, 512)": lambda: d.mt_spectrogram(x, 1024, 512),
    "conv(img[512x512], k[9x9])": lambda: d.conv(img, ker),
    "DF2TFilter(b[64]).filt(x[2^16])": lambda: df.filt(x[:1 << 16]),
    "FIRFilter(3//2).filt(x[2^20]) (held object)": lambda: ff.filt(x),
    "welch_pgram(x, config) (held object)": lambda: d.welch_pgram(x, cfg),
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
