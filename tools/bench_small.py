#!/usr/bin/env python3
"""Throughput of the spectral kernels at smaller transform sizes (real Float32 input, 2^28 samples): Welch at 50 % overlap and
75 % overlap, spectrogram at 75 % overlap, overlap-save filt with short filters.  Gsamples/s of input."""
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
from dsp_jl_amd.dspbase import OlsPlan
from dsp_jl_amd.periodograms import _StftPlan, compute_window

lib = _lib.lib()
_lib.check(lib.mdsp_init(0))
n = 1 << int(os.environ.get("SMALL_LOG2N", "28"))
stream = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda"); g.manual_seed(1776)
x = torch.randn(n, generator=g, device="cuda", dtype=torch.float32)


def ev():
    e = C.c_void_p(); _lib.check(lib.mdsp_event_create(C.byref(e))); return e


e0, e1 = ev(), ev()


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        _lib.check(lib.mdsp_event_record(e0, stream)); fn(); _lib.check(lib.mdsp_event_record(e1, stream))
        ms = C.c_float(); _lib.check(lib.mdsp_event_elapsed_ms(e0, e1, C.byref(ms))); ts.append(ms.value)
    return min(ts)


res = {}
for nfft in (256, 512, 1024, 2048, 4096, 8192):
    for name, nov in (("welch50", nfft // 2), ("welch75", 3 * nfft // 4)):
        cfg = d.WelchConfig(n, np.float32, n=nfft, noverlap=nov, window=d.hanning)
        psd = torch.empty(cfg.nout, dtype=torch.float32, device="cuda")
        ms = timeit(lambda: _lib.check(lib.mdsp_welch_exec(cfg._h, x.data_ptr(), n, 1, n, psd.data_ptr(), cfg.nout, stream)))
        res[f"{name}_{nfft}"] = round(n / ms / 1e6, 1)
    m = n >> 2                                                   # the spectrogram output is 2x the input at 75 % overlap
    win, norm2 = compute_window(d.hanning, nfft)
    K = d.frame_count(m, nfft, 3 * nfft // 4)
    plan = _StftPlan(nfft, 3 * nfft // 4, nfft, win, norm2, True, True, np.float32, d.ENGINE_AUTO)
    out = torch.empty((K, plan.nout), dtype=torch.float32, device="cuda")
    ms = timeit(lambda: _lib.check(lib.mdsp_stft_exec(plan._h, x.data_ptr(), m, 1, m, out.data_ptr(), plan.nout, K * plan.nout, stream)))
    res[f"spectrogram75_{nfft}"] = round(m / ms / 1e6, 1)
    del out
    nb = nfft // 8
    taps = d.design.lowpass_firwindow(0.25, d.hamming(nb), fs=1.0).astype(np.float32)
    p = OlsPlan(taps, nfft, n, 0, d.ENGINE_AUTO)
    y = torch.empty_like(x)
    ms = timeit(lambda: _lib.check(lib.mdsp_ols_exec(p._h, x.data_ptr(), n, 1, n, y.data_ptr(), n, n, stream)))
    res[f"filt_nb{nb}_{nfft}"] = round(n / ms / 1e6, 1)
    del y
for k, v in res.items():
    print(k, v, "Gsamples/s")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "small.json"), "w"), indent=1)
