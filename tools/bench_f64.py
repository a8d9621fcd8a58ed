#!/usr/bin/env python3
"""Float64 throughput of the two headline kernels (DSP.jl's default eltype): 2^28-sample stream, same filters as bench.py."""
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

lib = _lib.lib()
_lib.check(lib.mdsp_init(0))
n = 1 << int(os.environ.get("F64_LOG2N", "28"))
stream = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda"); g.manual_seed(1776)


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
for dt, tdt, sz in ((np.float32, torch.float32, 4), (np.float64, torch.float64, 8)):
    x = torch.randn(n, generator=g, device="cuda", dtype=tdt)
    y = torch.empty_like(x)
    taps = d.design.lowpass_firwindow(0.25, d.hamming(256), fs=1.0).astype(dt)
    for eng, en in ((d.ENGINE_FUSED, "fused"), (d.ENGINE_ROCFFT, "rocfft")):
        p = OlsPlan(taps, 2048, n, 0, eng)
        ms = timeit(lambda: _lib.check(lib.mdsp_ols_exec(p._h, x.data_ptr(), n, 1, n, y.data_ptr(), n, n, stream)))
        res[f"ols_{np.dtype(dt).name}_{en}"] = {"ms": round(ms, 4), "GBps_algorithmic": round(2 * sz * n / ms / 1e6, 1), "Gsamples_per_s": round(n / ms / 1e6, 2)}
        cfg = d.WelchConfig(n, dt, n=4096, noverlap=2048, window=d.hanning, engine=eng)
        psd = torch.empty(2049, dtype=tdt, device="cuda")
        ms = timeit(lambda: _lib.check(lib.mdsp_welch_exec(cfg._h, x.data_ptr(), n, 1, n, psd.data_ptr(), 2049, stream)))
        res[f"welch_{np.dtype(dt).name}_{en}"] = {"ms": round(ms, 4), "GBps_algorithmic": round(sz * n / ms / 1e6, 1), "Gsamples_per_s": round(n / ms / 1e6, 2)}
    del x, y
for k, v in res.items():
    print(k, v)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "f64.json"), "w"), indent=1)
