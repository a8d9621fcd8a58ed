#!/usr/bin/env python3
"""welch_pgram with a LARGE nfft on a long stream (many transforms per call, unlike the default n = length >> 3): the multi-pass engine's forms.
    WL_LOG2LEN=27  WL_LOG2N=17,18,19,20,21  WL_DTYPE=f32|f64   (set MDSP_BIG_WELCH_ROWS=0/1 in the environment)"""
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

lib = _lib.lib()
_lib.check(lib.mdsp_init(0))
stream = torch.cuda.current_stream().cuda_stream
e0, e1 = C.c_void_p(), C.c_void_p()
_lib.check(lib.mdsp_event_create(C.byref(e0))); _lib.check(lib.mdsp_event_create(C.byref(e1)))


def timeit(fn, reps=7):
    fn(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        _lib.check(lib.mdsp_event_record(e0, stream)); fn(); _lib.check(lib.mdsp_event_record(e1, stream))
        torch.cuda.synchronize()
        ms = C.c_float(); _lib.check(lib.mdsp_event_elapsed_ms(e0, e1, C.byref(ms))); ts.append(ms.value)
    return sorted(ts)[len(ts) // 2]


dt = os.environ.get("WL_DTYPE", "f32")
npdt, tdt, esz = (np.float32, torch.float32, 4) if dt == "f32" else (np.float64, torch.float64, 8)
length = 1 << int(os.environ.get("WL_LOG2LEN", "27"))
x = torch.randn(length, device="cuda", dtype=tdt)
out = {}
for l2 in [int(v) for v in os.environ.get("WL_LOG2N", "17,18,19,20,21").split(",")]:
    n = 1 << l2
    cfg = d.WelchConfig(length, npdt, n=n, noverlap=n // 2, nfft=n, window=d.hanning)
    psd = torch.empty(cfg.nout, dtype=tdt, device="cuda")
    ms = timeit(lambda: _lib.check(lib.mdsp_welch_exec(cfg._h, x.data_ptr(), length, 1, length, psd.data_ptr(), cfg.nout, stream)))
    out[str(n)] = {"ms": round(ms, 4), "TBps": round(esz * length / ms / 1e9, 4), "engine": cfg.engine}
    print(dt, n, out[str(n)], flush=True)
    del cfg, psd
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", os.environ.get("WL_OUT", "welch_large.json")), "w"), indent=1)
