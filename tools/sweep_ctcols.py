#!/usr/bin/env python3
"""Welch (Float32, 50 % overlap, 2^26 samples) at every nfft = R0 x S the compile-time row kernels take (csrc/spectral_ctcols.hip: R0 = 2 .. 4, S from its
size table), timed with the library's own choice.  Run it twice -- as is, and with MDSP_GX=4 (the run-time-schedule kernel instead) -- and compare: the
sizes where the compile-time rows lose are what ctcols_split leaves out.  SWEEP_OUT names the json under gpurun_out/."""
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
n = 1 << 26
x = torch.randn(n, device="cuda", dtype=torch.float32)
S = [2000, 2400, 2500, 2560, 3000, 3072, 3200, 3840, 4000, 4096, 4800, 5000, 5120, 6000, 6144, 6400, 8000, 8192, 4200, 4500, 4608, 5400, 5600, 6250, 6750, 7000, 7200,
     7500, 7680, 8100]
sizes = sorted({r * s for r in (2, 3, 4) for s in S if r * s > 8192})
if os.environ.get("SWEEP_SIZES"):   # any list of nfft instead (e.g. the single-workgroup sizes of csrc/spectral_ctbig.hip)
    sizes = [int(v) for v in os.environ["SWEEP_SIZES"].split(",")]


def ev():
    e = C.c_void_p(); _lib.check(lib.mdsp_event_create(C.byref(e))); return e


e0, e1 = ev(), ev()
res = {}
for nfft in sizes:
    cfg = d.WelchConfig(n, np.float32, n=nfft, noverlap=nfft // 2, nfft=nfft, window=d.hanning)
    psd = torch.empty(cfg.nout, dtype=torch.float32, device="cuda")
    run = lambda: _lib.check(lib.mdsp_welch_exec(cfg._h, x.data_ptr(), n, 1, n, psd.data_ptr(), cfg.nout, stream))
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        _lib.check(lib.mdsp_event_record(e0, stream)); run(); _lib.check(lib.mdsp_event_record(e1, stream)); torch.cuda.synchronize()
        ms = C.c_float(); _lib.check(lib.mdsp_event_elapsed_ms(e0, e1, C.byref(ms))); ts.append(ms.value)
    res[nfft] = round(4.0 * n / sorted(ts)[2] / 1e9, 3)
    del cfg, psd
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", os.environ.get("SWEEP_OUT", "sweep_ctcols.json")), "w"))
print(json.dumps(res))
