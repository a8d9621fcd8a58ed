#!/usr/bin/env python3
"""The FIRArbitrary row of bench.py (4 channels x 2^28 Float32, rate 160/147 as Float64, 1185 taps, warm trajectory) on its own: median / min / max
of REPS launches, for A/B runs of library builds (MDSP_LIB_TAG) and knobs (MDSP_ARB_TILE, MDSP_ARB_NCH, MDSP_ARB_PRIO) in separate processes.
    ARB_LOG2N=28 ARB_RATE=160/147 ARB_NCH=4 ARB_DTYPE=f32 REPS=9 python tools/bench_firarb.py"""
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
n = 1 << int(os.environ.get("ARB_LOG2N", "28"))
nch = int(os.environ.get("ARB_NCH", "4"))
rate = float(eval(os.environ.get("ARB_RATE", "160/147")))
reps = int(os.environ.get("REPS", "9"))
f64 = os.environ.get("ARB_DTYPE", "f32") == "f64"
tdt, ndt, code = (torch.float64, np.float64, _lib.F64) if f64 else (torch.float32, np.float32, _lib.F32)
ha = d.resample_filter(rate, 32).astype(ndt)
x = torch.randn((nch, n), dtype=tdt, device="cuda")
fa = C.c_void_p()
_lib.check(lib.mdsp_firarb_create(C.byref(fa), ha.ctypes.data_as(C.c_void_p), len(ha), rate, 32, code, code, nch))
ola = C.c_int64(); _lib.check(lib.mdsp_firarb_outputlength(fa, n, C.byref(ola)))
ya = torch.empty((nch, ola.value + 1), dtype=tdt, device="cuda")
nw = C.c_int64()


def arb():
    _lib.check(lib.mdsp_firarb_reset(fa))
    _lib.check(lib.mdsp_firarb_exec(fa, x.data_ptr(), n, n, ya.data_ptr(), ola.value + 1, ya.shape[1], C.byref(nw), stream))


for _ in range(3):
    arb()
torch.cuda.synchronize()
ts = []
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); arb(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
med = ts[len(ts) // 2]
b = (1 + rate) * x.element_size() * n * nch
print(json.dumps({"tag": os.environ.get("MDSP_LIB_TAG", ""), "knobs": {k: v for k, v in os.environ.items() if k.startswith("MDSP_ARB")}, "rate": rate, "nch": nch, "log2n": int(np.log2(n)),
                  "dtype": "f64" if f64 else "f32", "taps": len(ha), "ms": round(med, 4), "min_ms": round(ts[0], 4), "max_ms": round(ts[-1], 4), "GBps": round(b / med / 1e6, 1),
                  "frac": round(b / med / 1e6 / 8000, 4), "checksum": float(ya[:, : nw.value].double().abs().sum())}))
