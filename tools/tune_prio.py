#!/usr/bin/env python3
"""Wave priority around the memory instructions of the two headline kernels (MDSP_OLS_PRIO: 1 loads, 2 stores, 3 both; MDSP_SPEC_PRIO: 1 = Welch
loads), interleaved rounds in one process.  Writes gpurun_out/tune_prio.json"""
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
log2n = int(os.environ.get("TUNE_LOG2N", "30"))
n = 1 << log2n
stream = torch.cuda.current_stream().cuda_stream
x = torch.randn(n, device="cuda")
y = torch.empty_like(x)
psd = torch.empty(2049, dtype=torch.float32, device="cuda")
taps = (np.hanning(256) / 128).astype(np.float32)
p = OlsPlan(taps, 2048, n, 0, d.ENGINE_FUSED)
c = d.WelchConfig(n, np.float32, n=4096, noverlap=2048, window=d.hanning, engine=d.ENGINE_FUSED)
e0, e1 = C.c_void_p(), C.c_void_p()
_lib.check(lib.mdsp_event_create(C.byref(e0))); _lib.check(lib.mdsp_event_create(C.byref(e1)))


def timeit(fn):
    fn(); _lib.check(lib.mdsp_event_record(e0, stream)); fn(); fn(); fn(); _lib.check(lib.mdsp_event_record(e1, stream))
    torch.cuda.synchronize()
    ms = C.c_float(); _lib.check(lib.mdsp_event_elapsed_ms(e0, e1, C.byref(ms))); return ms.value / 3


ols = lambda: _lib.check(lib.mdsp_ols_exec(p._h, x.data_ptr(), n, 1, n, y.data_ptr(), n, n, stream))
wel = lambda: _lib.check(lib.mdsp_welch_exec(c._h, x.data_ptr(), n, 1, n, psd.data_ptr(), 2049, stream))
res = {"log2n": log2n, "ols": {}, "welch": {}}
ref = None
for v in (0, 1, 2, 3):
    _lib.set_tunable("MDSP_OLS_PRIO", v); ols(); torch.cuda.synchronize()
    ref = y.clone() if ref is None else ref
    res["ols"][v] = {"ms": [], "maxdiff": float((y - ref).abs().max())}
pref = None
for v in (0, 1):
    _lib.set_tunable("MDSP_SPEC_PRIO", v); wel(); torch.cuda.synchronize()
    pref = psd.clone() if pref is None else pref
    res["welch"][v] = {"ms": [], "relerr": float((psd - pref).norm() / pref.norm())}
for r in range(int(os.environ.get("TUNE_ROUNDS", "6"))):
    for v in (0, 1, 2, 3):
        _lib.set_tunable("MDSP_OLS_PRIO", v); res["ols"][v]["ms"].append(round(timeit(ols), 4))
    for v in (0, 1):
        _lib.set_tunable("MDSP_SPEC_PRIO", v); res["welch"][v]["ms"].append(round(timeit(wel), 4))
_lib.set_tunable("MDSP_OLS_PRIO", None); _lib.set_tunable("MDSP_SPEC_PRIO", None)
for k in ("ols", "welch"):
    for v, e in res[k].items():
        e["median_ms"] = float(np.median(e["ms"]))
        print(k, "prio", v, e["median_ms"], "ms", e)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "tune_prio.json"), "w"), indent=1)
