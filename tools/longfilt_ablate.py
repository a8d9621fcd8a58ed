#!/usr/bin/env python3
"""Where the partitioned overlap-save kernel's time goes: the MDSP_ABLATE bits (1 no input, 2 no transforms, 4 no stores, 8 no spectra loads) on
a -DMDSP_DEBUG_KNOBS build, interleaved rounds.  MDSP_LIB_TAG=dbg python tools/longfilt_ablate.py  ->  gpurun_out/longfilt_ablate.json"""
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
assert lib.mdsp_debug_knobs(), "needs the dbg build"
stream = torch.cuda.current_stream().cuda_stream
n = 1 << 28
x = torch.randn(n, device="cuda")
y = torch.empty_like(x)
e0, e1 = C.c_void_p(), C.c_void_p()
_lib.check(lib.mdsp_event_create(C.byref(e0))); _lib.check(lib.mdsp_event_create(C.byref(e1)))


def once(p):
    _lib.check(lib.mdsp_event_record(e0, stream)); _lib.check(lib.mdsp_ols_exec(p._h, x.data_ptr(), n, 1, n, y.data_ptr(), n, n, stream)); _lib.check(lib.mdsp_event_record(e1, stream))
    torch.cuda.synchronize()
    ms = C.c_float(); _lib.check(lib.mdsp_event_elapsed_ms(e0, e1, C.byref(ms))); return ms.value


res = {}
for nb in [int(v) for v in os.environ.get("LONGFILT_TAPS", "5120").split(",")]:
    taps = (np.random.default_rng(nb).standard_normal(nb) / np.sqrt(nb)).astype(np.float32)
    nfft = d.optimalfftfiltlength(nb, n)
    plans = {}
    for v in [int(v) for v in os.environ.get("LONGFILT_VARIANTS", "0,1").split(",")]:
        _lib.set_tunable("MDSP_OLS_VARIANT", v)
        plans[v] = OlsPlan(taps, nfft, n, _lib.OLS_FILT, d.ENGINE_FUSED)
    _lib.set_tunable("MDSP_OLS_VARIANT", None)
    abl = [int(v) for v in os.environ.get("LONGFILT_ABLATE", "0,1,8,9,2,4,5,13,15,6,10,11,14").split(",")]
    t = {(v, a): [] for v in plans for a in abl}
    for r in range(4):
        for a in abl:
            _lib.set_tunable("MDSP_ABLATE", a)
            for v, p in plans.items():
                once(p)
                t[(v, a)].append(min(once(p) for _ in range(3)))
    _lib.set_tunable("MDSP_ABLATE", None)
    for (v, a), ts in t.items():
        res[f"taps{nb}_v{v}_ablate{a}"] = {"min_ms": round(min(ts), 4), "median_ms": round(sorted(ts)[len(ts) // 2], 4)}
        print(nb, "variant", v, "ablate", a, res[f"taps{nb}_v{v}_ablate{a}"], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "longfilt_ablate.json"), "w"), indent=1)
