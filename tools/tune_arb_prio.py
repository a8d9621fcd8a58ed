#!/usr/bin/env python3
"""FIRArbitrary: the prologue of a workgroup at normal / raised wave priority (MDSP_ARB_PRIO), warm calls, interleaved; 4 channels x 2^26 / 2^28."""
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
res = {}
for log2n, rate in ((28, 160 / 147), (26, 147 / 160), (26, 0.3721)):
    n, nch = 1 << log2n, 4
    ha = d.resample_filter(rate, 32).astype(np.float32)
    x = torch.randn((nch, n), dtype=torch.float32, device="cuda")
    fa = C.c_void_p()
    _lib.check(lib.mdsp_firarb_create(C.byref(fa), ha.ctypes.data_as(C.c_void_p), len(ha), rate, 32, _lib.F32, _lib.F32, nch))
    ola = C.c_int64(); _lib.check(lib.mdsp_firarb_outputlength(fa, n, C.byref(ola)))
    ya = torch.empty((nch, ola.value + 1), dtype=torch.float32, device="cuda")
    nw = C.c_int64()

    def arb():
        _lib.check(lib.mdsp_firarb_reset(fa))
        _lib.check(lib.mdsp_firarb_exec(fa, x.data_ptr(), n, n, ya.data_ptr(), ola.value + 1, ola.value + 1, C.byref(nw), stream))

    def timeit():
        arb(); _lib.check(lib.mdsp_event_record(e0, stream)); arb(); _lib.check(lib.mdsp_event_record(e1, stream)); torch.cuda.synchronize()
        ms = C.c_float(); _lib.check(lib.mdsp_event_elapsed_ms(e0, e1, C.byref(ms))); return ms.value

    arb(); arb(); torch.cuda.synchronize()
    ref = ya.clone()
    t = {0: [], 1: []}
    for r in range(6):
        for v in (0, 1):
            _lib.set_tunable("MDSP_ARB_PRIO", v); t[v].append(round(timeit(), 4))
    _lib.set_tunable("MDSP_ARB_PRIO", 1); arb(); torch.cuda.synchronize()
    same = bool(torch.equal(ref, ya))
    _lib.set_tunable("MDSP_ARB_PRIO", None)
    key = f"rate={rate:.4f} 4x2^{log2n}"
    res[key] = {"normal_ms": float(np.median(t[0])), "raised_ms": float(np.median(t[1])), "bit_identical": same}
    print(key, res[key], flush=True)
    _lib.check(lib.mdsp_firarb_destroy(fa)); del x, ya
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "tune_arb_prio.json"), "w"), indent=1)
