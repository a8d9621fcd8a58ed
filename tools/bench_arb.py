#!/usr/bin/env python3
"""FIRArbitrary (arbitrary-rate resampler) kernel sweep: channels per group (MDSP_ARB_NCH) x channel count x rate, warm calls
(cached host anchors).  Prints ms, outputs/s and the algorithmic GB/s (4 B in + 4 B out per output sample at Float32)."""
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
n = 1 << int(os.environ.get("ARB_LOG2N", "26"))
out = {}
rates = [float(eval(r)) for r in os.environ.get("ARB_RATES", "160/147,147/160,0.3721").split(",")]
chans = [int(c) for c in os.environ.get("ARB_CHANNELS", "1,2,4,8").split(",")]
for rate in rates:
    ha = d.resample_filter(rate, 32).astype(np.float32)
    for nch in chans:
        x = torch.randn((nch, n), dtype=torch.float32, device="cuda")
        fa = C.c_void_p()
        _lib.check(lib.mdsp_firarb_create(C.byref(fa), ha.ctypes.data_as(C.c_void_p), len(ha), rate, 32, _lib.F32, _lib.F32, nch))
        ola = C.c_int64(); _lib.check(lib.mdsp_firarb_outputlength(fa, n, C.byref(ola)))
        ya = torch.empty((nch, ola.value + 1), dtype=torch.float32, device="cuda")
        nw = C.c_int64()
        ref = None
        # cold call: the trajectory has to be evaluated (device scan vs the serial host loop)
        import time
        cold = {}
        for scan in ("1", "0"):
            _lib.set_tunable("MDSP_ARB_SCAN", scan)
            fc = C.c_void_p()
            _lib.check(lib.mdsp_firarb_create(C.byref(fc), ha.ctypes.data_as(C.c_void_p), len(ha), rate, 32, _lib.F32, _lib.F32, nch))
            torch.cuda.synchronize(); t0 = time.perf_counter()
            _lib.check(lib.mdsp_firarb_exec(fc, x.data_ptr(), n, n, ya.data_ptr(), ola.value + 1, ola.value + 1, C.byref(nw), stream))
            torch.cuda.synchronize(); cold[scan] = time.perf_counter() - t0
            _lib.check(lib.mdsp_firarb_destroy(fc))
        _lib.set_tunable("MDSP_ARB_SCAN", "1")
        print(f"rate={rate:.4f} nch={nch} cold call: scan {cold['1']*1e3:.2f} ms, serial {cold['0']*1e3:.2f} ms  ({nw.value} outputs/channel)", flush=True)
        out[f"rate={rate:.4f} nch={nch} cold_ms"] = {"scan": round(cold["1"] * 1e3, 3), "serial": round(cold["0"] * 1e3, 3), "outputs": nw.value}
        for g in (1, 2, 4):
            if g > max(nch, 1) and g > 1:
                continue
            _lib.set_tunable("MDSP_ARB_NCH", str(g))

            def arb():
                _lib.check(lib.mdsp_firarb_reset(fa))
                _lib.check(lib.mdsp_firarb_exec(fa, x.data_ptr(), n, n, ya.data_ptr(), ola.value + 1, ola.value + 1, C.byref(nw), stream))
            arb(); torch.cuda.synchronize()
            if ref is None:
                ref = ya.clone()
            same = bool(torch.equal(ref, ya))
            ts = []
            for _ in range(4):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); arb(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
            ms = min(ts)
            key = f"rate={rate:.4f} taps={len(ha)} nch={nch} group={g}"
            out[key] = {"ms": round(ms, 3), "Gout_per_s": round(nw.value * nch / ms / 1e6, 2), "GBps": round((4 * n + 4 * nw.value) * nch / ms / 1e6, 1), "bit_identical_to_group1": same}
            print(key, out[key], flush=True)
        _lib.check(lib.mdsp_firarb_destroy(fa))
        del x, ya
        torch.cuda.empty_cache()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "arb_sweep.json"), "w"), indent=1)
