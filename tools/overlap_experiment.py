#!/usr/bin/env python3
"""Can the two headline kernels share the GPU?  filt (HBM-bound) and welch_pgram (VALU-bound) of the same stream are independent:
launch them on two HIP streams, with the persistent grids sized so that both are co-resident (MDSP_WG_PER_CU), and compare the
time of the pair with the back-to-back time.  Prints one JSON object (gpurun_out/overlap.json)."""
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
from bench import lowpass_taps

lib = _lib.lib()
_lib.check(lib.mdsp_init(0))
n = 1 << int(os.environ.get("LOG2N", "30"))
x = torch.randn(n, device="cuda", dtype=torch.float32)
y = torch.empty_like(x)
plan = OlsPlan(np.asarray(lowpass_taps(256)), 2048, n, _lib.OLS_FILT, d.ENGINE_FUSED)
cfg = d.WelchConfig(n, np.float32, n=4096, noverlap=2048, window=d.hanning, engine=d.ENGINE_FUSED)
psd = torch.empty((1, cfg.nout), dtype=torch.float32, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def ols(st):
    _lib.check(lib.mdsp_ols_exec(plan._h, x.data_ptr(), n, 1, n, y.data_ptr(), n, n, st.cuda_stream))


def welch(st):
    _lib.check(lib.mdsp_welch_exec(cfg._h, x.data_ptr(), n, 1, n, psd.data_ptr(), cfg.nout, st.cuda_stream))


def timed(fn, reps=8):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(torch.cuda.current_stream())
        fn()
        e1.record(torch.cuda.current_stream())
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return round(float(np.median(ts)), 4)


def serial():
    cur = torch.cuda.current_stream()
    ols(cur); welch(cur)


def overlapped():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    ols(s1); welch(s2)
    cur.wait_stream(s1); cur.wait_stream(s2)


res = {"samples": n}
ref = None
for wg in ("default", "1", "2", "3"):
    if wg == "default":
        _lib.set_tunable("MDSP_WG_PER_CU", None)
    else:
        _lib.set_tunable("MDSP_WG_PER_CU", wg)
    res[f"wg_per_cu={wg}"] = {"back_to_back_ms": timed(serial), "two_streams_ms": timed(overlapped),
                              "filt_alone_ms": timed(lambda: ols(torch.cuda.current_stream())), "welch_alone_ms": timed(lambda: welch(torch.cuda.current_stream()))}
    print(wg, res[f"wg_per_cu={wg}"], flush=True)
    torch.cuda.synchronize()
    if ref is None:
        ref = (y.clone(), psd.clone())
    else:
        assert torch.equal(ref[0], y) and torch.allclose(ref[1], psd, rtol=1e-5)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "overlap.json"), "w"), indent=1)
