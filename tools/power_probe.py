#!/usr/bin/env python3
"""Is a kernel bounded by the power budget?  Runs each headline kernel back to back for a few seconds while a thread samples `rocm-smi` (socket power,
shader clock, the power cap), and reports average power / clock / time per launch: the Welch kernel on random data, on an all-zero stream (no
toggling: the DVFS give-back of MI355X_MICROARCH.md), the overlap-save kernel, and the float4 copy.  Writes gpurun_out/power_probe.json.

    python tools/power_probe.py [LOG2N=30] [SECONDS=3] [WELCH_VARIANTS=0,30,35]
"""
import ctypes as C
import json
import os
import re
import subprocess
import sys
import threading
import time

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
secs = float(os.environ.get("SECONDS", "3"))
wvars = [int(v) for v in os.environ.get("WELCH_VARIANTS", "0,30,35").split(",")]
g = torch.Generator(device="cuda"); g.manual_seed(1776)
x = torch.randn(n, generator=g, device="cuda", dtype=torch.float32)
xz = torch.zeros_like(x)
y = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
plan = OlsPlan(np.asarray(lowpass_taps(256)), 2048, n, _lib.OLS_FILT, d.ENGINE_FUSED)
cfgs = {}
for v in wvars:
    _lib.set_tunable("MDSP_WELCH_VARIANT", str(v))
    cfgs[v] = d.WelchConfig(n, np.float32, n=4096, noverlap=2048, window=d.hanning, engine=d.ENGINE_FUSED)
_lib.set_tunable("MDSP_WELCH_VARIANT", None)
psd = torch.empty(2049, dtype=torch.float32, device="cuda")


def smi():
    """(power W, sclk MHz, cap W) from one rocm-smi call; None where a field is missing."""
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True, timeout=5).stdout
    except Exception:
        return None, None, None
    p = re.search(r"Current Socket Graphics Package Power \(W\):\s*([0-9.]+)", out) or re.search(r"Average Graphics Package Power \(W\):\s*([0-9.]+)", out)
    c = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", out)
    m = re.search(r"Max Graphics Package Power \(W\):\s*([0-9.]+)", out)
    return (float(p.group(1)) if p else None, float(c.group(1)) if c else None, float(m.group(1)) if m else None)


def probe(name, fn):
    fn(); torch.cuda.synchronize()
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append(smi())
            time.sleep(0.05)

    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(50):
            fn()
        torch.cuda.synchronize(); k += 50
    dt = time.perf_counter() - t0
    stop.set(); th.join()
    pw = [s[0] for s in samples if s[0] is not None]; ck = [s[1] for s in samples if s[1] is not None]; cap = [s[2] for s in samples if s[2] is not None]
    r = {"ms_per_launch": round(dt / k * 1e3, 4), "launches": k, "smi_samples": len(samples),
         "power_W_mean": round(sum(pw) / len(pw), 1) if pw else None, "power_W_max": max(pw) if pw else None,
         "sclk_MHz_mean": round(sum(ck) / len(ck), 0) if ck else None, "power_cap_W": cap[0] if cap else None}
    print(name, r, flush=True)
    return r


res = {"samples": n, "seconds_per_case": secs, "idle": dict(zip(("power_W", "sclk_MHz", "cap_W"), smi()))}
cases = os.environ.get("CASES", "welch,welch_zeros,ols,ols_zeros,copy").split(",")
if "welch" in cases:
    for v in wvars:
        res[f"welch_v{v}_random"] = probe(f"welch v{v} random", lambda v=v: _lib.check(lib.mdsp_welch_exec(cfgs[v]._h, x.data_ptr(), n, 1, n, psd.data_ptr(), 2049, st)))
if "welch_zeros" in cases:
    for v in wvars:     # (round 4: every variant on the all-zero stream, not only the first)
        res[f"welch_v{v}_zeros"] = probe(f"welch v{v} zeros", lambda v=v: _lib.check(lib.mdsp_welch_exec(cfgs[v]._h, xz.data_ptr(), n, 1, n, psd.data_ptr(), 2049, st)))
if "ols" in cases:
    res["ols_random"] = probe("ols random", lambda: _lib.check(lib.mdsp_ols_exec(plan._h, x.data_ptr(), n, 1, n, y.data_ptr(), n, n, st)))
if "ols_zeros" in cases:
    res["ols_zeros"] = probe("ols zeros", lambda: _lib.check(lib.mdsp_ols_exec(plan._h, xz.data_ptr(), n, 1, n, y.data_ptr(), n, n, st)))
if "copy" in cases:
    res["copy_float4"] = probe("copy", lambda: _lib.check(lib.mdsp_copy_bench(y.data_ptr(), x.data_ptr(), n * 4, st)))
    res["copy_nt_4wg"] = probe("copy nt 4wg", lambda: _lib.check(lib.mdsp_copy_bench_mode(y.data_ptr(), x.data_ptr(), n * 4, 2, 4, st)))
res["ablate"] = os.environ.get("MDSP_ABLATE", "")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", os.environ.get("OUT", "power_probe.json")), "w"), indent=1)
