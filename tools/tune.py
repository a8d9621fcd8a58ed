#!/usr/bin/env python3
"""Sweep the tuning variants of the fused kernels on the GPU box (within-process, interleaved rounds) and write
gpurun_out/tune.json.  Variants are selected at plan creation through MDSP_OLS_VARIANT / MDSP_WELCH_VARIANT."""
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
log2n = int(os.environ.get("TUNE_LOG2N", "28"))
rounds = int(os.environ.get("TUNE_ROUNDS", "5"))
n = 1 << log2n
g = torch.Generator(device="cuda"); g.manual_seed(1776)
x = torch.randn(n, generator=g, device="cuda", dtype=torch.float32)
y = torch.empty_like(x)
stream = torch.cuda.current_stream().cuda_stream
taps = d.design.lowpass_firwindow(0.25, d.hamming(256), fs=1.0).astype(np.float32)


def ev():
    e = C.c_void_p(); _lib.check(lib.mdsp_event_create(C.byref(e))); return e


e0, e1 = ev(), ev()


def timeit(fn):
    fn(); torch.cuda.synchronize()
    _lib.check(lib.mdsp_event_record(e0, stream)); fn(); _lib.check(lib.mdsp_event_record(e1, stream))
    ms = C.c_float(); _lib.check(lib.mdsp_event_elapsed_ms(e0, e1, C.byref(ms)))
    return ms.value


res = {"log2n": log2n, "ols": {}, "welch": {}, "stft": {}}
# ---- overlap-save ----
ols_variants = [int(v) for v in os.environ.get("TUNE_OLS", "0,1,2,3,4,5,6,7,8,9,10,11,12").split(",") if v != ""]
plans = {}
for v in ols_variants:
    _lib.set_tunable("MDSP_OLS_VARIANT", str(v))
    plans[("fused", v)] = OlsPlan(taps, 2048, n, 0, d.ENGINE_FUSED)
_lib.set_tunable("MDSP_OLS_VARIANT", "0")
plans[("rocfft", 0)] = OlsPlan(taps, 2048, n, 0, d.ENGINE_ROCFFT)
ref = None
for key, p in plans.items():
    _lib.check(lib.mdsp_ols_exec(p._h, x.data_ptr(), n, 1, n, y.data_ptr(), n, n, stream))
    torch.cuda.synchronize()
    if ref is None:
        ref = y.clone()
    err = float((y - ref).abs().max())
    res["ols"][f"{key[0]}:{key[1]}"] = {"maxdiff_vs_first": err, "ms": []}
for r in range(rounds):
    for key, p in plans.items():
        ms = timeit(lambda: _lib.check(lib.mdsp_ols_exec(p._h, x.data_ptr(), n, 1, n, y.data_ptr(), n, n, stream)))
        res["ols"][f"{key[0]}:{key[1]}"]["ms"].append(round(ms, 4))
for k, v in res["ols"].items():
    v["best_GBps"] = round(8.0 * n / (min(v["ms"]) * 1e-3) / 1e9, 1)
    v["median_ms"] = sorted(v["ms"])[len(v["ms"]) // 2]
del plans, ref
# ---- Welch ----
welch_variants = [int(v) for v in os.environ.get("TUNE_WELCH", "0,1,2,3,4,5,6,7,8,9").split(",") if v != ""]
cfgs = {}
for v in welch_variants:
    _lib.set_tunable("MDSP_WELCH_VARIANT", str(v))
    cfgs[("fused", v)] = d.WelchConfig(n, np.float32, n=4096, noverlap=2048, window=d.hanning, engine=d.ENGINE_FUSED)
_lib.set_tunable("MDSP_WELCH_VARIANT", "0")
cfgs[("rocfft", 0)] = d.WelchConfig(n, np.float32, n=4096, noverlap=2048, window=d.hanning, engine=d.ENGINE_ROCFFT)
psd = torch.empty(2049, dtype=torch.float32, device="cuda")
ref = None
for key, c in cfgs.items():
    _lib.check(lib.mdsp_welch_exec(c._h, x.data_ptr(), n, 1, n, psd.data_ptr(), 2049, stream)); torch.cuda.synchronize()
    if ref is None:
        ref = psd.clone()
    res["welch"][f"{key[0]}:{key[1]}"] = {"relerr_vs_first": float((psd - ref).norm() / ref.norm()), "ms": []}
for r in range(rounds):
    for key, c in cfgs.items():
        ms = timeit(lambda: _lib.check(lib.mdsp_welch_exec(c._h, x.data_ptr(), n, 1, n, psd.data_ptr(), 2049, stream)))
        res["welch"][f"{key[0]}:{key[1]}"]["ms"].append(round(ms, 4))
for k, v in res["welch"].items():
    v["best_GBps"] = round(4.0 * n / (min(v["ms"]) * 1e-3) / 1e9, 1)
    v["median_ms"] = sorted(v["ms"])[len(v["ms"]) // 2]
# ---- occupancy / schedule knobs (MDSP_WG_PER_CU and MDSP_CHUNK_LOG2 are read at every launch) ----
res["knobs"] = {}
_lib.set_tunable("MDSP_OLS_VARIANT", "0"); _lib.set_tunable("MDSP_WELCH_VARIANT", "0")
p0 = OlsPlan(taps, 2048, n, 0, d.ENGINE_FUSED)
c0 = d.WelchConfig(n, np.float32, n=4096, noverlap=2048, window=d.hanning, engine=d.ENGINE_FUSED)
for w in os.environ.get("TUNE_WGS", "1,2,4").split(","):
    for c in os.environ.get("TUNE_RUNS", "1,2,8,64").split(","):
        _lib.set_tunable("MDSP_WG_PER_CU", w)
        _lib.set_tunable("MDSP_RUNS_PER_SLOT", c)
        to = [timeit(lambda: _lib.check(lib.mdsp_ols_exec(p0._h, x.data_ptr(), n, 1, n, y.data_ptr(), n, n, stream))) for _ in range(rounds)]
        tw = [timeit(lambda: _lib.check(lib.mdsp_welch_exec(c0._h, x.data_ptr(), n, 1, n, psd.data_ptr(), 2049, stream))) for _ in range(rounds)]
        res["knobs"][f"wg{w}_chunk{c}"] = {"ols_ms": min(to), "ols_GBps": round(8.0 * n / (min(to) * 1e-3) / 1e9, 1), "welch_ms": min(tw),
                                          "welch_GBps": round(4.0 * n / (min(tw) * 1e-3) / 1e9, 1)}
        print("knobs wg", w, "runs_per_slot", c, res["knobs"][f"wg{w}_chunk{c}"])
_lib.set_tunable("MDSP_WG_PER_CU", None); _lib.set_tunable("MDSP_RUNS_PER_SLOT", None)
# ---- ablations (MDSP_ABLATE: 1 no loads, 2 no transforms, 4 no stores/accumulate) ----
res["ablate"] = {}
# only in a -DMDSP_DEBUG_KNOBS build:  python dsp.jl_amd/build.py --tag dbg --cflags -DMDSP_DEBUG_KNOBS ; MDSP_LIB_TAG=dbg python tools/tune.py
for ab in (os.environ.get("TUNE_ABLATE", "0,1,2,4,3,6,5,7").split(",") if lib.mdsp_debug_knobs() else []):
    _lib.set_tunable("MDSP_ABLATE", ab)
    to = [timeit(lambda: _lib.check(lib.mdsp_ols_exec(p0._h, x.data_ptr(), n, 1, n, y.data_ptr(), n, n, stream))) for _ in range(rounds)]
    tw = [timeit(lambda: _lib.check(lib.mdsp_welch_exec(c0._h, x.data_ptr(), n, 1, n, psd.data_ptr(), 2049, stream))) for _ in range(rounds)]
    res["ablate"][ab] = {"ols_ms": round(min(to), 4), "welch_ms": round(min(tw), 4)}
    print("ablate", ab, res["ablate"][ab])
_lib.set_tunable("MDSP_ABLATE", "0")
# ---- power / DVFS probe: the same two launches on an all-zero stream (no toggling data: MI355X_MICROARCH.md "DVFS give-back") ----
if os.environ.get("TUNE_ZERO"):
    xz = torch.zeros_like(x)
    to = [timeit(lambda: _lib.check(lib.mdsp_ols_exec(p0._h, xz.data_ptr(), n, 1, n, y.data_ptr(), n, n, stream))) for _ in range(rounds)]
    tw = [timeit(lambda: _lib.check(lib.mdsp_welch_exec(c0._h, xz.data_ptr(), n, 1, n, psd.data_ptr(), 2049, stream))) for _ in range(rounds)]
    tor = [timeit(lambda: _lib.check(lib.mdsp_ols_exec(p0._h, x.data_ptr(), n, 1, n, y.data_ptr(), n, n, stream))) for _ in range(rounds)]
    twr = [timeit(lambda: _lib.check(lib.mdsp_welch_exec(c0._h, x.data_ptr(), n, 1, n, psd.data_ptr(), 2049, stream))) for _ in range(rounds)]
    res["zero_input"] = {"ols_ms_zeros": min(to), "ols_ms_random": min(tor), "welch_ms_zeros": min(tw), "welch_ms_random": min(twr)}
    print("zero-input probe", res["zero_input"])
    del xz
# ---- copy yardstick ----
cms = [timeit(lambda: _lib.check(lib.mdsp_copy_bench(y.data_ptr(), x.data_ptr(), n * 4, stream))) for _ in range(5)]
res["copy_GBps"] = round(2 * 4.0 * n / (min(cms) * 1e-3) / 1e9, 1)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "tune.json"), "w"), indent=1)
for sec in ("ols", "welch"):
    for k, v in res[sec].items():
        print(sec, k, v["median_ms"], "ms", v["best_GBps"], "GB/s", {kk: vv for kk, vv in v.items() if "err" in kk or "diff" in kk})
print("copy", res["copy_GBps"], "GB/s")
