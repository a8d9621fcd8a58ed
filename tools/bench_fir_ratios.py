#!/usr/bin/env python3
"""Every ratio x signal type of the polyphase table (profiles/r0N_fir_all_ratios.json) in ONE process, knob variants interleaved round by round:
4 channels x 2^26 samples, taps = resample_filter(ratio), median of FIRR_ROUNDS.  Algorithmic bytes = (1 + L / M) x element size per input sample.

    FIRR_VARIANTS="default;MDSP_FIR_MM_RPX=1;MDSP_FIR_MM_TIEWAVES=1;MDSP_FIR_MM_RPX=1,MDSP_FIR_MM_TIEWAVES=1"
    FIRR_DTYPES=f32,f64,c32,c64   FIRR_RATIOS=160/147,...   FIRR_OUT=fir_ratios.json
Writes gpurun_out/$FIRR_OUT: per cell and variant median ms, GB/s, fraction of 8 TB/s, the kernel path taken, and max |difference| to the first variant."""
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
from fractions import Fraction

lib = _lib.lib()
_lib.check(lib.mdsp_init(0))
log2n = int(os.environ.get("FIRR_LOG2N", "26"))
rounds = int(os.environ.get("FIRR_ROUNDS", "3"))
nch, n = int(os.environ.get("FIRR_NCH", "4")), 1 << log2n
dtypes = os.environ.get("FIRR_DTYPES", "f32,f64,c32,c64").split(",")
ratios = os.environ.get("FIRR_RATIOS", "160/147,147/160,2/1,1/2,3/2,2/3,5/3,4/1,1/3,1/4,1/8,1/16,3/8,160/441,441/160").split(",")
variants = []
for v in os.environ.get("FIRR_VARIANTS", "default").split(";"):
    variants.append((v, {} if v == "default" else dict(kv.split("=") for kv in v.split(","))))
knobs = sorted({k for _, kv in variants for k in kv})
stream = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda"); g.manual_seed(1776)
TYPES = {"f32": (torch.float32, np.float32, 4, _lib.F32, _lib.F32), "f64": (torch.float64, np.float64, 8, _lib.F64, _lib.F64),
         "c32": (torch.complex64, np.float32, 8, _lib.F32, _lib.C32), "c64": (torch.complex128, np.float64, 16, _lib.F64, _lib.C64)}


def ev():
    e = C.c_void_p(); _lib.check(lib.mdsp_event_create(C.byref(e))); return e


e0, e1 = ev(), ev()


def select(kv):
    for k in knobs:
        if k in kv:
            os.environ[k] = kv[k]
        else:
            os.environ.pop(k, None)
    _lib.check(lib.mdsp_reload_tunables())


out = {"note": f"tools/bench_fir_ratios.py: {nch} channels x 2^{log2n} samples, median of {rounds}, variants interleaved in one process", "cells": {}}
for dt in dtypes:
    tdt, hdt, esz, lt, lx = TYPES[dt]
    x = torch.randn((nch, n), generator=g, device="cuda", dtype=tdt)
    for ratio in ratios:
        L, M = (int(v) for v in ratio.split("/"))
        h = np.asarray(d.resample_filter(Fraction(L, M)), dtype=hdt)
        fh = C.c_void_p()
        _lib.check(lib.mdsp_fir_create(C.byref(fh), h.ctypes.data_as(C.c_void_p), len(h), L, M, lt, lx, nch))
        ol = C.c_int64(); _lib.check(lib.mdsp_fir_outputlength(fh, n, C.byref(ol)))
        y = torch.empty((nch, ol.value), dtype=tdt, device="cuda")
        nw = C.c_int64()

        def run():
            _lib.check(lib.mdsp_fir_reset(fh))
            _lib.check(lib.mdsp_fir_exec(fh, x.data_ptr(), n, n, y.data_ptr(), ol.value, ol.value, C.byref(nw), stream))

        cell = {"taps": len(h), "variants": {}}
        ref = None
        for name, kv in variants:
            select(kv)
            y.zero_(); run(); torch.cuda.synchronize()
            path = C.c_int(-1)
            _lib.check(lib.mdsp_fir_kernel_path(fh, n, C.byref(path)))
            if ref is None:
                ref = y.clone()
            cell["variants"][name] = {"ms": [], "path": path.value, "maxdiff_vs_first": float((y - ref).abs().max())}
        for _ in range(rounds):
            for name, kv in variants:
                select(kv)
                run(); torch.cuda.synchronize()
                _lib.check(lib.mdsp_event_record(e0, stream)); run(); _lib.check(lib.mdsp_event_record(e1, stream))
                torch.cuda.synchronize()
                ms = C.c_float(); _lib.check(lib.mdsp_event_elapsed_ms(e0, e1, C.byref(ms)))
                cell["variants"][name]["ms"].append(round(ms.value, 4))
        bytes_alg = (esz + esz * L / M) * n * nch
        for name, e in cell["variants"].items():
            e["median_ms"] = float(np.median(e["ms"]))
            e["GBps"] = round(bytes_alg / e["median_ms"] / 1e6, 1)
            e["frac_of_8TBps"] = round(e["GBps"] / 8000, 3)
        key = f"{dt}_{ratio.replace('/', '_')}"
        out["cells"][key] = cell
        print(key, {k: (v["median_ms"], v["frac_of_8TBps"], v["path"]) for k, v in cell["variants"].items()}, flush=True)
        _lib.check(lib.mdsp_fir_destroy(fh))
        del y, ref
    del x
select({})
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", os.environ.get("FIRR_OUT", "fir_ratios.json")), "w"), indent=1)
