#!/usr/bin/env python3
"""A/B of the polyphase kernels on BASELINE config 5's single-GPU share (160//147, 5120 taps, 4 channels x 2^28 Float32):
register-tap kernel (MDSP_FIR_MM=0) against the matrix-core kernel (MDSP_FIR_MM=1), interleaved rounds in one process.
Writes gpurun_out/tune_fir.json.   TUNE_FIR="mm,wg_per_cu,p;..."  TUNE_LOG2N=28  TUNE_RATIO=160/147  TUNE_TAPS=5120"""
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
log2n = int(os.environ.get("TUNE_LOG2N", "28"))
rounds = int(os.environ.get("TUNE_ROUNDS", "5"))
nch, n = int(os.environ.get("TUNE_NCH", "4")), 1 << log2n
L, M = (int(v) for v in os.environ.get("TUNE_RATIO", "160/147").split("/"))
variants = [tuple(int(t) for t in v.split(",")) for v in os.environ.get("TUNE_FIR", "0,0,0;1,0,0;1,2,0;1,1,0").split(";")]
# MDSP_FIR_MM_VERIFY_CHOICE=1: the library's own choice of form (tile by cost, padded runs, register taps, memory-wave priority) against
# round 2's rule (largest tile, taps fetched per tile beyond the round-2 register limits, memory waves at normal priority) on THIS box, for
# the ratio / type given by TUNE_RATIO / TUNE_DTYPE; exit status 3 and a REGRESSION line if the choice is more than 2 % slower than the rule it replaced.
VERIFY = os.environ.get("MDSP_FIR_MM_VERIFY_CHOICE", "0") == "1"
if VERIFY:
    variants = [(-1, 0, 0), (1, 0, 0, 0, 0, 8, 0, -1, -1, 1, 0, 0, 0, 0)]   # round 2: NG = 8 (largest tile), PRIO 0, T64 0, NBLK 0; row staging left to the library
g = torch.Generator(device="cuda"); g.manual_seed(1776)
DT = os.environ.get("TUNE_DTYPE", "f32")   # f32 | f64 | c32 | c64 (complex signal, real taps)
tdt, xdt, ydt, esz, lt, lx = {"f32": (torch.float32, np.float32, torch.float32, 4, _lib.F32, _lib.F32), "f64": (torch.float64, np.float64, torch.float64, 8, _lib.F64, _lib.F64),
                              "c32": (torch.complex64, np.float32, torch.complex64, 8, _lib.F32, _lib.C32),
                              "c64": (torch.complex128, np.float64, torch.complex128, 16, _lib.F64, _lib.C64)}[DT]
x = torch.randn((nch, n), generator=g, device="cuda", dtype=tdt)
stream = torch.cuda.current_stream().cuda_stream
h = np.asarray(d.resample_filter(Fraction(L, M)), dtype=xdt)
ntaps = int(os.environ.get("TUNE_TAPS", "5120" if (L, M) == (160, 147) else "0"))   # config 5: 5120 taps = 32 per phase
if ntaps:
    h = np.concatenate([h, np.zeros(ntaps - len(h), xdt)]) if len(h) < ntaps else h[:ntaps].copy()
fh = C.c_void_p()
_lib.check(lib.mdsp_fir_create(C.byref(fh), h.ctypes.data_as(C.c_void_p), len(h), L, M, lt, lx, nch))
ol = C.c_int64(); _lib.check(lib.mdsp_fir_outputlength(fh, n, C.byref(ol)))
ldy = ol.value + int(os.environ.get("TUNE_LDPAD", "0"))
y = torch.empty((nch, ldy), dtype=ydt, device="cuda")
nw = C.c_int64()


def ev():
    e = C.c_void_p(); _lib.check(lib.mdsp_event_create(C.byref(e))); return e


e0, e1 = ev(), ev()


def run():
    _lib.check(lib.mdsp_fir_reset(fh))
    _lib.check(lib.mdsp_fir_exec(fh, x.data_ptr(), n, n, y.data_ptr(), ol.value, ldy, C.byref(nw), stream))


def select(v):
    _lib.set_tunable("MDSP_FIR_MM", str(v[0])); _lib.set_tunable("MDSP_WG_PER_CU", str(v[1])); _lib.set_tunable("MDSP_FIR_P", str(v[2]))
    _lib.set_tunable("MDSP_FIR_MM_ND", str(v[3]) if len(v) > 3 else "0"); _lib.set_tunable("MDSP_FIR_MM_NS", str(v[4]) if len(v) > 4 else "0")
    _lib.set_tunable("MDSP_FIR_MM_NG", str(v[5]) if len(v) > 5 else "0")   # cap on the 64-row groups per tile (0: the library's choice)
    _lib.set_tunable("MDSP_FIR_MM_CH", str(v[6]) if len(v) > 6 else "0")   # cap on the 16-row chunks per multiplying wave
    _lib.set_tunable("MDSP_FIR_MM_PAD", str(v[7]) if len(v) > 7 else "-1")  # 0: unpadded output rows
    _lib.set_tunable("MDSP_FIR_MM_ROWS", str(v[8]) if len(v) > 8 else "-1")  # 0 / 1: tile staged as one run / row by row
    _lib.set_tunable("MDSP_FIR_MM_NBLK", str(v[13]) if len(v) > 13 else "1")  # 0: L > 192 fetches its taps per tile
    _lib.set_tunable("MDSP_FIR_MM_T64", str(v[12]) if len(v) > 12 else "1")   # 0: 49 .. 64 k-steps fetched per tile
    _lib.set_tunable("MDSP_FIR_MM_PRIO", str(v[11]) if len(v) > 11 else "-1")   # 0 / 1: memory waves at normal / raised priority (-1: the library's choice)
    _lib.set_tunable("MDSP_FIR_MM_RPAD", str(v[10]) if len(v) > 10 else "0")   # dwords of padding behind a granule of a padded run
    _lib.set_tunable("MDSP_FIR_MM_VSTORE", str(v[9]) if len(v) > 9 else "1")  # 0: element-by-element stores of rows that are not whole vectors


def timeit():
    run(); torch.cuda.synchronize()
    _lib.check(lib.mdsp_event_record(e0, stream)); run(); _lib.check(lib.mdsp_event_record(e1, stream))
    ms = C.c_float(); _lib.check(lib.mdsp_event_elapsed_ms(e0, e1, C.byref(ms)))
    return ms.value


def vkey(v):
    return "mm={} wg_per_cu={} p={}".format(*v[:3]) + (" nd={} ns={}".format(*v[3:5]) if len(v) > 4 else "") + (" ng={}".format(v[5]) if len(v) > 5 else "") + (" ch={}".format(v[6]) if len(v) > 6 else "") + (" pad={}".format(v[7]) if len(v) > 7 else "") + (" rows={}".format(v[8]) if len(v) > 8 else "") + (" vstore={}".format(v[9]) if len(v) > 9 else "") + (" rpad={}".format(v[10]) if len(v) > 10 else "") + (" prio={}".format(v[11]) if len(v) > 11 else "") + (" t64={}".format(v[12]) if len(v) > 12 else "") + (" nblk={}".format(v[13]) if len(v) > 13 else "")


res = {"dtype": DT, "log2n": log2n, "nch": nch, "ratio": f"{L}//{M}", "taps": len(h), "nout": ol.value, "variants": {}}
ref = None
for v in variants:
    select(v); y.zero_(); run(); torch.cuda.synchronize()
    if ref is None:
        ref = y.clone()
    key = vkey(v)
    res["variants"][key] = {"maxdiff_vs_first": float((y - ref).abs().max()), "ms": []}
    del_ref = None
for r in range(rounds):
    for v in variants:
        select(v)
        res["variants"][vkey(v)]["ms"].append(round(timeit(), 4))
bytes_alg = (esz + esz * L / M) * n * nch
for k, e in res["variants"].items():
    e["median_ms"] = float(np.median(e["ms"]))
    e["GBps"] = round(bytes_alg / e["median_ms"] / 1e6, 1)
    print(k, e["median_ms"], "ms", e["GBps"], "GB/s  maxdiff", e["maxdiff_vs_first"], flush=True)
if VERIFY:
    (kc, ec), (kr, er) = list(res["variants"].items())
    ratio = ec["median_ms"] / er["median_ms"]
    res["verify_choice"] = {"choice_ms": ec["median_ms"], "round2_rule_ms": er["median_ms"], "choice_over_rule": round(ratio, 4), "regression": bool(ratio > 1.02)}
    print(("REGRESSION" if ratio > 1.02 else "ok") + f": {L}//{M} {DT}: library's choice {ec['median_ms']:.4f} ms, round-2 rule {er['median_ms']:.4f} ms ({ratio:.3f})", flush=True)
# TUNE_PERSIST=1: what THIS box measured goes into the choice file MDSP_FIR_CHOICE_FILE names (common.h FirChoice; the library reads it only when that variable
# is set -- opt-in since round 6 -- and only behind its key line): the first variant is the library's own rule; when another one is more
# than 2 % faster its knobs become the shape's line (only knobs the file may carry; MDSP_WG_PER_CU / _RPAD / _VSTORE are process-wide), otherwise a line
# left by an earlier run is removed.  The library applies a line to filters of exactly this (L, M, ntaps, dtypes) and to nothing else.
if os.environ.get("TUNE_PERSIST") == "1" and len(variants) > 1:
    import math
    path = os.environ.get("MDSP_FIR_CHOICE_FILE")
    if not path:
        raise SystemExit("TUNE_PERSIST=1 needs MDSP_FIR_CHOICE_FILE (the library reads no default location)")
    gq = math.gcd(L, M)
    shape = f"{L // gq} {M // gq} {len(h)} {lt} {lx}"
    keys = list(res["variants"].keys())
    best = min(keys, key=lambda k: res["variants"][k]["median_ms"])
    names = {0: ("MDSP_FIR_MM", -1), 2: ("MDSP_FIR_P", 0), 3: ("MDSP_FIR_MM_ND", 0), 4: ("MDSP_FIR_MM_NS", 0), 5: ("MDSP_FIR_MM_NG", 0), 6: ("MDSP_FIR_MM_CH", 0), 7: ("MDSP_FIR_MM_PAD", -1),
             8: ("MDSP_FIR_MM_ROWS", -1), 11: ("MDSP_FIR_MM_PRIO", -1), 12: ("MDSP_FIR_MM_T64", 1), 13: ("MDSP_FIR_MM_NBLK", 1)}
    lines = []
    if os.path.exists(path):
        lines = [ln for ln in open(path).read().splitlines() if ln.strip() and not ln.startswith(shape + " ")]
    v = variants[keys.index(best)]
    knobs = [f"{nm}={v[i]}" for i, (nm, dflt) in names.items() if i < len(v) and v[i] != dflt]
    if best != keys[0] and res["variants"][best]["median_ms"] < 0.98 * res["variants"][keys[0]]["median_ms"] and knobs:
        lines.append(shape + " " + ",".join(knobs))
        res["persisted"] = lines[-1]
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    open(path, "w").write(f"# mi355dsp-fir-choices v{lib.mdsp_version()} gfx950\n# polyphase choices measured on this box (tools/tune_fir.py TUNE_PERSIST=1): L M ntaps taps_dtype x_dtype KNOB=value,...\n" + "\n".join(ln for ln in lines if not ln.startswith("#")) + "\n")
    print("choice file", path, "->", res.get("persisted", "(the library's rule stands for this shape)"), flush=True)
select((-1, 0, 0)); _lib.set_tunable("MDSP_FIR_MM", None); _lib.set_tunable("MDSP_WG_PER_CU", None); _lib.set_tunable("MDSP_FIR_P", None)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "tune_fir.json"), "w"), indent=1)
if VERIFY and res["verify_choice"]["regression"]:
    sys.exit(3)
