#!/usr/bin/env python3
"""Round 4: the three-pass composite-radix schedules of the nextfastfft sizes (MDSP_GEN_WIDE=1, spectral_gen.h MDSP_GEN_CT_WIDE_SIZES) against
round 3's small-radix schedules (MDSP_GEN_WIDE=0), the SAME plans in one process, alternating.  Welch 50 % (4 B/sample), ComplexF32 STFT 75 %
(40 B/sample), real spectrogram 50 % (~8 B/sample); 2^27 samples; algorithmic GB/s.  Writes gpurun_out/wide.json.
    WIDE_SIZES=3000,1500 WIDE_LOG2N=27 REPS=7 python tools/bench_wide.py"""
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
from dsp_jl_amd.periodograms import _StftPlan, compute_window

lib = _lib.lib()
_lib.check(lib.mdsp_init(0))
stream = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda"); g.manual_seed(1776)
n = 1 << int(os.environ.get("WIDE_LOG2N", "27"))
reps = int(os.environ.get("REPS", "7"))


def timeit(fn):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


res = {}
xr = torch.randn(n, generator=g, device="cuda", dtype=torch.float32)
xc = torch.complex(xr[: n // 2].clone(), torch.randn(n // 2, generator=g, device="cuda", dtype=torch.float32))
for nfft in [int(v) for v in os.environ.get("WIDE_SIZES", "3000").split(",")]:
    cfg = d.WelchConfig(n, np.float32, n=nfft, noverlap=nfft // 2, nfft=nfft, window=d.hanning, engine=d.ENGINE_FUSED)
    psd = torch.empty(cfg.nout, dtype=torch.float32, device="cuda")
    m = n // 2
    win, norm2 = compute_window(d.hanning, nfft)
    hop = nfft // 4
    Kc = d.frame_count(m, nfft, nfft - hop)
    pc = _StftPlan(nfft, nfft - hop, nfft, win, norm2, False, 0, np.complex64, d.ENGINE_FUSED)
    oc = torch.empty((Kc, nfft), dtype=torch.complex64, device="cuda")
    pr = _StftPlan(nfft, nfft // 2, nfft, win, norm2, True, 1, np.float32, d.ENGINE_FUSED)
    Kr = d.frame_count(n, nfft, nfft // 2)
    orr = torch.empty((Kr, pr.nout), dtype=torch.float32, device="cuda")
    cases = {
        "welch50": (lambda: _lib.check(lib.mdsp_welch_exec(cfg._h, xr.data_ptr(), n, 1, n, psd.data_ptr(), cfg.nout, stream)), 4.0 * n, lambda: psd.clone()),
        "stft75_c64": (lambda: _lib.check(lib.mdsp_stft_exec(pc._h, xc.data_ptr(), m, 1, m, oc.data_ptr(), nfft, Kc * nfft, stream)), 40.0 * m, lambda: oc[:64].clone()),
        "spectrogram50": (lambda: _lib.check(lib.mdsp_stft_exec(pr._h, xr.data_ptr(), n, 1, n, orr.data_ptr(), pr.nout, Kr * pr.nout, stream)),
                          (4.0 + 4.0 * pr.nout / (nfft // 2)) * n, lambda: orr[:64].clone()),
    }
    row = {}
    for name, (fn, nbytes, snap) in cases.items():
        r = {}
        outs = {}
        for rnd in range(2):
            for wide in (0, 1):
                _lib.set_tunable("MDSP_GEN_WIDE", wide)
                ms = timeit(fn)
                r.setdefault(f"wide{wide}", []).append(round(ms, 4))
                torch.cuda.synchronize(); outs[wide] = snap()
        a, b = outs[0].float() if outs[0].dtype != torch.complex64 else torch.view_as_real(outs[0]), outs[1].float() if outs[1].dtype != torch.complex64 else torch.view_as_real(outs[1])
        rel = float((a - b).abs().max() / a.abs().max())
        row[name] = {"ms_small": min(r["wide0"]), "ms_wide": min(r["wide1"]), "GBps_small": round(nbytes / min(r["wide0"]) / 1e6, 1),
                     "GBps_wide": round(nbytes / min(r["wide1"]) / 1e6, 1), "max_rel_diff": rel}
    res[str(nfft)] = row
    print(nfft, {k: (v["GBps_small"], v["GBps_wide"], f"{v['max_rel_diff']:.1e}") for k, v in row.items()}, flush=True)
    del cfg, pc, pr, oc, orr
_lib.set_tunable("MDSP_GEN_WIDE", None)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
outp = os.path.join(ROOT, "gpurun_out", os.environ.get("OUT", "wide.json"))
os.makedirs(os.path.dirname(outp), exist_ok=True)
json.dump(res, open(outp, "w"), indent=1)
