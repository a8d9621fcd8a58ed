#!/usr/bin/env python3
"""The run-time-schedule spectral kernel (csrc/gx_kernels.h) against the Float64 oracle and the rocFFT pipeline: Welch (50 % overlap) at 7-smooth sizes without
a compile-time schedule, 256 ... 65536 points.  GX_SIZES, GX_LOG2N (stream length), GX_DTYPES (f32,f64,c32,c64), GX_CHECK=0 skips the oracle,
GX_COLUMNS=1 adds the column modes (stft 75 % overlap of complex signals: 8 + 32 B per sample; spectrogram 50 % of real ones: ~8 B per sample; timing only).
The fused engine is whatever AUTO-with-engine=FUSED resolves to (MDSP_GX=0 in the environment: the round-5 choice).  Writes gpurun_out/gx.json."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import dsp_jl_amd as d
from dsp_jl_amd import _lib

lib = _lib.lib()
_lib.check(lib.mdsp_init(0))
stream = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda"); g.manual_seed(1776)
n = 1 << int(os.environ.get("GX_LOG2N", "26"))
check = os.environ.get("GX_CHECK", "1") != "0"


def ev():
    e = C.c_void_p(); _lib.check(lib.mdsp_event_create(C.byref(e))); return e


e0, e1 = ev(), ev()


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        _lib.check(lib.mdsp_event_record(e0, stream)); fn(); _lib.check(lib.mdsp_event_record(e1, stream))
        torch.cuda.synchronize()
        ms = C.c_float(); _lib.check(lib.mdsp_event_elapsed_ms(e0, e1, C.byref(ms))); ts.append(ms.value)
    return sorted(ts)[len(ts) // 2]


NP = {"f32": np.float32, "f64": np.float64, "c32": np.complex64, "c64": np.complex128}
TD = {"f32": torch.float32, "f64": torch.float64, "c32": torch.complex64, "c64": torch.complex128}
res = {}
sizes = [int(v) for v in os.environ.get("GX_SIZES", "1125,4802,8400,12500,16384,20000,40000,65536,125000").split(",")]
for dt in os.environ.get("GX_DTYPES", "f32").split(","):
    cplx = dt.startswith("c")
    nn = n // (2 if cplx else 1) // (2 if dt.endswith("64") else 1)
    xr = torch.randn(nn, generator=g, device="cuda", dtype=torch.float32 if dt.endswith("32") else torch.float64)
    if cplx:
        x = torch.complex(xr, torch.randn(nn, generator=g, device="cuda", dtype=xr.dtype))
    else:
        x = xr + 0.5 * torch.sin(2 * np.pi * 0.1234 * torch.arange(nn, device="cuda", dtype=xr.dtype))
    bps = x.element_size()
    for nfft in sizes:
        row = {}
        outs = {}
        for eng, ename in ((d.ENGINE_FUSED, "fused"), (d.ENGINE_ROCFFT, "rocfft")):
            try:
                cfg = d.WelchConfig(nn, NP[dt], n=nfft, noverlap=nfft // 2, nfft=nfft, window=d.hanning, engine=eng, onesided=not cplx)
            except Exception as ex:   # noqa: BLE001
                row[ename] = {"error": str(ex)[:200]}
                continue
            psd = torch.empty(cfg.nout, dtype=xr.dtype, device="cuda")
            ms = timeit(lambda: _lib.check(lib.mdsp_welch_exec(cfg._h, x.data_ptr(), nn, 1, nn, psd.data_ptr(), cfg.nout, stream)))
            row[ename] = {"ms": round(ms, 4), "Gsamples_per_s": round(nn / ms / 1e6, 1), "TBps_algorithmic": round(bps * nn / ms / 1e9, 3)}
            outs[ename] = psd.cpu().numpy().astype(np.float64)
        if os.environ.get("GX_COLUMNS") == "1":
            from dsp_jl_amd.periodograms import _StftPlan, compute_window
            win, norm2 = compute_window(d.hanning, nfft)
            for eng, ename in ((d.ENGINE_FUSED, "fused"), (d.ENGINE_ROCFFT, "rocfft")):
                try:
                    if cplx:
                        hop = nfft // 4
                        K = d.frame_count(nn, nfft, nfft - hop)
                        plan = _StftPlan(nfft, nfft - hop, nfft, win, norm2, False, 0, NP[dt], eng)
                        out = torch.empty((K, nfft), dtype=TD[dt], device="cuda")
                        ms = timeit(lambda: _lib.check(lib.mdsp_stft_exec(plan._h, x.data_ptr(), nn, 1, nn, out.data_ptr(), nfft, K * nfft, stream)))
                        row[f"stft75_{ename}"] = {"ms": round(ms, 4), "TBps_algorithmic": round(5.0 * bps * nn / ms / 1e9, 3)}
                    else:
                        plan = _StftPlan(nfft, nfft // 2, nfft, win, norm2, True, 1, NP[dt], eng)
                        K = d.frame_count(nn, nfft, nfft // 2)
                        out = torch.empty((K, plan.nout), dtype=xr.dtype, device="cuda")
                        ms = timeit(lambda: _lib.check(lib.mdsp_stft_exec(plan._h, x.data_ptr(), nn, 1, nn, out.data_ptr(), plan.nout, K * plan.nout, stream)))
                        row[f"spectrogram50_{ename}"] = {"ms": round(ms, 4), "TBps_algorithmic": round((1.0 + plan.nout / (nfft // 2)) * bps * nn / ms / 1e9, 3)}
                    del out, plan
                except Exception as ex:   # noqa: BLE001
                    row[f"columns_{ename}"] = {"error": str(ex)[:200]}
        if "fused" in outs and "rocfft" in outs:
            row["fused_vs_rocfft"] = float(np.linalg.norm(outs["fused"] - outs["rocfft"]) / np.linalg.norm(outs["rocfft"]))
        if check and "fused" in outs:
            from oracle import periodograms as opg, windows as ow
            m = min(nn, 40 * nfft)   # the oracle on a prefix: the same plan on the same prefix
            cfg = d.WelchConfig(m, NP[dt], n=nfft, noverlap=nfft // 2, nfft=nfft, window=d.hanning, engine=d.ENGINE_FUSED, onesided=not cplx)
            psd = torch.empty(cfg.nout, dtype=xr.dtype, device="cuda")
            _lib.check(lib.mdsp_welch_exec(cfg._h, x.data_ptr(), m, 1, m, psd.data_ptr(), cfg.nout, stream))
            torch.cuda.synchronize()
            ref = opg.welch_pgram(x[:m].cpu().numpy(), nfft, nfft // 2, window=ow.hanning, onesided=not cplx, dtype=np.float64).power
            got = psd.cpu().numpy().astype(np.float64)
            row["vs_oracle"] = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
            row["vs_oracle_max_ulps_of_max"] = float(np.max(np.abs(got - ref)) / (np.max(ref) * np.finfo(xr.cpu().numpy().dtype).eps))
        res.setdefault(dt, {})[str(nfft)] = row
        print(dt, nfft, json.dumps(row), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", os.environ.get("GX_OUT", "gx.json")), "w"), indent=1)
