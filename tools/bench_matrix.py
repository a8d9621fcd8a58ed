#!/usr/bin/env python3
"""Throughput matrix over element types and transform sizes (fused engine, 2^27 samples, one channel): overlap-save filt,
Welch at 50 % overlap, spectrogram at 50 % overlap.  Gsamples/s; used to spot configurations that fall off the curve."""
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
from dsp_jl_amd.periodograms import _StftPlan, compute_window

lib = _lib.lib()
_lib.check(lib.mdsp_init(0))
n = 1 << int(os.environ.get("MATRIX_LOG2N", "27"))
stream = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda"); g.manual_seed(1776)


def ev():
    e = C.c_void_p(); _lib.check(lib.mdsp_event_create(C.byref(e))); return e


e0, e1 = ev(), ev()


def timeit(fn, reps=4):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        _lib.check(lib.mdsp_event_record(e0, stream)); fn(); _lib.check(lib.mdsp_event_record(e1, stream))
        ms = C.c_float(); _lib.check(lib.mdsp_event_elapsed_ms(e0, e1, C.byref(ms))); ts.append(ms.value)
    return min(ts)


res = {}
TD = {np.float32: torch.float32, np.float64: torch.float64, np.complex64: torch.complex64, np.complex128: torch.complex128}
for dt in (np.float32, np.float64, np.complex64, np.complex128):
    cplx = np.dtype(dt).kind == "c"
    rdt = np.float32 if dt in (np.float32, np.complex64) else np.float64
    x = torch.randn(n, generator=g, device="cuda", dtype=TD[rdt])
    if cplx:
        x = torch.complex(x, torch.randn(n, generator=g, device="cuda", dtype=TD[rdt]))
    name = np.dtype(dt).name
    for nfft in (256, 1024, 2048, 4096, 8192):
        if nfft == 8192 and rdt == np.float64:
            continue
        row = {}
        taps = d.design.lowpass_firwindow(0.25, d.hamming(nfft // 8), fs=1.0).astype(rdt)
        try:
            p = OlsPlan(taps if not cplx else taps.astype(dt), nfft, n, 0, d.ENGINE_FUSED)
            y = torch.empty_like(x)
            ms = timeit(lambda: _lib.check(lib.mdsp_ols_exec(p._h, x.data_ptr(), n, 1, n, y.data_ptr(), n, n, stream)))
            row["filt"] = round(n / ms / 1e6, 1)
            if os.environ.get("MATRIX_PREFETCH"):      # the same plan with the software prefetch switched on (MDSP_OLS_PREFETCH is read at launch)
                _lib.set_tunable("MDSP_OLS_PREFETCH", 1)
                ms = timeit(lambda: _lib.check(lib.mdsp_ols_exec(p._h, x.data_ptr(), n, 1, n, y.data_ptr(), n, n, stream)))
                _lib.set_tunable("MDSP_OLS_PREFETCH", None)
                row["filt_prefetch"] = round(n / ms / 1e6, 1)
            del y, p
        except Exception as e:
            row["filt"] = str(e)[:40]
        if os.environ.get("MATRIX_ONLY") == "filt":
            res[f"{name}_{nfft}"] = row
            print(name, nfft, row, flush=True)
            continue
        cfg = d.WelchConfig(n, dt, n=nfft, noverlap=nfft // 2, window=d.hanning, engine=d.ENGINE_FUSED)
        psd = torch.empty(cfg.nout, dtype=TD[rdt], device="cuda")
        ms = timeit(lambda: _lib.check(lib.mdsp_welch_exec(cfg._h, x.data_ptr(), n, 1, n, psd.data_ptr(), cfg.nout, stream)))
        row["welch50"] = round(n / ms / 1e6, 1)
        m = n >> 1
        win, norm2 = compute_window(d.hanning, nfft)
        K = d.frame_count(m, nfft, nfft // 2)
        plan = _StftPlan(nfft, nfft // 2, nfft, win, norm2, not cplx, True, dt, d.ENGINE_FUSED)
        out = torch.empty((K, plan.nout), dtype=TD[rdt], device="cuda")
        ms = timeit(lambda: _lib.check(lib.mdsp_stft_exec(plan._h, x.data_ptr(), m, 1, m, out.data_ptr(), plan.nout, K * plan.nout, stream)))
        row["spectrogram50"] = round(m / ms / 1e6, 1)
        del out, plan, psd, cfg
        res[f"{name}_{nfft}"] = row
        print(name, nfft, row, flush=True)
    del x
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "matrix.json"), "w"), indent=1)
