#!/usr/bin/env python3
"""welch_pgram / spectrogram / stft / periodogram called with the reference's DEFAULT arguments (n = length >> 3, noverlap = n >> 1,
nfft = nextfastfft(n): periodograms.jl:560, :647, :828, :872) on device-resident Float32 streams: the multi-pass fused engine (bigfft.hip)
against the rocFFT pipeline these sizes took up to round 4.  Algorithmic bytes: Welch / periodogram 4 B per sample; spectrogram 4 B in + 4 B per
PSD bin out; stft 4 B in + 8 B per bin out.  Writes gpurun_out/default_spectral.json.

    DEFSPEC_LENGTHS=1048576,1000000,16777216,10000000,134217728   DEFSPEC_CHUNKS=16,64,256 (MDSP_BIG_CHUNK_MIB sweep, Welch only)"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import dsp_jl_amd as d
from dsp_jl_amd import _lib, util
from dsp_jl_amd.periodograms import _StftPlan, compute_window

lib = _lib.lib()
_lib.check(lib.mdsp_init(0))
stream = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda"); g.manual_seed(1776)


def ev():
    e = C.c_void_p(); _lib.check(lib.mdsp_event_create(C.byref(e))); return e


e0, e1 = ev(), ev()


def timeit(fn, reps=7):
    fn(); fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        _lib.check(lib.mdsp_event_record(e0, stream)); fn(); _lib.check(lib.mdsp_event_record(e1, stream))
        torch.cuda.synchronize()
        ms = C.c_float(); _lib.check(lib.mdsp_event_elapsed_ms(e0, e1, C.byref(ms))); ts.append(ms.value)
    return sorted(ts)[len(ts) // 2]


def row(ms, nbytes):
    return {"ms": round(ms, 4), "TBps_algorithmic": round(nbytes / ms / 1e9, 4)}


lengths = [int(v) for v in os.environ.get("DEFSPEC_LENGTHS", "1048576,1000000,16777216,10000000,134217728").split(",")]
chunks = [int(v) for v in os.environ.get("DEFSPEC_CHUNKS", "").split(",") if v]
dt_name = os.environ.get("DEFSPEC_DTYPE", "f32")
npdt, tdt, esz = (np.float32, torch.float32, 4) if dt_name == "f32" else (np.float64, torch.float64, 8)
res = {"dtype": dt_name, "rows": {}}
for length in lengths:
    x = torch.randn(length, generator=g, device="cuda", dtype=tdt)
    n = length >> 3
    nov = n >> 1
    nfft = util.nextfastfft(n)
    K = d.frame_count(length, n, nov)
    r = {"n": n, "nfft": nfft, "frames": K}
    for eng, ename in [e for e in ((d.ENGINE_AUTO, "auto"), (d.ENGINE_ROCFFT, "rocfft")) if e[1] in os.environ.get("DEFSPEC_ENGINES", "auto,rocfft").split(",")]:
        cfg = d.WelchConfig(length, npdt, n=n, noverlap=nov, nfft=nfft, window=d.hanning, engine=eng)
        psd = torch.empty(cfg.nout, dtype=tdt, device="cuda")
        ms = timeit(lambda: _lib.check(lib.mdsp_welch_exec(cfg._h, x.data_ptr(), length, 1, length, psd.data_ptr(), cfg.nout, stream)))
        r[f"welch_{ename}"] = dict(row(ms, esz * length), engine=cfg.engine)
        if ename == "auto" and cfg.engine == d.ENGINE_FUSED:
            for c in chunks:
                os.environ["MDSP_BIG_CHUNK_MIB"] = str(c)
                _lib.check(lib.mdsp_reload_tunables())
                ms = timeit(lambda: _lib.check(lib.mdsp_welch_exec(cfg._h, x.data_ptr(), length, 1, length, psd.data_ptr(), cfg.nout, stream)))
                r[f"welch_auto_chunk{c}"] = row(ms, esz * length)
            if chunks:
                del os.environ["MDSP_BIG_CHUNK_MIB"]
                _lib.check(lib.mdsp_reload_tunables())
        del cfg, psd
        if os.environ.get("DEFSPEC_WELCH_ONLY"):
            continue
        win, norm2 = compute_window(d.hanning, n)
        for psd_only, name, osz in ((1, "spectrogram", esz), (0, "stft", 2 * esz)):
            plan = _StftPlan(n, nov, nfft, win, norm2, True, psd_only, npdt, eng)
            out = torch.empty((K, plan.nout), dtype=tdt if psd_only else (torch.complex64 if esz == 4 else torch.complex128), device="cuda")
            ms = timeit(lambda: _lib.check(lib.mdsp_stft_exec(plan._h, x.data_ptr(), length, 1, length, out.data_ptr(), plan.nout, K * plan.nout, stream)))
            r[f"{name}_{ename}"] = dict(row(ms, esz * length + osz * K * plan.nout), engine=plan.engine)
            del plan, out
        # periodogram(s): one frame of nextfastfft(length) points
        nf1 = util.nextfastfft(length)
        win1, norm1 = compute_window(None, length)
        plan = _StftPlan(length, 0, nf1, win1, norm1, True, 1, npdt, eng)
        out = torch.empty((1, plan.nout), dtype=tdt, device="cuda")
        ms = timeit(lambda: _lib.check(lib.mdsp_stft_exec(plan._h, x.data_ptr(), length, 1, length, out.data_ptr(), plan.nout, plan.nout, stream)))
        r[f"periodogram_{ename}"] = dict(row(ms, esz * length), engine=plan.engine, nfft=nf1)
        del plan, out
    res["rows"][str(length)] = r
    print(length, {k: (v["TBps_algorithmic"] if isinstance(v, dict) else v) for k, v in r.items()}, flush=True)
    del x
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", os.environ.get("DEFSPEC_OUT", "default_spectral.json")), "w"), indent=1)
