#!/usr/bin/env python3
"""Measure the other SURVEY section-8 rows at (single-GPU shares of) their BASELINE configs and write gpurun_out/rows.json:
   config 4: stft / spectrogram, nfft=1024 hop=256 hanning, ComplexF32, 8 channels x 2^26 (the 1-GPU share of 64 channels on 8 GPUs)
   config 5: FIRFilter 160//147, 5120 taps (32/phase), Float32, 4 channels x 2^28 (the 1-GPU share of 32 channels)
   config 1: filt(b, 1, x) 127 taps, 10^6 Float64 (time-domain path)
Reported: ms per call (HIP events, device-resident data), algorithmic GB/s (SURVEY 8d bytes/sample), Gsamples/s."""
import ctypes as C
import json
import math
import os
import sys
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import dsp_jl_amd as d
from dsp_jl_amd import _lib, _dev
from dsp_jl_amd.periodograms import _StftPlan, compute_window

lib = _lib.lib()
_lib.check(lib.mdsp_init(0))
stream = torch.cuda.current_stream().cuda_stream
reps = int(os.environ.get("ROWS_REPS", "5"))
scale = int(os.environ.get("ROWS_SHRINK", "0"))          # shrink lengths by 2^scale for quick runs


def ev():
    e = C.c_void_p(); _lib.check(lib.mdsp_event_create(C.byref(e))); return e


e0, e1 = ev(), ev()


def timeit(fn):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        _lib.check(lib.mdsp_event_record(e0, stream)); fn(); _lib.check(lib.mdsp_event_record(e1, stream))
        ms = C.c_float(); _lib.check(lib.mdsp_event_elapsed_ms(e0, e1, C.byref(ms))); ts.append(ms.value)
    return sorted(ts)[len(ts) // 2], min(ts)


res = {}
g = torch.Generator(device="cuda"); g.manual_seed(1776)

# ---------------- config 4: STFT / spectrogram ----------------
nch, n = 8, 1 << (26 - scale)
if os.environ.get("ROWS_SKIP_STFT") or os.environ.get("ROWS_ONLY") == "fir":
    nch = 0
z = torch.randn((max(nch, 1), n if nch else 8, 2), generator=g, device="cuda", dtype=torch.float32) * math.sqrt(0.5)
s = torch.view_as_complex(z)                                    # (nch, n) C64: channel = contiguous column
win, norm2 = compute_window(d.hanning, 1024)
K = d.frame_count(n, 1024, 768)
for name, psd, outdt, bps in ((("stft", 0, torch.complex64, 40.0), ("spectrogram", 1, torch.float32, 24.0)) if nch else ()):
    for eng, ename in ((d.ENGINE_FUSED, "fused"), (d.ENGINE_ROCFFT, "rocfft")):
        if os.environ.get("ROWS_ENGINE", ename) != ename:
            continue
        plan = _StftPlan(1024, 768, 1024, win, 1.0 * norm2, False, psd, np.complex64, eng)
        out = torch.empty((nch, K, 1024), dtype=outdt, device="cuda")
        f = lambda: _lib.check(lib.mdsp_stft_exec(plan._h, s.data_ptr(), n, nch, n, out.data_ptr(), 1024, K * 1024, stream))
        med, best = timeit(f)
        res[f"config4_{name}_{ename}"] = {"ms": round(med, 4), "best_ms": round(best, 4), "channels": nch, "samples_per_channel": n,
                                          "GBps_algorithmic": round(bps * n * nch / (best * 1e-3) / 1e9, 1),
                                          "Gsamples_per_s": round(n * nch / (best * 1e-3) / 1e9, 2)}
        print(name, ename, res[f"config4_{name}_{ename}"], flush=True)
        del out, plan
del s, z
torch.cuda.empty_cache()
if os.environ.get("ROWS_ONLY") == "stft":
    sys.exit(0)

# ---------------- config 5: polyphase resampler 160//147 ----------------
nch, n = 4, 1 << (28 - scale)
h = d.resample_filter(Fraction(160, 147))
h = (np.concatenate([h, np.zeros(5120 - len(h))]) if len(h) < 5120 else h[:5120]).astype(np.float32)
x = torch.randn((nch, n), generator=g, device="cuda", dtype=torch.float32)
fh = C.c_void_p()
_lib.check(lib.mdsp_fir_create(C.byref(fh), h.ctypes.data_as(C.c_void_p), len(h), 160, 147, _lib.F32, _lib.F32, nch))
ol = C.c_int64(); _lib.check(lib.mdsp_fir_outputlength(fh, n, C.byref(ol)))
y = torch.empty((nch, ol.value), dtype=torch.float32, device="cuda")
nw = C.c_int64()


def fir():
    _lib.check(lib.mdsp_fir_reset(fh))
    _lib.check(lib.mdsp_fir_exec(fh, x.data_ptr(), n, n, y.data_ptr(), ol.value, ol.value, C.byref(nw), stream))


med, best = timeit(fir)
res["config5_resample_160_147"] = {"ms": round(med, 4), "best_ms": round(best, 4), "channels": nch, "samples_per_channel": n, "out_per_channel": ol.value,
                                   "GBps_algorithmic": round((4 + 4 * 160 / 147) * n * nch / (best * 1e-3) / 1e9, 1),
                                   "Gsamples_per_s": round(n * nch / (best * 1e-3) / 1e9, 2)}
print("resample", res["config5_resample_160_147"], flush=True)
if os.environ.get("ROWS_FIR_SWEEP"):
    for kib in (8, 12, 20, 32, 48):
        for wg in (2, 4, 8):
            _lib.set_tunable("MDSP_FIR_LDS_KIB", str(kib)); _lib.set_tunable("MDSP_WG_PER_CU", str(wg))
            med, best = timeit(fir)
            print("fir sweep lds_kib", kib, "wg_per_cu", wg, "ms", round(best, 3), "GB/s", round((4 + 4 * 160 / 147) * n * nch / (best * 1e-3) / 1e9, 1), flush=True)
    _lib.set_tunable("MDSP_FIR_LDS_KIB", None); _lib.set_tunable("MDSP_WG_PER_CU", None)
del y
if os.environ.get("ROWS_ONLY") == "fir":
    sys.exit(0)
# ---------------- next row 1: arbitrary-rate resampler (FIRArbitrary), rate 160/147 as a Float64, Nphi = 32 ----------------
import time
rate = 160 / 147
ha = d.resample_filter(rate, 32).astype(np.float32)
fa = C.c_void_p()
_lib.check(lib.mdsp_firarb_create(C.byref(fa), ha.ctypes.data_as(C.c_void_p), len(ha), rate, 32, _lib.F32, _lib.F32, nch))
ola = C.c_int64(); _lib.check(lib.mdsp_firarb_outputlength(fa, n, C.byref(ola)))
ya = torch.empty((nch, ola.value + 1), dtype=torch.float32, device="cuda")


def arb():
    _lib.check(lib.mdsp_firarb_reset(fa))
    _lib.check(lib.mdsp_firarb_exec(fa, x.data_ptr(), n, n, ya.data_ptr(), ola.value + 1, ola.value + 1, C.byref(nw), stream))


t0 = time.perf_counter(); arb(); torch.cuda.synchronize(); cold = time.perf_counter() - t0     # includes the serial host recurrence
med, best = timeit(arb)                                                                       # same (state, length): cached anchors
res["next1_resample_arbitrary"] = {"rate": rate, "taps": len(ha), "cold_call_s": round(cold, 4), "ms": round(med, 4), "best_ms": round(best, 4),
                                   "channels": nch, "samples_per_channel": n, "out_per_channel": nw.value,
                                   "GBps_algorithmic": round((4 + 4 * rate) * n * nch / (best * 1e-3) / 1e9, 1),
                                   "Gsamples_per_s": round(n * nch / (best * 1e-3) / 1e9, 2),
                                   "cold_Gsamples_per_s": round(n * nch / cold / 1e9, 2)}
print("arbitrary", res["next1_resample_arbitrary"], flush=True)
_lib.check(lib.mdsp_firarb_destroy(fa))
del x, ya
torch.cuda.empty_cache()

# ---------------- config 1: time-domain FIR 127 taps, 1e6 Float64 ----------------
b = d.design.lowpass_firwindow(0.25, d.hamming(127), fs=1.0)
x1 = torch.randn(10 ** 6, generator=g, device="cuda", dtype=torch.float64)
y1 = torch.empty_like(x1)
f1 = lambda: _lib.check(lib.mdsp_tdfir_exec(b.ctypes.data_as(C.c_void_p), 127, _lib.F64, x1.data_ptr(), 10 ** 6, 1, 10 ** 6, y1.data_ptr(), 10 ** 6, stream))
med, best = timeit(f1)
res["config1_tdfilt_127_f64"] = {"ms": round(med, 4), "best_ms": round(best, 4), "Msamples_per_s": round(1.0 / (best * 1e-3), 1)}
print("tdfilt", res["config1_tdfilt_127_f64"], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "rows.json"), "w"), indent=1)
