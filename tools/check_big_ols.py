#!/usr/bin/env python3
"""Overlap-save beyond the partitioned kernels (blocks on the multi-pass engine, bigfft.hip run_ols): correctness against the rocFFT engine and a
Float64 host convolution, then throughput against the host-summed segments it replaces.  Writes gpurun_out/big_ols.json.
    BIGOLS_TAPS=32768,65536,131072   BIGOLS_LOG2N=0,18,19,20,21 (MDSP_BIG_OLS_LOG2N sweep)"""
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
stream = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda"); g.manual_seed(1776)
res = {"check": {}, "bench": {}}
SEG = {4: 16384, 8: 8192}     # the partitioned kernels' range


def segments(taps, cols, nx, seg):
    """What hosts did up to round 4: filt(b, x) = sum_k delay(filt(b[k seg : (k + 1) seg], x), k seg), every segment on the partitioned kernels."""
    out = None
    for k in range(-(-len(taps) // seg)):
        hk = np.ascontiguousarray(taps[k * seg:(k + 1) * seg])
        plan = OlsPlan(hk, max(256, 1 << (2 * len(hk) - 1).bit_length()), nx, _lib.OLS_FILT, _lib.ENGINE_FUSED, cached=True)
        t = plan.exec(cols, nx)
        if out is None:
            out = t
        else:
            out[..., k * seg:] += t[..., :nx - k * seg]   # (up to round 5 a library entry, mdsp_shift_add; no host needs it since the long-filter plans)
    return out


# correctness: every dtype, both modes, a length that is not a multiple of anything, against numpy in Float64
rng = np.random.default_rng(5)
for name, hdt, tdt in () if os.environ.get("BIGOLS_SKIP_CHECK") else (("f32", np.float32, torch.float32), ("f64", np.float64, torch.float64), ("c32", np.complex64, torch.complex64), ("c64", np.complex128, torch.complex128)):
    for nb in (20001, 40000):
        nx = 1_300_017
        cplx = np.dtype(hdt).kind == "c"
        b = rng.standard_normal(nb) + (1j * rng.standard_normal(nb) if cplx else 0)
        b = (b / np.sqrt(nb)).astype(hdt)
        xh = rng.standard_normal(nx) + (1j * rng.standard_normal(nx) if cplx else 0)
        xh = xh.astype(hdt)
        x = torch.from_numpy(xh).cuda()
        nf = 1 << 22
        full = np.fft.ifft(np.fft.fft(xh.astype(np.complex128), nf) * np.fft.fft(b.astype(np.complex128), nf))[:nx + nb - 1]
        if not cplx:
            full = full.real
        for mode, nout in ((_lib.OLS_FILT, nx), (_lib.OLS_CONV, nx + nb - 1)):
            p = OlsPlan(b, d.optimalfftfiltlength(nb, nx), nx, mode, d.ENGINE_FUSED)
            en, el, ep = C.c_int64(), C.c_int64(), C.c_int()
            _lib.check(lib.mdsp_ols_plan_geometry(p._h, C.byref(en), C.byref(el), C.byref(ep)))
            y = torch.full((nout,), float("nan"), dtype=tdt, device="cuda")
            _lib.check(lib.mdsp_ols_exec(p._h, x.data_ptr(), nx, 1, nx, y.data_ptr(), nout, nout, stream))
            torch.cuda.synchronize()
            err = float(np.abs(y.cpu().numpy() - full[:nout]).max() / np.abs(full).max())
            res["check"][f"{name}_{nb}_{'filt' if mode == _lib.OLS_FILT else 'conv'}"] = {"exec_nfft": en.value, "block": el.value, "rel_err": err}
            print(name, nb, mode, en.value, el.value, "rel err", err, flush=True)
            del p, y
        del x

e0, e1 = C.c_void_p(), C.c_void_p()
_lib.check(lib.mdsp_event_create(C.byref(e0))); _lib.check(lib.mdsp_event_create(C.byref(e1)))


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        _lib.check(lib.mdsp_event_record(e0, stream)); fn(); _lib.check(lib.mdsp_event_record(e1, stream))
        torch.cuda.synchronize()
        ms = C.c_float(); _lib.check(lib.mdsp_event_elapsed_ms(e0, e1, C.byref(ms))); ts.append(ms.value)
    return sorted(ts)[len(ts) // 2]


TAPS = [int(v) for v in os.environ.get("BIGOLS_TAPS", "32768,65536,131072").split(",")]
LOG2N = [int(v) for v in os.environ.get("BIGOLS_LOG2N", "0").split(",")]
for dt, tdt, log2n in ((np.float32, torch.float32, 28), (np.float64, torch.float64, 27)):
    n = 1 << log2n
    x = torch.randn((1, n), generator=g, device="cuda", dtype=tdt)
    y = torch.empty_like(x)
    for nb in TAPS:
        taps = (np.random.default_rng(nb).standard_normal(nb) / np.sqrt(nb)).astype(dt)
        row = {}
        seg = SEG[np.dtype(dt).itemsize]
        yseg = segments(taps, x, n, seg)
        if not os.environ.get("BIGOLS_SKIP_SEGMENTS"):
            ms = timeit(lambda: segments(taps, x, n, seg), reps=3)
            row["segments"] = {"ms": round(ms, 3), "TBps": round(2 * x.element_size() * n / ms / 1e9, 4)}
        for l2 in LOG2N:
            if l2 and (1 << l2) < 2 * nb:
                continue
            _lib.set_tunable("MDSP_BIG_OLS_LOG2N", str(l2))
            p = OlsPlan(taps, d.optimalfftfiltlength(nb, n), n, _lib.OLS_FILT, d.ENGINE_FUSED)
            en = C.c_int64(); _lib.check(lib.mdsp_ols_plan_geometry(p._h, C.byref(en), None, None))
            ms = timeit(lambda: _lib.check(lib.mdsp_ols_exec(p._h, x.data_ptr(), n, 1, n, y.data_ptr(), n, n, stream)))
            row[f"big_{l2}"] = {"ms": round(ms, 3), "TBps": round(2 * x.element_size() * n / ms / 1e9, 4), "exec_nfft": en.value,
                                "maxdiff_vs_segments": float((y - yseg).abs().max())}
            del p
        _lib.set_tunable("MDSP_BIG_OLS_LOG2N", "0")
        res["bench"][f"{np.dtype(dt).name}_{nb}"] = row
        print(np.dtype(dt).name, nb, row, flush=True)
        del yseg
    del x, y
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", os.environ.get("BIGOLS_OUT", "big_ols.json")), "w"), indent=1)
