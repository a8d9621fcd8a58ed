#!/usr/bin/env python3
"""Minimal driver for profiling the mixed-radix kernel: Welch + complex STFT at one nfft (argv[1], default 1000), a few launches each."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import dsp_jl_amd as d
from dsp_jl_amd import _lib
from dsp_jl_amd.periodograms import _StftPlan, compute_window
lib = _lib.lib(); _lib.check(lib.mdsp_init(0))
st = torch.cuda.current_stream().cuda_stream
nfft = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = 1 << 26
x = torch.randn(n, device="cuda"); xc = torch.complex(x[: n // 2].clone(), x[n // 2:].clone())
cfg = d.WelchConfig(n, np.float32, n=nfft, noverlap=nfft // 2, nfft=nfft, window=d.hanning, engine=d.ENGINE_FUSED)
psd = torch.empty(cfg.nout, device="cuda")
win, norm2 = compute_window(d.hanning, nfft)
m = n // 2; hop = nfft // 4; K = d.frame_count(m, nfft, nfft - hop)
plan = _StftPlan(nfft, nfft - hop, nfft, win, norm2, False, 0, np.complex64, d.ENGINE_FUSED)
out = torch.empty((K, nfft), dtype=torch.complex64, device="cuda")
for _ in range(3):
    _lib.check(lib.mdsp_welch_exec(cfg._h, x.data_ptr(), n, 1, n, psd.data_ptr(), cfg.nout, st))
    _lib.check(lib.mdsp_stft_exec(plan._h, xc.data_ptr(), m, 1, m, out.data_ptr(), nfft, K * nfft, st))
torch.cuda.synchronize()
