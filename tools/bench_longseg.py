import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import dsp_jl_amd as d
from dsp_jl_amd import _lib
_lib.check(_lib.lib().mdsp_init(0))
for dt, log2n in ((np.float32, 28), (np.float64, 27)):
    n = 1 << log2n
    x = torch.randn(n, device="cuda", dtype=torch.float32 if dt == np.float32 else torch.float64)
    for nb in (16384, 32768, 65536):
        b = (np.random.default_rng(nb).standard_normal(nb) / np.sqrt(nb)).astype(dt)
        y = d.filt(b, x); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); y = d.filt(b, x); e.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(e))
        ms = sorted(ts)[1]
        yr = d.fftfilt(b, x, d.optimalfftfiltlength(nb, n), engine=d.ENGINE_ROCFFT) if nb > (16384 if dt == np.float32 else 8192) else None
        diff = float((y - yr).abs().max() / yr.abs().max()) if yr is not None else None
        print(np.dtype(dt).name, nb, "taps: filt(b, x)", round(ms, 3), "ms =", round(2 * x.element_size() * n / ms / 1e6, 1), "GB/s algorithmic; max diff vs rocFFT engine", diff, flush=True)
        del y, yr
