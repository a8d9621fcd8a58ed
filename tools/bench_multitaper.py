import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import dsp_jl_amd as d
n = 2**26
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randn(n, generator=g, device="cuda", dtype=torch.float32)
for spw, nov in ((1024, 512), (4096, 2048), (256, 128)):
    mc = d.MTConfig(np.float32, spw, fs=1.0, nw=4)
    cfg = d.MTSpectrogramConfig(n, mc, nov)
    d.mt_spectrogram(x, cfg); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        r = d.mt_spectrogram(x, cfg)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
    sp = d.spectrogram(x, spw, nov, window=d.hanning); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        sp = d.spectrogram(x, spw, nov, window=d.hanning)
    torch.cuda.synchronize(); ds = (time.perf_counter() - t) / 3
    print(f"n={spw} ntapers={mc.ntapers}: mt_spectrogram {dt*1e3:.2f} ms ({n/dt/1e9:.1f} Gsamples/s), plain spectrogram {ds*1e3:.2f} ms -> ratio {dt/ds:.1f}")
