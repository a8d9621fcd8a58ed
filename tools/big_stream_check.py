import os, sys, math
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import dsp_jl_amd as d
from oracle import dspbase as odsp, periodograms as opg, windows as ow
from fractions import Fraction
n = 2**32 + 12345
g = torch.Generator(device="cuda"); g.manual_seed(7)
x = torch.empty(n, device="cuda", dtype=torch.float32)
step = 2**28
for i in range(0, n, step):
    x[i:i+step] = torch.randn(min(step, n-i), generator=g, device="cuda")
b = d.design.lowpass_firwindow(0.25, d.hamming(256), fs=1.0).astype(np.float32)
y = d.fftfilt(b, x, 2048)
assert y.shape == (n,)
L = 2048 - 255
for s in (0, 2**31 - 300, 2**31 + 77, 2**32 - 500, n - 700):
    lo = max(0, s - 255)
    xs = x[lo:s+600].cpu().numpy().astype(np.float64)
    ref = odsp.filt_ba(b.astype(np.float64), 1.0, xs)[s-lo:]
    got = y[s:s+600].cpu().numpy()
    err = np.linalg.norm(got[:len(ref)] - ref[:len(got)]) / np.linalg.norm(ref[:len(got)])
    print("filt spot", s, err); assert err < 5e-6
del y
p = d.welch_pgram(x, 4096, 2048, window=d.hanning).power
K = (n - 4096)//2048 + 1
print("welch K", K, "mean level", float(p[5:2040].mean()))
assert abs(float(p[5:2040].mean())/2 - 1) < 5e-3
# tail frames only: Welch of the last 2^20 samples region equals oracle
seg = x[n - 2**20:]
ps = d.welch_pgram(seg, 4096, 2048, window=d.hanning).power.cpu().numpy()
ro = opg.welch_pgram(seg.cpu().numpy().astype(np.float64), 4096, 2048, window=ow.hanning).power
print("welch tail seg err", np.linalg.norm(ps-ro)/np.linalg.norm(ro))
# resample a > 2^31 channel
h = d.resample_filter(Fraction(3, 2)).astype(np.float32)
m = 2**31 + 1001
z = d.resample(x[:m], Fraction(3, 2), h)
print("resample len", z.shape, math.ceil(m * 1.5)); assert z.shape[0] == math.ceil(m*Fraction(3,2))
from oracle import stream_filt as osf
tail = 3000
# compare the END region: oracle on the last chunk of input (enough history)
print("ok")
