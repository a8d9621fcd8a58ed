# debug aid for the hand-allocated Welch kernel: one unit, error pattern by (lane, kt)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import dsp_jl_amd as d
from dsp_jl_amd import _lib
from oracle import periodograms as opg, windows as ow
_lib.check(_lib.lib().mdsp_init(0))
rng = np.random.default_rng(5)
for length in (6144, 6144, 8192 + 2048, 4096 * 9):
    s = rng.standard_normal(length).astype(np.float32)
    _lib.set_tunable("MDSP_WELCH_VARIANT", "42")
    cfg = d.WelchConfig(length, np.float32, n=4096, noverlap=2048, window=d.hanning, engine=d.ENGINE_FUSED)
    outs = [np.asarray(d.welch_pgram(s, cfg).power, dtype=np.float64) for _ in range(4)]
    ref = opg.welch_pgram(s, 4096, 2048, window=ow.hanning, dtype=np.float64).power
    print("length", length, "relerr", [float(np.linalg.norm(o - ref) / np.linalg.norm(ref)) for o in outs], "runs equal", [bool(np.array_equal(outs[0], o)) for o in outs[1:]])
    o = outs[0]
    bad = np.abs(o - ref) > 1e-4 * ref.max()
    k = np.flatnonzero(bad)
    print("  bad bins", len(k), "of", len(ref), "lanes", sorted(set((k % 64).tolist()))[:70], "kts", sorted(set((k // 64).tolist()))[:40])
    diff = np.flatnonzero(outs[0] != outs[1])
    print("  bins differing between runs", len(diff), sorted(set((diff % 64).tolist()))[:70], sorted(set((diff // 64).tolist()))[:40])
