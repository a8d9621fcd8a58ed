#!/usr/bin/env python3
"""One line per kernel from a prof_summary.py --pmc json: duration, clock, LDS-array busy fraction, share of LDS cycles that are bank
conflicts, VALU busy fraction, waves per SIMD.   python tools/pmc_brief.py gpurun_out/x_pmc.json [name-filter]"""
import json
import sys

d = json.load(open(sys.argv[1]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k, v in d.items():
    if not isinstance(v, dict) or "counters" not in v:   # e.g. the "commit" entry tools/gpu_profile.sh adds
        continue
    c = v["counters"]
    if flt not in k or "SQ_BUSY_CYCLES" not in c or v["avg_ns"] < 20000:
        continue
    clk = c["SQ_BUSY_CYCLES"] / 32.0                       # per-CU cycles of the launch (32 SQ instances report)
    ghz = clk / v["avg_ns"]
    lds = c.get("SQ_LDS_IDX_ACTIVE", 0) / 256.0 / clk
    conf = c.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, c.get("SQ_LDS_IDX_ACTIVE", 1))
    valu = c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (clk * 1024)
    waves = c.get("SQ_WAVE_CYCLES", 0) * 4 / (clk * 1024)
    print(f"{k[:70]:70s} {v['avg_ns']/1e6:8.3f} ms  {ghz:4.2f} GHz  LDS busy {lds:5.1%} (conflicts {conf:5.1%})  VALU busy {valu:5.1%}  waves/SIMD {waves:4.1f}  vgpr {v['vgpr']} lds {v['lds']}")
