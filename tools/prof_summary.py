#!/usr/bin/env python3
"""Condense rocprofv3 (rocpd sqlite) outputs under gpurun_out/ into small text/json summaries for profiles/.

    python tools/prof_summary.py gpurun_out/prof_stats/bench_results.db            # kernel-trace --stats summary
    python tools/prof_summary.py --pmc gpurun_out/prof_FETCH_SIZE/bench_results.db  # per-kernel mean counter values
"""
import json
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(<[^()]*>)?)", name)
    s = m.group(1) if m else name
    return s[:110]


def stats(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    out = []
    for n, c, tot, avg, pct in rows:
        out.append({"kernel": short(n), "calls": c, "total_us": round(tot, 1), "avg_us": round(avg, 2), "pct": round(pct, 2)})
    return out


def pmc(path):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration), max(vgpr_count), max(accum_vgpr_count), "
                      "max(sgpr_count), max(lds_block_size), max(scratch_size), max(grid_size), max(workgroup_size) "
                      "from counters_collection group by kernel_name, counter_name").fetchall()
    out = {}
    for kn, cn, v, cnt, dur, vg, ag, sg, lds, scr, grid, wg in rows:
        k = short(kn)
        e = out.setdefault(k, {"dispatches": cnt, "avg_ns": round(dur, 0), "vgpr": vg, "agpr": ag, "sgpr": sg, "lds": lds, "scratch": scr,
                               "grid": grid, "wg": wg, "counters": {}})
        e["counters"][cn] = v
    return out


if __name__ == "__main__":
    if sys.argv[1] == "--pmc":
        print(json.dumps(pmc(sys.argv[2]), indent=1))
    else:
        for r in stats(sys.argv[1]):
            print(f"{r['pct']:6.2f}%  calls={r['calls']:4d}  avg={r['avg_us']:10.2f} us  total={r['total_us']:12.1f} us  {r['kernel']}")
