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


if __name__ == "__main__" and sys.argv[1] != "--traffic":
    if sys.argv[1] == "--pmc":
        print(json.dumps(pmc(sys.argv[2]), indent=1))
    else:
        for r in stats(sys.argv[1]):
            print(f"{r['pct']:6.2f}%  calls={r['calls']:4d}  avg={r['avg_us']:10.2f} us  total={r['total_us']:12.1f} us  {r['kernel']}")


def traffic(fetch_db, write_db, out_path):
    """HBM bytes per launch of the fused kernels from separate --pmc FETCH_SIZE / WRITE_SIZE passes.
    FETCH_SIZE (KB) is doubled: on gfx950 it tallies 128-B read requests as 64 B (MI355X_MICROARCH.md, HBM section);
    the float4 copy kernel in the same run is the calibration (reported next to its known byte count)."""
    f, w = pmc(fetch_db), pmc(write_db)

    def get(d, key, cname):
        for k, v in d.items():
            if key in k:
                return v["counters"].get(cname), k
        return None, None

    out = {"method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024",
           "workload": "bench.py default (2^30 Float32 samples per launch)"}
    for name, key in (("ols_fused", "ols_fused_kernel"), ("welch_fused", "welch_half_kernel"), ("welch_fused_generic", "welch_fused_kernel"),
                      ("stft", "stft_fused_kernel<float, 1024, 16, 4, 1, 4, true, false"), ("spectrogram", "stft_fused_kernel<float, 1024, 16, 4, 1, 4, true, true"),
                      ("resample", "polyphase_"), ("firarb", "arbitrary_fir_kernel"), ("copy", "mdsp_copy_kernel")):
        fv, fk = get(f, key, "FETCH_SIZE")
        wv, _ = get(w, key, "WRITE_SIZE")
        if fv is None or wv is None:
            continue
        out[f"{name}_kernel"] = fk
        out[f"{name}_fetch_bytes_corrected"] = int(2 * fv * 1024)
        out[f"{name}_write_bytes"] = int(wv * 1024)
        out[f"{name}_bytes_per_launch"] = int(2 * fv * 1024 + wv * 1024)
    try:
        import subprocess
        out["commit"] = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except Exception:
        out["commit"] = None
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--traffic":
    traffic(sys.argv[2], sys.argv[3], sys.argv[4])
