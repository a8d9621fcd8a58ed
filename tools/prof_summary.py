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


# bench.py launches a marker before every measured row: mdsp_fill_kernel with (MARK_BASE + row index) workgroups.  Dispatches between two markers
# belong to the row of the first, so a row's counters never mix with another row's dispatches of the same template instantiation.
MARK_KERNEL, MARK_BASE = "mdsp_fill_kernel", 100
ROWS = ("step", "yardsticks", "stft", "spectrogram", "resample", "firarb", "resample_f64", "resample_c32", "interp2_f32", "decim2_f32",
        "welch_3000", "welch_1536", "filt_5120", "decim8_f32", "resample_147_160_f32", "resample_160_441_f64",
        "decim16_f32", "resample_441_160_c64", "welch_default", "welch_default_2p24", "spectrogram_default", "filt_32768", "filt_f64", "welch_f64",
        "welch_f64_5000", "welch_f64_8000", "mt_pgram", "hilbert", "conv2d", "filtfilt", "welch_2p19",
        "welch_8192", "welch_12500", "welch_16384", "welch_65536", "welch_125000", "welch_200000", "spectrogram_8400")   # == bench.py Marks.ROWS

def rows(path, min_ns=20000):
    """Per bench row: the kernels dispatched in that row's marker segment with their mean counter values -> {row: {kernel: {...}}}.
    Databases of commands without markers come back as the single row "unmarked"."""
    db = sqlite3.connect(path)
    q = db.execute("select dispatch_id, kernel_name, counter_name, value, duration, grid_size, workgroup_size, lds_block_size, vgpr_count, accum_vgpr_count, start "
                   "from counters_collection order by start, dispatch_id").fetchall()
    out, row = {}, "unmarked"
    for did, kn, cn, v, dur, grid, wg, lds, vg, ag, _ in q:
        if MARK_KERNEL in kn:
            i = grid // max(1, wg) - MARK_BASE
            row = ROWS[i] if 0 <= i < len(ROWS) else f"row{i}"
            continue
        if dur < min_ns:
            continue
        e = out.setdefault(row, {}).setdefault(short(kn), {"dispatches": set(), "ns": {}, "vgpr": vg, "agpr": ag, "lds": lds, "grid": grid, "wg": wg, "sum": {}})
        e["dispatches"].add(did)
        e["ns"][did] = dur
        e["sum"][cn] = e["sum"].get(cn, 0.0) + v
    for r in out.values():
        for e in r.values():
            n = len(e["dispatches"])
            e["counters"] = {c: t / n for c, t in e.pop("sum").items()}
            e["avg_ns"] = round(sum(e.pop("ns").values()) / n, 0)
            e["dispatches"] = n
    return out


def dominant(rowd, pat=""):
    """The kernel of a row with the largest time share (optionally restricted to names containing `pat`)."""
    c = [(e["avg_ns"] * e["dispatches"], k) for k, e in rowd.items() if pat in k]
    return max(c)[1] if c else None


ROW_KERNELS = (("ols_fused", "step", "ols_fused_kernel"), ("welch_fused", "step", "welch_"), ("copy", "yardsticks", "mdsp_copy_kernel"),
               ("stft", "stft", "stft_"), ("spectrogram", "spectrogram", "stft_"), ("resample", "resample", "polyphase_"), ("firarb", "firarb", "arbitrary_fir_kernel"),
               ("resample_f64", "resample_f64", "polyphase_"), ("resample_c32", "resample_c32", "polyphase_"), ("interp2_f32", "interp2_f32", "polyphase_"),
               ("decim2_f32", "decim2_f32", ""), ("welch_3000", "welch_3000", "gen_"), ("welch_1536", "welch_1536", "gen_"),
               ("filt_5120", "filt_5120", "upols"), ("decim8_f32", "decim8_f32", ""), ("resample_147_160_f32", "resample_147_160_f32", "polyphase_"),
               ("resample_160_441_f64", "resample_160_441_f64", "polyphase_"),
               # round 5 (pattern "": the row's dominant kernel whatever its name -- several rows run more than one kernel: the figure is that kernel's alone)
               ("decim16_f32", "decim16_f32", ""), ("resample_441_160_c64", "resample_441_160_c64", ""), ("filt_f64", "filt_f64", "ols_"), ("welch_f64", "welch_f64", "welch_"),
               ("welch_f64_5000", "welch_f64_5000", "gen_"), ("welch_f64_8000", "welch_f64_8000", "gen_"), ("hilbert", "hilbert", ""))

def traffic_rows(fetch_db, write_db):
    """HBM bytes per launch of each row's dominant kernel from separate --pmc FETCH_SIZE / WRITE_SIZE passes of a marker-carrying bench.py run
    (FETCH_SIZE doubled: gfx950 correction, calibrated on the float4 copy of the same run)."""
    f, w = rows(fetch_db), rows(write_db)
    out = {}
    for name, row, pat in ROW_KERNELS:
        k = None
        for cand in (row, "step", "unmarked"):      # --config stft / resample runs carry their kernel in the "step" row; old databases have no markers
            fr, wr = f.get(cand) or {}, w.get(cand) or {}
            k = dominant(fr, pat)
            if k is not None and k in wr:
                break
            k = None
        if k is None:
            continue
        fv, wv = fr[k]["counters"].get("FETCH_SIZE"), wr[k]["counters"].get("WRITE_SIZE")
        if fv is None or wv is None:
            continue
        out[f"{name}_kernel"] = k
        out[f"{name}_fetch_bytes_corrected"] = int(2 * fv * 1024)
        out[f"{name}_write_bytes"] = int(wv * 1024)
        out[f"{name}_bytes_per_launch"] = int(2 * fv * 1024 + wv * 1024)
    return out


if __name__ == "__main__" and sys.argv[1] == "--rows":
    r = rows(sys.argv[2])
    print(json.dumps(r, indent=1))
    sys.exit(0)

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
    out = {"method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes), bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024; rows separated by "
                     "bench.py's marker dispatches (tools/prof_summary.py rows())",
           "workload": "bench.py default (2^30 Float32 samples per launch; rows at their section-8 sizes)"}
    out.update(traffic_rows(fetch_db, write_db))
    try:
        import subprocess
        out["commit"] = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except Exception:
        out["commit"] = None
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--traffic":
    traffic(sys.argv[2], sys.argv[3], sys.argv[4])
