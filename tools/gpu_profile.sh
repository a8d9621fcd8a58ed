#!/bin/bash
# Profile protocol for the headline bench on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats            -> gpurun_out/prof_stats      (per-kernel durations)
#   2. rocprofv3 --pmc FETCH_SIZE                  -> gpurun_out/prof_FETCH_SIZE (HBM reads; x2 correction on gfx950)
#   3. rocprofv3 --pmc WRITE_SIZE                  -> gpurun_out/prof_WRITE_SIZE (HBM writes)
#   4. rocprofv3 --pmc <SQ counters>               -> gpurun_out/prof_sq         (VALU / LDS activity, bank conflicts)
# Counter passes are separate runs with --kernel-trace only (never combined with sys/runtime traces).
# Afterwards, locally:  python tools/prof_summary.py ...  and copy the summaries into profiles/.
set -u
REPO="$(pwd)"
OUT="$REPO/gpurun_out"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --no-cpu-baseline --no-live-pmc --no-host"   # the step kernels AND the section-8 rows (stft, spectrogram, resample, firarb) in one process
rm -rf "$OUT"/prof_stats "$OUT"/prof_FETCH_SIZE "$OUT"/prof_WRITE_SIZE "$OUT"/prof_sq
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_stats" -o bench -- $BENCH --steps 10 --warmup 3 > "$OUT/prof_stats.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c -d "$OUT/prof_$c" -o bench -- $BENCH --steps 3 --warmup 1 > "$OUT/prof_$c.log" 2>&1
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU \
    -d "$OUT/prof_sq" -o bench -- $BENCH --steps 3 --warmup 1 > "$OUT/prof_sq.log" 2>&1
cd "$REPO"
find "$OUT" -name "*.db" | sort
grep -h '"metric"' "$OUT/prof_stats.log" | head -1 | cut -c1-400
# condensed summaries (copy the ones to keep into profiles/)
python tools/prof_summary.py "$(find "$OUT/prof_stats" -name '*.db' | head -1)" > "$OUT/kernel_stats.txt" 2>/dev/null
python tools/prof_summary.py --pmc "$(find "$OUT/prof_sq" -name '*.db' | head -1)" > "$OUT/pmc_sq.json" 2>/dev/null
python tools/prof_summary.py --traffic "$(find "$OUT/prof_FETCH_SIZE" -name '*.db' | head -1)" "$(find "$OUT/prof_WRITE_SIZE" -name '*.db' | head -1)" "$OUT/pmc_traffic.json" > /dev/null 2>&1
python tools/pmc_brief.py "$OUT/pmc_sq.json" > "$OUT/pmc_brief.txt" 2>/dev/null
python tools/prof_summary.py --rows "$(find "$OUT/prof_sq" -name '*.db' | head -1)" > "$OUT/pmc_sq_rows.json" 2>/dev/null
head -30 "$OUT/kernel_stats.txt"; cat "$OUT/pmc_brief.txt"
# the rocpd databases stay on the box (gpurun copies back at most 64 MiB): the summaries above are what profiles/ keeps
rm -rf "$OUT"/prof_stats "$OUT"/prof_FETCH_SIZE "$OUT"/prof_WRITE_SIZE "$OUT"/prof_sq
# the un-profiled lines of the same snapshot: the driver's command, and the two 8-GPU configs at their single-GPU shares
python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench_line.err"
python bench.py --config stft > "$OUT/bench_config4.json" 2> /dev/null
python bench.py --config resample > "$OUT/bench_config5.json" 2> /dev/null
# RULE: this script is the LAST GPU session of a round -- `git rev-parse HEAD > .build_commit` before sending the snapshot, no kernel
# commit after it.  Every summary carries that hash.
python - <<'PY'
import json, os
out = os.path.join(os.getcwd(), "gpurun_out")
try:
    commit = open(".build_commit").read().strip()
except Exception:
    commit = "unknown (no .build_commit in the snapshot)"
for f in ("pmc_sq.json", "pmc_traffic.json", "pmc_sq_rows.json"):
    p = os.path.join(out, f)
    try:
        d = json.load(open(p))
        if isinstance(d, dict):
            d["commit"] = commit
            json.dump(d, open(p, "w"), indent=1)
    except Exception as e:
        print("no", f, e)
for f in ("kernel_stats.txt", "pmc_brief.txt"):
    p = os.path.join(out, f)
    try:
        t = open(p).read()
        open(p, "w").write(f"# commit {commit}; tools/gpu_profile.sh\n" + t)
    except Exception as e:
        print("no", f, e)
for f in ("bench_line.json", "bench_config4.json", "bench_config5.json"):
    try:
        d = json.loads(open(os.path.join(out, f)).read().strip().splitlines()[-1])
        print(f, d["value"], d["unit"], d.get("ms_per_step"), d.get("commit"))
    except Exception as e:
        print("no", f, e)
PY
