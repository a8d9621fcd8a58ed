#!/bin/bash
# rocprofv3 kernel stats + one SQ counter pass of an arbitrary command, condensed by tools/prof_summary.py / pmc_brief.py into $O (gpurun_out/NAME).
#   bash tools/prof_cmd.sh NAME 'GX_CHECK=0 python tools/bench_gx.py'      (run from the repository root on the GPU box)
set -u
REPO="$(pwd)"; NAME=${1:?name}; shift; CMD="$*"
O="$REPO/gpurun_out/$NAME"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_st /tmp/prof_sq /tmp/prof_sq2
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_st -o p -- bash -c "cd $REPO && $CMD" > "$O/stats.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU \
    -d /tmp/prof_sq -o p -- bash -c "cd $REPO && $CMD" > "$O/sq.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES \
    -d /tmp/prof_sq2 -o p -- bash -c "cd $REPO && $CMD" > "$O/sq2.log" 2>&1
cd "$REPO"
python tools/prof_summary.py "$(find /tmp/prof_st -name '*.db' | head -1)" > "$O/kernel_stats.txt" 2>/dev/null
python tools/prof_summary.py --pmc "$(find /tmp/prof_sq -name '*.db' | head -1)" > "$O/pmc_sq.json" 2>/dev/null
python tools/prof_summary.py --pmc "$(find /tmp/prof_sq2 -name '*.db' | head -1)" > "$O/pmc_sq2.json" 2>/dev/null
python tools/pmc_brief.py "$O/pmc_sq.json" > "$O/pmc_brief.txt" 2>/dev/null
head -12 "$O/kernel_stats.txt"; cat "$O/pmc_brief.txt"
python - "$O/pmc_sq2.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if not isinstance(v, dict) or v.get("avg_ns", 0) < 20000: continue
    c = v["counters"]; clk = c.get("SQ_BUSY_CYCLES", 0) / 32.0
    if not clk: continue
    print(k[:60], {n: round(x / clk, 1) for n, x in c.items() if n != "SQ_BUSY_CYCLES"}, "scratch", v.get("scratch"), "sgpr", v.get("sgpr"))
PY
