#!/bin/bash
# Round 5, session 2: kernel trace + SQ counters of the multi-pass engine on the default Welch call (2^27, 2^24, 2^20 samples).
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s2; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for L in 134217728 16777216 1048576; do
  cd /tmp
  DEFSPEC_ENGINES=auto DEFSPEC_WELCH_ONLY=1 DEFSPEC_LENGTHS=$L DEFSPEC_OUT=r05s2/defspec_$L.json timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$L -o p -- python $R/tools/bench_default_spectral.py > $R/$O/prof_$L.log 2>&1
  cd $R
  python tools/prof_summary.py $(find $O/prof_$L -name "*.db" | head -1) > $O/stats_$L.txt 2>&1; head -12 $O/stats_$L.txt | cut -c1-200
  rm -rf $O/prof_$L
done
L=134217728
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  cd /tmp
  DEFSPEC_ENGINES=auto DEFSPEC_WELCH_ONLY=1 DEFSPEC_LENGTHS=$L DEFSPEC_OUT=r05s2/tmp.json timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/$O/pmc_$n -o p -- python $R/tools/bench_default_spectral.py > $R/$O/pmc_$n.log 2>&1
  cd $R
  python tools/prof_summary.py --pmc $(find $O/pmc_$n -name "*.db" | head -1) > $O/pmc_$n.json 2>&1
  rm -rf $O/pmc_$n
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05s2/pmc_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    for k,v in d.items():
        if "big" in k: print(f.split("/")[-1], k[:70], v.get("avg_ns"), {a: round(b) for a,b in v["counters"].items()})
PY
