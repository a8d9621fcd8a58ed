#!/bin/bash
# Round 5, session 4: ablation of the pass kernel's phases (MDSP_BIG_ABLATE) + SQ counters, default Welch at 2^27.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s4; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
L=134217728
for ab in 0 1 2 4 8 3 6 7 15; do
  echo -n "ablate=$ab: "; MDSP_BIG_ABLATE=$ab DEFSPEC_ENGINES=auto DEFSPEC_WELCH_ONLY=1 DEFSPEC_LENGTHS=$L DEFSPEC_OUT=r05s4/ab$ab.json timeout 300 python tools/bench_default_spectral.py 2>&1 | grep -v amdgpu.ids | cut -c1-300
done
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  n=$(echo $c | cut -d' ' -f1)
  cd /tmp
  DEFSPEC_ENGINES=auto DEFSPEC_WELCH_ONLY=1 DEFSPEC_LENGTHS=$L DEFSPEC_OUT=r05s4/tmp.json timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/$O/pmc_$n -o p -- python $R/tools/bench_default_spectral.py > $R/$O/pmc_$n.log 2>&1
  cd $R
  python tools/prof_summary.py --pmc $(find $O/pmc_$n -name "*.db" | head -1) > $O/pmc_$n.json 2>&1
  rm -rf $O/pmc_$n
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05s4/pmc_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    for k,v in d.items():
        if "big_pass" in k: print(k[:60], v.get("avg_ns"), "vgpr", v.get("vgpr"), "lds", v.get("lds"), {a: round(b) for a,b in v["counters"].items()})
PY
