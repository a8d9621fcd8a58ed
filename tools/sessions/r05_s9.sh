#!/bin/bash
# Round 5, session 9: ablation of the decimator kernel (1//16 and 1//4, Float32 and ComplexF64) + SQ counters.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s9; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
FIRR_DTYPES=f32,c64 FIRR_RATIOS=1/4,1/16 FIRR_VARIANTS="default;MDSP_FIR_DEC_ABLATE=1;MDSP_FIR_DEC_ABLATE=2;MDSP_FIR_DEC_ABLATE=4;MDSP_FIR_DEC_ABLATE=3;MDSP_FIR_DEC_ABLATE=6;MDSP_FIR_DEC_ABLATE=5" FIRR_OUT=r05s9/fir_dec_ablate.json timeout 900 python tools/bench_fir_ratios.py 2>&1 | grep -v amdgpu.ids | cut -c1-400
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"; do
  n=$(echo $c | cut -d' ' -f1)
  cd /tmp
  FIRR_DTYPES=f32 FIRR_RATIOS=1/16 FIRR_ROUNDS=1 FIRR_OUT=r05s9/tmp.json timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/$O/pmc_$n -o p -- python $R/tools/bench_fir_ratios.py > $R/$O/pmc_$n.log 2>&1
  cd $R
  python tools/prof_summary.py --pmc $(find $O/pmc_$n -name "*.db" | head -1) > $O/pmc_$n.json 2>&1
  rm -rf $O/pmc_$n
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05s9/pmc_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    for k,v in d.items():
        if "decimator" in k: print(k[:60], v.get("avg_ns"), "vgpr", v.get("vgpr"), "lds", v.get("lds"), "grid", v.get("grid"), {a: round(b) for a,b in v["counters"].items()})
PY
