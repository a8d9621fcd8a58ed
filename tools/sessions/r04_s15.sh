#!/bin/bash
# Round 4, GPU session 15: SQ counters of the FIRArbitrary kernel after the record / prologue rework (LDS-array busy, bank conflicts, VALU busy).
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/s15; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
R=/root/repo
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $R/$OUT/prof_sq -o t -- python $R/tools/bench_firarb.py > $R/$OUT/prof_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU -d $R/$OUT/prof_sq2 -o t -- python $R/tools/bench_firarb.py > $R/$OUT/prof_sq2.log 2>&1
cd $R
python tools/prof_summary.py --pmc "$(find $OUT/prof_sq -name '*.db' | head -1)" > $OUT/pmc1.json 2>/dev/null
python tools/prof_summary.py --pmc "$(find $OUT/prof_sq2 -name '*.db' | head -1)" > $OUT/pmc2.json 2>/dev/null
python tools/pmc_brief.py $OUT/pmc1.json arbitrary
python - <<'PY'
import json
for f in ("gpurun_out/s15/pmc1.json", "gpurun_out/s15/pmc2.json"):
    d = json.load(open(f))
    for k, v in d.items():
        if "arbitrary" in k:
            print(f, k[:60], json.dumps(v))
PY
rm -rf $OUT/prof_sq $OUT/prof_sq2
