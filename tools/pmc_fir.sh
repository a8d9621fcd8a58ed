#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > $O/mfma_counters.txt
export TUNE_FIR="1,0,0" TUNE_ROUNDS=2
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $O/prof_fir1 -o t -- python $R/tools/tune_fir.py > $O/prof_fir1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/prof_fir2 -o t -- python $R/tools/tune_fir.py > $O/prof_fir2.log 2>&1
python $R/tools/prof_summary.py --pmc "$(find $O/prof_fir1 -name '*.db' | head -1)" > $O/fir_pmc1.json 2>/dev/null
python $R/tools/prof_summary.py --pmc "$(find $O/prof_fir2 -name '*.db' | head -1)" > $O/fir_pmc2.json 2>/dev/null
python $R/tools/pmc_brief.py $O/fir_pmc1.json polyphase
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/fir_pmc2.json'))
for k,v in d.items():
    if 'polyphase' in k: print(k[:60], v['avg_ns'], v['counters'])
PY
cat $O/mfma_counters.txt | head; tail -3 $O/prof_fir2.log
