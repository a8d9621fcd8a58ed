#!/bin/bash
# Round 4, GPU session 32: the SQ counters of the Welch kernels, old and new, in one session: welch_half3_kernel (variant 30),
# mdsp_welch_w64_asm (42: re-reads the shared half-frame), mdsp_welch_w64c_asm (43, default) -- headline step only.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/s32; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
for v in 30 42 43; do
  cd /tmp
  MDSP_WELCH_VARIANT=$v timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU \
      -d $R/$OUT/prof_$v -o b -- python $R/bench.py --no-rows --no-cpu-baseline --no-live-pmc --no-host --steps 5 --warmup 2 > /dev/null 2>&1
  cd $R
  python tools/prof_summary.py --pmc "$(find $OUT/prof_$v -name '*.db' | head -1)" > $OUT/pmc_$v.json 2>/dev/null
  echo "== MDSP_WELCH_VARIANT=$v"; python tools/pmc_brief.py $OUT/pmc_$v.json welch
  rm -rf $OUT/prof_$v
done
