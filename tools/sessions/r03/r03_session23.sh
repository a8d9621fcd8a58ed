#!/bin/bash
# round 3, session 23: 147//160 (48 kHz -> 44.1 kHz), ten column blocks: half-size tiles that let two workgroups share a CU
#   fields: mm,wg_per_cu,p,nd,ns,ng,ch,pad,rows
V="1,0,0;1,0,0,2,2,0,1,-1,1;1,0,0,0,0,0,1,-1,1;1,0,0,2,2,0,1,-1,0;1,0,0,2,2,0,2,-1,1;1,2,0,2,2,0,1,-1,1;1,3,0,2,2,0,1,-1,1"
for c in f32:147/160 f64:147/160 c32:147/160; do
  dt=${c%%:*}; r=${c##*:}
  echo "== $dt $r"
  TUNE_DTYPE=$dt TUNE_RATIO=$r TUNE_LOG2N=26 TUNE_ROUNDS=5 TUNE_FIR="$V" timeout 200 python tools/tune_fir.py 2>&1 | grep "mm="
  cp gpurun_out/tune_fir.json gpurun_out/ng/wg2b_${dt}_${r/\//_}.json
done
