#!/bin/bash
# round 3, session 38: taps in registers with fewer chunks per wave (default) against round 2's limits (MDSP_FIR_MM_T64=0), one process per shape
#   fields: mm,wg,p,nd,ns,ng,ch,pad,rows,vstore,rpad,prio,t64
mkdir -p gpurun_out/regtaps
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_boundary.py -x -q -m gpu -k "polyphase or matrix_core or resample" 2>&1 | tail -3
V="1,0,0,0,0,0,0,-1,-1,1,0,-1,0;1,0,0"
for c in f32:1/8 f32:1/4 f32:1/6 f64:3/8 f64:1/3 c64:1/2 c32:1/6 f32:3/16 c32:1/5 f64:2/5; do
  dt=${c%%:*}; r=${c##*:}
  echo "== $dt $r"
  TUNE_DTYPE=$dt TUNE_RATIO=$r TUNE_LOG2N=26 TUNE_ROUNDS=5 TUNE_FIR="$V" timeout 200 python tools/tune_fir.py 2>&1 | grep "mm=" | awk '{print "   ", $(NF-6), $(NF-5), $(NF-4), $(NF-3), $(NF-1), $NF}'
  cp gpurun_out/tune_fir.json gpurun_out/regtaps/${dt}_${r/\//_}.json
done
