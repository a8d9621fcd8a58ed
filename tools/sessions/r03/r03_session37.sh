#!/bin/bash
# round 3, session 37: Float32 signals with 49 .. 64 k-steps: taps in registers with two chunks per wave (default) against fetched per tile
#   fields: mm,wg,p,nd,ns,ng,ch,pad,rows,vstore,rpad,prio,t64
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_boundary.py -x -q -m gpu -k "polyphase or matrix_core or resample" 2>&1 | tail -3
V="1,0,0,0,0,0,0,-1,-1,1,0,-1,0;1,0,0"
for c in 1/4:0 1/4:200 2/9:300 1/5:250; do
  r=${c%%:*}; t=${c##*:}
  echo "== f32 $r taps $t"
  TUNE_TAPS=$t TUNE_DTYPE=f32 TUNE_RATIO=$r TUNE_LOG2N=26 TUNE_ROUNDS=5 TUNE_FIR="$V" timeout 200 python tools/tune_fir.py 2>&1 | grep "mm=" | awk '{print "   ", $(NF-6), $(NF-5), $(NF-4), $(NF-3), $(NF-1), $NF}'
done
