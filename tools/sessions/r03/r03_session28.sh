#!/bin/bash
# round 3, session 28: memory waves of the matrix-core polyphase kernel at raised priority (s_setprio 3): A/B over ratios and signal types
#   fields: mm,wg,p,nd,ns,ng,ch,pad,rows,vstore,rpad,prio
mkdir -p gpurun_out/prio
echo "== tests"
[ -n "$PRIO_SKIP_TESTS" ] || timeout 900 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_fullsize.py -x -q -m gpu -k "polyphase or matrix_core or resample or fir or config5" 2>&1 | tail -3
V="${PRIO_V:-1,0,0,0,0,0,0,-1,-1,1,0,0;1,0,0,0,0,0,0,-1,-1,1,0,1}"
for c in ${PRIO_CASES:-f32:160/147:28 f32:147/160:26 f32:2/1:26 f32:1/2:26 f32:3/2:26 f32:2/3:26 f32:4/1:26 f32:1/4:26 f32:1/8:26 f32:441/160:26 f64:160/147:26 f64:2/1:26 f64:1/2:26 c32:160/147:26 c32:2/1:26 c32:1/2:26 c32:147/160:26 c64:160/147:26 c64:2/1:26}; do
  IFS=: read dt r lg <<< "$c"
  echo "== $dt $r 2^$lg"
  TUNE_DTYPE=$dt TUNE_RATIO=$r TUNE_LOG2N=$lg TUNE_ROUNDS=7 TUNE_FIR="$V" timeout 200 python tools/tune_fir.py 2>&1 | grep "mm=" | awk '{print "   ", $(NF-6), $(NF-5), $(NF-4), $(NF-3)}'
  cp gpurun_out/tune_fir.json gpurun_out/prio/${dt}_${r/\//_}.json
done
