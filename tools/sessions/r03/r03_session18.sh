#!/bin/bash
# round 3, session 18: config 5's shape (160//147, 5120 taps) with half-size tiles, unpadded output rows: two workgroups per CU
mkdir -p gpurun_out/ng
V="1,0,0;1,0,0,0,0,0,2;1,0,0,0,0,0,2,0;1,0,0,0,0,0,1,0;1,0,0,0,0,0,4,0"
for c in f32:160/147:28 f32:160/147:26 f32:147/160:26 f64:160/147:26; do
  IFS=: read dt r lg <<< "$c"
  echo "== $dt $r 2^$lg"
  TUNE_DTYPE=$dt TUNE_RATIO=$r TUNE_LOG2N=$lg TUNE_ROUNDS=5 TUNE_FIR="$V" timeout 200 python tools/tune_fir.py 2>&1 | grep "mm="
  cp gpurun_out/tune_fir.json gpurun_out/ng/wg2_${dt}_${r/\//_}_$lg.json
done
