#!/bin/bash
# round 3, session 15: decimators on the matrix-core polyphase kernel -- tap prefetch (in the build), chunks per wave x groups per tile
mkdir -p gpurun_out/ng
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_boundary.py -x -q -m gpu -k "polyphase or matrix_core or resample or fir" 2>&1 | tail -5
V="1,0,0,0,0,0,0;1,0,0,0,0,8,2;1,0,0,0,0,8,1;1,0,0,0,0,4,2;1,0,0,0,0,4,4;1,0,0,0,0,4,1"
for c in f32:1/8 f32:1/4 f32:3/8 f32:1/3 f32:1/16 f64:1/8 f32:1/2 c32:1/8; do
  dt=${c%%:*}; r=${c##*:}
  echo "== $dt $r"
  TUNE_DTYPE=$dt TUNE_RATIO=$r TUNE_LOG2N=26 TUNE_ROUNDS=5 TUNE_FIR="$V" python tools/tune_fir.py 2>&1 | grep "mm="
  cp gpurun_out/tune_fir.json gpurun_out/ng/ch_${dt}_${r/\//_}.json
done
