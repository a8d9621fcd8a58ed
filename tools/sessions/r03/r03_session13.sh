#!/bin/bash
# round 3, session 13: tile size (64-row groups per tile -> workgroups per CU) of the matrix-core polyphase kernel for the small ratios
mkdir -p gpurun_out/ng
for c in f32:2/1 f32:1/2 f32:4/1 f32:1/4 f32:3/2 f64:2/1 c32:2/1; do
  dt=${c%%:*}; r=${c##*:}
  echo "== $dt $r"
  TUNE_DTYPE=$dt TUNE_RATIO=$r TUNE_LOG2N=26 TUNE_ROUNDS=5 TUNE_FIR="1,0,0,0,0,0;1,0,0,0,0,6;1,0,0,0,0,4;1,0,0,0,0,3;1,0,0,0,0,2;1,0,0,0,0,1" python tools/tune_fir.py 2>&1 | grep "mm="
  cp gpurun_out/tune_fir.json gpurun_out/ng/${dt}_${r/\//_}.json
done
