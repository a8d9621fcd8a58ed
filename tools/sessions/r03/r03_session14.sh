#!/bin/bash
# round 3, session 14: memory-wave split of the matrix-core polyphase kernel at four 64-row groups per tile (interpolators)
mkdir -p gpurun_out/ng
V="1,0,0,0,0,4;1,0,0,1,2,4;1,0,0,1,3,4;1,0,0,2,3,4;1,0,0,2,4,4;1,0,0,1,4,4;1,0,0,2,6,4;1,0,0,2,2,8;1,0,0,2,4,8;1,0,0,2,6,8"
for c in f32:2/1 f32:4/1 c32:2/1 f32:1/1; do
  dt=${c%%:*}; r=${c##*:}
  echo "== $dt $r"
  TUNE_DTYPE=$dt TUNE_RATIO=$r TUNE_LOG2N=26 TUNE_ROUNDS=5 TUNE_FIR="$V" python tools/tune_fir.py 2>&1 | grep "mm="
  cp gpurun_out/tune_fir.json gpurun_out/ng/ns_${dt}_${r/\//_}.json
done
