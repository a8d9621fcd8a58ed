#!/bin/bash
# round 3, session 24: gathered wide stores for output rows that are not whole vectors (147//160, 441//160, ...): A/B against the element stores
mkdir -p gpurun_out/ng
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_boundary.py -x -q -m gpu -k "polyphase or matrix_core or resample" 2>&1 | tail -3
V="1,0,0,0,0,0,0,-1,-1,0;1,0,0"
for c in f32:147/160 f64:147/160 c32:147/160 f64:441/160 f32:147/80 f64:21/20 f32:49/48; do
  dt=${c%%:*}; r=${c##*:}
  echo "== $dt $r"
  TUNE_DTYPE=$dt TUNE_RATIO=$r TUNE_LOG2N=26 TUNE_ROUNDS=5 TUNE_FIR="$V" timeout 200 python tools/tune_fir.py 2>&1 | grep "mm="
  cp gpurun_out/tune_fir.json gpurun_out/ng/vstore_${dt}_${r/\//_}.json
done
