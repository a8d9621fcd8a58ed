#!/bin/bash
# round 3, session 34: branch-free tap fetches in the long-filter mode of the matrix-core polyphase kernel (taps fetched per tile)
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_boundary.py -x -q -m gpu -k "polyphase or matrix_core or resample or fir" 2>&1 | tail -3
for c in f32:1/8 c32:1/8 f64:1/8 c64:1/8 f32:1/16 c32:1/16 f64:1/16 f32:3/8 c32:3/8 f64:3/8 c64:3/8 f32:1/4 c32:1/4 f64:441/160 f32:160/441 c64:1/4; do
  dt=${c%%:*}; r=${c##*:}
  echo -n "$dt $r  "
  TUNE_DTYPE=$dt TUNE_RATIO=$r TUNE_LOG2N=26 TUNE_ROUNDS=5 TUNE_FIR="1,0,0" timeout 200 python tools/tune_fir.py 2>&1 | grep "mm=" | awk '{print $4, $5, $6, $7}'
done
