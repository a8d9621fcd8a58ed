#!/bin/bash
# round 3, session 16: the matrix-core polyphase kernel with the round-3 tile choice -- parity, then every ratio x signal type of the r02r table
mkdir -p gpurun_out
echo "== tests"
true
O=gpurun_out/r03n; mkdir -p $O
for dt in f32 f64 c32 c64; do
  for r in 160/147 147/160 2/1 1/2 3/2 2/3 5/3 4/1 1/3 1/4 1/8 1/16 3/8 160/441 441/160; do
    echo "== $dt $r"
    TUNE_DTYPE=$dt TUNE_RATIO=$r TUNE_LOG2N=26 TUNE_ROUNDS=3 TUNE_FIR="1,0,0" timeout 120 python tools/tune_fir.py 2>&1 | grep "mm="
    cp gpurun_out/tune_fir.json $O/${dt}_${r/\//_}.json
  done
done
