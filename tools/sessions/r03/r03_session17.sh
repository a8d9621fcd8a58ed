#!/bin/bash
# round 3, session 17: A/B of the tile choice (round-2 rule: MDSP_FIR_MM_NG=8 against the round-3 default), decimating ratios, every signal type
O=gpurun_out/r03h_ab; mkdir -p $O
for dt in f32 f64 c32 c64; do
  for r in 1/2 2/3 1/3 1/4 3/8 1/8 1/16 147/160; do
    echo "== $dt $r"
    TUNE_DTYPE=$dt TUNE_RATIO=$r TUNE_LOG2N=26 TUNE_ROUNDS=5 TUNE_FIR="1,0,0,0,0,8;1,0,0" timeout 120 python tools/tune_fir.py 2>&1 | grep "mm="
    cp gpurun_out/tune_fir.json $O/${dt}_${r/\//_}.json
  done
done
