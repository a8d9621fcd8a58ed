#!/bin/bash
# round 3, session 26: where the padded-run form loses against the row-staged one at 147//160 (debug-knob build, MDSP_ABLATE bits)
export MDSP_LIB_TAG=dbg
for ab in 6 14 22; do
  echo "ablate=$ab"
  MDSP_ABLATE=$ab TUNE_DTYPE=f32 TUNE_RATIO=147/160 TUNE_LOG2N=26 TUNE_ROUNDS=5 TUNE_FIR="1,0,0,0,0,0,0,-1,1;1,0,0,0,0,0,0,-1,2" timeout 200 python tools/tune_fir.py 2>&1 | grep "mm=" | awk '{print "   ", $10, $11, $12}'
done
