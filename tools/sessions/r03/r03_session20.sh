#!/bin/bash
# round 3, session 20: small streaming chunks -- matrix-core polyphase kernel against the register-tap / generic kernels, one channel
for r in 2/1 160/147 1/2; do
  for lg in 8 10 12 14 16; do
    echo "== f32 $r 2^$lg"
    TUNE_NCH=1 TUNE_DTYPE=f32 TUNE_RATIO=$r TUNE_LOG2N=$lg TUNE_ROUNDS=21 TUNE_FIR="0,0,0;1,0,0" timeout 100 python tools/tune_fir.py 2>&1 | grep "mm=" | cut -c1-60
  done
done
