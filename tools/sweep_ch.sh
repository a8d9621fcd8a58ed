#!/bin/bash
# Chunks per multiplying wave of the matrix-core polyphase kernel (the library's rule, then MDSP_FIR_MM_CH = 1, 2) over the ratios up to one, per signal type:
# median ms per shape, one line each (tools/tune_fir.py; what TUNE_PERSIST=1 would write into the choice file where a variant wins by more than 2 %).
for dt in f32 c32 f64 c64; do for r in 1/2 1/3 2/3 3/8 147/160 160/441 1/5 1/6 1/7 3/4 5/6; do
  echo -n "$dt $r : "
  TUNE_DTYPE=$dt TUNE_RATIO=$r TUNE_TAPS=0 TUNE_LOG2N=25 TUNE_ROUNDS=3 TUNE_FIR="-1,0,0;-1,0,0,0,0,0,1;-1,0,0,0,0,0,2" timeout 120 python tools/tune_fir.py 2>&1 | grep -o "[0-9.]* ms" | tr '\n' ' '; echo
done; done
