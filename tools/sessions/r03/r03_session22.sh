#!/bin/bash
# round 3, session 22: does the uneven split of ten multiplying waves over four SIMDs cost config 5?  Same taps per phase (32), similar ratios,
# eight / ten / twelve column blocks: 128//117 (4096 taps), 160//147 (5120), 192//175 (6144); 4 channels x 2^28 Float32
for c in 128/117:4096 160/147:5120 192/175:6144 144/133:4608; do
  r=${c%%:*}; t=${c##*:}
  echo "== $r taps $t"
  TUNE_RATIO=$r TUNE_TAPS=$t TUNE_LOG2N=28 TUNE_ROUNDS=5 TUNE_FIR="1,0,0" timeout 200 python tools/tune_fir.py 2>&1 | grep "mm="
done
