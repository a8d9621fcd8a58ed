#!/bin/bash
# Round 4, GPU session 21: kernel trace of the headline step only (which launches make up the Welch stage since the hand-allocated kernel).
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/s21; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o step -- python $R/bench.py --no-rows --no-cpu-baseline --no-live-pmc --no-host --steps 10 --warmup 3 > $R/$OUT/bench.log 2>&1
cd $R

python tools/prof_summary.py $(find $OUT/prof -name "*.db" | head -1) | grep -i "welch\|w64\|reduce\|ols_fused\|copy" | cut -c1-160
tail -2 $OUT/bench.log | cut -c1-600
rm -rf $OUT/prof
