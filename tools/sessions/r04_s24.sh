#!/bin/bash
# Round 4, GPU session 24: kernel trace of the whole bench (step + rows): every launch a row makes, to find auxiliary kernels that cost more than they should.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/s24; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o rows -- python $R/bench.py --no-cpu-baseline --no-live-pmc --no-host --steps 3 --warmup 1 > $R/$OUT/bench.log 2>&1
cd $R
python tools/prof_summary.py $(find $OUT/prof -name "*.db" | head -1) | grep -v "at::native" | head -60 | cut -c1-190
python tools/prof_summary.py --rows $(find $OUT/prof -name "*.db" | head -1) > $OUT/rows.json 2>/dev/null
rm -rf $OUT/prof
