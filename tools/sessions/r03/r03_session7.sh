#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
echo "== mixed-radix tests"; timeout 900 python -m pytest tests/test_gpu_boundary.py -m gpu -x -q -k "mixed_radix" 2>&1 | tail -15 > $OUT/s7_gpu_tests.log; tail -4 $OUT/s7_gpu_tests.log
echo "== mixed bench, all compile-time sizes"; MIXED_SIZES=1000,1200,1280,1500,1536,1600,1920,2000,2400,2500,2560,3000,3072,3200,3840,4000,4800,5000,5120,6000,6144,6400,8000 MIXED_LOG2N=27 timeout 900 python tools/bench_mixed.py > $OUT/s7_mixed.log 2>&1; cp $OUT/mixed.json $OUT/s7_mixed_ct.json; tail -14 $OUT/s7_mixed.log
