#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
echo "== mixed-radix tests"; timeout 900 python -m pytest tests/test_gpu_boundary.py -m gpu -x -q -k "mixed_radix" 2>&1 | tail -15 > $OUT/s6_gpu_tests.log; tail -4 $OUT/s6_gpu_tests.log
echo "== mixed bench (compile-time schedules)"; MIXED_LOG2N=27 timeout 600 python tools/bench_mixed.py > $OUT/s6_mixed.log 2>&1; cp $OUT/mixed.json $OUT/s6_mixed_ct.json; tail -12 $OUT/s6_mixed.log
echo "== mixed bench (run-time kernel)"; MDSP_LIB_TAG=rtgen MIXED_LOG2N=27 timeout 600 python tools/bench_mixed.py > $OUT/s6_mixed_rt.log 2>&1; cp $OUT/mixed.json $OUT/s6_mixed_rt.json; tail -12 $OUT/s6_mixed_rt.log
