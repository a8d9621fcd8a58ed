#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
echo "== tests of the new variants"; timeout 900 python -m pytest tests/test_gpu_boundary.py -m gpu -x -q -k "round3 or lds_dma" 2>&1 | tail -15 > $OUT/s5_gpu_tests.log; tail -4 $OUT/s5_gpu_tests.log
echo "== tune"; TUNE_LOG2N=30 TUNE_ROUNDS=10 TUNE_OLS=0 TUNE_WELCH=0,30,32,35,36 TUNE_WGS=2 TUNE_RUNS=1 timeout 900 python tools/tune.py > $OUT/s5_tune.log 2>&1; cp $OUT/tune.json $OUT/s5_tune.json; grep -E "^(ols|welch|copy)" $OUT/s5_tune.log
