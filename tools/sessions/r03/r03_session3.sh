#!/bin/bash
# Round-3 GPU session 3: deeper prefetch -- Welch with two units in flight (variants 33 / 34), overlap-save with LDS-DMA staging (36 / 37).
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
echo "== tests of the new variants"; timeout 900 python -m pytest tests/test_gpu_boundary.py -m gpu -x -q -k "round3 or lds_dma or plan_cache" 2>&1 | tail -15 > $OUT/s3_gpu_tests.log; tail -4 $OUT/s3_gpu_tests.log
echo "== tune"; TUNE_LOG2N=30 TUNE_ROUNDS=10 TUNE_OLS=0,36,37 TUNE_WELCH=0,30,32,33,34 TUNE_WGS=2 TUNE_RUNS=1 timeout 900 python tools/tune.py > $OUT/s3_tune.log 2>&1; cp $OUT/tune.json $OUT/s3_tune.json; grep -E "^(ols|welch|copy)" $OUT/s3_tune.log
echo "== welch phase profile, two units in flight"; MDSP_LIB_TAG=prof TUNE_LOG2N=30 TUNE_ROUNDS=2 TUNE_OLS= TUNE_WELCH=30,33 TUNE_WGS=2 TUNE_RUNS=1 timeout 300 python tools/tune.py > $OUT/s3_prof.log 2>&1; grep WELCHPROF $OUT/s3_prof.log | tail -6
