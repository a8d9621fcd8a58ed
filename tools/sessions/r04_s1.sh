#!/bin/bash
# Round 4, GPU session 1: the one-wavefront-per-transform Welch kernel (variant 40) -- parity test, then interleaved timing against the round-3
# default (30), on the product build and on the -DMDSP_PK_NATIVE=1 build (compiler-native packed forms where it can express them).
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/s1; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_boundary.py -q -x -k "welch_round3_kernel and 40" > $OUT/pytest_v40.log 2>&1; tail -5 $OUT/pytest_v40.log
for tag in "" native; do
  MDSP_LIB_TAG=$tag TUNE_LOG2N=30 TUNE_ROUNDS=8 TUNE_OLS=0 TUNE_WELCH=30,40,18 TUNE_WGS=2 TUNE_RUNS=1 TUNE_ZERO=1 timeout 600 python tools/tune.py > $OUT/tune_${tag:-product}.log 2>&1
  mv gpurun_out/tune.json $OUT/tune_${tag:-product}.json
  tail -12 $OUT/tune_${tag:-product}.log
done
MDSP_LIB_TAG=native timeout 600 python -m pytest tests/test_gpu_boundary.py -q -x -k "welch_round3_kernel and (40 or 30)" > $OUT/pytest_native.log 2>&1; tail -3 $OUT/pytest_native.log
