#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
echo "== welch phase profile (finer), 2 WG/CU"; MDSP_LIB_TAG=prof TUNE_LOG2N=30 TUNE_ROUNDS=2 TUNE_OLS= TUNE_WELCH=30,33 TUNE_WGS=2 TUNE_RUNS=1 timeout 300 python tools/tune.py > $OUT/s4_prof2.log 2>&1; grep WELCHPROF $OUT/s4_prof2.log | tail -6
echo "== 1 WG/CU"; MDSP_WG_PER_CU=1 MDSP_LIB_TAG=prof TUNE_LOG2N=30 TUNE_ROUNDS=1 TUNE_OLS= TUNE_WELCH=30,33 TUNE_WGS=1 TUNE_RUNS=1 timeout 300 python tools/tune.py > $OUT/s4_prof1.log 2>&1; grep WELCHPROF $OUT/s4_prof1.log | tail -4
