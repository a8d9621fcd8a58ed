#!/bin/bash
# Round-3 GPU session 1: parity of the new kernels, A/B against the round-2 forms, co-residency experiment, a bench line.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
echo "== gpu tests" ; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/s1_gpu_tests.log; tail -3 $OUT/s1_gpu_tests.log
echo "== tune (new lib)"; TUNE_LOG2N=30 TUNE_ROUNDS=8 TUNE_OLS=0 TUNE_WELCH=0,18,30,31,32 TUNE_WGS=2 TUNE_RUNS=1 timeout 600 python tools/tune.py > $OUT/s1_tune_new.log 2>&1; cp $OUT/tune.json $OUT/s1_tune_new.json; grep -E "^(ols|welch|copy)" $OUT/s1_tune_new.log
echo "== tune (round-2 I/O + VGPR constants)"; MDSP_LIB_TAG=r2io TUNE_LOG2N=30 TUNE_ROUNDS=8 TUNE_OLS=0 TUNE_WELCH=0,18,30 TUNE_WGS=2 TUNE_RUNS=1 timeout 600 python tools/tune.py > $OUT/s1_tune_r2io.log 2>&1; cp $OUT/tune.json $OUT/s1_tune_r2io.json; grep -E "^(ols|welch|copy)" $OUT/s1_tune_r2io.log
echo "== tune again (new lib), order effects"; TUNE_LOG2N=30 TUNE_ROUNDS=8 TUNE_OLS=0 TUNE_WELCH=0,30 TUNE_WGS=2 TUNE_RUNS=1 timeout 600 python tools/tune.py > $OUT/s1_tune_new2.log 2>&1; cp $OUT/tune.json $OUT/s1_tune_new2.json; grep -E "^(ols|welch|copy)" $OUT/s1_tune_new2.log
echo "== coresident"; timeout 600 python tools/coresident.py > $OUT/s1_coresident.log 2>&1; tail -40 $OUT/s1_coresident.log
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/s1_bench.json 2> $OUT/s1_bench.err; cut -c1-600 $OUT/s1_bench.json
