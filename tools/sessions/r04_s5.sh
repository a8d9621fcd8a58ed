#!/bin/bash
# Round 4, GPU session 5: the hand-allocated Welch kernel (variant 42, csrc/welch_w64_asm.s) -- parity, then interleaved timing against 30.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/s5; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_boundary.py -q -x -k "welch_round3_kernel and 42" > $OUT/pytest.log 2>&1; tail -25 $OUT/pytest.log
TUNE_LOG2N=30 TUNE_ROUNDS=10 TUNE_OLS=0 TUNE_WELCH=30,42 TUNE_WGS=2 TUNE_RUNS=1 timeout 600 python tools/tune.py > $OUT/tune.log 2>&1
mv gpurun_out/tune.json $OUT/tune.json; tail -8 $OUT/tune.log
