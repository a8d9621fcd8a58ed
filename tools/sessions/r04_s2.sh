#!/bin/bash
# Round 4, GPU session 2: Welch variant 41 (one wavefront per transform, two waves per SIMD) -- parity, then interleaved timing against 30 and 40.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/s2; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_boundary.py -q -x -k "welch_round3_kernel and (41 or 40)" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
TUNE_LOG2N=30 TUNE_ROUNDS=8 TUNE_OLS=0 TUNE_WELCH=30,40,41 TUNE_WGS=2 TUNE_RUNS=1 timeout 600 python tools/tune.py > $OUT/tune.log 2>&1
mv gpurun_out/tune.json $OUT/tune.json; tail -8 $OUT/tune.log
