#!/bin/bash
# Round 4, GPU session 8: the hand-allocated overlap-save kernel (mdsp_ols_w64_asm) -- every overlap-save / conv / filt parity test, config 2 at full
# size, the host pipeline, then interleaved timing: default (asm for interior units) against variant 30 (ols_fused_kernel everywhere).
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/s8; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -m gpu -q -x -k "ols or fftfilt or conv or filt or config2 or host or overlap" > $OUT/pytest.log 2>&1; tail -12 $OUT/pytest.log
TUNE_LOG2N=30 TUNE_ROUNDS=10 TUNE_OLS=0,30 TUNE_WELCH=0 TUNE_WGS=2 TUNE_RUNS=1 timeout 600 python tools/tune.py > $OUT/tune.log 2>&1
mv gpurun_out/tune.json $OUT/tune.json; tail -8 $OUT/tune.log
