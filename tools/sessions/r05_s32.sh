#!/bin/bash
# Round 5, session 32: the 128-point two-stage pass as 16 x 8 (MDSP_BIG_FAST=1, new) against 8 x 16 (=2), Float32.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s32; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bigfft.py -m gpu -x -q 2>&1 | tail -3
for f in 1 2; do
  echo "== MDSP_BIG_FAST=$f"
  MDSP_BIG_FAST=$f DEFSPEC_ENGINES=auto DEFSPEC_LENGTHS=8388608,16777216,33554432,67108864 DEFSPEC_OUT=r05s32/def_fast$f.json timeout 600 python tools/bench_default_spectral.py 2>&1 | grep -v amdgpu.ids | cut -c1-330
  MDSP_BIG_FAST=$f MDSP_BIG_WELCH_ROWS=2 WL_LOG2N=20 WL_OUT=r05s32/wl_fast$f.json timeout 600 python tools/bench_welch_large.py 2>&1 | grep -v amdgpu.ids
  MDSP_BIG_FAST=$f MDSP_BIG_WELCH_ROWS=0 WL_LOG2N=20,21 WL_OUT=r05s32/wl3_fast$f.json timeout 600 python tools/bench_welch_large.py 2>&1 | grep -v amdgpu.ids
  MDSP_BIG_FAST=$f BIGOLS_SKIP_CHECK=1 BIGOLS_SKIP_SEGMENTS=1 BIGOLS_TAPS=32768,131072 BIGOLS_LOG2N=19,20 BIGOLS_OUT=r05s32/ols_fast$f.json timeout 600 python tools/check_big_ols.py 2>&1 | grep "^float32" | cut -c1-400
done
