#!/bin/bash
# Round 5, session 30: Welch with a large nfft on a LONG stream (255 .. 2047 frames), rows form against three passes.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s30; mkdir -p $O
export TMPDIR=/tmp
for rows in 1 0; do
  echo "== MDSP_BIG_WELCH_ROWS=$rows"
  MDSP_BIG_WELCH_ROWS=$rows WL_OUT=r05s30/f32_rows$rows.json WL_LOG2N=18,19,20,21 timeout 600 python tools/bench_welch_large.py 2>&1 | grep -v amdgpu.ids
  MDSP_BIG_WELCH_ROWS=$rows WL_DTYPE=f64 WL_LOG2LEN=26 WL_LOG2N=17,18,19,20 WL_OUT=r05s30/f64_rows$rows.json timeout 600 python tools/bench_welch_large.py 2>&1 | grep -v amdgpu.ids
done
