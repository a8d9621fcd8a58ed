#!/bin/bash
# Round 4, GPU session 17: nfft 3000 as 5 x 24 x 25 (three passes, composite butterflies) against 3 x 5 x 5 x 5 x 8; parity first.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/s17; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -k "mixed_radix or compile_time" 2>&1 | tail -4
WIDE_SIZES=3000 OUT=s17/wide.json timeout 600 python tools/bench_wide.py 2>&1 | grep -v amdgpu.ids
