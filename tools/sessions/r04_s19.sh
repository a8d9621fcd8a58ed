#!/bin/bash
# Round 4, GPU session 19: sixteen three-pass composite-radix schedules against round 3's; parity first.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/s19; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -k "mixed_radix or compile_time" 2>&1 | tail -3
WIDE_SIZES=1000,1200,1500,1600,1920,2000,2400,2500,3000,3200,3840,4800,5000,6000,6400,8000 REPS=5 OUT=s19/wide.json timeout 900 python tools/bench_wide.py 2>&1 | grep -v amdgpu.ids
