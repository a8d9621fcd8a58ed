#!/bin/bash
# Round 4, GPU session 18: compile-time schedules after the descriptor loads (all 23 sizes, against profiles/r03d_mixed_ct.json) and nfft 3000 as
# 5 x 24 x 25 in place with two waves per SIMD; parity first.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/s18; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -k "mixed_radix or compile_time" 2>&1 | tail -3
WIDE_SIZES=3000,1000,1200,1280,1500,1536,1600,1920,2000,2400,2500,2560,3072,3200,3840,4000,4800,5000,5120,6000,6144,6400,8000 REPS=5 OUT=s18/wide.json timeout 900 python tools/bench_wide.py 2>&1 | grep -v amdgpu.ids
