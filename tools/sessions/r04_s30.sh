#!/bin/bash
# Round 4, GPU session 30: one element of padding behind every output group of a pass in the compile-time schedules (CtSched::padded) against a build
# without it (-DMDSP_GEN_CT_PAD=0), alternating processes, all 23 sizes; parity of every size and signal type first.
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -k "mixed_radix or compile_time" 2>&1 | tail -2
for tag in nopad "" nopad ""; do echo "== tag=$tag"; MDSP_LIB_TAG=$tag WIDE_SIZES=1000,1200,1280,1500,1536,1600,1920,2000,2400,2500,2560,3000,3072,3200,3840,4000,4800,5000,5120,6000,6144,6400,8000 REPS=5 OUT=s30/wide_$tag.json timeout 900 python tools/bench_wide.py 2>&1 | grep -v amdgpu.ids | awk '{print}' | cut -c1-200; done
