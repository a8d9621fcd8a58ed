#!/bin/bash
# Round 4, GPU session 28: one LDS buffer (CtSched flag 16) on the SMALL-radix schedules too -- a build with the flag on the whole list against the product
# build, alternating processes.  Result: no gain (Welch -1 ... -5 %), not adopted.
set -u
cd "$(dirname "$0")/../.."
for tag in "" smallinpl ""; do echo "== tag=$tag"; MDSP_LIB_TAG=$tag WIDE_SIZES=1000,1280,1536,1600,2560,3072,3200,4000,5120,6144,6400,8000 REPS=5 OUT=s28/wide_$tag.json timeout 900 python tools/bench_wide.py 2>&1 | grep -v amdgpu.ids | cut -c1-200; done
