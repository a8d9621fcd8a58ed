#!/bin/bash
# Round 5, session 20: the whole polyphase table (15 ratios x 4 signal types) with the round's kernels.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s20; mkdir -p $O
export TMPDIR=/tmp
FIRR_OUT=r05s20/fir_all_ratios.json timeout 1500 python tools/bench_fir_ratios.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
