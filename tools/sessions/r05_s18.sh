#!/bin/bash
# Round 5, session 18: overlap-save on the multi-pass engine -- correctness of every dtype / mode, then against the host-summed segments.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s18; mkdir -p $O
export TMPDIR=/tmp
BIGOLS_LOG2N=0,17,18,19,20,21 BIGOLS_OUT=r05s18/big_ols.json timeout 1200 python tools/check_big_ols.py 2>&1 | grep -v amdgpu.ids | cut -c1-400
