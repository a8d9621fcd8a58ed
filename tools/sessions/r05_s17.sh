#!/bin/bash
# Round 5, session 17: the decimator kernel forced (MDSP_FIR_DEC=3) at the M where the default still takes the matrix cores.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s17; mkdir -p $O
export TMPDIR=/tmp
FIRR_RATIOS=1/2,1/3,1/5,1/6,1/7,1/9,1/10 FIRR_VARIANTS="MDSP_FIR_DEC=3;MDSP_FIR_DEC=0" FIRR_OUT=r05s17/fir_dec_forced.json timeout 900 python tools/bench_fir_ratios.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
