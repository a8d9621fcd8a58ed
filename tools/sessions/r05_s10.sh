#!/bin/bash
# Round 5, session 10: decimator kernel with execution-masked tail steps: tests + decimating cells against MDSP_FIR_DEC=0 (M = 2 .. 16, 5, 6, 12, 32).
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s10; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_boundary.py -x -q -k "decimator" > $O/pytest_dec.log 2>&1; echo "pytest dec rc=$?" | tee -a $O/rc.txt
tail -3 $O/pytest_dec.log | cut -c1-250
FIRR_RATIOS=1/2,1/3,1/4,1/5,1/6,1/8,1/12,1/16,1/32 FIRR_VARIANTS="default;MDSP_FIR_DEC=0" FIRR_OUT=r05s10/fir_dec_ab.json timeout 900 python tools/bench_fir_ratios.py 2>&1 | grep -v amdgpu.ids | cut -c1-250
