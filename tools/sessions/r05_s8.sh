#!/bin/bash
# Round 5, session 8: decimator kernel, compile-time M forms against the run-time M form and the older kernels.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s8; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_boundary.py -x -q -k "decimator" > $O/pytest_dec.log 2>&1; echo "pytest dec rc=$?" | tee -a $O/rc.txt
tail -3 $O/pytest_dec.log | cut -c1-250
FIRR_RATIOS=1/2,1/4,1/8,1/16 FIRR_VARIANTS="default;MDSP_FIR_DEC=2;MDSP_FIR_DEC=0" FIRR_OUT=r05s8/fir_dec_ab.json timeout 900 python tools/bench_fir_ratios.py 2>&1 | grep -v amdgpu.ids | cut -c1-250
