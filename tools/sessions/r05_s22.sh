#!/bin/bash
# Round 5, session 22: the decimator kernel as a persistent workgroup with the next tile's samples in registers.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s22; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_boundary.py -m gpu -x -q -k "decimator or matrix_core_kernel_equals" 2>&1 | tail -4
FIRR_RATIOS=1/4,1/8,1/16 FIRR_VARIANTS="default;MDSP_FIR_DEC_WGS=1;MDSP_FIR_DEC_WGS=4" FIRR_OUT=r05s22/fir_dec.json timeout 900 python tools/bench_fir_ratios.py 2>&1 | grep -v amdgpu.ids | cut -c1-250
