#!/bin/bash
# Round 5, session 24: decimators after the split (Float64 persistent with the next tile in registers, Float32 one tile per workgroup) + FIR tests.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s24; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_boundary.py -m gpu -x -q -k "decimator or matrix_core_kernel_equals or fir or resample" 2>&1 | tail -3
FIRR_RATIOS=1/4,1/8,1/16,1/5,1/32 FIRR_OUT=r05s24/fir_dec.json timeout 900 python tools/bench_fir_ratios.py 2>&1 | grep -v amdgpu.ids | cut -c1-120
