#!/bin/bash
# Round 5, session 21: the 14-step ComplexF64 form (160//147 with the default resampling filter).
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s21; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_boundary.py -m gpu -x -q -k "matrix_core_kernel_equals and 160-147" 2>&1 | tail -5
FIRR_DTYPES=c64 FIRR_RATIOS=160/147,3/2 FIRR_VARIANTS="default;MDSP_FIR_MM_TIGHT=0" FIRR_OUT=r05s21/fir_c64.json timeout 600 python tools/bench_fir_ratios.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
