#!/bin/bash
# Round 5, session 7: the decimator kernel meets hardware: parity / streaming / non-finite tests, then decimating cells of the ratio table against MDSP_FIR_DEC=0.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s7; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_boundary.py -x -q -k "decimator or kernel_choice or nonfinite or matrix_core_kernel_fuzz" > $O/pytest_dec.log 2>&1; echo "pytest dec rc=$?" | tee -a $O/rc.txt
tail -25 $O/pytest_dec.log | cut -c1-250
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "firfilter_kernels or resample" > $O/pytest_fir.log 2>&1; echo "pytest fir rc=$?" | tee -a $O/rc.txt
tail -5 $O/pytest_fir.log | cut -c1-250
FIRR_RATIOS=1/2,1/3,1/4,1/8,1/16 FIRR_VARIANTS="default;MDSP_FIR_DEC=0" FIRR_OUT=r05s7/fir_dec_ab.json timeout 900 python tools/bench_fir_ratios.py 2>&1 | grep -v amdgpu.ids | cut -c1-200
