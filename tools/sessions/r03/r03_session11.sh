#!/bin/bash
# round 3, session 11: partitioned overlap-save with LDS-resident half spectra -- tests, ablation, sweep
mkdir -p gpurun_out
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_boundary.py -x -q -m gpu -k "partitioned or long_filters" 2>&1 | tail -15 | tee gpurun_out/s11_tests.log
echo "== ablation"
MDSP_LIB_TAG=dbg LONGFILT_VARIANTS=0,1,5,6,7 LONGFILT_ABLATE=0,1,8,2,4,13 timeout 300 python tools/longfilt_ablate.py 2>&1 | tail -18 | tee gpurun_out/s11_ablate.log
echo "== long filters"
LONGFILT_VARIANTS=0,1,2,3,5,6,7 LONGFILT_TAPS=5120,8192 LONGFILT_DTYPES=float32 LONGFILT_NO_ROCFFT=1 timeout 600 python tools/bench_longfilt.py 2>&1 | tail -8 | tee gpurun_out/s11_longfilt.log
