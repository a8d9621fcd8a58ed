#!/bin/bash
# round 3, session 10: partitioned overlap-save -- LDS ring variants, block ranges; long-filter sweep
mkdir -p gpurun_out
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_boundary.py -x -q -m gpu -k "partitioned or long_filters or block_range or host_pipeline_overlap" 2>&1 | tail -15 | tee gpurun_out/s10_tests.log
echo "== long filters"
LONGFILT_VARIANTS=0,1,2,3,4 LONGFILT_TAPS=5120,8192,12000,16384 LONGFILT_DTYPES=float32 LONGFILT_NO_ROCFFT=1 timeout 600 python tools/bench_longfilt.py 2>&1 | tail -8 | tee gpurun_out/s10_longfilt.log
cp gpurun_out/longfilt.json gpurun_out/s10_longfilt.json
