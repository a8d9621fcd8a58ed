#!/bin/bash
# round 3, session 12: partitioned overlap-save, round-3 form at the other geometries
mkdir -p gpurun_out
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_boundary.py -x -q -m gpu -k "partitioned or long_filters" 2>&1 | tail -15 | tee gpurun_out/s12_tests.log
echo "== long filters"
LONGFILT_VARIANTS=0,1 timeout 600 python tools/bench_longfilt.py 2>&1 | tail -20 | tee gpurun_out/s12_longfilt.log
