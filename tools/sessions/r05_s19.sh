#!/bin/bash
# Round 5, session 19: the long-filter tests on the multi-pass engine, the filter / conv parity files, the long-filter sweep.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s19; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_parity.py -m gpu -x -q -k "filt or conv or ols or long or xcorr" 2>&1 | tail -15 | tee $O/pytest.log
LONGFILT_TAPS=8192,16384,20000,32768,65536,131072,262144,1000000 LONGFILT_NO_ROCFFT=1 timeout 900 python tools/bench_longfilt.py 2>&1 | grep -v amdgpu.ids | cut -c1-330
cp gpurun_out/longfilt.json $O/
