#!/bin/bash
# Round 5, session 14: the AUTO rule of the multi-pass engine (tests), and the spectral / boundary suites after it.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s14; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_bigfft.py tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
tail -6 $O/pytest.log | cut -c1-300
