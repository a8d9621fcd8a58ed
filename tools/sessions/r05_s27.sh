#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_parity.py -m gpu -x -q -k "filt or conv or ols or long or xcorr" 2>&1 | tail -15
