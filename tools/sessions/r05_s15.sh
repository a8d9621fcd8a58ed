#!/bin/bash
# Round 5, session 15: whole GPU suite after the pruning.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s15; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" | tee -a $O/rc.txt
tail -12 $O/pytest_gpu.log | cut -c1-300
