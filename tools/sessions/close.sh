#!/bin/bash
# Closing session of a round -- the LAST GPU session, at the last library commit (`git rev-parse --short=12 HEAD > .build_commit` before sending):
# the whole GPU suite, smoke(), tools/gpu_profile.sh (rocprofv3 kernel stats, FETCH / WRITE / SQ counter passes, the un-profiled bench lines).
# Afterwards the builder copies gpurun_out/{pytest_gpu.log,kernel_stats.txt,pmc_*.json,pmc_brief.txt,bench_line.json,bench_config*.json} to profiles/rNNz_*.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/close; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/gpu_profile.sh > $O/gpu_profile.log 2>&1; tail -4 $O/gpu_profile.log
