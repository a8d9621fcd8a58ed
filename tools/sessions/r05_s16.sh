#!/bin/bash
# Round 5, session 16: Float64 compile-time schedules from 4800 points with two-level LDS twiddles instead of register twiddles: parity, then rates.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s16; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_boundary.py -x -q -k "compile_time_mixed_radix" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/rc.txt
tail -4 $O/pytest.log | cut -c1-300
LOG2N=27 SIZES=4096,4800,5000,5120,6000,6144,6400,8000 timeout 600 python tools/bench_f64_sizes.py 2>&1 | grep -v amdgpu.ids | cut -c1-300; cp gpurun_out/f64_sizes.json $O/f64_tw2l.json
