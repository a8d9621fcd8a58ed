#!/bin/bash
# Round 5, session 12: Float64 Welch / STFT at 4800 .. 8000 points: the single-workgroup compile-time schedules (which spill) against the multi-pass engine.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s12; mkdir -p $O
export TMPDIR=/tmp
echo "--- default"; LOG2N=27 SIZES=4096,4800,5000,6000,6400,7200,8000 timeout 600 python tools/bench_f64_sizes.py 2>&1 | grep -v amdgpu.ids | cut -c1-300; cp gpurun_out/f64_sizes.json $O/f64_default.json
echo "--- MDSP_GEN_CT_F64_MAX=4096"; MDSP_GEN_CT_F64_MAX=4096 LOG2N=27 SIZES=4800,5000,6000,6400,7200,8000 timeout 600 python tools/bench_f64_sizes.py 2>&1 | grep -v amdgpu.ids | cut -c1-300; cp gpurun_out/f64_sizes.json $O/f64_big.json
