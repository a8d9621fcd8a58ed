#!/bin/bash
# Round 5, session 13: where does the multi-pass engine beat the rocFFT pipeline?  Welch / STFT / spectrogram at 8400 .. 200000 points, Float32 and Float64.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s13; mkdir -p $O
export TMPDIR=/tmp
echo "--- f32"; MIXED_LOG2N=26 MIXED_SIZES=8400,9000,10000,12500,16000,16384,20000,25000,32768,40000,50000,65536,100000,125000,200000 timeout 900 python tools/bench_mixed.py 2>&1 | grep -v amdgpu.ids | cut -c1-330; cp gpurun_out/mixed.json $O/mixed_f32.json
echo "--- f64"; LOG2N=26 SIZES=7200,8192,9000,10000,12500,16384,20000,32768,50000,65536,100000,125000 timeout 900 python tools/bench_f64_sizes.py 2>&1 | grep -v amdgpu.ids | cut -c1-300; cp gpurun_out/f64_sizes.json $O/f64_sizes.json
