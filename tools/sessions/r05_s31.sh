#!/bin/bash
# Round 5, session 31: workgroups per CU of the Float64 two-stage passes on the three-pass spectral engine (default 1) -- Welch default calls.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s31; mkdir -p $O
export TMPDIR=/tmp
for w in 0 2 3 4; do
  echo "== MDSP_BIG_WGS=$w"
  MDSP_BIG_WGS=$w DEFSPEC_DTYPE=f64 DEFSPEC_WELCH_ONLY=1 DEFSPEC_ENGINES=auto DEFSPEC_LENGTHS=1048576,33554432,67108864,134217728 DEFSPEC_OUT=r05s31/f64_wgs$w.json timeout 600 python tools/bench_default_spectral.py 2>&1 | grep -v amdgpu.ids
  MDSP_BIG_WGS=$w DEFSPEC_WELCH_ONLY=1 DEFSPEC_ENGINES=auto DEFSPEC_LENGTHS=1048576,67108864,134217728 DEFSPEC_OUT=r05s31/f32_wgs$w.json timeout 600 python tools/bench_default_spectral.py 2>&1 | grep -v amdgpu.ids
done
