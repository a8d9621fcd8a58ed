#!/bin/bash
# Round 4, GPU session 16: FIRArbitrary with the remainder taps batched (4 + 2 + 1) against the previous build, alternating processes.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/s16; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "arb or firarb or arbitrary" 2>&1 | tail -3
for round in 1 2 3; do
  for tag in arbnopipe ""; do
    MDSP_LIB_TAG=$tag timeout 300 python tools/bench_firarb.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.jsonl
  done
done
for tag in arbnopipe ""; do
  MDSP_LIB_TAG=$tag ARB_RATE=147/160 ARB_LOG2N=26 timeout 300 python tools/bench_firarb.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.jsonl
  MDSP_LIB_TAG=$tag ARB_DTYPE=f64 ARB_LOG2N=26 timeout 300 python tools/bench_firarb.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.jsonl
  MDSP_LIB_TAG=$tag ARB_NCH=1 ARB_LOG2N=26 timeout 300 python tools/bench_firarb.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.jsonl
  MDSP_LIB_TAG=$tag ARB_NCH=2 ARB_LOG2N=26 timeout 300 python tools/bench_firarb.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.jsonl
  MDSP_LIB_TAG=$tag ARB_RATE=0.3721 ARB_LOG2N=26 timeout 300 python tools/bench_firarb.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.jsonl
done
