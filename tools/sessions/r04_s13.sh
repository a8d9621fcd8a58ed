#!/bin/bash
# Round 4, GPU session 13: FIRArbitrary compute loop with the next batch's LDS reads in flight (batches of 8 and of 4) against the burst form,
# alternating processes; parity tests of the resampler first.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/s13; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "arb or firarb or arbitrary" 2>&1 | tail -5
MDSP_LIB_TAG=arbb4 timeout 900 python -m pytest tests -m gpu -x -q -k "arb or firarb or arbitrary" 2>&1 | tail -2
for round in 1 2 3; do
  for tag in arbnopipe "" arbb4; do
    MDSP_LIB_TAG=$tag timeout 300 python tools/bench_firarb.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.jsonl
  done
done
for tag in arbnopipe "" arbb4; do
  MDSP_LIB_TAG=$tag ARB_RATE=147/160 ARB_LOG2N=26 timeout 300 python tools/bench_firarb.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.jsonl
  MDSP_LIB_TAG=$tag ARB_DTYPE=f64 ARB_LOG2N=26 timeout 300 python tools/bench_firarb.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.jsonl
  MDSP_LIB_TAG=$tag ARB_NCH=1 ARB_LOG2N=26 timeout 300 python tools/bench_firarb.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.jsonl
  MDSP_LIB_TAG=$tag ARB_NCH=2 ARB_LOG2N=26 timeout 300 python tools/bench_firarb.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.jsonl
done
