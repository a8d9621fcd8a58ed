#!/bin/bash
# Round 4, GPU session 11: FIRArbitrary with 12-byte trajectory records (four workgroups per CU) and a one-latency prologue, A/B against the previous
# build (libmi355dsp_arbold.so) in alternating processes; parity tests of the resampler first.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/s11; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "arb or firarb or arbitrary" 2>&1 | tail -5
for round in 1 2 3; do
  for tag in arbold ""; do
    MDSP_LIB_TAG=$tag timeout 300 python tools/bench_firarb.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.jsonl
  done
done
for tag in arbold ""; do
  MDSP_LIB_TAG=$tag ARB_RATE=147/160 ARB_LOG2N=26 timeout 300 python tools/bench_firarb.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.jsonl
  MDSP_LIB_TAG=$tag ARB_DTYPE=f64 ARB_LOG2N=26 timeout 300 python tools/bench_firarb.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.jsonl
  MDSP_LIB_TAG=$tag ARB_NCH=1 ARB_LOG2N=26 timeout 300 python tools/bench_firarb.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.jsonl
done
