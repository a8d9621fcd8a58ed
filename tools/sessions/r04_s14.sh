#!/bin/bash
# Round 4, GPU session 14: FIRArbitrary -- interior staging path, shorter replay chain, priority on the replaying wave only; timeline of the
# uncontended prologue (one tap per output) and of the full kernel.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/s14; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "arb or firarb or arbitrary" 2>&1 | tail -3
for round in 1 2; do
  for tag in arbnopipe ""; do
    MDSP_LIB_TAG=$tag timeout 300 python tools/bench_firarb.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.jsonl
  done
done
for prio in 1 2; do MDSP_ARB_PRIO=$prio timeout 300 python tools/bench_firarb.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.jsonl; done
MDSP_LIB_TAG=dbg MDSP_ARB_PROF=1 REPS=2 timeout 300 python tools/bench_firarb.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $OUT/prof.txt
MDSP_LIB_TAG=dbg MDSP_ARB_PROF=1 MDSP_ARB_PRIO=2 REPS=2 timeout 300 python tools/bench_firarb.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $OUT/prof.txt
MDSP_LIB_TAG=dbg MDSP_ARB_PROF=1 MDSP_ABLATE=2 REPS=2 timeout 300 python tools/bench_firarb.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $OUT/prof.txt
MDSP_LIB_TAG=dbg MDSP_ARB_PROF=1 MDSP_ABLATE=6 REPS=2 timeout 300 python tools/bench_firarb.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $OUT/prof.txt
MDSP_LIB_TAG=dbg MDSP_ABLATE=2 REPS=5 timeout 300 python tools/bench_firarb.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $OUT/prof.txt
for tag in arbnopipe ""; do
  MDSP_LIB_TAG=$tag ARB_RATE=147/160 ARB_LOG2N=26 timeout 300 python tools/bench_firarb.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.jsonl
  MDSP_LIB_TAG=$tag ARB_DTYPE=f64 ARB_LOG2N=26 timeout 300 python tools/bench_firarb.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.jsonl
  MDSP_LIB_TAG=$tag ARB_NCH=1 ARB_LOG2N=26 timeout 300 python tools/bench_firarb.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.jsonl
done
