#!/bin/bash
# Round 4, GPU session 12: FIRArbitrary per-workgroup timeline (clock stamps, debug-knob build) and tile sizes / priority on the product build.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/s12; mkdir -p $OUT
export TMPDIR=/tmp
MDSP_LIB_TAG=dbg MDSP_ARB_PROF=1 REPS=3 timeout 300 python tools/bench_firarb.py 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $OUT/prof.txt
for abl in 1 2 4 8; do
  echo "ablate $abl" | tee -a $OUT/prof.txt
  MDSP_LIB_TAG=dbg MDSP_ABLATE=$abl REPS=5 timeout 300 python tools/bench_firarb.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $OUT/prof.txt
done
for tile in 0 768 512 896; do
  for prio in 0 1; do
    MDSP_ARB_TILE=$tile MDSP_ARB_PRIO=$prio timeout 300 python tools/bench_firarb.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/knobs.jsonl
  done
done
