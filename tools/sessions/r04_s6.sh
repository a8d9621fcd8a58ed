#!/bin/bash
# Round 4, GPU session 6: where the hand-allocated Welch kernel's time goes -- builds of it without its loads / LDS operations / lane swaps /
# accumulation (tools/gen_welch_asm.py --ablate), all-zero stream (full clock) and random data, under the power probe.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/s6; mkdir -p $OUT
export TMPDIR=/tmp
for tag in "" ab_loads ab_lds ab_perm ab_acc; do
  MDSP_LIB_TAG=$tag WELCH_VARIANTS=42 CASES=welch,welch_zeros SECONDS=2 OUT=s6/power_${tag:-full}.json python tools/power_probe.py 2>&1 | grep -v amdgpu.ids | sed "s/^/${tag:-full}: /"
done
