#!/bin/bash
# Round 4, GPU session 9: where the hand-allocated overlap-save kernel's time goes -- builds without its loads / stores / LDS operations and with
# nontemporal stores, under the power probe (random data and an all-zero stream).
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/s9; mkdir -p $OUT
export TMPDIR=/tmp
for tag in "" ob_loads ob_stores ob_lds ob_nt; do
  MDSP_LIB_TAG=$tag CASES=ols,ols_zeros SECONDS=2 OUT=s9/power_${tag:-full}.json python tools/power_probe.py 2>&1 | grep -v amdgpu.ids | sed "s/^/${tag:-full}: /"
done
MDSP_OLS_VARIANT=30 CASES=ols,ols_zeros SECONDS=2 OUT=s9/power_cpp.json python tools/power_probe.py 2>&1 | grep -v amdgpu.ids | sed "s/^/cpp30: /"
