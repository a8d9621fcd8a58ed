#!/bin/bash
# Round 5, session 25: long filters in the rows form (column pass + row kernel + column pass back) against three passes each way.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s25; mkdir -p $O
export TMPDIR=/tmp
echo "== rows form"
BIGOLS_LOG2N=0,18,19,20,21 BIGOLS_OUT=r05s25/big_ols_rows.json timeout 1200 python tools/check_big_ols.py 2>&1 | grep -v amdgpu.ids | cut -c1-700
echo "== three passes each way"
MDSP_BIG_OLS_ROWS=0 BIGOLS_LOG2N=0 BIGOLS_OUT=r05s25/big_ols_3pass.json timeout 1200 python tools/check_big_ols.py 2>&1 | grep -v amdgpu.ids | grep "^float" | cut -c1-300
