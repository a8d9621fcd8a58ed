#!/bin/bash
# Round 5, session 26: workgroups per CU of the column passes in the rows form.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s26; mkdir -p $O
export TMPDIR=/tmp
for w in 1 2 3 4 6; do
  echo "== MDSP_BIG_WGS=$w"
  MDSP_BIG_WGS=$w BIGOLS_SKIP_CHECK=1 BIGOLS_SKIP_SEGMENTS=1 BIGOLS_TAPS=32768,131072 BIGOLS_LOG2N=18,19,20,21 BIGOLS_OUT=r05s26/wgs$w.json timeout 600 python tools/check_big_ols.py 2>&1 | grep "^float" | python -c "
import sys,ast
for l in sys.stdin:
    k=l.split(' {',1); d=ast.literal_eval('{'+k[1]); print(k[0], {a:b['TBps'] for a,b in d.items()})"
done
