#!/bin/bash
# Round 5, session 29: Welch in the rows form (column pass + single-workgroup Welch kernel over the rows) -- parity, then against three passes.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s29; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_bigfft.py -m gpu -x -q 2>&1 | tail -8
for rows in 1 0; do
  echo "== MDSP_BIG_WELCH_ROWS=$rows"
  MDSP_BIG_WELCH_ROWS=$rows DEFSPEC_WELCH_ONLY=1 DEFSPEC_ENGINES=auto DEFSPEC_LENGTHS=2097152,4194304,8388608,16777216 DEFSPEC_OUT=r05s29/welch_rows$rows.json timeout 600 python tools/bench_default_spectral.py 2>&1 | grep -v amdgpu.ids
  MDSP_BIG_WELCH_ROWS=$rows DEFSPEC_DTYPE=f64 DEFSPEC_WELCH_ONLY=1 DEFSPEC_ENGINES=auto DEFSPEC_LENGTHS=1048576,2097152,4194304,8388608 DEFSPEC_OUT=r05s29/welch_rows${rows}_f64.json timeout 600 python tools/bench_default_spectral.py 2>&1 | grep -v amdgpu.ids
done
