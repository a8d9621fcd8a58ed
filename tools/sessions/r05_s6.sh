#!/bin/bash
# Round 5, session 6: ablation of the two-stage pass kernel, default Welch at 2^27 (chunk 2048 MiB: every pass one launch over all 8 transforms).
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s6; mkdir -p $O
export TMPDIR=/tmp
L=134217728
for ab in 0 2 4 6 8 16 18 20 22 30; do
  echo -n "ablate=$ab: "; MDSP_BIG_CHUNK_MIB=2048 MDSP_BIG_ABLATE=$ab DEFSPEC_ENGINES=auto DEFSPEC_WELCH_ONLY=1 DEFSPEC_LENGTHS=$L DEFSPEC_OUT=r05s6/ab$ab.json timeout 300 python tools/bench_default_spectral.py 2>&1 | grep -v amdgpu.ids | cut -c1-300
done
