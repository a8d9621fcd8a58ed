#!/bin/bash
# Round 5, session 23: Float32 decimator forms with the next tile in registers -- outputs per block (P) x resident workgroups x tile size, one
# library build each (MDSP_LIB_TAG), same box, back to back, twice.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s23; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
for tag in "" p8w3 p8w4 p16w3k24; do
  [ -n "$tag" ] && [ ! -f dsp.jl_amd/libmi355dsp_$tag.so ] && continue
  echo "== build '${tag:-default p16 w2 k40}' (pass $rep)"
  MDSP_LIB_TAG=$tag FIRR_DTYPES=f32,c32 FIRR_RATIOS=1/4,1/8,1/16 FIRR_OUT=r05s23/fir_dec_${tag:-default}_$rep.json timeout 600 python tools/bench_fir_ratios.py 2>&1 | grep -v amdgpu.ids | cut -c1-120
done
done
for tag in p8w3 p8w4 p16w3k24; do
  [ -f dsp.jl_amd/libmi355dsp_$tag.so ] || continue
  MDSP_LIB_TAG=$tag timeout 600 python -m pytest tests/test_gpu_boundary.py -m gpu -x -q -k "decimator" 2>&1 | tail -2
done
