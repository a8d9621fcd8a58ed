#!/bin/bash
# The same measurement against several builds of the library, same box, back to back, two passes:
#   python dsp.jl_amd/build.py --tag p8w3 --cflags "-DMDSP_DEC_F32_P=8"        (on the builder: the tagged .so travels with the snapshot)
#   gpurun -- "bash tools/sessions/ab_tags.sh 'FIRR_RATIOS=1/8 python tools/bench_fir_ratios.py' '' p8w3"
# ('' = the product build; the tool sees MDSP_LIB_TAG and AB_OUT = gpurun_out/ab/<tag>_<pass>.json to name its output after)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/ab; mkdir -p $O
export TMPDIR=/tmp
TOOL=${1:?command}; shift
for rep in 1 2; do
  for tag in "$@"; do
    [ -n "$tag" ] && [ ! -f dsp.jl_amd/libmi355dsp_$tag.so ] && { echo "no build tagged $tag"; continue; }
    echo "== build '${tag:-product}' (pass $rep)"
    MDSP_LIB_TAG=$tag AB_OUT=$O/${tag:-product}_$rep.json timeout 900 bash -c "$TOOL" 2>&1 | grep -v amdgpu.ids | cut -c1-200
  done
done
