#!/bin/bash
# Where a tile's time goes in the matrix-core polyphase kernel: MDSP_ABLATE bits (1 no tile DMA, 2 no matrix products, 4 no output stores) on a
# -DMDSP_DEBUG_KNOBS build, per shape.   gpurun -- bash tools/fir_ablate.sh  ->  gpurun_out/fir_ablate/<dtype>_<L>_<M>_<bits>.json
O=gpurun_out/fir_ablate; mkdir -p $O
export MDSP_LIB_TAG=dbg
for c in ${FIR_ABLATE_CASES:-f32:147/160:26 f32:160/147:28 f32:2/1:26 f32:1/8:26 f32:1/2:26}; do
  IFS=: read dt r lg <<< "$c"
  for ab in ${FIR_ABLATE_BITS:-0 1 2 4 3 5 6 7}; do
    echo -n "$dt $r 2^$lg ablate=$ab  "
    MDSP_ABLATE=$ab TUNE_DTYPE=$dt TUNE_RATIO=$r TUNE_LOG2N=$lg TUNE_ROUNDS=5 TUNE_FIR="1,0,0" timeout 200 python tools/tune_fir.py 2>&1 | grep "mm=" | awk '{print $4, "ms"}'
    cp gpurun_out/tune_fir.json $O/${dt}_${r/\//_}_$ab.json
  done
done
