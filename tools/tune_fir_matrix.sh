#!/bin/bash
# Register-tap / generic polyphase kernels against the matrix-core kernel over ratios and signal types (4 channels x 2^26 samples;
# config 5's own shape at 2^28 Float32).  One JSON per case under gpurun_out/: r02r_tune_fir_<dtype>_<L>_<M>.json
O=gpurun_out; mkdir -p $O
TUNE_FIR="0,0,0;1,0,0" python tools/tune_fir.py | grep "mm=" && cp $O/tune_fir.json $O/r02r_tune_fir_f32_160_147_2p28.json
for dt in f32 f64 c32 c64; do
  for r in 160/147 147/160 2/1 1/2 3/2 2/3 5/3 4/1 1/3 1/4 1/8 3/8 160/441 441/160; do
    echo "== $dt $r"
    TUNE_DTYPE=$dt TUNE_RATIO=$r TUNE_LOG2N=26 TUNE_ROUNDS=3 TUNE_FIR="0,0,0;1,0,0" python tools/tune_fir.py 2>&1 | grep "mm="
    cp $O/tune_fir.json $O/r02r_tune_fir_${dt}_${r/\//_}.json
  done
done
