#!/bin/bash
# round 3, session 39: register taps with fewer chunks / shorter rows (default) against round 2's limits (MDSP_FIR_MM_T64=0): every ratio whose
# geometry changed, four signal types, one process per shape
mkdir -p gpurun_out/regtaps
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_fullsize.py -x -q -m gpu -k "polyphase or matrix_core or resample or fir or config5" 2>&1 | tail -2
V="1,0,0,0,0,0,0,-1,-1,1,0,-1,0;1,0,0"
for dt in f32 f64 c32 c64; do
  for r in 1/8 1/4 3/8 1/3 1/2 1/16 2/3 1/6; do
    echo -n "$dt $r  "
    TUNE_DTYPE=$dt TUNE_RATIO=$r TUNE_LOG2N=26 TUNE_ROUNDS=5 TUNE_FIR="$V" timeout 200 python tools/tune_fir.py 2>&1 | grep "mm=" | awk '{printf "%s %s  ", $(NF-5), $(NF-3)} END {print ""}'
    cp gpurun_out/tune_fir.json gpurun_out/regtaps/ab_${dt}_${r/\//_}.json
  done
done
