#!/bin/bash
# round 3, session 25: padded runs (MDSP_FIR_MM_ROWS=2, the default where the row stride is bank-hostile) against the row-staged form (=1) and the
# plain linear run (=0); parity first.   fields: mm,wg,p,nd,ns,ng,ch,pad,rows
mkdir -p gpurun_out/ng
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_fullsize.py -x -q -m gpu -k "polyphase or matrix_core or resample or fir or config5" 2>&1 | tail -3
V="1,0,0,0,0,0,0,-1,1;1,0,0"
for c in ${RPAD_CASES:-f32:147/160 f64:147/160 c32:147/160 f32:147/80 f32:147/320 f64:147/320 f32:441/320 f32:49/48}; do
  dt=${c%%:*}; r=${c##*:}
  echo "== $dt $r"
  TUNE_DTYPE=$dt TUNE_RATIO=$r TUNE_LOG2N=26 TUNE_ROUNDS=5 TUNE_FIR="$V" timeout 200 python tools/tune_fir.py 2>&1 | grep "mm=\|rror" | cut -c1-110
  cp gpurun_out/tune_fir.json gpurun_out/ng/rpad_${dt}_${r/\//_}.json
done
