#!/bin/bash
# round 3, session 42: L > 192 -- the taps of a wave's two / three column blocks in registers (default) against fetched per tile
# (MDSP_FIR_MM_NBLK=0) and, Float32, against the register-tap kernel (mm=0)
#   fields: mm,wg,p,nd,ns,ng,ch,pad,rows,vstore,rpad,prio,t64,nblk
mkdir -p gpurun_out/nblk
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_boundary.py -x -q -m gpu -k "polyphase or matrix_core or resample" 2>&1 | tail -2
V="0,0,0;1,0,0,0,0,0,0,-1,-1,1,0,-1,1,0;1,0,0"
for c in f32:441/160 f64:441/160 c32:441/160 f32:320/147 f64:320/147 f32:250/249 f64:640/441; do
  dt=${c%%:*}; r=${c##*:}
  echo "== $dt $r"
  TUNE_DTYPE=$dt TUNE_RATIO=$r TUNE_LOG2N=26 TUNE_ROUNDS=5 TUNE_FIR="$V" timeout 200 python tools/tune_fir.py 2>&1 | grep "mm=" | awk '{print "   ", $1, $(NF-6), $(NF-5), $(NF-4), $(NF-3), $(NF-1), $NF}'
  cp gpurun_out/tune_fir.json gpurun_out/nblk/${dt}_${r/\//_}.json
done
