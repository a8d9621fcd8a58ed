#!/bin/bash
# round 3, session 36: config 5's shape, DMA / store wave counts with the memory waves at raised priority
#   fields: mm,wg,p,nd,ns
V="1,0,0;1,0,0,1,2;1,0,0,2,3;1,0,0,2,4;1,0,0,1,3;1,0,0,3,3;1,0,0,1,4;1,0,0,3,2"
TUNE_RATIO=160/147 TUNE_LOG2N=28 TUNE_ROUNDS=5 TUNE_FIR="$V" timeout 300 python tools/tune_fir.py 2>&1 | grep "mm=" | cut -c1-90
echo "== 147/160"
TUNE_RATIO=147/160 TUNE_LOG2N=26 TUNE_ROUNDS=5 TUNE_FIR="1,0,0;1,0,0,2,2;1,0,0,2,3;1,0,0,1,4;1,0,0,3,3;1,0,0,1,2" timeout 300 python tools/tune_fir.py 2>&1 | grep "mm=" | cut -c1-90
echo "== 2/1"
TUNE_RATIO=2/1 TUNE_LOG2N=26 TUNE_ROUNDS=5 TUNE_FIR="1,0,0;1,0,0,1,2;1,0,0,1,3;1,0,0,2,3" timeout 300 python tools/tune_fir.py 2>&1 | grep "mm=" | cut -c1-90
