#!/bin/bash
# round 3, session 31: config 5 through bench.py with the memory waves at normal / raised priority, alternating processes on one box
for v in 0 1 0 1 0 1; do
  echo -n "MDSP_FIR_MM_PRIO=$v  "
  MDSP_FIR_MM_PRIO=$v timeout 200 python bench.py --config resample --steps 10 --warmup 3 --no-cpu-baseline --no-live-pmc --no-host 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
echo "== tune_fir same box"
TUNE_RATIO=160/147 TUNE_LOG2N=28 TUNE_ROUNDS=5 TUNE_FIR="1,0,0,0,0,0,0,-1,-1,1,0,0;1,0,0" python tools/tune_fir.py 2>&1 | grep "mm=" | cut -c1-120
