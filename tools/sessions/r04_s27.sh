#!/bin/bash
# Round 4, GPU session 27: mdsp_welch_w64c_asm (shared half-frame carried, twiddles in LDS) against mdsp_welch_w64_asm: parity, alternating timing, traffic.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/s27; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -k "welch_round3_kernel_vs_oracle or config3 or welch" 2>&1 | tail -3
for r in 1 2 3; do
  for v in 42 43 30; do
    MDSP_WELCH_VARIANT=$v timeout 600 python bench.py --no-rows --no-cpu-baseline --no-live-pmc --no-host 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant $v', d['value'], d['ms_per_step'], d['config'].get('stages_ms'))"
  done
done
CASES=welch,welch_zeros SECONDS=2 OUT=s27/power.json python tools/power_probe.py 2>&1 | grep -v amdgpu.ids | tail -12
R=$PWD; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/$OUT/prof_$c -o b -- python $R/bench.py --no-rows --no-cpu-baseline --no-live-pmc --no-host --steps 3 --warmup 1 > /dev/null 2>&1
done
cd $R
python tools/prof_summary.py --traffic "$(find $OUT/prof_FETCH_SIZE -name '*.db' | head -1)" "$(find $OUT/prof_WRITE_SIZE -name '*.db' | head -1)" $OUT/traffic.json 2>/dev/null | grep -i "welch_fused" 
rm -rf $OUT/prof_FETCH_SIZE $OUT/prof_WRITE_SIZE
