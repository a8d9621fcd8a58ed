#!/bin/bash
# Round 4, GPU session 23: two-step slice reduction for every Welch path (reduce_partials); spectral parity, nextfastfft sizes, headline step.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/s23; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -k "welch or mixed_radix or compile_time or periodogram or multitaper or mt_" 2>&1 | tail -3
WIDE_SIZES=1000,1536,2000,3000,4096,6000 REPS=7 OUT=s23/wide.json timeout 900 python tools/bench_wide.py 2>&1 | grep -v amdgpu.ids
for i in 1 2; do
  timeout 600 python bench.py --no-rows --no-cpu-baseline --no-live-pmc --no-host > $OUT/bench_$i.json 2> $OUT/bench_$i.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$i.json").read().strip().splitlines()[-1])
print($i, d["value"], d["ms_per_step"], d["config"].get("stages_ms"))
PY
done
MDSP_WELCH_VARIANT=30 timeout 600 python bench.py --no-rows --no-cpu-baseline --no-live-pmc --no-host 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant 30', d['value'], d['ms_per_step'], d['config'].get('stages_ms'))"
