#!/bin/bash
# Round 4, GPU session 22: two-step reduction of the hand-allocated Welch kernel's rows; Welch parity, then the headline step twice.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/s22; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -k "welch" 2>&1 | tail -3
for i in 1 2; do
  timeout 600 python bench.py --no-rows --no-cpu-baseline --no-live-pmc --no-host > $OUT/bench_$i.json 2> $OUT/bench_$i.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$i.json").read().strip().splitlines()[-1])
print($i, d["value"], d["ms_per_step"], d["config"].get("stages_ms"))
PY
done
