#!/bin/bash
# Round 4, GPU session 26: the whole GPU suite, then the full bench line (rows included) twice.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/s26; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
for i in 1 2; do
  timeout 900 python bench.py --no-cpu-baseline --no-live-pmc --no-host > $OUT/bench_$i.json 2> $OUT/bench_$i.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$i.json").read().strip().splitlines()[-1])
print($i, d["value"], d["ms_per_step"], d["config"].get("stages_ms"), {k:(v["ms"], v["frac"]) for k,v in d["kernels"].items() if isinstance(v,dict) and "ms" in v})
PY
done
