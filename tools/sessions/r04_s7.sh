#!/bin/bash
# Round 4, GPU session 7: the whole -m gpu suite with the hand-allocated Welch kernel as the default of the headline shape and fir.hip built without
# the SI load/store optimizer; then the bench line twice (no live PMC).
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/s7; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1; tail -6 $OUT/pytest_gpu.log; grep -h "element-wise error" $OUT/pytest_gpu.log
for rep in 1 2; do
  timeout 900 python bench.py --steps 20 --warmup 5 --no-live-pmc --no-cpu-baseline --no-host > $OUT/bench_$rep.json 2> $OUT/bench_$rep.err
  python - <<PY
import json
d=json.load(open("$OUT/bench_$rep.json"))
k=d["kernels"]
print($rep, d["value"], d["ms_per_step"], d["config"].get("stages_ms"), {n:(v["ms_per_launch"], v["frac"]) for n,v in k.items() if isinstance(v,dict) and "ms_per_launch" in v})
PY
done
