#!/bin/bash
# Round 4, GPU session 3: the whole -m gpu suite on the fused-statement build (new tests: element-wise ulp bounds, non-finite samples, plan-cache
# churn, Welch variants 40 / 41), then bench.py (new rows) on the fused build and on the -DMDSP_PK_FUSED=0 build, interleaved twice.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/s3; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q -s > $OUT/pytest_gpu.log 2>&1; tail -15 $OUT/pytest_gpu.log; grep -h "element-wise error" $OUT/pytest_gpu.log
for rep in 1 2; do
  for tag in "" nofuse; do
    MDSP_LIB_TAG=$tag timeout 900 python bench.py --steps 20 --warmup 5 --no-live-pmc --no-cpu-baseline --no-host > $OUT/bench_${tag:-fused}_$rep.json 2> $OUT/bench_${tag:-fused}_$rep.err
    python - <<PY
import json
d=json.load(open("$OUT/bench_${tag:-fused}_$rep.json"))
k=d["kernels"]
print("${tag:-fused}", $rep, d["value"], d["ms_per_step"], d["config"].get("stages_ms"), {n:(v["ms_per_launch"], v["frac"]) for n,v in k.items() if isinstance(v,dict) and "ms_per_launch" in v})
PY
  done
done
