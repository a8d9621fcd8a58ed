#!/bin/bash
# Round 4, GPU session 4: the whole -m gpu suite (no -x), then bench.py on the default build and on the build without the SI load/store optimizer
# (-target-feature -load-store-opt: no ds_read2_b64 merging, which halves the LDS read rate of the merged pairs), interleaved twice.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/s4; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1; tail -15 $OUT/pytest_gpu.log; grep -h "element-wise error" $OUT/pytest_gpu.log
for rep in 1 2; do
  for tag in "" nolso; do
    MDSP_LIB_TAG=$tag timeout 900 python bench.py --steps 20 --warmup 5 --no-live-pmc --no-cpu-baseline --no-host > $OUT/bench_${tag:-default}_$rep.json 2> $OUT/bench_${tag:-default}_$rep.err
    python - <<PY
import json
d=json.load(open("$OUT/bench_${tag:-default}_$rep.json"))
k=d["kernels"]
print("${tag:-default}", $rep, d["value"], d["ms_per_step"], d["config"].get("stages_ms"), {n:(v["ms_per_launch"], v["frac"]) for n,v in k.items() if isinstance(v,dict) and "ms_per_launch" in v})
PY
  done
done
MDSP_LIB_TAG=nolso timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py -m gpu -q -x -k "firarb or mixed or compile_time or welch or resample or polyphase" > $OUT/pytest_nolso.log 2>&1; tail -4 $OUT/pytest_nolso.log
