#!/bin/bash
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
echo "== full gpu suite"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $OUT/s8_gpu_tests.log; tail -5 $OUT/s8_gpu_tests.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/s8_bench.json 2> $OUT/s8_bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/s8_bench.json')); print(d['value'], d['ms_per_step'], d['config']['stages_ms'], d['roofline']['frac'], d['kernels']['other']['frac'])
PY
