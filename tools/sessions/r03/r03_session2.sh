#!/bin/bash
# Round-3 GPU session 2: new boundary tests (plan cache per thread, stft / fir host pipelines), Welch phase profile, host-path rates.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
echo "== boundary tests"; timeout 900 python -m pytest tests/test_gpu_boundary.py -m gpu -x -q 2>&1 | tail -25 > $OUT/s2_gpu_tests.log; tail -6 $OUT/s2_gpu_tests.log
echo "== welch phase profile (2 WG/CU = default)"; MDSP_LIB_TAG=prof TUNE_LOG2N=30 TUNE_ROUNDS=3 TUNE_OLS= TUNE_WELCH=30,32 TUNE_WGS=2 TUNE_RUNS=1 timeout 300 python tools/tune.py > $OUT/s2_prof2.log 2>&1; grep WELCHPROF $OUT/s2_prof2.log | tail -8
echo "== welch phase profile (1 WG/CU)"; MDSP_WG_PER_CU=1 MDSP_LIB_TAG=prof TUNE_LOG2N=30 TUNE_ROUNDS=3 TUNE_OLS= TUNE_WELCH=30 TUNE_WGS=1 TUNE_RUNS=1 timeout 300 python tools/tune.py > $OUT/s2_prof1.log 2>&1; grep WELCHPROF $OUT/s2_prof1.log | tail -4
echo "== bench (host path incl.)"; timeout 600 python bench.py --steps 10 --warmup 3 --no-rows --no-cpu-baseline --no-live-pmc > $OUT/s2_bench.json 2> $OUT/s2_bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/s2_bench.json')); print(d['value'], d['config']['stages_ms']); print(json.dumps(d.get('host_path'), indent=1))
PY
