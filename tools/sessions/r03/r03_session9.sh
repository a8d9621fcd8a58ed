#!/bin/bash
# Round-3 closing profile at HEAD: rocprofv3 kernel stats + PMC traffic + SQ counters of bench.py (headline + rows), and the --config lines.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
bash tools/gpu_profile.sh > $OUT/s9_profile.log 2>&1; tail -30 $OUT/s9_profile.log
for c in stft resample; do
  timeout 600 python bench.py --config $c --steps 10 --warmup 3 > $OUT/s9_config_$c.json 2> $OUT/s9_config_$c.err; cut -c1-300 $OUT/s9_config_$c.json
done
