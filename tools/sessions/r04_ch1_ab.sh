#!/bin/bash
# Round 4, evidence only (no library change): decimating shapes with single-chunk multiplying waves (MDSP_FIR_MM_CH=1: more, smaller waves per tile --
# more k-steps in flight per SIMD) against the tile the cost line of fir_mm_geo picks.  tools/tune_fir.py, 4 channels x 2^26 samples, median of 3.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/ch1; mkdir -p $O
export TMPDIR=/tmp
for sh in "f32 1/16" "f32 3/8" "c32 3/8" "f32 1/4" "c32 1/4" "c32 1/8"; do
  set -- $sh
  TUNE_DTYPE=$1 TUNE_RATIO=$2 TUNE_LOG2N=26 TUNE_ROUNDS=3 TUNE_FIR="-1,0,0;-1,0,0,0,0,0,1" timeout 60 python tools/tune_fir.py > /dev/null 2>&1
  cp gpurun_out/tune_fir.json $O/${1}_${2/\//_}.json 2>/dev/null
done
python - <<'PY'
import glob, json, os
out = {}
for f in sorted(glob.glob("gpurun_out/ch1/*_*.json")):
    d = json.load(open(f))
    out[os.path.basename(f)[:-5]] = {"taps": d["taps"], **{k: v["median_ms"] for k, v in d["variants"].items()}}
json.dump(out, open("gpurun_out/ch1/summary.json", "w"), indent=1)
for k, v in out.items(): print(k, v)
PY
