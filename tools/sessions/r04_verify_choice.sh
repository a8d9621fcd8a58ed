#!/bin/bash
# Round 4: MDSP_FIR_MM_VERIFY_CHOICE over the polyphase shapes bench.py times -- the library's choice of form against round 2's rule on THIS box.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/verify; mkdir -p $OUT
export TMPDIR=/tmp
run() {  # ratio dtype log2n taps
  MDSP_FIR_MM_VERIFY_CHOICE=1 TUNE_RATIO=$1 TUNE_DTYPE=$2 TUNE_LOG2N=$3 TUNE_TAPS=$4 TUNE_ROUNDS=5 timeout 300 python tools/tune_fir.py 2>&1 | grep -E "^(ok|REGRESSION)"
  cp gpurun_out/tune_fir.json $OUT/$(echo $1 | tr / _)_$2.json 2>/dev/null
}
run 160/147 f32 28 5120
run 160/147 f64 26 0
run 160/147 c32 26 0
run 2/1 f32 26 0
run 1/2 f32 26 0
run 1/8 f32 26 0
run 147/160 f32 26 0
run 160/441 f64 26 0
python - <<'PY'
import glob, json, os
out = {}
for f in sorted(glob.glob("gpurun_out/verify/*.json")):
    d = json.load(open(f))
    out[os.path.basename(f)[:-5]] = {"ratio": d["ratio"], "dtype": d["dtype"], **d.get("verify_choice", {})}
json.dump(out, open("gpurun_out/verify/summary.json", "w"), indent=1)
PY
