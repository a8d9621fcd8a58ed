#!/bin/bash
# Round 4, evidence only (no library change): is the SI load/store optimizer (off for fir.hip since this round: merged ds_read2_b64 run at half rate)
# what the four Float32 / ComplexF32 polyphase shapes that read 9 - 17 % slower than round 3 in r04_fir_all_ratios.json pay for?  The product library
# against a variant built with the pass ON for fir.hip (python: build.EXTRA_CFLAGS = {}; build.build(tag="lso")), same box, alternating.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/lso; mkdir -p $O
export TMPDIR=/tmp
for sh in "f32 147/160" "f32 1/16" "c32 147/160" "f32 160/441" "f32 2/1" "f32 160/147"; do
  set -- $sh
  for tag in "" lso; do
    MDSP_LIB_TAG=$tag TUNE_DTYPE=$1 TUNE_RATIO=$2 TUNE_LOG2N=26 TUNE_ROUNDS=3 TUNE_FIR="-1,0,0" timeout 60 python tools/tune_fir.py > /dev/null 2>&1
    cp gpurun_out/tune_fir.json $O/${1}_${2/\//_}_${tag:-off}.json 2>/dev/null
  done
done
python - <<'PY'
import glob, json, os
out = {}
for f in sorted(glob.glob("gpurun_out/lso/*.json")):
    if f.endswith("summary.json"): continue
    d = json.load(open(f)); v = list(d["variants"].values())[0]
    out[os.path.basename(f)[:-5]] = {"taps": d["taps"], "median_ms": v["median_ms"], "GBps": v["GBps"]}
json.dump(out, open("gpurun_out/lso/summary.json", "w"), indent=1)
for k, v in out.items(): print(k, v["median_ms"])
PY
