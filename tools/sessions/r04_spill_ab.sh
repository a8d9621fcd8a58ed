#!/bin/bash
# Round 4, evidence only: every matrix-core polyphase instantiation IN USE that spills VGPRs (llvm-readelf notes of fir.hip at
# 9dd1442: <float,1,1,96> 16, <float,2,1,96> 16, <double,1,1,48> 13, <double,1,1,16,.,3> 13, <double,1,1,40> 2, <float,1,4,48> 1) against the nearest
# non-spilling form in the same process: taps fetched per tile (MDSP_FIR_MM_T64=0) / per-tile taps for several column blocks (MDSP_FIR_MM_NBLK=0).
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/spill; mkdir -p $O
export TMPDIR=/tmp
D="-1,0,0"; T0="-1,0,0,0,0,0,0,-1,-1,1,0,-1,0,1"; N0="-1,0,0,0,0,0,0,-1,-1,1,0,-1,1,0"
for sh in "f32 1/8 $D;$T0" "c32 1/8 $D;$T0" "f64 1/4 $D;$T0" "f64 160/441 $D;$T0" "f32 3/8 $D;$T0" "f64 441/160 $D;$N0"; do
  set -- $sh
  TUNE_DTYPE=$1 TUNE_RATIO=$2 TUNE_LOG2N=26 TUNE_ROUNDS=3 TUNE_FIR="$3" timeout 60 python tools/tune_fir.py > /dev/null 2>&1
  cp gpurun_out/tune_fir.json $O/${1}_${2/\//_}.json 2>/dev/null
done
python - <<'PY'
import glob, json, os
out = {}
for f in sorted(glob.glob("gpurun_out/spill/*_*.json")):
    d = json.load(open(f))
    out[os.path.basename(f)[:-5]] = {"taps": d["taps"], **{k: v["median_ms"] for k, v in d["variants"].items()}}
json.dump(out, open("gpurun_out/spill/summary.json", "w"), indent=1)
for k, v in out.items(): print(k, v)
PY
