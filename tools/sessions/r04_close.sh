#!/bin/bash
# Round 4, closing session (the LAST GPU session of the round; `git rev-parse HEAD > .build_commit` before sending): the whole GPU suite, smoke(),
# tools/gpu_profile.sh, then the shapes that round 4's last-resort tiles moved from the generic to the matrix-core polyphase kernel, each with
# MDSP_FIR_MM_TIGHT=0 (the generic kernel, as before) next to the default -> gpurun_out/close/.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/close; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash tools/gpu_profile.sh > $O/gpu_profile.log 2>&1; tail -4 $O/gpu_profile.log
for sh in "c32 160/441" "c64 1/16" "c64 1/32" "f64 1/32"; do
  set -- $sh
  for tight in 0 1; do
    MDSP_FIR_MM_TIGHT=$tight TUNE_DTYPE=$1 TUNE_RATIO=$2 TUNE_LOG2N=26 TUNE_ROUNDS=3 TUNE_FIR="-1,0,0" timeout 120 python tools/tune_fir.py > /dev/null 2>&1
    cp gpurun_out/tune_fir.json $O/${1}_${2/\//_}_tight$tight.json 2>/dev/null
  done
done
python - <<'PY'
import glob, json, os
out = {}
for f in sorted(glob.glob("gpurun_out/close/*_tight*.json")):
    d = json.load(open(f)); v = list(d["variants"].values())[0]
    out[os.path.basename(f)[:-5]] = {"taps": d["taps"], "median_ms": v["median_ms"], "GBps": v["GBps"], "frac_of_8TBps": round(v["GBps"] / 8000, 3)}
json.dump(out, open("gpurun_out/close/tight_summary.json", "w"), indent=1)
print(json.dumps(out))
PY
