#!/bin/bash
# Round 4: every ratio x signal type of the r03n table once more (the matrix-core polyphase kernel as the library chooses it; fir.hip now built without the
# SI load/store optimizer), 4 channels x 2^26 samples, median of 3 -> gpurun_out/r04n/summary.json (with round 3's numbers next to each).
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r04n; mkdir -p $O
export TMPDIR=/tmp
for dt in f32 f64 c32 c64; do
  for r in 160/147 147/160 2/1 1/2 3/2 2/3 5/3 4/1 1/3 1/4 1/8 1/16 3/8 160/441 441/160; do
    TUNE_DTYPE=$dt TUNE_RATIO=$r TUNE_LOG2N=26 TUNE_ROUNDS=3 TUNE_FIR="-1,0,0" timeout 100 python tools/tune_fir.py > /dev/null 2>&1
    cp gpurun_out/tune_fir.json $O/${dt}_${r/\//_}.json 2>/dev/null
  done
done
python - <<'PY'
import glob, json, os
old = json.load(open("profiles/r03n_fir_all_ratios.json"))["ratios"]
out = {}
for f in sorted(glob.glob("gpurun_out/r04n/*_*.json")):
    k = os.path.basename(f)[:-5]
    d = json.load(open(f))
    v = list(d["variants"].values())[0]
    out[k] = {"taps": d["taps"], "median_ms": v["median_ms"], "GBps": v["GBps"], "frac_of_8TBps": round(v["GBps"] / 8000, 3),
              "round3_ms": old.get(k, {}).get("median_ms"), "ratio_to_round3": round(v["median_ms"] / old[k]["median_ms"], 3) if k in old else None}
json.dump({"note": "tools/sessions/r04_all_ratios.sh at the round's last library commit; 4 channels x 2^26 samples, median of 3, one box (boxes differ by +-5 %); round3_ms = profiles/r03n_fir_all_ratios.json (another box)", "ratios": out}, open("gpurun_out/r04n/summary.json", "w"), indent=1)
rs = [v["ratio_to_round3"] for v in out.values() if v["ratio_to_round3"]]
print(len(out), "shapes; time vs round 3: min %.3f median %.3f max %.3f" % (min(rs), sorted(rs)[len(rs) // 2], max(rs)))
PY
