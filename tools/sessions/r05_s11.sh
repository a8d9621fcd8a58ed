#!/bin/bash
# Round 5, session 11: the whole GPU suite after the round's changes so far, then bench.py with the new compact line and rows.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s11; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" | tee -a $O/rc.txt
tail -15 $O/pytest_gpu.log | cut -c1-300
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
wc -c $O/bench.json; tail -3 $O/bench.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05s11/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value","ms_per_step")}, d["config"].get("stage_ms_filt"), d["config"].get("stage_ms_welch"))
print("roofline", {k:v for k,v in d["roofline"].items() if not k.startswith("f_")})
for k,v in d["kernels"].items(): print(k, v)
print("traffic_source:", d.get("traffic_source"))
PY
