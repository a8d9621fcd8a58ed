#!/bin/bash
# Round 5, session 28: whole GPU suite + smoke + the driver's bench line (a mid-round check before the closing session).
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s28; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; wc -c $O/bench.json
