#!/bin/bash
# The GPU suite (or the tests a -k expression selects), smoke(), and the driver's bench line -> gpurun_out/suite/.
#   gpurun --timeout 3000 -- 'bash tools/sessions/suite.sh'            gpurun -- 'bash tools/sessions/suite.sh "filt or conv"'
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/suite; mkdir -p $O
export TMPDIR=/tmp
if [ $# -gt 0 ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q -k "$1" 2>&1 | tail -8 | tee $O/pytest_gpu.log
else
  timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
  timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; wc -c $O/bench.json
fi
