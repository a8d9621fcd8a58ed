#!/bin/bash
# One GPU session: any list of commands, run from the repository root with TMPDIR=/tmp and an output directory of their own.
#   gpurun --timeout 1500 -- "bash tools/sessions/run.sh r05s26 'MDSP_BIG_WGS=4 BIGOLS_OUT=\$O/wgs4.json python tools/check_big_ols.py' 'python bench.py > \$O/bench.json'"
# $O = gpurun_out/NAME (what gpurun merges back).  tools/sessions/LOG.md has every session of rounds 3-5 in this form.
set -u
cd "$(dirname "$0")/../.."
NAME=${1:?session name}; shift
export O=gpurun_out/$NAME TMPDIR=/tmp
mkdir -p "$O"
for cmd in "$@"; do
  echo "== $cmd"
  timeout "${SESSION_CMD_TIMEOUT:-1500}" bash -c "$cmd" 2>&1 | grep -v amdgpu.ids
done
