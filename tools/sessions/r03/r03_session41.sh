#!/bin/bash
# round 3, session 41: the headline step through bench.py with the load priority of the two kernels off / on, alternating processes on one box
for v in "0 0" "1 1" "0 0" "1 1" "0 1" "1 0"; do
  set -- $v
  echo -n "MDSP_OLS_PRIO=$1 MDSP_SPEC_PRIO=$2  "
  MDSP_OLS_PRIO=$1 MDSP_SPEC_PRIO=$2 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rows --no-host --no-live-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['stages_ms'])"
done
