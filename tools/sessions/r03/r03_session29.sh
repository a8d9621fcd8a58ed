#!/bin/bash
# round 3, session 29: wave priority around the STFT kernel's loads (1) / stores (2) -- bench.py --config stft, alternating processes
for v in 0 1 2 3 0 1 2 3; do
  echo -n "MDSP_SPEC_PRIO=$v  "
  MDSP_SPEC_PRIO=$v timeout 200 python bench.py --config stft --steps 10 --warmup 3 --no-cpu-baseline --no-live-pmc --no-host 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('stages_ms'))"
done
