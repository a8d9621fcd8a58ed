#!/bin/bash
# Cache policy of the streaming buffer loads / stores (MDSP_IO_AUX_LOAD / MDSP_IO_AUX_STORE in csrc/devio.h): default vs nontemporal
# stores (nts) vs nontemporal loads + stores (ntls).  Build the variants first:
#   python dsp.jl_amd/build.py --tag nts  --cflags "-DMDSP_IO_AUX_STORE=2"
#   python dsp.jl_amd/build.py --tag ntls --cflags "-DMDSP_IO_AUX_STORE=2 -DMDSP_IO_AUX_LOAD=2"
for tag in "" nts ntls ""; do
  export MDSP_LIB_TAG=$tag
  echo "== lib tag '${tag}'"
  timeout 100 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value', d['value'], d['config']['stages_ms'], 'copy', d['kernels']['copy_float4_GBps'])
"
  ROWS_ENGINE=fused ROWS_REPS=5 timeout 200 python tools/bench_rows.py 2>&1 | grep "fused\|resample\|arbitrary" | cut -c1-75
done
