#!/bin/bash
# FIRArbitrary kernel: outputs per workgroup (MDSP_ARB_TILE) x channels per group (MDSP_ARB_NCH, swept by bench_arb.py)
for t in 1024 512; do echo "== tile $t"; MDSP_ARB_TILE=$t ARB_RATES=${ARB_RATES:-160/147} ARB_CHANNELS=${ARB_CHANNELS:-4} timeout 100 python tools/bench_arb.py 2>&1 | grep group; done
