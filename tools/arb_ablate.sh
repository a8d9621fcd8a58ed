#!/bin/bash
# Where the FIRArbitrary kernel's time goes: MDSP_ABLATE bits 1 (no phase-A replay), 2 (one tap), 4 (no staging loads), 8 (no tap copy)
# needs the debug-knob build: python dsp.jl_amd/build.py --tag dbg --cflags -DMDSP_DEBUG_KNOBS
export MDSP_LIB_TAG=dbg
for a in 0 1 2 4 8 3 6 15; do echo "== MDSP_ABLATE=$a"; MDSP_ABLATE=$a MDSP_ARB_NCH=4 ARB_RATES=160/147 ARB_CHANNELS=4 timeout 100 python tools/bench_arb.py 2>&1 | grep "group=4" | cut -c1-120; done
