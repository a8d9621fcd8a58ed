#!/bin/bash
# Round 5, session 1: (a) the multi-pass spectral engine meets hardware: parity tests, then the default-argument bench with a work-buffer sweep;
# (b) A/B of the merged r05-prep branch (padded runs for long windows / fetched taps, tie-break by multiplying waves) over all 60 polyphase cells.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s1; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bigfft.py -x -q -k "not 2p27" > $O/pytest_bigfft.log 2>&1; echo "pytest bigfft rc=$?" | tee -a $O/rc.txt
tail -15 $O/pytest_bigfft.log
DEFSPEC_CHUNKS=16,32,128,256 DEFSPEC_LENGTHS=1048576,1000000,16777216,134217728 timeout 600 python tools/bench_default_spectral.py > $O/defspec.log 2>&1; echo "defspec rc=$?" | tee -a $O/rc.txt
cp gpurun_out/default_spectral.json $O/ 2>/dev/null; tail -6 $O/defspec.log
FIRR_VARIANTS="default;MDSP_FIR_MM_RPX=1;MDSP_FIR_MM_TIEWAVES=1;MDSP_FIR_MM_RPX=1,MDSP_FIR_MM_TIEWAVES=1" FIRR_OUT=r05s1/fir_ratios_ab.json timeout 900 python tools/bench_fir_ratios.py > $O/fir_ab.log 2>&1; echo "fir ab rc=$?" | tee -a $O/rc.txt
tail -62 $O/fir_ab.log | cut -c1-220
