#!/bin/bash
# Round 5, session 3: the multi-pass engine after the first rework (factored twiddles, next item's samples prefetched, persistent tile lanes):
# parity again, default-argument bench (work-buffer and workgroup sweeps), kernel trace at 2^27.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r05s3; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_gpu_bigfft.py -x -q > $O/pytest_bigfft.log 2>&1; echo "pytest bigfft rc=$?" | tee -a $O/rc.txt
tail -3 $O/pytest_bigfft.log
DEFSPEC_CHUNKS=16,128,512,2048 DEFSPEC_LENGTHS=1048576,1000000,16777216,134217728 DEFSPEC_OUT=r05s3/defspec.json timeout 600 python tools/bench_default_spectral.py 2>&1 | grep -v amdgpu.ids | cut -c1-900
for w in 1 3 4; do MDSP_BIG_WGS=$w DEFSPEC_ENGINES=auto DEFSPEC_WELCH_ONLY=1 DEFSPEC_LENGTHS=16777216,134217728 DEFSPEC_OUT=r05s3/defspec_wgs$w.json timeout 300 python tools/bench_default_spectral.py 2>&1 | grep -v amdgpu.ids | cut -c1-300; done
L=134217728
cd /tmp
DEFSPEC_ENGINES=auto DEFSPEC_WELCH_ONLY=1 DEFSPEC_LENGTHS=$L DEFSPEC_OUT=r05s3/tmp.json timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$L -o p -- python $R/tools/bench_default_spectral.py > $R/$O/prof_$L.log 2>&1
cd $R
python tools/prof_summary.py $(find $O/prof_$L -name "*.db" | head -1) > $O/stats_$L.txt 2>&1; head -8 $O/stats_$L.txt | cut -c1-200
rm -rf $O/prof_$L
