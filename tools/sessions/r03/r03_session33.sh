#!/bin/bash
# round 3, session 33: partitioned overlap-save, loads at normal / raised wave priority (MDSP_OLS_PRIO is read per launch: alternate processes)
for v in 0 1 0 1; do
  echo "MDSP_OLS_PRIO=$v"
  MDSP_OLS_PRIO=$v LONGFILT_VARIANTS=0 LONGFILT_TAPS=5120,6000 LONGFILT_DTYPES=float32 LONGFILT_NO_ROCFFT=1 timeout 300 python tools/bench_longfilt.py 2>&1 | grep float32 | sed "s/.*'ms': \([0-9.]*\).*GBps_algorithmic': \([0-9.]*\).*/    \1 ms \2 GB\/s/"
done
