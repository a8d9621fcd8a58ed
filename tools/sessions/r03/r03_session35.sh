#!/bin/bash
# round 3, session 35: long-filter mode, fetched taps carried raw into the next group (default build) against masked next to the fetch
# (libmi355dsp_tc0.so: -DMDSP_FIR_TAP_CARRY=0), alternating processes on one box
for c in f32:1/8 c32:1/8 f64:1/8 c64:1/8 f32:1/16 f64:1/16 c32:1/16 c64:3/8 c64:1/4 f64:441/160; do
  dt=${c%%:*}; r=${c##*:}
  for tag in "" tc0 "" tc0; do
    echo -n "$dt $r [${tag:-carry}]  "
    MDSP_LIB_TAG=$tag TUNE_DTYPE=$dt TUNE_RATIO=$r TUNE_LOG2N=26 TUNE_ROUNDS=5 TUNE_FIR="1,0,0" timeout 200 python tools/tune_fir.py 2>&1 | grep "mm=" | awk '{print $4, $5}'
  done
done
