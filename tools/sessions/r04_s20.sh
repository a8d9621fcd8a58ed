#!/bin/bash
# Round 4, GPU session 20: the thirteen composite-radix schedules kept (per-mode flags) against round 3's; parity first; the library's whole GPU suite
# for the spectral files.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/s20; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -k "mixed_radix or compile_time or welch or stft or spectrogram or periodogram" 2>&1 | tail -3
WIDE_SIZES=1200,1500,1920,2000,2400,2500,3000,3200,3840,4800,5000,6000,6400 REPS=7 OUT=s20/wide.json timeout 900 python tools/bench_wide.py 2>&1 | grep -v amdgpu.ids
