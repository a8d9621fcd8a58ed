#!/bin/bash
# rocFFT-engine STFT / spectrogram (config 4 share) against the chunk size of its intermediates
for c in 32 64 128 192 256 512 1024; do
  echo "== MDSP_ROCFFT_CHUNK_MIB=$c"
  MDSP_ROCFFT_CHUNK_MIB=$c ROWS_REPS=3 ROWS_ONLY=stft ROWS_ENGINE=rocfft timeout 120 python tools/bench_rows.py 2>&1 | grep -i "rocfft" | cut -c1-220
done
