#!/usr/bin/env python3
"""Pack the reference's own golden data files for the hot path into one compact fixture.

Run in the authoring container (where ``/root/reference`` is mounted):

    python tests/golden/make_golden.py

Reads the tab/newline-delimited decimal text files that DSP.jl's tests load with
``read_reference_data`` (``test/FilterTestHelpers.jl:8``) and writes ``tests/golden/dsp_golden.npz``.
The GPU box has no ``/root/reference``; tests there read only the committed ``.npz``.

Files and the reference tests that consume them:
  spectrogram_{x,p,f,t}.txt           test/periodograms.jl:25-36   (MATLAB spectrogram)
  stft_x.txt, stft_S_{real,imag}.txt  test/periodograms.jl:332-344 (MATLAB stft)
  resample_x.txt, resample_taps_*.txt, resample_y_*.txt   test/resample.jl:8-24 (MATLAB resample)
  hanning128.txt                      test/windows.jl:55-59
  digitalfilter_hamming_12{8,9}_lowpass[_scaled]_fc0.25_fs1.0.txt   test/filter_design.jl:988-1060 (SciPy firwin)
  dpss128,4.txt                       test/windows.jl:34-36 (MATLAB dpss)
  {hamming,triang,bartlett,bartlett_hann,blackman,blackmanharris_*,nuttall_*,kaiser,flattop,gaussian,tukey,lanczos,cosine}128*.txt   test/windows.jl:45-128
  mt_pgram.txt, pmtm_{x,fx,pxx}.txt   test/periodograms.jl:381-440 (MATLAB pmtm)
  csd_array_multitaper_{frequencies,values_re,values_im}.txt, noise.txt   test/multitaper.jl:254-300 (MNE-Python)
"""
import os
import sys

import numpy as np

REF = os.environ.get("DSP_REFERENCE", "/root/reference")
DATA = os.path.join(REF, "test", "data")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dsp_golden.npz")

FILES = [
    "spectrogram_x", "spectrogram_p", "spectrogram_f", "spectrogram_t",
    "stft_x", "stft_S_real", "stft_S_imag",
    "resample_x",
    "resample_taps_1_2", "resample_taps_2_1", "resample_taps_3_2", "resample_taps_2_3",
    "resample_y_1_2", "resample_y_2_1", "resample_y_3_2", "resample_y_2_3",
    "hanning128",
    "digitalfilter_hamming_128_lowpass_fc0.25_fs1.0",
    "digitalfilter_hamming_128_lowpass_scaled_fc0.25_fs1.0",
    "digitalfilter_hamming_129_lowpass_fc0.25_fs1.0",
    "digitalfilter_hamming_129_lowpass_scaled_fc0.25_fs1.0",
    "dpss128,4", "mt_pgram", "pmtm_x", "pmtm_fx", "pmtm_pxx",
    "csd_array_multitaper_frequencies", "csd_array_multitaper_values_re", "csd_array_multitaper_values_im", "noise",
    "hamming128", "triang128", "bartlett128", "bartlett_hann128", "blackman128", "blackmanharris_3term_128", "blackmanharris_4term_128",
    "nuttall_3term_128", "nuttall_4term_128", "kaiser128,0.4", "flattop", "gaussian128,0.2", "tukey128,0.4", "lanczos128", "cosine128",
]


def read_reference_data(name):
    rows = []
    with open(os.path.join(DATA, name + ".txt")) as f:
        for line in f:
            line = line.strip()
            if line:
                rows.append([float(tok) for tok in line.replace(",", " ").split()])
    a = np.array(rows, dtype=np.float64)
    return a[:, 0] if a.shape[1] == 1 else (a[0] if a.shape[0] == 1 else a)


def main():
    if not os.path.isdir(DATA):
        sys.exit(f"reference data directory not found: {DATA}")
    arrays = {}
    for name in FILES:
        key = name.replace(".", "p").replace(",", "_")
        arrays[key] = read_reference_data(name)
        print(f"{name:60s} {arrays[key].shape}")
    np.savez_compressed(OUT, **arrays)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
