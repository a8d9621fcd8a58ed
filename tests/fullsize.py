"""Oracle evaluation at BASELINE sizes (test infrastructure; imports oracle/).

The CPU oracle materialises every frame / runs a Python loop per output sample, so at 2^28..2^30 samples it is applied
PIECEWISE: to chunks whose frames tile the reference's frame set exactly (Welch), to the frames of a window of columns
(STFT) and to a window of outputs whose filter state is set by the oracle's own closed form of the polyphase recurrence
(resample).  tests/test_fullsize_helpers.py pins each helper to the one-shot oracle on sizes where both run.
"""
from __future__ import annotations

import math
from fractions import Fraction

import numpy as np

from oracle import filt as oflt
from oracle import periodograms as opg
from oracle import stream_filt as osf


def oracle_welch_chunked(get, length: int, n: int, noverlap: int, window, chunk_frames: int = 1 << 14, fs: float = 1.0):
    """Float64 one-sided Welch PSD of a real stream of ``length`` samples, ``get(lo, hi)`` -> host array of samples [lo, hi).

    periodograms.jl:746-759 evaluates sum_k fft2pow(frame_k) / (K fs sum(w^2)); the sum over frames is split into chunks of
    ``chunk_frames`` consecutive frames (chunk c covers samples [k0*hop, (k1-1)*hop + n)) and each chunk is the oracle's own
    welch_pgram (Float64), weighted by its frame count."""
    hop = n - noverlap
    K = opg.frame_count(length, n, noverlap)
    acc = np.zeros(n // 2 + 1, dtype=np.float64)
    for k0 in range(0, K, chunk_frames):
        k1 = min(K, k0 + chunk_frames)
        seg = np.asarray(get(k0 * hop, (k1 - 1) * hop + n))
        assert opg.frame_count(len(seg), n, noverlap) == k1 - k0
        acc += opg.welch_pgram(seg, n, noverlap, window=window, fs=fs, dtype=np.float64).power.astype(np.float64) * (k1 - k0)
    return acc / K, K


def oracle_stft_columns(get, n: int, noverlap: int, f0: int, count: int, window, psdonly: bool = False, fs: float = 1.0):
    """Columns f0 .. f0+count-1 of stft / spectrogram of a stream (``get(lo, hi)`` -> host samples), Float64 arithmetic.
    Frame k starts at sample k*hop (periodograms.jl:57-69)."""
    hop = n - noverlap
    seg = np.asarray(get(f0 * hop, (f0 + count - 1) * hop + n))
    out = opg.stft(seg, n, noverlap, psdonly=psdonly, nfft=n, fs=fs, window=window,
                   dtype=np.float64)
    assert out.shape[1] == count
    return out


def resample_initial_state(h, ratio: Fraction):
    """(phi_idx, input_deficit) of the filter `resample` builds: FIRFilter(h, ratio) after undelay! (stream_filt.jl:706-714)."""
    f = osf.FIRFilter(np.asarray(h, dtype=np.float64), Fraction(ratio))
    f.setphase(f.timedelay())
    return f.phi_idx, f.input_deficit, f.taps_per_phi


def oracle_resample_window(get, nx: int, ratio: Fraction, h, m0: int, count: int):
    """Outputs m0 .. m0+count-1 (0-based) of resample(x, ratio, h) for a rational ratio with L, M > 1, in Float64.

    The state the reference's loop (stream_filt.jl:496-509) has when it writes output m0 is given in closed form by
    oracle.stream_filt.polyphase_closed_form; a fresh oracle filter is put in that state with an input deficit of
    tapsPerPhase (its window then lies wholly inside the slice handed to it, no history) and run on the slice of the
    zero-padded stream that the ``count`` outputs read.  ``get(lo, hi)`` -> host samples [lo, hi) of x, lo >= 0, hi <= nx."""
    ratio = Fraction(ratio)
    L, M = ratio.numerator, ratio.denominator
    h64 = np.asarray(h, dtype=np.float64)
    phi0, d0, tpp = resample_initial_state(h64, ratio)
    phi_m, idx_m = osf.polyphase_closed_form(phi0, d0, L, M, np.asarray([m0, m0 + count - 1]))
    first = int(idx_m[0]) - tpp            # 0-based index into the padded stream of the oldest sample output m0 reads
    last = int(idx_m[1])                   # one past the newest sample output m0+count-1 reads (1-based index = 0-based end)
    lo, hi = max(first, 0), min(last, nx)
    seg = np.zeros(last - first, dtype=np.float64)
    if hi > lo:
        seg[lo - first:hi - first] = np.asarray(get(lo, hi), dtype=np.float64)   # zeros in front (history) and behind (_zeropad)
    g = osf.FIRFilter(h64, ratio)
    g.phi_idx = int(phi_m[0])
    g.input_deficit = tpp
    y = g.filt(seg)
    assert len(y) >= count, (len(y), count)
    return y[:count]


def resample_output_length(nx: int, ratio: Fraction) -> int:
    return math.ceil(nx * Fraction(ratio))


def oracle_filt_chunked(get, nx: int, b, chunk: int = 1 << 22):
    """Float64 ``filt(b, x)`` of a whole stream (``get(lo, hi)`` -> host samples [lo, hi)), chunk by chunk: yields (lo, hi, y[lo:hi]).

    Filters/filt.jl:479-521 computes y[n] = sum_k b[k] x[n-k] with zero initial state; outputs [lo, hi) depend on x[lo - (nb-1), hi) only, so chunk c is the
    oracle's own fftfilt on that slice (zeros in front of the stream) with the first nb - 1 outputs dropped."""
    b64 = np.asarray(b, dtype=np.float64)
    nb = len(b64)
    for lo in range(0, nx, chunk):
        hi = min(nx, lo + chunk)
        first = lo - (nb - 1)
        seg = np.zeros(hi - first, dtype=np.float64)
        src_lo = max(first, 0)
        seg[src_lo - first:] = np.asarray(get(src_lo, hi), dtype=np.float64)
        y = oflt.fftfilt(b64, seg)
        yield lo, hi, y[nb - 1:]
