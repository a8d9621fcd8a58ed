// Welch sums at nfft = R0 x S with a compile-time schedule for the S-point rows (spectral_ctcols.hip).  Internal to the library.
#pragma once

#include "common.h"

namespace mdsp {
struct CtColsPlan {   // what a Welch plan keeps for it: roots of S and of nfft, the window in working precision
    bool ready = false;
    DevBuf roots, rootsN, win;
};
// the column factor R0 in 2 .. 8 with nfft / R0 among the instantiated row sizes (0: none)
int ctcols_split(int dtype, int64_t nfft);
// partial[(group, ch)][nfft] (+ reduce by the caller): Float64 sums of |Z|^2 per bin, natural order
int ctcols_welch(CtColsPlan& cp, int dtype, const void* s, int64_t lds_, int64_t K, int64_t hop, int64_t nch, int n, int64_t nfft, const double* win_dev,
                 hipStream_t st, int64_t* ngroups, DevBuf* partial);
// spectral_ctcols_big.hip: rows of 8193 .. 16384 points (the single-workgroup schedules of ctbig_sizes.h); ctcols_split / ctcols_welch route to them
bool ctcols_big_row_ok(int dtype, int64_t S);
int ctcols_big_welch(CtColsPlan& cp, int dtype, const void* s, int64_t lds_, int64_t K, int64_t hop, int64_t nch, int n, int64_t nfft, int R0,
                     const double* win_dev, hipStream_t st, int64_t* ngroups, DevBuf* partial);
// spectral_ctcols_f64.hip: Float64 / ComplexF64 rows of 4097 .. 9600 points (the same tables), R0 = 2 .. 8
bool ctcols64_row_ok(int64_t S);
int ctcols64_welch(CtColsPlan& cp, bool cplx, const void* s, int64_t lds_, int64_t K, int64_t hop, int64_t nch, int n, int64_t nfft, int R0, const double* win_dev,
                   hipStream_t st, int64_t* ngroups, DevBuf* partial);
// spectral_ctbig.hip: nfft between 8193 and 16384 points with a single-workgroup compile-time schedule (Float32 / ComplexF32); cp holds the nfft roots
bool ctbig_ok(int dtype, int64_t nfft);
bool ctbig_preferred(int dtype, int64_t nfft);   // a size that also has an all-mode compile-time schedule, Welch sums faster here
// (accumulate: the partial rows of an earlier launch with at least as many slots are added to instead of overwritten -- spectral_ctrows.hip)
int ctbig_welch(CtColsPlan& cp, int dtype, const void* s, int64_t lds_, int64_t K, int64_t hop, int64_t nch, int n, int64_t nfft, const double* win_dev,
                hipStream_t st, int64_t* nslots, DevBuf* partial, int accumulate = 0);
// spectral_ctbig_cols.hip: STFT / spectrogram / periodogram columns on the same single-workgroup schedules (complex signals to 16384 points, real ones to 9600)
struct CtBigColsArgs {
    const void* s;
    void* out;
    const double* win;   // n doubles or nullptr (the window of THIS launch)
    int64_t len, lds_, K, hop, nch, ldo, chs;
    int n;
    int64_t nfft;
    int nout, onesided, psd, accumulate;
    double r;
};
bool ctbig_cols_ok(int dtype, int64_t nfft);
int ctbig_stft(CtColsPlan& cp, int dtype, const CtBigColsArgs& a, hipStream_t st);
// spectral_ctbig_f64.hip: the same for Float64 / ComplexF64 (sizes whose one buffer of 16-byte elements fits: to 9600 points); ctbig_ok / ctbig_welch / ctbig_cols_ok /
// ctbig_stft route there
bool ctbig64_ok(int64_t nfft);
int ctbig64_welch(CtColsPlan& cp, bool cplx, const void* s, int64_t lds_, int64_t K, int64_t hop, int64_t nch, int n, int64_t nfft, const double* win_dev, hipStream_t st,
                  int64_t* nslots, DevBuf* partial, int accumulate);
int ctbig64_stft(CtColsPlan& cp, bool cplx, const CtBigColsArgs& c, hipStream_t st);
}  // namespace mdsp
