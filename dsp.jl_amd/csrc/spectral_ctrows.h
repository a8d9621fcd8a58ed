// Welch sums at nfft = R0 x S in two kernels -- a column kernel into a work buffer, the single-workgroup Welch kernel over the rows (spectral_ctrows.hip).
// Internal to the library.
#pragma once

#include "common.h"
#include "spectral_ctcols.h"

namespace mdsp {
struct CtRowsPlan {   // what a Welch plan keeps for it
    bool ready = false;
    DevBuf rootsN, win, work, partial;
    CtColsPlan rows;   // the row kernel's tables (spectral_ctbig.hip)
};
// the smallest column factor R0 >= r0_min (a radix of fft_lds.h, up to 32) whose rows nfft / R0 have a single-workgroup schedule above 8192 points (0: none)
int ctrows_split(int dtype, int64_t nfft, int r0_min);
// acc[ch][nfft] (+)= sums of |X[k]|^2 over the K frames, natural order, Float64
int ctrows_welch(CtRowsPlan& rp, int dtype, int R0, const void* s, int64_t lds_, int64_t K, int64_t hop, int64_t nch, int n, int64_t nfft, const double* win_dev,
                 double* acc, bool fresh, hipStream_t st);
}  // namespace mdsp
