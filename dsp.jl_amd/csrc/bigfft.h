// Multi-pass fused spectral engine for transforms no single workgroup holds (bigfft.hip; the passes are bigfft_pass.h).  Internal to the
// library: spectral.hip routes Welch / STFT / spectrogram / periodogram plans with a large nfft here.
#pragma once

#include "common.h"

namespace mdsp {
namespace big {

// nfft the engine plans: 7-smooth, above the single-workgroup kernels of spectral.hip, splitting into 2..4 factors of at most 512
bool size_ok(int dtype, int64_t nfft);

struct Engine;
struct EngineHolder {   // a plan's engines (created on first use; one holder per plan, so one owner and one stream at a time like the plan itself)
    Engine* p = nullptr;      // the passes form (natural-order spectra, 2 .. 4 passes)
    Engine* rows = nullptr;   // the rows form (column pass + single-workgroup row kernel): kept NEXT to the other -- a streaming plan whose chunk sizes
                              // vary takes either per call, and up to round 5 every flip freed one engine's work buffers and rebuilt the other's
    EngineHolder() = default;
    EngineHolder(const EngineHolder&) = delete;
    EngineHolder& operator=(const EngineHolder&) = delete;
    ~EngineHolder();
};

// Every entry: one channel.  `s` is the channel's first sample (dtype of the plan), K frames of n samples, `hop` apart, zero-padded to nfft;
// win_dev: n doubles on the device or nullptr.
//
// Welch: acc[k] (+)= sum over the frames of |Z[k]|^2, k < nfft, Float64 -- the pair-packed full spectrum for real signals (two frames per
// transform, folded by welch_finalize_kernel modes 3 / 4), the plain one for complex signals.  fresh: overwrite acc instead of adding.
int welch(EngineHolder& h, int dtype, int64_t n, int64_t nfft, const void* s, int64_t K, int64_t hop, const double* win_dev, double* acc, bool fresh,
          hipStream_t st);
// STFT / spectrogram / periodogram columns: out[f * ldo + j], j < nout -- raw spectra (complex) or |X|^2 m (real; accumulate: added to what is
// there), one- or two-sided exactly as the single-workgroup kernels produce them.
int stft(EngineHolder& h, int dtype, int64_t n, int64_t nfft, const void* s, int64_t K, int64_t hop, const double* win_dev, void* out, int64_t ldo,
         int64_t nout, int onesided, int psd, int accumulate, double r, hipStream_t st);

// Overlap-save with blocks no single workgroup holds (ols.hip routes filters beyond the partitioned kernels here).  ols_size: the transform
// length for a filter of nb taps and (a hint, 0: unknown) nout outputs per column (0: not this engine).  ols: blocks [g0, g1) of ONE column's block grid (block length N - nb + 1; g0 even for real
// signals) -- x / y are the column's (possibly virtual) bases as in mdsp_ols_exec_range, H the filter's full natural-order spectrum, 1 / N folded in.
int64_t ols_size(int dtype, int64_t nb, int64_t nout_hint);
// > 0: a transform of N points runs in the rows form (N = R0 x S, S = 8192 Float32 / 4096 Float64: column pass, one row kernel on the single-workgroup
// transforms of ols.hip, column pass back) and H is expected ROW-MAJOR in its layout, H'[k1 S + k2] = H[k1 + R0 k2]; 0: natural order, three passes each way
int ols_rows_r0(int dtype, int64_t N);
int ols(EngineHolder& h, int dtype, int64_t N, int R0 /* ols_rows_r0 when H was laid out */, const void* x, int64_t nx, const void* H, int64_t nb, void* y, int64_t nout,
        int64_t g0, int64_t g1, hipStream_t st);

}  // namespace big
}  // namespace mdsp
