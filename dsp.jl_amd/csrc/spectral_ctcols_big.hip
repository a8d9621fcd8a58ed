// Welch sums at nfft = R0 x S with rows of 8193 .. 16384 points (round 6): the row kernel of spectral_ctcols.hip instantiated over the single-workgroup
// schedules of spectral_ctbig.hip (ctbig_sizes.h), every one in its lean form -- Float32 sums flushed to the Float64 partials, column twiddles loaded beside
// the samples, twiddles as products of ~2 sqrt(R) table values -- one workgroup per CU.  32768 = 2 x 16384, 25000 = 2 x 12500, 65536 = 4 x 16384: half the
// column factor of the 8192-point rows, i.e. half the reads of a frame per point (R0 workgroups read every frame; R0 - 1 of the reads come from the L2).
// Float32 / ComplexF32.  Reference loops: periodograms.jl:746-759 (welch_pgram_helper!), :57-69 (ArraySplit), :142-172 (fft2pow!).
#include <algorithm>

#include "common.h"
#include "devio.h"
#include "fft_lds.h"
#include "hostfft.h"
#include "spectral_ctcols.h"

using namespace mdsp;
using mdsp::fft::cx;

namespace {
#include "spectral_gen.h"

#include "spectral_ctcols_kernel.h"
#include "ctbig_sizes.h"

#ifndef MDSP_ROW_TOUCH
#define MDSP_ROW_TOUCH 32768   // every row touches its share of the next unit's span into the L2 (A/B r06s40)
#endif
constexpr int ROW_FLAGS = 4096 | 8192 | 16384 | MDSP_ROW_TOUCH;   // lean rows whatever form the size takes as a whole transform

template <typename R, bool CPLX> int big_rows_dispatch(ColsArgs& ca, int64_t nch, hipStream_t st, int64_t* ngroups, DevBuf* partial) {
    switch (ca.g.N) {
#define MDSP_X(N, T, F, ...) \
    case N: return cols_launch<R, CPLX, CtSched<N, T, ((F) & ~32768) | ROW_FLAGS, __VA_ARGS__>>(ca, nch, st, ngroups, partial);
        MDSP_CTBIG_SIZES(MDSP_X)
        MDSP_CTBIG_LEAN_SIZES(MDSP_X)
#undef MDSP_X
        default: break;
    }
    MDSP_FAIL(MDSP_ERR_ASSERTION, "no compile-time row schedule of %d points", ca.g.N);
}
}  // namespace

namespace mdsp {
bool ctcols_big_row_ok(int dtype, int64_t S) {
    if (dtype_is_double(dtype)) return false;
    switch (S) {
#define MDSP_X(N, ...) case N:
        MDSP_CTBIG_SIZES(MDSP_X)
        MDSP_CTBIG_LEAN_SIZES(MDSP_X)
#undef MDSP_X
        return true;
        default: return false;
    }
}

int ctcols_big_welch(CtColsPlan& cp, int dtype, const void* s, int64_t lds_, int64_t K, int64_t hop, int64_t nch, int n, int64_t nfft, int R0,
                     const double* win_dev, hipStream_t st, int64_t* ngroups, DevBuf* partial) {
    const int64_t S = nfft / R0;
    if (nfft % R0 || !ctcols_big_row_ok(dtype, S)) MDSP_FAIL(MDSP_ERR_ASSERTION, "nfft=%lld is not %d x a single-workgroup row size", (long long)nfft, R0);
    const bool cplx = dtype_is_complex(dtype);
    if (!cp.ready) {
        MDSP_TRY(upload_roots_n<float>(cp.roots, S));
        MDSP_TRY(upload_roots_n<float>(cp.rootsN, nfft));
        MDSP_TRY(cp.win.reserve(sizeof(float) * (size_t)nfft));
        cp.ready = true;
    }
    hipLaunchKernelGGL(cols_window_kernel<float>, dim3((unsigned)cdiv(nfft, 256)), dim3(256), 0, st, win_dev, cp.win.as<float>(), n, (int)nfft);
    MDSP_LAUNCH_CHECK();
    ColsArgs ca{};
    ca.g.s = s; ca.g.roots = cp.roots.p; ca.g.lds_ = lds_; ca.g.K = K; ca.g.hop = hop; ca.g.nch = nch;
    ca.g.units_per_ch = cplx ? K : cdiv(K, 2);
    ca.g.n = n; ca.g.N = (int)S;
    ca.winf = cp.win.p; ca.rootsN = cp.rootsN.p; ca.nfft = (int)nfft; ca.R0 = R0;
    return cplx ? big_rows_dispatch<float, true>(ca, nch, st, ngroups, partial) : big_rows_dispatch<float, false>(ca, nch, st, ngroups, partial);
}
}  // namespace mdsp
