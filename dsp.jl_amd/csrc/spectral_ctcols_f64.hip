// Float64 / ComplexF64 Welch sums at nfft = R0 x S with the single-workgroup schedules of ctbig_sizes.h as rows (round 6): the row kernel of spectral_ctcols.hip
// (column step fused into the loads) over every table size above 4096 points whose one buffer of 16-byte elements fits 160 KiB (to 9600 points), R0 = 2 .. 8 --
// 16800 = 2 x 8400, 19200 = 2 x 9600, 50000 = 8 x 6250 ... : DSP.jl's default element type at the sizes between 9601 and 76800 points, which ran the run-time
// schedule's column form (0.4 - 0.55 TB/s) or, from 32768 / 50000 points, the multi-pass engine.  Lean rows: column twiddles loaded beside the samples, derived table
// twiddles, no group padding.  Reference loops: periodograms.jl:746-759 (welch_pgram_helper!), :57-69 (ArraySplit), :142-172 (fft2pow!).
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

constexpr int f64_row_flags(int f) { return ((f) & ~(512 | 1024 | 8192 | 32768 | 65536)) | 16 | 2048 | 4096 | 16384; }
template <typename S> constexpr bool f64_fits() { return S::N > 4096 && sizeof(cx<double>) * ((size_t)S::NP + S::TWS + S::NTWHI) <= (size_t)160 * 1024; }

template <bool CPLX> int rows64_dispatch(ColsArgs& ca, int64_t nch, hipStream_t st, int64_t* ngroups, DevBuf* partial) {
    switch (ca.g.N) {
#define MDSP_X(N, T, F, ...)                                                                                     \
    case N: {                                                                                                    \
        using S = CtSched<N, T, f64_row_flags(F), __VA_ARGS__>;                                                  \
        if constexpr (f64_fits<S>()) return cols_launch<double, CPLX, S>(ca, nch, st, ngroups, partial);         \
        break;                                                                                                   \
    }
        MDSP_CTBIG_SIZES(MDSP_X)
        MDSP_CTBIG_LEAN_SIZES(MDSP_X)
        MDSP_CTBIG_SMALL_SIZES(MDSP_X)
#undef MDSP_X
        default: break;
    }
    MDSP_FAIL(MDSP_ERR_ASSERTION, "no Float64 compile-time row schedule of %d points", ca.g.N);
}
}  // namespace

namespace mdsp {
bool ctcols64_row_ok(int64_t S) {
    switch (S) {
#define MDSP_X(N, T, F, ...) \
    case N: return f64_fits<CtSched<N, T, f64_row_flags(F), __VA_ARGS__>>();
        MDSP_CTBIG_SIZES(MDSP_X)
        MDSP_CTBIG_LEAN_SIZES(MDSP_X)
        MDSP_CTBIG_SMALL_SIZES(MDSP_X)
#undef MDSP_X
        default: return false;
    }
}

int ctcols64_welch(CtColsPlan& cp, bool cplx, const void* s, int64_t lds_, int64_t K, int64_t hop, int64_t nch, int n, int64_t nfft, int R0, const double* win_dev,
                   hipStream_t st, int64_t* ngroups, DevBuf* partial) {
    const int64_t S = nfft / R0;
    if (nfft % R0 || !ctcols64_row_ok(S)) MDSP_FAIL(MDSP_ERR_ASSERTION, "nfft=%lld is not %d x a Float64 single-workgroup row size", (long long)nfft, R0);
    if (!cp.ready) {
        MDSP_TRY(upload_roots_n<double>(cp.roots, S));
        MDSP_TRY(upload_roots_n<double>(cp.rootsN, nfft));
        MDSP_TRY(cp.win.reserve(sizeof(double) * (size_t)nfft));
        cp.ready = true;
    }
    hipLaunchKernelGGL(cols_window_kernel<double>, dim3((unsigned)cdiv(nfft, 256)), dim3(256), 0, st, win_dev, cp.win.as<double>(), n, (int)nfft);
    MDSP_LAUNCH_CHECK();
    ColsArgs ca{};
    ca.g.s = s; ca.g.roots = cp.roots.p; ca.g.lds_ = lds_; ca.g.K = K; ca.g.hop = hop; ca.g.nch = nch;
    ca.g.units_per_ch = cplx ? K : cdiv(K, 2);
    ca.g.n = n; ca.g.N = (int)S;
    ca.winf = cp.win.p; ca.rootsN = cp.rootsN.p; ca.nfft = (int)nfft; ca.R0 = R0;
    return cplx ? rows64_dispatch<true>(ca, nch, st, ngroups, partial) : rows64_dispatch<false>(ca, nch, st, ngroups, partial);
}
}  // namespace mdsp
