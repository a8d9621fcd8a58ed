// STFT / spectrogram / periodogram columns on the single-workgroup compile-time schedules of ctbig_sizes.h (round 6): the column modes of gen_ct_kernel
// (spectral_gen.h MODE 1) over the same size tables as the Welch sums of spectral_ctbig.hip -- complex signals at every size up to 16384 points (the last pass is
// consumed from registers: one LDS buffer), real signals (two frames per transform; the last pass leaves the natural-order spectrum in the same buffer, from which the two
// frames are untangled) likewise.
// The sizes that also have an all-mode schedule in ct_sched.h keep it.  Float32 / ComplexF32.
// Reference loops: periodograms.jl:872-897 (stft), :828-860 (spectrogram), :57-69 (ArraySplit), :142-172 / :234-244 (fft2pow!, fft2oneortwosided!).
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

#include "ctbig_sizes.h"

constexpr int REAL_COLUMNS_MAX = 16384;   // (real-signal columns on ONE buffer, CtSched flag 65536: up to round 6's first form two buffers, 9600 points)
constexpr int col_flags(int f, bool cplx) { return cplx ? (f & ~(8192 | 32768)) : ((f & ~(4096 | 8192 | 32768)) | 65536); }   // no sums; real signals: the window in registers, one buffer

template <typename R, bool CPLX> int cols_dispatch(GenArgs& a, int64_t nch, hipStream_t st) {
    int64_t nslots = 0;
    switch (a.N) {
#define MDSP_X(N, T, F, ...)                                                                                               \
    case N:                                                                                                                \
        if constexpr (CPLX || N <= REAL_COLUMNS_MAX)                                                                       \
            return gen_ct_launch<R, CPLX, 1, CtSched<N, T, col_flags(F, CPLX), __VA_ARGS__>>(a, nch, st, &nslots, nullptr); \
        break;
        MDSP_CTBIG_SIZES(MDSP_X)
        MDSP_CTBIG_LEAN_SIZES(MDSP_X)
        MDSP_CTBIG_SMALL_SIZES(MDSP_X)
#undef MDSP_X
        default: break;
    }
    MDSP_FAIL(MDSP_ERR_ASSERTION, "no single-workgroup column schedule of %d points", a.N);
}

__global__ __launch_bounds__(256) void cols_window_kernel(const double* __restrict__ win, float* __restrict__ out, int n, int nfft) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i < nfft) out[i] = i < n ? (win ? (float)win[i] : 1.0f) : 0.0f;
}
}  // namespace

namespace mdsp {
bool ctbig_cols_ok(int dtype, int64_t nfft) {
    if (dtype_is_double(dtype)) return tunables().gx != 3 && ctbig64_ok(nfft);
    const bool cplx = dtype_is_complex(dtype);
    switch (nfft) {
#define MDSP_X(N, ...) case N:
        MDSP_CTBIG_SIZES(MDSP_X)
        MDSP_CTBIG_LEAN_SIZES(MDSP_X)
        MDSP_CTBIG_SMALL_SIZES(MDSP_X)
#undef MDSP_X
        return cplx || nfft <= REAL_COLUMNS_MAX;
        default: return false;
    }
}

int ctbig_stft(CtColsPlan& cp, int dtype, const CtBigColsArgs& c, hipStream_t st) {
    if (!ctbig_cols_ok(dtype, c.nfft)) MDSP_FAIL(MDSP_ERR_ASSERTION, "nfft=%lld has no single-workgroup column schedule", (long long)c.nfft);
    if (dtype_is_double(dtype)) return ctbig64_stft(cp, dtype_is_complex(dtype), c, st);
    if (!cp.ready) {
        std::vector<cx<float>> w((size_t)c.nfft);
        for (int64_t k = 0; k < c.nfft; ++k) {
            const zd r = unit_root(k, c.nfft, -1);
            w[(size_t)k] = {(float)r.real(), (float)r.imag()};
        }
        MDSP_TRY(cp.roots.reserve(sizeof(cx<float>) * (size_t)c.nfft));
        MDSP_HIP(hipMemcpy(cp.roots.p, w.data(), sizeof(cx<float>) * (size_t)c.nfft, hipMemcpyHostToDevice));
        MDSP_TRY(cp.win.reserve(sizeof(float) * (size_t)c.nfft));
        cp.ready = true;
    }
    const bool cplx = dtype_is_complex(dtype);
    if (cplx) {   // (per launch: the window of a multitaper plan changes between tapers; the lean forms load it beside the samples)
        hipLaunchKernelGGL(cols_window_kernel, dim3((unsigned)cdiv(c.nfft, 256)), dim3(256), 0, st, c.win, cp.win.as<float>(), c.n, (int)c.nfft);
        MDSP_LAUNCH_CHECK();
    }
    GenArgs g{};
    g.s = c.s; g.out = c.out; g.roots = cp.roots.p; g.win = c.win; g.winr = cp.win.p;
    g.len = c.len; g.lds_ = c.lds_; g.K = c.K; g.hop = c.hop; g.nch = c.nch; g.ldo = c.ldo; g.chs = c.chs;
    g.units_per_ch = cplx ? c.K : cdiv(c.K, 2);
    g.n = c.n; g.N = (int)c.nfft; g.nout = c.nout; g.onesided = c.onesided; g.psd = c.psd; g.accumulate = c.accumulate; g.r = c.r;
    return cplx ? cols_dispatch<float, true>(g, c.nch, st) : cols_dispatch<float, false>(g, c.nch, st);
}
}  // namespace mdsp
