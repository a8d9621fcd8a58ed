// Float64 / ComplexF64 on the single-workgroup compile-time schedules of ctbig_sizes.h (round 6): Welch sums and the column modes for every table size whose
// ONE buffer of 16-byte elements and twiddle tables fit 160 KiB (up to 9600 points) -- DSP.jl's default element type at the 7-smooth sizes that have no schedule in
// ct_sched.h's tables, which ran the run-time schedule (0.5 - 0.9 TB/s).  The lean form throughout: window loaded beside the samples (register-consumed modes),
// table twiddles derived from ~2 sqrt(R) values, no group padding (the Float64 rule of spectral_gen.h), real-signal columns on the one buffer.
// Reference loops: periodograms.jl:746-759 (welch_pgram_helper!), :872-897 (stft), :57-69 (ArraySplit), :142-172 / :234-244 (fft2pow!, fft2oneortwosided!).
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

constexpr int f64_flags(int f) { return ((f) & ~(512 | 1024 | 8192 | 32768 | 65536)) | 16 | 2048 | 4096 | 16384; }
constexpr int f64_flags_real_cols(int f) { return (f64_flags(f) & ~4096) | 65536; }
template <typename S> constexpr bool f64_fits() { return sizeof(cx<double>) * ((size_t)S::NP + S::TWS + S::NTWHI) <= (size_t)160 * 1024; }

// MODE 0: Welch sums; 1: columns
template <bool CPLX, int MODE> int dispatch64(GenArgs& a, int64_t nch, hipStream_t st, int64_t* nslots, DevBuf* partial) {
    switch (a.N) {
#define MDSP_X(N, T, F, ...)                                                                                              \
    case N: {                                                                                                             \
        using S = CtSched<N, T, (MODE == 1 && !CPLX) ? f64_flags_real_cols(F) : f64_flags(F), __VA_ARGS__>;               \
        if constexpr (f64_fits<S>()) return gen_ct_launch<double, CPLX, MODE, S>(a, nch, st, nslots, partial);            \
        break;                                                                                                            \
    }
        MDSP_CTBIG_SIZES(MDSP_X)
        MDSP_CTBIG_LEAN_SIZES(MDSP_X)
        MDSP_CTBIG_SMALL_SIZES(MDSP_X)
#undef MDSP_X
        default: break;
    }
    MDSP_FAIL(MDSP_ERR_ASSERTION, "no Float64 single-workgroup compile-time schedule of %d points", a.N);
}

__global__ __launch_bounds__(256) void window64_kernel(const double* __restrict__ win, double* __restrict__ out, int n, int nfft) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i < nfft) out[i] = i < n ? (win ? win[i] : 1.0) : 0.0;
}

int prepare64(CtColsPlan& cp, int64_t nfft) {
    if (cp.ready) return MDSP_OK;
    std::vector<cx<double>> w((size_t)nfft);
    for (int64_t k = 0; k < nfft; ++k) {
        const zd r = unit_root(k, nfft, -1);
        w[(size_t)k] = {r.real(), r.imag()};
    }
    MDSP_TRY(cp.roots.reserve(sizeof(cx<double>) * (size_t)nfft));
    MDSP_HIP(hipMemcpy(cp.roots.p, w.data(), sizeof(cx<double>) * (size_t)nfft, hipMemcpyHostToDevice));
    MDSP_TRY(cp.win.reserve(sizeof(double) * (size_t)nfft));
    cp.ready = true;
    return MDSP_OK;
}
}  // namespace

namespace mdsp {
bool ctbig64_ok(int64_t nfft) {
    switch (nfft) {
#define MDSP_X(N, T, F, ...) \
    case N: return f64_fits<CtSched<N, T, f64_flags(F), __VA_ARGS__>>();
        MDSP_CTBIG_SIZES(MDSP_X)
        MDSP_CTBIG_LEAN_SIZES(MDSP_X)
        MDSP_CTBIG_SMALL_SIZES(MDSP_X)
#undef MDSP_X
        default: return false;
    }
}

int ctbig64_welch(CtColsPlan& cp, bool cplx, const void* s, int64_t lds_, int64_t K, int64_t hop, int64_t nch, int n, int64_t nfft, const double* win_dev, hipStream_t st,
                  int64_t* nslots, DevBuf* partial, int accumulate) {
    if (!ctbig64_ok(nfft)) MDSP_FAIL(MDSP_ERR_ASSERTION, "nfft=%lld has no Float64 single-workgroup compile-time schedule", (long long)nfft);
    const bool fresh = !cp.ready;
    MDSP_TRY(prepare64(cp, nfft));
    if (fresh) {   // (a Welch plan's window is fixed)
        hipLaunchKernelGGL(window64_kernel, dim3((unsigned)cdiv(nfft, 256)), dim3(256), 0, st, win_dev, cp.win.as<double>(), n, (int)nfft);
        MDSP_LAUNCH_CHECK();
    }
    GenArgs g{};
    g.s = s; g.roots = cp.roots.p; g.win = win_dev; g.winr = cp.win.p;
    g.lds_ = lds_; g.K = K; g.hop = hop; g.nch = nch; g.units_per_ch = cplx ? K : cdiv(K, 2);
    g.n = n; g.N = (int)nfft; g.accumulate = accumulate;
    return cplx ? dispatch64<true, 0>(g, nch, st, nslots, partial) : dispatch64<false, 0>(g, nch, st, nslots, partial);
}

int ctbig64_stft(CtColsPlan& cp, bool cplx, const CtBigColsArgs& c, hipStream_t st) {
    if (!ctbig64_ok(c.nfft)) MDSP_FAIL(MDSP_ERR_ASSERTION, "nfft=%lld has no Float64 single-workgroup column schedule", (long long)c.nfft);
    MDSP_TRY(prepare64(cp, c.nfft));
    if (cplx) {   // (per launch: the window of a multitaper plan changes between tapers)
        hipLaunchKernelGGL(window64_kernel, dim3((unsigned)cdiv(c.nfft, 256)), dim3(256), 0, st, c.win, cp.win.as<double>(), c.n, (int)c.nfft);
        MDSP_LAUNCH_CHECK();
    }
    GenArgs g{};
    g.s = c.s; g.out = c.out; g.roots = cp.roots.p; g.win = c.win; g.winr = cp.win.p;
    g.len = c.len; g.lds_ = c.lds_; g.K = c.K; g.hop = c.hop; g.nch = c.nch; g.ldo = c.ldo; g.chs = c.chs;
    g.units_per_ch = cplx ? c.K : cdiv(c.K, 2);
    g.n = c.n; g.N = (int)c.nfft; g.nout = c.nout; g.onesided = c.onesided; g.psd = c.psd; g.accumulate = c.accumulate; g.r = c.r;
    int64_t nslots = 0;
    return cplx ? dispatch64<true, 1>(g, c.nch, st, &nslots, nullptr) : dispatch64<false, 1>(g, c.nch, st, &nslots, nullptr);
}
}  // namespace mdsp
