// Welch sums at 7-smooth nfft between 8193 and 16384 points in ONE workgroup with a COMPILE-TIME schedule (round 6): the mixed-radix kernel of spectral_gen.h
// (gen_ct_kernel: three passes of composite radices 14 .. 32, radices / strides / padding as template constants, last pass consumed from registers) on ONE LDS
// buffer of up to 128 KiB with table twiddles -- the sizes whose three factors give every one of 512 threads at most two butterflies per pass.  A workgroup
// holds the whole transform, so a frame is read once (no column step, no R0 reads per point): what welch_pgram(s) asks for on an 80 000 .. 130 000-sample
// signal (nfft = nextfastfft(length >> 3), periodograms.jl:560, :647; util.jl:134).  Float32 / ComplexF32.
// Reference loops: periodograms.jl:746-759 (welch_pgram_helper!), :57-69 (ArraySplit), :142-172 (fft2pow!).
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

template <typename R, bool CPLX> int big_dispatch(GenArgs& a, int64_t nch, hipStream_t st, int64_t* nslots, DevBuf* partial) {
    switch (a.N) {
#define MDSP_X(N, T, F, ...) \
    case N: return gen_ct_launch<R, CPLX, 0, CtSched<N, T, F, __VA_ARGS__>>(a, nch, st, nslots, partial);
        MDSP_CTBIG_SIZES(MDSP_X)
        MDSP_CTBIG_LEAN_SIZES(MDSP_X)
        MDSP_CTBIG_SMALL_SIZES(MDSP_X)
        MDSP_CTBIG_PREF_SIZES(MDSP_X)
#undef MDSP_X
        default: MDSP_FAIL(MDSP_ERR_ASSERTION, "no single-workgroup compile-time schedule of %d points", a.N);
    }
}

// the window in the working precision for the lean schedules (ct_pass0_lean): ones without a window, zeros behind n
__global__ __launch_bounds__(256) void big_window_kernel(const double* __restrict__ win, float* __restrict__ out, int n, int nfft) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i < nfft) out[i] = i < n ? (win ? (float)win[i] : 1.0f) : 0.0f;
}

template <typename R> int upload_roots_n(DevBuf& buf, int64_t n) {
    std::vector<cx<R>> w((size_t)n);
    for (int64_t k = 0; k < n; ++k) {
        const zd r = unit_root(k, n, -1);
        w[(size_t)k] = {(R)r.real(), (R)r.imag()};
    }
    MDSP_TRY(buf.reserve(sizeof(cx<R>) * (size_t)n));
    MDSP_HIP(hipMemcpy(buf.p, w.data(), sizeof(cx<R>) * (size_t)n, hipMemcpyHostToDevice));
    return MDSP_OK;
}

}  // namespace

namespace mdsp {
bool ctbig_ok(int dtype, int64_t nfft) {
    if (dtype_is_double(dtype)) return tunables().gx != 3 && ctbig64_ok(nfft);   // (MDSP_GX=3: Float64 stays on the run-time schedule, A/B)
    switch (nfft) {
#define MDSP_X(N, ...) case N:
        MDSP_CTBIG_SIZES(MDSP_X)
        MDSP_CTBIG_LEAN_SIZES(MDSP_X)
        MDSP_CTBIG_SMALL_SIZES(MDSP_X)
        MDSP_CTBIG_PREF_SIZES(MDSP_X)
#undef MDSP_X
        return true;
        default: return false;
    }
}

bool ctbig_preferred(int dtype, int64_t nfft) {   // ... in front of the all-mode schedule the size also has (spectral.hip use_gx)
    if (dtype_is_double(dtype)) return false;
    switch (nfft) {
#define MDSP_X(N, ...) case N:
        MDSP_CTBIG_PREF_SIZES(MDSP_X)
#undef MDSP_X
        return true;
        default: return false;
    }
}

int ctbig_welch(CtColsPlan& cp, int dtype, const void* s, int64_t lds_, int64_t K, int64_t hop, int64_t nch, int n, int64_t nfft, const double* win_dev,
                hipStream_t st, int64_t* nslots, DevBuf* partial, int accumulate) {
    if (!ctbig_ok(dtype, nfft)) MDSP_FAIL(MDSP_ERR_ASSERTION, "nfft=%lld has no single-workgroup compile-time schedule", (long long)nfft);
    if (dtype_is_double(dtype)) return ctbig64_welch(cp, dtype_is_complex(dtype), s, lds_, K, hop, nch, n, nfft, win_dev, st, nslots, partial, accumulate);
    if (!cp.ready) {
        MDSP_TRY(upload_roots_n<float>(cp.roots, nfft));
        MDSP_TRY(cp.win.reserve(sizeof(float) * (size_t)nfft));
        hipLaunchKernelGGL(big_window_kernel, dim3((unsigned)cdiv(nfft, 256)), dim3(256), 0, st, win_dev, cp.win.as<float>(), n, (int)nfft);
        MDSP_LAUNCH_CHECK();
        cp.ready = true;
    }
    const bool cplx = dtype_is_complex(dtype);
    GenArgs g{};
    g.s = s; g.roots = cp.roots.p; g.win = win_dev; g.winr = cp.win.p;
    g.lds_ = lds_; g.K = K; g.hop = hop; g.nch = nch; g.units_per_ch = cplx ? K : cdiv(K, 2);
    g.n = n; g.N = (int)nfft; g.accumulate = accumulate;
    return cplx ? big_dispatch<float, true>(g, nch, st, nslots, partial) : big_dispatch<float, false>(g, nch, st, nslots, partial);
}
}  // namespace mdsp
