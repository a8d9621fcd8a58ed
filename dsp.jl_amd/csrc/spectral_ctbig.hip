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

// flags: 16 one LDS buffer, 512 group padding, 2048 table twiddles (+ MDSP_CTBIG_PREF: 8, the next unit's samples in registers through the passes)
#ifndef MDSP_CTBIG_PREF
#define MDSP_CTBIG_PREF 0   // (8: the next unit's samples in registers through the passes -- 130 - 270 spilled registers at these sizes)
#endif
#define MDSP_CTBIG_F (16 | 512 | 2048 | MDSP_CTBIG_PREF)
#define MDSP_CTBIG_SIZES(X)                                                                                                                    \
    X(8400, 512, MDSP_CTBIG_F, 20, 20, 21) X(8505, 512, MDSP_CTBIG_F, 27, 15, 21) X(8640, 512, MDSP_CTBIG_F, 18, 20, 24)                       \
    X(8748, 512, MDSP_CTBIG_F, 27, 18, 18) X(8750, 512, MDSP_CTBIG_F, 25, 14, 25) X(8820, 512, MDSP_CTBIG_F, 28, 15, 21)                       \
    X(8960, 512, MDSP_CTBIG_F, 28, 16, 20) X(9000, 512, MDSP_CTBIG_F, 18, 20, 25) X(9072, 512, MDSP_CTBIG_F, 28, 18, 18)                       \
    X(9216, 512, MDSP_CTBIG_F, 24, 16, 24) X(9375, 512, MDSP_CTBIG_F, 25, 15, 25) X(9408, 512, MDSP_CTBIG_F, 28, 16, 21)                       \
    X(9450, 512, MDSP_CTBIG_F, 21, 18, 25) X(9600, 512, MDSP_CTBIG_F, 20, 20, 24) X(9720, 512, MDSP_CTBIG_F, 27, 18, 20)                       \
    X(9800, 512, MDSP_CTBIG_F, 28, 14, 25) X(10000, 512, MDSP_CTBIG_F, 20, 20, 25) X(10080, 512, MDSP_CTBIG_F, 28, 18, 20)                     \
    X(10125, 512, MDSP_CTBIG_F, 27, 15, 25) X(10368, 512, MDSP_CTBIG_F, 27, 16, 24)                    \
    X(10500, 512, MDSP_CTBIG_F, 28, 15, 25) X(10752, 512, MDSP_CTBIG_F, 28, 16, 24) X(10800, 512, MDSP_CTBIG_F, 24, 18, 25)                    \
    X(10935, 512, MDSP_CTBIG_F, 27, 15, 27) X(10976, 512, MDSP_CTBIG_F, 28, 14, 28) X(11200, 512, MDSP_CTBIG_F, 28, 16, 25)                    \
    X(11250, 512, MDSP_CTBIG_F, 25, 18, 25)   /* (11520 = 24 20 24 and 12000 = 24 20 25 lost to 3 x 3840 / 2 x 6000: profiles/r06_ctbig.json) */   \
    X(12500, 512, MDSP_CTBIG_F, 25, 20, 25)   /* (a first pass of radix 32 -- 10240, 12288, 12800, 16384 -- spills 150 - 210 registers: those stay R0 x S) */

// ... and the 7-smooth sizes from 2100 to 8192 points that have no schedule in ct_sched.h's table (which carries every mode for 23 sizes): generated -- the
// three factors out of {4 .. 30} and the thread count out of {256 .. 512} that give every thread at most ONE butterfly per pass with the fewest idle lanes
// (93 of the 109 sizes have such a triple; the others stay on the run-time schedule).  Welch sums only.
#define MDSP_CTBIG_SMALL_SIZES(X) \
    X(2100, 256, MDSP_CTBIG_F, 10, 10, 21) X(2160, 256, MDSP_CTBIG_F, 9, 10, 24) X(2187, 256, MDSP_CTBIG_F, 9, 9, 27) X(2205, 320, MDSP_CTBIG_F, 7, 15, 21) \
    X(2240, 256, MDSP_CTBIG_F, 10, 14, 16) X(2250, 256, MDSP_CTBIG_F, 9, 10, 25) X(2268, 256, MDSP_CTBIG_F, 9, 9, 28) X(2304, 256, MDSP_CTBIG_F, 9, 16, 16) \
    X(2352, 256, MDSP_CTBIG_F, 12, 14, 14) X(2430, 320, MDSP_CTBIG_F, 9, 9, 30) X(2450, 384, MDSP_CTBIG_F, 7, 14, 25) X(2520, 256, MDSP_CTBIG_F, 10, 12, 21) \
    X(2592, 256, MDSP_CTBIG_F, 12, 12, 18) X(2625, 384, MDSP_CTBIG_F, 7, 15, 25) X(2646, 320, MDSP_CTBIG_F, 9, 14, 21) X(2688, 256, MDSP_CTBIG_F, 12, 14, 16) \
    X(2700, 256, MDSP_CTBIG_F, 12, 15, 15) X(2744, 256, MDSP_CTBIG_F, 14, 14, 14) X(2800, 320, MDSP_CTBIG_F, 10, 10, 28) X(2835, 320, MDSP_CTBIG_F, 9, 15, 21) \
    X(2880, 256, MDSP_CTBIG_F, 12, 12, 20) X(2916, 384, MDSP_CTBIG_F, 9, 12, 27) X(2940, 256, MDSP_CTBIG_F, 14, 14, 15) X(3024, 256, MDSP_CTBIG_F, 12, 12, 21) \
    X(3087, 448, MDSP_CTBIG_F, 7, 21, 21) X(3136, 256, MDSP_CTBIG_F, 14, 14, 16) X(3150, 256, MDSP_CTBIG_F, 14, 15, 15) X(3240, 320, MDSP_CTBIG_F, 12, 15, 18) \
    X(3360, 256, MDSP_CTBIG_F, 14, 15, 16) X(3375, 256, MDSP_CTBIG_F, 15, 15, 15) X(3402, 384, MDSP_CTBIG_F, 9, 14, 27) X(3456, 320, MDSP_CTBIG_F, 12, 12, 24) \
    X(3500, 384, MDSP_CTBIG_F, 10, 14, 25) X(3528, 256, MDSP_CTBIG_F, 14, 14, 18) X(3584, 256, MDSP_CTBIG_F, 14, 16, 16) X(3600, 256, MDSP_CTBIG_F, 15, 15, 16) \
    X(3645, 448, MDSP_CTBIG_F, 9, 15, 27) X(3750, 384, MDSP_CTBIG_F, 10, 15, 25) X(3780, 320, MDSP_CTBIG_F, 12, 15, 21) X(3888, 384, MDSP_CTBIG_F, 12, 12, 27) \
    X(3920, 320, MDSP_CTBIG_F, 14, 14, 20) X(3969, 448, MDSP_CTBIG_F, 9, 21, 21) X(4032, 320, MDSP_CTBIG_F, 14, 16, 18) X(4050, 320, MDSP_CTBIG_F, 15, 15, 18) \
    X(4116, 320, MDSP_CTBIG_F, 14, 14, 21) X(4200, 320, MDSP_CTBIG_F, 14, 15, 20) X(4320, 320, MDSP_CTBIG_F, 15, 16, 18) X(4374, 512, MDSP_CTBIG_F, 9, 18, 27) \
    X(4410, 320, MDSP_CTBIG_F, 14, 15, 21) X(4480, 320, MDSP_CTBIG_F, 14, 16, 20) X(4500, 320, MDSP_CTBIG_F, 15, 15, 20) X(4536, 384, MDSP_CTBIG_F, 12, 14, 27) \
    X(4608, 320, MDSP_CTBIG_F, 16, 16, 18) X(4704, 384, MDSP_CTBIG_F, 14, 14, 24) X(4725, 320, MDSP_CTBIG_F, 15, 15, 21) X(4860, 384, MDSP_CTBIG_F, 15, 18, 18) \
    X(4900, 384, MDSP_CTBIG_F, 14, 14, 25) X(5040, 384, MDSP_CTBIG_F, 14, 15, 24) X(5184, 384, MDSP_CTBIG_F, 16, 18, 18) X(5250, 384, MDSP_CTBIG_F, 14, 15, 25) \
    X(5292, 384, MDSP_CTBIG_F, 14, 14, 27) X(5376, 384, MDSP_CTBIG_F, 14, 16, 24) X(5400, 384, MDSP_CTBIG_F, 15, 15, 24) X(5488, 448, MDSP_CTBIG_F, 14, 14, 28) \
    X(5600, 448, MDSP_CTBIG_F, 14, 16, 25) X(5625, 384, MDSP_CTBIG_F, 15, 15, 25) X(5670, 384, MDSP_CTBIG_F, 15, 18, 21) X(5760, 384, MDSP_CTBIG_F, 15, 16, 24) \
    X(5832, 384, MDSP_CTBIG_F, 18, 18, 18) X(5880, 448, MDSP_CTBIG_F, 14, 14, 30) X(6048, 384, MDSP_CTBIG_F, 16, 18, 21) X(6075, 448, MDSP_CTBIG_F, 15, 15, 27) \
    X(6174, 448, MDSP_CTBIG_F, 14, 21, 21) X(6272, 448, MDSP_CTBIG_F, 14, 16, 28) X(6300, 448, MDSP_CTBIG_F, 15, 15, 28) X(6480, 384, MDSP_CTBIG_F, 18, 18, 20) \
    X(6615, 448, MDSP_CTBIG_F, 15, 21, 21) X(6720, 448, MDSP_CTBIG_F, 15, 16, 28) X(6750, 512, MDSP_CTBIG_F, 15, 15, 30) X(6804, 384, MDSP_CTBIG_F, 18, 18, 21) \
    X(6912, 448, MDSP_CTBIG_F, 16, 16, 27) X(7000, 512, MDSP_CTBIG_F, 14, 20, 25) X(7056, 448, MDSP_CTBIG_F, 16, 21, 21) X(7168, 448, MDSP_CTBIG_F, 16, 16, 28) \
    X(7200, 448, MDSP_CTBIG_F, 18, 20, 20) X(7290, 512, MDSP_CTBIG_F, 15, 18, 27) X(7500, 512, MDSP_CTBIG_F, 15, 20, 25) X(7560, 448, MDSP_CTBIG_F, 18, 20, 21) \
    X(7680, 512, MDSP_CTBIG_F, 16, 16, 30) X(7776, 448, MDSP_CTBIG_F, 18, 18, 24) X(7938, 448, MDSP_CTBIG_F, 18, 21, 21) X(8064, 512, MDSP_CTBIG_F, 16, 18, 28) \
    X(8100, 512, MDSP_CTBIG_F, 18, 18, 25)

template <typename R, bool CPLX> int big_dispatch(GenArgs& a, int64_t nch, hipStream_t st, int64_t* nslots, DevBuf* partial) {
    switch (a.N) {
#define MDSP_X(N, T, F, ...) \
    case N: return gen_ct_launch<R, CPLX, 0, CtSched<N, T, F, __VA_ARGS__>>(a, nch, st, nslots, partial);
        MDSP_CTBIG_SIZES(MDSP_X)
        MDSP_CTBIG_SMALL_SIZES(MDSP_X)
#undef MDSP_X
        default: MDSP_FAIL(MDSP_ERR_ASSERTION, "no single-workgroup compile-time schedule of %d points", a.N);
    }
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
    if (dtype_is_double(dtype)) return false;
    switch (nfft) {
#define MDSP_X(N, ...) case N:
        MDSP_CTBIG_SIZES(MDSP_X)
        MDSP_CTBIG_SMALL_SIZES(MDSP_X)
#undef MDSP_X
        return true;
        default: return false;
    }
}

int ctbig_welch(CtColsPlan& cp, int dtype, const void* s, int64_t lds_, int64_t K, int64_t hop, int64_t nch, int n, int64_t nfft, const double* win_dev,
                hipStream_t st, int64_t* nslots, DevBuf* partial) {
    if (!ctbig_ok(dtype, nfft)) MDSP_FAIL(MDSP_ERR_ASSERTION, "nfft=%lld has no single-workgroup compile-time schedule", (long long)nfft);
    if (!cp.ready) {
        MDSP_TRY(upload_roots_n<float>(cp.roots, nfft));
        cp.ready = true;
    }
    const bool cplx = dtype_is_complex(dtype);
    GenArgs g{};
    g.s = s; g.roots = cp.roots.p; g.win = win_dev;
    g.lds_ = lds_; g.K = K; g.hop = hop; g.nch = nch; g.units_per_ch = cplx ? K : cdiv(K, 2);
    g.n = n; g.N = (int)nfft;
    return cplx ? big_dispatch<float, true>(g, nch, st, nslots, partial) : big_dispatch<float, false>(g, nch, st, nslots, partial);
}
}  // namespace mdsp
