// Welch sums at nfft = R0 x S with a COMPILE-TIME schedule for the S-point row transforms (round 6): the decimation-in-frequency step of gx_kernels.h
//   X[k1 + R0 k2] = FFT_S( y_k1 )[k2],   y_k1[i] = W_nfft^{i k1} sum_{n1 < R0} x[S n1 + i] w[S n1 + i] W_R0^{n1 k1}
// fused into the first pass of the mixed-radix kernels of spectral_gen.h (gen_ct_kernel: radices, strides and padding as template constants, twiddles in
// registers or two LDS tables, last pass consumed from registers).  One workgroup per (frame sequence, k1); the R0 workgroups of a sequence sit on one XCD,
// so R0 - 1 of the R0 reads of a frame come from the L2.  The run-time-schedule kernel does the same with 2 - 2.5 x the vector instructions per point
// (run-time indexing, table twiddles): 16384 = 2 x 8192 measured 0.51 TB/s there against 2.1 for the 8192-point kernel this one is built from.
// (Tried and dropped, session r06s17: every row keeping ONE segment of the next unit in flight through the passes, the column twiddles in LDS to make room:
// 16384 1.06 -> 0.73 TB/s, 14000 0.86 -> 0.46 -- the second LDS array halves the residency of the smaller rows and the prefetch registers spill.)
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

#include "spectral_ctcols_kernel.h"

// the row sizes with a COLS instantiation: the compile-time schedules from 2000 points and two powers of two (16384 = 2 x 8192, 12288 = 3 x 4096 ...).
// Flags: group padding (512) wherever a pass's output groups alias the banks (the 8192-point rows ran with 58 % of their LDS cycles in conflicts without it);
// from 4096 points ONE LDS buffer and table twiddles (the register form holds sum_p M(p) (R_p - 1) complex
// values next to the column step's operands)
#define MDSP_CTCOLS_SIZES(X)                                                                                                            \
    X(2000, 256, 0, 5, 5, 5, 16) X(2400, 256, 0, 3, 5, 5, 4, 8) X(2500, 256, 0, 5, 5, 5, 5, 4) X(2560, 320, 0, 5, 8, 8, 8)                  \
    X(3000, 384, 0, 3, 5, 5, 5, 8) X(3072, 256, 512, 3, 16, 8, 8) X(3200, 256, 0, 5, 5, 8, 16) X(3840, 256, 0, 3, 5, 16, 16)                \
    X(4000, 512, 0, 5, 5, 5, 4, 8) X(4096, 256, 2576, 16, 16, 16) X(4800, 512, 2064, 3, 5, 5, 8, 8) X(5000, 512, 2064, 5, 5, 5, 5, 8)       \
    X(5120, 320, 2064, 5, 16, 8, 8) X(6000, 512, 2064, 3, 5, 5, 5, 16) X(6144, 512, 2576, 3, 16, 16, 8) X(6400, 448, 2064, 5, 5, 16, 16)    \
    X(8000, 512, 2064, 5, 5, 5, 8, 8) X(8192, 512, 2576, 16, 32, 16)                                                                        \
    /* halves of the other nextfastfft sizes between 8193 and 16384 (three passes, composite radices; one LDS buffer, table twiddles) */   \
    /* (4200 = 25 24 7, 5600 = 7 32 25, 7500 = 12 25 25 and 8100 = 12 27 25 were tried and lost to the run-time schedule at every R0: profiles/r06_ctcols.json) */ \
    X(4500, 320, 2064, 15, 15, 20) X(4608, 512, 2576, 9, 16, 32) X(5400, 512, 2576, 8, 27, 25) X(6250, 320, 2064, 10, 25, 25)                \
    X(6750, 512, 2064, 15, 18, 25) X(7000, 512, 2576, 14, 20, 25) X(7200, 512, 2576, 16, 18, 25) X(7680, 512, 2576, 15, 16, 32)

// Float64 / ComplexF64 rows: up to 4096 points on ONE buffer of 16-byte elements with table twiddles (the single-workgroup Float64 schedules from 4800 points
// spill 100 - 184 registers and move 1.8 - 2.6 x the algorithmic bytes: 5000 = 2 x 2500, 8000 = 2 x 4000 run here without scratch)
#define MDSP_CTCOLS_SIZES_F64(X)                                                                                                        \
    X(2000, 256, 2064, 5, 5, 5, 16) X(2400, 256, 2064, 3, 5, 5, 4, 8) X(2500, 256, 2064, 5, 5, 5, 5, 4) X(2560, 320, 2064, 5, 8, 8, 8)      \
    X(3000, 384, 2064, 3, 5, 5, 5, 8) X(3072, 256, 2576, 3, 16, 8, 8) X(3200, 256, 2064, 5, 5, 8, 16) X(4000, 512, 2064, 5, 5, 5, 4, 8)     \
    X(4096, 256, 2576, 16, 16, 16)

template <typename R, bool CPLX> int cols_dispatch(ColsArgs& ca, int64_t nch, hipStream_t st, int64_t* ngroups, DevBuf* partial) {
#define MDSP_X(N, T, F, ...) \
    case N: return cols_launch<R, CPLX, CtSched<N, T, F, __VA_ARGS__>>(ca, nch, st, ngroups, partial);
    if constexpr (sizeof(R) == 4) {
        switch (ca.g.N) {
            MDSP_CTCOLS_SIZES(MDSP_X)
            default: break;
        }
    } else {
        switch (ca.g.N) {
            MDSP_CTCOLS_SIZES_F64(MDSP_X)
            default: break;
        }
    }
#undef MDSP_X
    MDSP_FAIL(MDSP_ERR_ASSERTION, "no compile-time row schedule of %d points", ca.g.N);
}

}  // namespace

namespace mdsp {
int ctcols_split(int dtype, int64_t nfft) {
    if (dtype_is_double(dtype)) {
        for (int R0 = 2; R0 <= 8; ++R0) {
            if (nfft % R0) continue;
            if (tunables().gx != 6 && tunables().gx != 3 && ctcols64_row_ok(nfft / R0)) return R0;   // rows of 4097 .. 9600 points (spectral_ctcols_f64.hip)
            if (R0 > 4) continue;
            switch (nfft / R0) {
#define MDSP_X(N, ...) case N:
                MDSP_CTCOLS_SIZES_F64(MDSP_X)
#undef MDSP_X
                return R0;
                default: break;
            }
        }
        return 0;
    }
    // R0 = 2 .. 4, and 8 x 8192 (measured, profiles/r06_ctcols.json: 16384 = 2 x 8192 1.0 TB/s, 32768 = 4 x 8192 0.79, 65536 = 8 x 8192 0.56 against the
    // multi-pass engine's 0.44; from R0 = 5 the R0 reads per point cost what the row kernel saves: 40000 = 5 x 8000 0.30 against 0.34 on the run-time schedule)
    // round 6, later: rows of 8193 .. 16384 points (spectral_ctcols_big.hip) come first -- the smallest column factor wins (MDSP_GX=6: without them)
    // (measured, profiles/r06_ctcols_big.json: 2 x S 0.71 - 1.01 TB/s against 0.42 - 0.75 without them, 65536 = 4 x 16384 0.80 against 0.49 as 8 x 8192, and up to
    // R0 = 8 -- 81920 = 5 x 16384 0.65, 100000 = 8 x 12500 0.46, 131072 = 8 x 16384 0.53 -- against the multi-pass engine's 0.18 - 0.35)
    const int big_r0_max = 8;
    for (int R0 = 2; R0 <= 8; ++R0) {
        if (nfft % R0) continue;
        if (tunables().gx != 6 && R0 <= big_r0_max && ctcols_big_row_ok(dtype, nfft / R0)) return R0;
        if (R0 > 4 && !(R0 == 8 && nfft == 65536)) continue;
        switch (nfft / R0) {
#define MDSP_X(N, ...) case N:
            MDSP_CTCOLS_SIZES(MDSP_X)
#undef MDSP_X
            return R0;
            default: break;
        }
    }
    return 0;
}

int ctcols_welch(CtColsPlan& cp, int dtype, const void* s, int64_t lds_, int64_t K, int64_t hop, int64_t nch, int n, int64_t nfft, const double* win_dev,
                 hipStream_t st, int64_t* ngroups, DevBuf* partial) {
    const int R0 = ctcols_split(dtype, nfft);
    if (R0 == 0) MDSP_FAIL(MDSP_ERR_ASSERTION, "nfft=%lld is not R0 x a compile-time row size", (long long)nfft);
    const int64_t S = nfft / R0;
    if (tunables().gx != 6 && ctcols_big_row_ok(dtype, S)) return ctcols_big_welch(cp, dtype, s, lds_, K, hop, nch, n, nfft, R0, win_dev, st, ngroups, partial);
    if (dtype_is_double(dtype) && tunables().gx != 6 && tunables().gx != 3 && ctcols64_row_ok(S))
        return ctcols64_welch(cp, dtype_is_complex(dtype), s, lds_, K, hop, nch, n, nfft, R0, win_dev, st, ngroups, partial);
    const bool dbl = dtype_is_double(dtype), cplx = dtype_is_complex(dtype);
    if (!cp.ready) {
        MDSP_TRY(dbl ? upload_roots_n<double>(cp.roots, S) : upload_roots_n<float>(cp.roots, S));
        MDSP_TRY(dbl ? upload_roots_n<double>(cp.rootsN, nfft) : upload_roots_n<float>(cp.rootsN, nfft));
        MDSP_TRY(cp.win.reserve((dbl ? sizeof(double) : sizeof(float)) * (size_t)nfft));
        cp.ready = true;
    }
    if (dbl) hipLaunchKernelGGL(cols_window_kernel<double>, dim3((unsigned)cdiv(nfft, 256)), dim3(256), 0, st, win_dev, cp.win.as<double>(), n, (int)nfft);
    else hipLaunchKernelGGL(cols_window_kernel<float>, dim3((unsigned)cdiv(nfft, 256)), dim3(256), 0, st, win_dev, cp.win.as<float>(), n, (int)nfft);
    MDSP_LAUNCH_CHECK();
    ColsArgs ca{};
    ca.g.s = s; ca.g.roots = cp.roots.p; ca.g.lds_ = lds_; ca.g.K = K; ca.g.hop = hop; ca.g.nch = nch;
    ca.g.units_per_ch = cplx ? K : cdiv(K, 2);
    ca.g.n = n; ca.g.N = (int)S;
    ca.winf = cp.win.p; ca.rootsN = cp.rootsN.p; ca.nfft = (int)nfft; ca.R0 = R0;
    if (dbl) return cplx ? cols_dispatch<double, true>(ca, nch, st, ngroups, partial) : cols_dispatch<double, false>(ca, nch, st, ngroups, partial);
    return cplx ? cols_dispatch<float, true>(ca, nch, st, ngroups, partial) : cols_dispatch<float, false>(ca, nch, st, ngroups, partial);
}
}  // namespace mdsp
