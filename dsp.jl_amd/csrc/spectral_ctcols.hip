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

struct ColsArgs {
    GenArgs g;             // s, out (partials), roots (of S), lds_, K, hop, nch, units_per_ch, per_slot, n, N = S
    const void* winf;      // R window[nfft] (ones without a window, zero tail)
    const void* rootsN;    // nfft forward roots, cx<R>
    int nfft, R0;
};

template <typename R, bool CPLX, typename S>
__global__ __launch_bounds__(S::T, 2) void gen_ct_cols_kernel(ColsArgs ca) {
    const GenArgs& a = ca.g;
    using TT = std::conditional_t<CPLX, cx<R>, R>;
    constexpr int N = S::N, T = S::T;
    constexpr int PL = S::P - 1, RL = S::radix(PL), ML = S::M(PL), NBL = S::nbf(PL);
    constexpr int R0r = S::radix(0), M0 = S::M(0), NB0 = S::nbf(0), W0 = M0 * R0r;
    constexpr bool INPL = S::INPLACE;
    constexpr int SZ = (int)sizeof(TT), WZ = (int)sizeof(R);
    __shared__ __attribute__((aligned(16))) cx<R> buf[INPL ? S::NP : 2 * S::NP];
    cx<R>*bufA = buf, *bufB = INPL ? buf : buf + S::NP;
    const int t = threadIdx.x;
    const int64_t ch = blockIdx.y;
    const TT* sc = static_cast<const TT*>(a.s) + ch * a.lds_;
    // group of workgroups (one frame sequence) and row k1; the R0 rows of a group on one XCD (workgroups go to the XCDs round-robin)
    const unsigned b = blockIdx.x, xcd = b & 7u, wq = b >> 3;
    const int k1 = (int)(wq % (unsigned)ca.R0);
    const int64_t gslot = (int64_t)(wq / (unsigned)ca.R0) * 8 + xcd;
    const int64_t u0 = gslot * a.per_slot;
    cx<R> tw[S::NTW];
    ct_load_twiddles<S, 0>(tw, static_cast<const cx<R>*>(a.roots), t);
    __shared__ __attribute__((aligned(16))) cx<R> twlo[S::TW2L ? S::TWS : 1], twhi[S::TW2L ? S::NTWHI : 1];
    const CtTw<R> t2{twlo, twhi};
    if constexpr (S::TW2L) {
        const cx<R>* g = static_cast<const cx<R>*>(a.roots);
        for (int i = t; i < S::TWS; i += T) fft::st2(twlo + i, g[i]);
        for (int i = t; i < S::NTWHI; i += T) fft::st2(twhi + i, g[(unsigned)i * S::TWS]);
    }
    // W_nfft^{i k1} at the points of this thread's first-pass butterflies: loop invariants
    const cx<R>* rootsN = static_cast<const cx<R>*>(ca.rootsN);
    cx<R> twc[W0];
#pragma unroll
    for (int m = 0; m < M0; ++m)
#pragma unroll
        for (int q = 0; q < R0r; ++q) {
            const unsigned i = (unsigned)(t + T * m + NB0 * q);
            twc[m * R0r + q] = rootsN[(unsigned)(((unsigned long long)(i < (unsigned)N ? i : 0u) * (unsigned)k1) % (unsigned)ca.nfft)];
        }
    double acc[ML * RL];
#pragma unroll
    for (int i = 0; i < ML * RL; ++i) acc[i] = 0.0;
    const __amdgpu_buffer_rsrc_t dw = io::make_rsrc(ca.winf, (long long)ca.nfft * WZ);
    for (int64_t it = 0; it < a.per_slot; ++it) {
        const int64_t u = u0 + it;
        const bool live = u < a.units_per_ch;
        const int64_t f0 = live ? (CPLX ? u : 2 * u) : 0;
        const bool haveB = !CPLX && live && (f0 + 1) < a.K;
        const TT* fa = sc + f0 * a.hop;
        const __amdgpu_buffer_rsrc_t da = io::make_rsrc(fa, live ? (long long)a.n * SZ : 0);
        const __amdgpu_buffer_rsrc_t db = io::make_rsrc(fa + (CPLX ? 0 : a.hop), haveB ? (long long)a.n * SZ : 0);
        // ---- first pass: the k1-th combination of the R0 segments of the windowed frame (pair), formed while loading
        cx<R> v0[W0];
#pragma unroll
        for (int i = 0; i < W0; ++i) v0[i] = cx<R>{(R)0, (R)0};
        int off = t * SZ, offw = t * WZ;
        asm volatile("" : "+v"(off), "+v"(offw));
        unsigned cidx = 0;   // n1 k1 mod R0
        for (int n1 = 0; n1 < ca.R0; ++n1) {
            const cx<R> c = rootsN[(unsigned)cidx * (unsigned)N];   // W_R0^{n1 k1} = W_nfft^{S (n1 k1 mod R0)}: wave-uniform
            cidx += (unsigned)k1;
            if (cidx >= (unsigned)ca.R0) cidx -= (unsigned)ca.R0;
            TT ra[W0], rb[CPLX ? 1 : W0];
            R w[W0];
#pragma unroll
            for (int m = 0; m < M0; ++m)
#pragma unroll
                for (int q = 0; q < R0r; ++q) {
                    const int e = T * m + NB0 * q;
                    ra[m * R0r + q] = io::Ld<TT>::load(da, off + e * SZ);
                    if constexpr (!CPLX) rb[m * R0r + q] = io::Ld<TT>::load(db, off + e * SZ);
                    w[m * R0r + q] = io::Ld<R>::load(dw, offw + e * WZ);
                }
#pragma unroll
            for (int i = 0; i < W0; ++i) {
                cx<R> z;
                if constexpr (CPLX) z = {ra[i].x * w[i], ra[i].y * w[i]};
                else z = {ra[i] * w[i], rb[i] * w[i]};
                v0[i] = fft::cadd(v0[i], fft::cmul(z, c));
            }
            off += N * SZ;
            offw += N * WZ;
        }
#pragma unroll
        for (int m = 0; m < M0; ++m) {
            const int j = t + T * m;
            if ((m + 1) * T <= NB0 || j < NB0) {
                cx<R> v[R0r];
#pragma unroll
                for (int q = 0; q < R0r; ++q) v[q] = k1 == 0 ? v0[m * R0r + q] : fft::cmul(v0[m * R0r + q], twc[m * R0r + q]);
                fft::gen_bfly<R0r>(v);
                cx<R>* o = bufA + (unsigned)j * (unsigned)(R0r + (S::padded(0) ? 1 : 0));
#pragma unroll
                for (int q = 0; q < R0r; ++q) fft::st2(o + q, v[q]);
            }
        }
        __syncthreads();
        // ---- the other passes exactly as gen_ct_kernel's register-consumed modes
        const cx<R>* src = bufA;
        if constexpr (INPL) ct_passes_inplace<S, 1, S::P - 1>(bufA, tw, t, t2);
        else src = ct_passes<S, 1, S::P - 1>(bufA, bufB, tw, t, t2);
        ct_last_pass_regs<S>(src, tw, t, t2, [&](int m, int q, int, cx<R> z) { acc[m * RL + q] += (double)(z.x * z.x + z.y * z.y); });   // (a unit that does not exist transformed zeros)
        __syncthreads();
    }
    double* part = static_cast<double*>(a.out) + (gslot * a.nch + ch) * (int64_t)ca.nfft + k1;
#pragma unroll
    for (int m = 0; m < ML; ++m) {
        const int j = t + T * m;
        if ((m + 1) * T <= NBL || j < NBL) {
#pragma unroll
            for (int q = 0; q < RL; ++q) part[(int64_t)(j + NBL * q) * ca.R0] = acc[m * RL + q];
        }
    }
}

template <typename R> __global__ __launch_bounds__(256) void cols_window_kernel(const double* __restrict__ win, R* __restrict__ out, int n, int nfft) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nfft) out[i] = i < n ? (win ? (R)win[i] : (R)1) : (R)0;
}

template <typename R, bool CPLX, typename S> int cols_launch(ColsArgs& ca, int64_t nch, hipStream_t st, int64_t* ngroups, DevBuf* partial) {
    auto kern = gen_ct_cols_kernel<R, CPLX, S>;
    GenArgs& a = ca.g;
    hipFuncAttributes fa{};
    MDSP_HIP(hipFuncGetAttributes(&fa, (const void*)kern));
    const int regs = std::max(8, (fa.numRegs + 7) / 8 * 8), waves = S::T / 64;
    const size_t lds_bytes = sizeof(cx<R>) * ((S::INPLACE ? 1 : 2) * (size_t)S::NP + (S::TW2L ? S::TWS + S::NTWHI : 0));
    int per_cu = std::min<int>({32 / waves, (512 / regs) * 4 / waves, (int)((size_t)160 * 1024 / std::max<size_t>(lds_bytes, 1))});
    if (per_cu < 1) per_cu = 1;
    if (tunables().wg_per_cu > 0) per_cu = tunables().wg_per_cu;
    const int64_t resident = std::max<int64_t>(1, (int64_t)device_cu_count() * per_cu / std::max<int64_t>(1, nch));
    int64_t groups = std::max<int64_t>(1, std::min<int64_t>(a.units_per_ch, resident / ca.R0));
    groups = std::max<int64_t>(8, groups / 8 * 8);   // the XCD mapping walks groups in eights; rounded DOWN: 85 -> 88 groups of three workgroups are 264 on 256 CUs, a second round for 8
    a.per_slot = cdiv(a.units_per_ch, groups);
    *ngroups = groups;
    MDSP_TRY(partial->reserve(sizeof(double) * (size_t)groups * (size_t)nch * (size_t)ca.nfft));
    a.out = partial->p;
    hipLaunchKernelGGL(kern, dim3((unsigned)(groups * ca.R0), (unsigned)nch), dim3(S::T), 0, st, ca);
    MDSP_LAUNCH_CHECK();
    return MDSP_OK;
}

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
int ctcols_split(int dtype, int64_t nfft) {
    if (dtype_is_double(dtype)) {
        for (int R0 = 2; R0 <= 4; ++R0) {
            if (nfft % R0) continue;
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
    for (int R0 = 2; R0 <= 8; ++R0) {
        if (nfft % R0 || (R0 > 4 && !(R0 == 8 && nfft == 65536))) continue;
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
