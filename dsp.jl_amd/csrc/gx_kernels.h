// Run-time-schedule spectral kernel (round 6; the kernels: compiled by gx_inst_*.hip, launched through gx_run; planned and dispatched by spectral_gx.h inside spectral.hip): Welch / STFT / spectrogram / periodogram for
// EVERY 7-smooth transform size a workgroup can hold on ONE LDS buffer -- up to 16384 points Float32 / ComplexF32 (128 KiB + group padding), 8192
// Float64 / ComplexF64 -- and, with a fused column step in front, for nfft = R0 x S up to R0 = 32.  It closes the gap between the compile-time
// schedules of spectral_gen.h (23 sizes up to 8000 / 8192) and the multi-pass engine of bigfft.hip: welch_pgram(s) / periodogram / spectrogram / stft
// default to nfft = nextfastfft(length(s) >> 3) (periodograms.jl:560, :647, :828, :872; util.jl:134), so every default call on a 65 537 .. 10^6-sample
// signal asks for a 7-smooth size between 8193 and 131072 -- until this round the un-fused rocFFT pipeline at 0.01 - 0.02 of the HBM roofline.
//
//   frame(s) straight from the signal (buffer descriptors: hardware zero tail; the next unit's samples in flight in registers), windowed into LDS
//   -> passes  read | twiddle, butterfly | write  between two LDS buffers (or in place with a second barrier where only one fits; gx_pass.h)
//   -> last pass consumed from registers: |Z|^2 added to the sums, which live in LDS too (Welch), or the output column.
//
// The radix of a pass is a template constant behind a wave-uniform switch (butterflies fully unrolled, registers statically indexed); butterfly
// counts, strides and padding are run-time numbers of the plan's gx::Sched; twiddles come from a two-level table in LDS (2 x 128 .. 1024 entries).
// Nothing a thread keeps across that switch depends on the radix: samples, window and sums are held in NATURAL order (point t + T e) -- the first
// form kept them per first- / last-pass butterfly and the compiler reconciled thirteen register assignments at every join (42 vector instructions
// per point and pass, half of them moves; profiles/r06_gx_sessions.json).
//
// nfft = R0 x S (R0 > 1; sizes above the buffer, and the few below it whose factors do not fill 512 threads evenly): decimation in frequency by R0 --
//   X[k1 + R0 k2] = FFT_S( y_k1 )[k2],   y_k1[i] = W_nfft^{i k1} sum_{n1 < R0} x[S n1 + i] W_R0^{n1 k1}
// -- one workgroup per (frame sequence, k1): it forms its y_k1 while loading (R0 loads per point, R0 - 1 of them served by the L2: the R0 workgroups
// of a frame sequence run on the same XCD), then transforms S points as above.  No work buffer in HBM, no second kernel; bins leave interleaved.
#pragma once

#include "common.h"
#include "devio.h"
#include "gx_pass.h"
#include "gx_plan.h"

struct GxArgs {
    const void* s;
    void* out;             // Welch: double partials [group][ch][nfft];  columns: the output matrices
    const void* tw;        // cx<R>: lo1_S[128], hi_S[nhs]; R0 > 1: + lo1_nfft[128], hi_nfft[nhn], W_R0^{i} [R0]
    const void* win;       // R window[nfft] (ones without a window, zero tail)
    int64_t lds_, K, hop, nch, ldo, chs;
    int64_t units_per_ch;  // frames, or frame pairs (real signals in the pair-packed modes)
    int64_t per_slot;      // consecutive units per group of workgroups
    int n, nfft, nout, onesided, psd, accumulate;
    int R0, nhs, nhn;
    int flush;             // Welch: units between two flushes of the sums in LDS (working precision) into the group's Float64 partial row
    int pairs;             // real signal, two frames per transform (z = w (a + i b))
    int nbuf;              // 2: passes ping-pong between two LDS buffers (one barrier per pass); 1: in place (read | barrier | write | barrier)
    mdsp::gx::Sched sc;
    double r;
};

namespace gxk {
using mdsp::fft::cx;
using mdsp::gx::Sched;

template <typename T> __device__ __forceinline__ T load_so(__amdgpu_buffer_rsrc_t r, int voff, int soff);
template <> __device__ __forceinline__ float load_so<float>(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
template <> __device__ __forceinline__ double load_so<double>(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return __hiloint2double((int)v.y, (int)v.x);
}
template <> __device__ __forceinline__ cx<float> load_so<cx<float>>(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return {__uint_as_float(v.x), __uint_as_float(v.y)};
}
template <> __device__ __forceinline__ cx<double> load_so<cx<double>>(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return {__hiloint2double((int)v.y, (int)v.x), __hiloint2double((int)v.w, (int)v.z)};
}
template <typename R, typename TT, bool PAIR> __device__ __forceinline__ cx<R> windowed(TT xa, TT xb, R w) {
    if constexpr (sizeof(TT) == 2 * sizeof(R)) return {xa.x * w, xa.y * w};
    else if constexpr (PAIR) return {xa * w, xb * w};
    else return {xa * w, (R)0};
}
}  // namespace gxk

#define MDSP_GX_SWITCH(r, F)                           \
    switch (r) {                                       \
        MDSP_GX_RADIX_CASES_16(F)                      \
        default: break;                                \
    }

// MODE 0: Welch sums; 1: columns (STFT raw or PSD).  PAIR: real signal, two frames per transform (Welch always; columns when R0 == 1).
template <typename R> struct GxGeo {
    static constexpr int EMAX = 16;   // points per thread and pass
    static constexpr int LB = 512;    // threads per workgroup at most
    static constexpr int MINW = sizeof(R) == 8 ? 2 : 4;   // waves per SIMD the kernel is compiled for: Float32 128 registers per thread (two 512-thread workgroups per CU where the LDS admits them), Float64 256
};
// COL: nfft = R0 x S with R0 > 1 -- the decimation-in-frequency step in front of the first pass (its own instantiation: neither form carries the other's registers)
template <typename R, bool CPLX, int MODE, bool PAIR, bool COL>
__global__ __launch_bounds__(GxGeo<R>::LB, GxGeo<R>::MINW) void gx_kernel(GxArgs a) {
    using namespace mdsp;
    using mdsp::fft::cx;
    using TT = std::conditional_t<CPLX, cx<R>, R>;
    constexpr int EMAX = GxGeo<R>::EMAX, SZ = (int)sizeof(TT), WZ = (int)sizeof(R);
    static_assert(!(CPLX && PAIR), "pairs are two REAL frames");
    extern __shared__ __attribute__((aligned(16))) unsigned char gx_smem[];
    const gx::Sched& s = a.sc;
    const int S = s.N, T = s.T, t = threadIdx.x, PL = s.P - 1;
    // LDS: the root tables first (fixed addresses: a twiddle fetch is  and / shift + an immediate), then the buffer(s), then the Welch sums
    cx<R>* lo1 = reinterpret_cast<cx<R>*>(gx_smem);
    cx<R>* hiS = lo1 + gx::TWS;
    cx<R>* lo1N = hiS + a.nhs;                                     // (R0 > 1 only: the three tables of the column step)
    cx<R>* hiN = lo1N + gx::TWS;
    cx<R>* cw = hiN + a.nhn;
    cx<R>* buf0 = a.R0 > 1 ? cw + a.R0 : lo1N;
    cx<R>* buf1 = a.nbuf == 2 ? buf0 + s.np : buf0;
    R* accl = reinterpret_cast<R*>(buf0 + a.nbuf * s.np);          // MODE 0: S sums, natural bin order
    {
        const int ntab = gx::TWS + a.nhs + (a.R0 > 1 ? gx::TWS + a.nhn + a.R0 : 0);
        const cx<R>* g = static_cast<const cx<R>*>(a.tw);
        for (int i = t; i < ntab; i += T) fft::st2(lo1 + i, g[i]);
        if constexpr (MODE == 0)
            for (int i = t; i < S; i += T) accl[i] = (R)0;
    }
    __syncthreads();
    // group of workgroups g (one frame sequence), row k1 of the decimation in frequency; the R0 rows of a group on one XCD (workgroups go to the XCDs round-robin)
    int64_t g = blockIdx.x;
    int k1 = 0;
    if (a.R0 > 1) {
        const unsigned b = blockIdx.x, xcd = b & 7u, w = b >> 3;
        k1 = (int)(w % (unsigned)a.R0);
        g = (int64_t)(w / (unsigned)a.R0) * 8 + xcd;
    }
    const int64_t ch = blockIdx.y;
    const TT* sc = static_cast<const TT*>(a.s) + ch * a.lds_;
    const int64_t u0 = g * a.per_slot;
    constexpr bool col = COL;
    const int E = (S + T - 1) / T;   // natural-order points of a thread: t + T e, e < E <= EMAX
    double* part = MODE == 0 ? static_cast<double*>(a.out) + (g * a.nch + ch) * (int64_t)a.nfft + k1 : nullptr;
    bool flushed = false;
    const R m1 = (R)(1.0 / a.r), m2 = (R)(2.0 / a.r);
    const __amdgpu_buffer_rsrc_t dw = io::make_rsrc(a.win, (long long)a.nfft * WZ);

    auto unit_frame = [&](int64_t it, const TT*& fa, bool& live, bool& haveB) {
        const int64_t u = u0 + it;
        live = it < a.per_slot && u < a.units_per_ch;
        const int64_t f0 = live ? (PAIR ? 2 * u : u) : 0;
        haveB = PAIR && live && (f0 + 1) < a.K;
        fa = sc + f0 * a.hop;
    };
    // the samples of a unit in natural order, one VGPR offset + scalar offsets; a frame that does not exist reads zeros through an empty descriptor
    TT pa[COL ? 1 : EMAX], pb[(PAIR && !COL) ? EMAX : 1];
    R w[COL ? 1 : EMAX];
    auto issue = [&](int64_t it) __attribute__((always_inline)) {
        if constexpr (!COL) {
        const TT* fa;
        bool live, haveB;
        unit_frame(it, fa, live, haveB);
        const __amdgpu_buffer_rsrc_t da = io::make_rsrc(fa, live ? (long long)a.n * SZ : 0);
        int off = t * SZ;
        asm volatile("" : "+v"(off));
        // (groups of four elements behind one wave-uniform branch -- a branch per element makes every load its own basic block; offsets past the frame are
        // outside the descriptor and read zeros)
#pragma unroll
        for (int e0 = 0; e0 < EMAX; e0 += 4)
            if (e0 < E) {
#pragma unroll
                for (int e = e0; e < e0 + 4; ++e) pa[e] = gxk::load_so<TT>(da, off, T * e * SZ);
            }
        if constexpr (PAIR) {
            const __amdgpu_buffer_rsrc_t db = io::make_rsrc(fa + a.hop, haveB ? (long long)a.n * SZ : 0);
#pragma unroll
            for (int e0 = 0; e0 < EMAX; e0 += 4)
                if (e0 < E) {
#pragma unroll
                    for (int e = e0; e < e0 + 4; ++e) pb[e] = gxk::load_so<TT>(db, off, T * e * SZ);
                }
        }
        }
    };
    if constexpr (!col) {
#pragma unroll
        for (int e = 0; e < EMAX; ++e) w[e] = gxk::load_so<R>(dw, t * WZ, T * e * WZ);   // (past nfft: 0 -- those points are never stored)
        issue(0);
    }
    int since_flush = 0, cur = 0;
    for (int64_t it = 0; it < a.per_slot; ++it) {
        const int64_t u = u0 + it;
        const bool live = u < a.units_per_ch;
        // Everything below is a function of the thread index and loop invariants: left alone, the compiler hoists the address arithmetic of every radix case,
        // every store predicate and every twiddle index of the loop in front of it -- hundreds of live values, 600 - 15000 spill instructions.  An opaque
        // copy of the thread index per unit keeps that arithmetic where it is used.
        int tl = t;
        MDSP_OPAQUE_INT(tl);
        const int64_t o0 = ch * a.chs + (PAIR ? 2 * u : u) * a.ldo;
        // ---- the windowed frame (pair), natural order, into the buffer the first pass reads
        {
            cx<R>* dst = cur ? buf1 : buf0;
            if constexpr (!col) {
#pragma unroll
                for (int e0 = 0; e0 < EMAX; e0 += 4)
                    if (e0 < E) {
#pragma unroll
                        for (int e = e0; e < e0 + 4; ++e)
                            if (tl + T * e < S) fft::st2(dst + tl + T * e, gxk::windowed<R, TT, PAIR>(pa[e], pb[PAIR ? e : 0], w[e]));
                    }
                issue(it + 1);   // next unit: in flight through the passes
            } else {
                // y_k1[i] = W_nfft^{i k1} sum_n1 x[S n1 + i] w[S n1 + i] W_R0^{n1 k1}
                const TT* fa;
                bool lv, hb;
                unit_frame(it, fa, lv, hb);
                const __amdgpu_buffer_rsrc_t da = io::make_rsrc(fa, lv ? (long long)a.n * SZ : 0);
                const __amdgpu_buffer_rsrc_t db = io::make_rsrc(fa + (PAIR ? a.hop : 0), (PAIR && hb) ? (long long)a.n * SZ : 0);
                cx<R> y[EMAX];
#pragma unroll
                for (int e = 0; e < EMAX; ++e) y[e] = cx<R>{(R)0, (R)0};
                unsigned cidx = 0;   // n1 k1 mod R0
                for (int n1 = 0; n1 < a.R0; ++n1) {
                    const cx<R> c = fft::ld2(cw + cidx);
                    cidx += (unsigned)k1;
                    if (cidx >= (unsigned)a.R0) cidx -= (unsigned)a.R0;
                    TT xa[EMAX], xb[PAIR ? EMAX : 1];
                    R wv[EMAX];
#pragma unroll
                    for (int e0 = 0; e0 < EMAX; e0 += 4)
                        if (e0 < E) {
#pragma unroll
                            for (int e = e0; e < e0 + 4; ++e) {
                                // (a point past S belongs to the next segment: read, combined, never stored)
                                const int so = n1 * S + T * e;
                                xa[e] = gxk::load_so<TT>(da, tl * SZ, so * SZ);
                                if constexpr (PAIR) xb[e] = gxk::load_so<TT>(db, tl * SZ, so * SZ);
                                wv[e] = gxk::load_so<R>(dw, tl * WZ, so * WZ);
                            }
                        }
#pragma unroll
                    for (int e0 = 0; e0 < EMAX; e0 += 4)
                        if (e0 < E) {
#pragma unroll
                            for (int e = e0; e < e0 + 4; ++e) y[e] = fft::cadd(y[e], fft::cmul(gxk::windowed<R, TT, PAIR>(xa[e], xb[PAIR ? e : 0], wv[e]), c));
                        }
                }
#pragma unroll
                for (int e0 = 0; e0 < EMAX; e0 += 4)
                    if (e0 < E) {
#pragma unroll
                        for (int e = e0; e < e0 + 4; ++e)
                            if (tl + T * e < S) {
                                const unsigned i = (unsigned)(tl + T * e);
                                fft::st2(dst + i, k1 == 0 ? y[e] : fft::cmul(y[e], gx::tw2(lo1N, hiN, i * (unsigned)k1)));
                            }
                    }
            }
        }
        __syncthreads();
        // ---- the passes
        for (int p = 0; p <= PL; ++p) {
            cx<R>* src = cur ? buf1 : buf0;
            cx<R>* dst = (cur ^ (a.nbuf - 1)) ? buf1 : buf0;
            const bool inplace = a.nbuf == 1;
            const gx::Pass ps = gx::pass_of(s, p);
            const int radix = s.radix[p];
#define MDSP_GX_F(RR)                                                                                                              \
    {                                                                                                                              \
        cx<R> v[EMAX];                                                                                                             \
        gx::pass_read<RR, EMAX>(ps, tl, src, v);                                                                                  \
        if (inplace) __syncthreads();   /* every operand is in registers: the buffer may be overwritten */                          \
        gx::pass_butterflies<RR, EMAX>(ps, tl, v, lo1, hiS);                                                                      \
        if (p < PL) {                                                                                                              \
            gx::pass_write<RR, EMAX>(ps, tl, dst, v);                                                                             \
        } else if constexpr (MODE == 0) {                                                                                          \
            /* (a unit that does not exist transformed zeros and adds nothing) */                                                  \
            gx::last_consume<RR, EMAX>(ps, tl, v, [&](int, unsigned bin, cx<R> z) {                                                  \
                accl[bin] += z.x * z.x + z.y * z.y;   /* the bin is this thread's alone (an LDS atomic runs at a fraction of the rate) */           \
            });                                                                                                                    \
        } else if constexpr (!PAIR) {   /* complex signal (or a real one, one frame per transform, R0 > 1): bins straight from registers */ \
            gx::last_consume<RR, EMAX>(ps, tl, v, [&](int, unsigned bin, cx<R> z) {                                                  \
                const int k = k1 + (int)bin * a.R0;                                                                                \
                if (live && k < a.nout) {                                                                                          \
                    if (a.psd) {                                                                                                   \
                        R m = m1;                                                                                                  \
                        if (a.onesided && !(k == 0 || (k == a.nout - 1 && a.nfft % 2 == 0))) m = m2;                               \
                        R* o = static_cast<R*>(a.out) + o0 + k;                                                                    \
                        const R pw = z.x * z.x + z.y * z.y;                                                                        \
                        *o = a.accumulate ? fma(pw, m, *o) : pw * m;                                                               \
                    } else static_cast<cx<R>*>(a.out)[o0 + k] = z;                                                                 \
                }                                                                                                                  \
            });                                                                                                                    \
        } else {   /* real pairs: the natural-order spectrum goes through LDS once more (below) */                                \
            gx::pass_write_natural<RR, EMAX>(ps, tl, dst, v);                                                                        \
        }                                                                                                                          \
    }
            MDSP_GX_SWITCH(radix, MDSP_GX_F)
#undef MDSP_GX_F
            if (p < PL) {
                __syncthreads();
                cur ^= a.nbuf - 1;
            }
        }
        // two buffers: the next unit's frame goes where nobody reads any more -- not the buffer the last pass just read (slow threads may still be at it)
        if constexpr (MODE == 0) {
            cur ^= a.nbuf - 1;
            if (++since_flush == a.flush || it + 1 == a.per_slot) {
                // the sums of at most `flush` units (working precision, in LDS) go into the group's Float64 partial row -- its own row: no other workgroup touches these bins
                __syncthreads();
                for (int i = tl; i < S; i += T) {
                    double* o = part + (int64_t)i * a.R0;
                    const double add = (double)accl[i];
                    *o = flushed ? *o + add : add;
                    accl[i] = (R)0;
                }
                flushed = true;
                since_flush = 0;
                __syncthreads();
            }
        } else if constexpr (PAIR) {
            // A[k] = (Z[k] + conj Z[N-k]) / 2, B[k] = (Z[k] - conj Z[N-k]) / (2i): the mirror bin belongs to another thread
            __syncthreads();
            const cx<R>* nat = (cur ^ (a.nbuf - 1)) ? buf1 : buf0;
            if (live) {
                const int N = S;
                const bool haveB = (2 * u + 1) < a.K;
                for (int j = tl; j < a.nout; j += T) {
                    const bool mirror = j > N / 2;   // real -> two-sided: X[N-k] = conj(X[k]) (fft2oneortwosided!, periodograms.jl:234-244)
                    const int k = mirror ? N - j : j;
                    const cx<R> zk = fft::ld2(nat + k), zm = fft::ld2(nat + (k == 0 ? 0 : N - k));
                    cx<R> A = {(R)0.5 * (zk.x + zm.x), (R)0.5 * (zk.y - zm.y)};
                    cx<R> B = {(R)0.5 * (zk.y + zm.y), (R)0.5 * (zm.x - zk.x)};
                    if (a.psd) {
                        R m = m1;
                        if (a.onesided && !(j == 0 || (j == a.nout - 1 && N % 2 == 0))) m = m2;
                        R* o = static_cast<R*>(a.out) + o0 + j;
                        const R pa_ = A.x * A.x + A.y * A.y;
                        *o = a.accumulate ? fma(pa_, m, *o) : pa_ * m;
                        if (haveB) {
                            const R pb_ = B.x * B.x + B.y * B.y;
                            o[a.ldo] = a.accumulate ? fma(pb_, m, o[a.ldo]) : pb_ * m;
                        }
                    } else {
                        if (mirror) {
                            A.y = -A.y;
                            B.y = -B.y;
                        }
                        cx<R>* o = static_cast<cx<R>*>(a.out) + o0 + j;
                        *o = A;
                        if (haveB) o[a.ldo] = B;
                    }
                }
            }
            if (a.nbuf == 1) __syncthreads();   // (two buffers: the next frame goes into the one the last pass read, which nobody touches any more)
        } else {
            cur ^= a.nbuf - 1;
        }
    }
}

// ---- one kernel per translation unit (gx_inst_<id>.hip: a gx_kernel instantiation takes minutes to compile -- 22 unrolled radix cases at three sites) -----
// id = 10 * (R0 > 1) + 5 * double + {0: Welch real pairs, 1: Welch complex, 2: columns complex, 3: columns real pairs, 4: columns real, one frame per
// transform (R0 > 1 only)}
namespace mdsp {
int gx_run(int id, const GxArgs& a, unsigned grid_x, unsigned grid_y, int threads, size_t lds_bytes, hipStream_t st);
template <typename K> inline int gx_launch_kernel(K kern, const GxArgs& a, unsigned grid_x, unsigned grid_y, int threads, size_t lds_bytes, hipStream_t st) {
    if (lds_bytes > 48 * 1024) MDSP_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(kern, dim3(grid_x, grid_y), dim3(threads), lds_bytes, st, a);
    MDSP_LAUNCH_CHECK();
    return MDSP_OK;
}
#define MDSP_GX_DECL(ID) int gx_run_##ID(const GxArgs& a, unsigned grid_x, unsigned grid_y, int threads, size_t lds_bytes, hipStream_t st);
MDSP_GX_DECL(0) MDSP_GX_DECL(1) MDSP_GX_DECL(2) MDSP_GX_DECL(3) MDSP_GX_DECL(5) MDSP_GX_DECL(6) MDSP_GX_DECL(7) MDSP_GX_DECL(8)
MDSP_GX_DECL(10) MDSP_GX_DECL(11) MDSP_GX_DECL(12) MDSP_GX_DECL(14) MDSP_GX_DECL(15) MDSP_GX_DECL(16) MDSP_GX_DECL(17) MDSP_GX_DECL(19)
#undef MDSP_GX_DECL
}  // namespace mdsp
