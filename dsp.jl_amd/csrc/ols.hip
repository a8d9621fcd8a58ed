// Overlap-save FIR filtering / convolution (mdsp_ols_*).
//
// Reference loop being replaced (per block, serial, one buffer):  Filters/filt.jl:504-518
//     fill!(tmp1,0); copyto!(tmp1, npadbefore+1, x, xstart, n)      K1  segment (nb-1 history | L new)
//     mul!(tmp2, p1, tmp1)                                          F1  rfft
//     tmp2 .*= filterft                                             K2  spectral multiply (1/nfft folded in)
//     mul!(tmp1, p2, tmp2)                                          F2  unnormalised brfft
//     copyto!(out, off, tmp1, nb, min(L, nx-off+1))                 K3  save the L valid samples
// and its conv twin unsafe_conv_kern_os! (dspbase.jl:583-606, padded edge blocks :441-483).
//
// Block geometry (0-based here): block g covers outputs [g*L, g*L+L) and reads x[g*L-(nb-1) .. g*L-(nb-1)+nfft),
// zero outside [0,nx).  That single rule reproduces npadbefore / n of filt.jl:505-507 for the leading blocks,
// the trailing clamp, and the pad_before / pad_after / u_deficit arithmetic of the conv edge blocks.
//
// Two engines:
//   FUSED  : one persistent kernel; each workgroup slot packs TWO real blocks into one complex nfft-point FFT
//            (z = a + i b; linear filtering with real taps keeps them separable), does FFT -> *H -> IFFT in
//            registers/LDS and stores the valid samples.  HBM traffic = read x once (+ (nb-1)/L overlap, L2) and
//            write y once.
//   ROCFFT : K1 -> rocFFT R2C -> K2 -> rocFFT C2R -> K3 over chunks sized to stay in the 256 MiB Infinity Cache.
#include <algorithm>
#include <atomic>
#include <cstdlib>

// The butterflies' constant roots stay in VGPRs here: these kernels run both transform directions next to the filter spectrum and already
// use all 106 SGPRs -- with the roots in SGPR pairs (fft_lds.h) hipcc spills 14 more SGPRs into VGPR lanes and reads them back inside the
// loop (18 v_readlane per unit in the headline instantiation), which costs more than the 10 VGPRs it frees (occupancy unchanged).
#ifndef MDSP_FFT_SGPR_CONST
#define MDSP_FFT_SGPR_CONST 0
#endif

#include "common.h"
#include "devio.h"
#include "fft_lds.h"
#include "fft_wg.h"
#include "hostfft.h"
#include "rocfft_wrap.h"
#include "ols_plan.h"

using namespace mdsp;
using mdsp::fft::cx;

namespace {

// ======================================================================================================
// rocFFT-engine kernels
// ======================================================================================================
// K1: unit u = col*nblocks + g  ->  td[(u-u0)*nfft + i]
template <typename T>
__global__ __launch_bounds__(256) void ols_segment_kernel(const T* __restrict__ x, T* __restrict__ td, int64_t nx, int64_t ldx,
                                                          int64_t nblocks, int64_t L, int nb, int nfft, int64_t u0, int64_t nunits) {
    const int64_t u = u0 + blockIdx.x;
    if (u >= nunits) return;
    const int64_t col = u / nblocks, g = u - col * nblocks;
    const int64_t start = g * L - (nb - 1);
    const T* xc = x + col * ldx;
    T* dst = td + (int64_t)blockIdx.x * nfft;
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < nfft; i += gridDim.y * blockDim.x) {
        const int64_t idx = start + i;
        T v{};
        if (idx >= 0 && idx < nx) v = xc[idx];
        dst[i] = v;
    }
}

// K2: fd[b*nspec + k] *= H[k]
template <typename R>
__global__ __launch_bounds__(256) void ols_cmul_kernel(cx<R>* __restrict__ fd, const cx<R>* __restrict__ H, int nspec, int64_t count) {
    const int64_t b = blockIdx.x;
    if (b >= count) return;
    cx<R>* row = fd + b * nspec;
    for (int k = blockIdx.y * blockDim.x + threadIdx.x; k < nspec; k += gridDim.y * blockDim.x) row[k] = fft::cmul(row[k], H[k]);
}

// K3: y[col*ldy + g*L + j] = td[(u-u0)*nfft + nb-1 + j]
template <typename T>
__global__ __launch_bounds__(256) void ols_save_kernel(const T* __restrict__ td, T* __restrict__ y, int64_t nout, int64_t ldy,
                                                       int64_t nblocks, int64_t L, int nb, int nfft, int64_t u0, int64_t nunits) {
    const int64_t u = u0 + blockIdx.x;
    if (u >= nunits) return;
    const int64_t col = u / nblocks, g = u - col * nblocks;
    const int64_t off = g * L;
    const int64_t cnt = std::min<int64_t>(L, nout - off);
    const T* src = td + (int64_t)blockIdx.x * nfft + (nb - 1);
    T* yc = y + col * ldy + off;
    for (int64_t j = blockIdx.y * blockDim.x + threadIdx.x; j < cnt; j += (int64_t)gridDim.y * blockDim.x) yc[j] = src[j];
}

// ======================================================================================================
// Fused kernel
// ======================================================================================================
struct OlsFusedArgs {
    const void* x;
    void* y;
    const void* table;  // N forward roots
    const void* H;      // N-point filter spectrum, 1/N folded in
    int64_t nx, nout, ldx, ldy, L;
    int64_t nblocks;        // per column
    int64_t units_per_col;  // pairs (real) or blocks (complex)
    int64_t nunits;         // total (one past the last unit of this launch)
    int64_t u_begin;        // first unit of this launch (0 unless a block range is executed, mdsp_ols_exec_range)
    int nb;
    int64_t run_len;        // a slot takes runs of run_len consecutive units ...
    int64_t niter;          // ... runs_per_slot * run_len iterations in total (same for every slot)
    int ablate;             // profiling aid, -DMDSP_DEBUG_KNOBS builds only (MDSP_ABLATE): 1 skip HBM loads, 2 skip transforms, 4 skip stores
    int memprio;            // MDSP_OLS_PRIO: 1 loads, 2 stores, 3 both issued at raised wave priority (s_setprio)
    // ROWS form (the middle of the long-filter convolution, mdsp::ols_rows): block g is row g mod hrows of a two-pass transform of hrows x N points --
    // its own spectrum row H + (g mod hrows) N, and the inverse half of the inter-pass twiddle, conj(W^{(g mod hrows) n}), on the way out
    int hrows;
    const void *rt0, *rt1;  // W_{hrows N}^m = rt1[m >> rlogS] * rt0[m & (2^rlogS - 1)]
    int rlogS;
};

// Raw samples of one unit as they come from HBM: two real blocks (a, b) or one complex block.
template <typename R, int E, bool CPLX> struct OlsRaw {
    std::conditional_t<CPLX, cx<R>, R> a[E];
    std::conditional_t<CPLX, char, R> b[CPLX ? 1 : E];
};

// Position of a unit inside the (nx, ncols) problem: column, pair/block index within the column, liveness.
// All wave-uniform (SALU).  Unit schedule: slot s of S walks runs of `run_len` CONSECUTIVE units (run g of slot s
// starts at unit (g*S + s)*run_len).  With one run per slot (the default) every workgroup streams one contiguous
// piece of x and y: the nb-1 sample overlap between consecutive blocks and the prefetch of the next unit hit
// L1/L2, and reads and writes are sequential per workgroup (measured +15 % over a strided schedule).
struct OlsPos {
    int64_t col, p;
    bool live;
};
struct OlsWalk {
    int64_t base, j;   // current unit = base + j
};
__device__ __forceinline__ OlsPos ols_pos(const OlsFusedArgs& a, const OlsWalk& w, bool more) {
    const int64_t u = w.base + w.j;
    OlsPos q;
    q.live = more && u < a.nunits;
    if (a.units_per_col >= a.nunits) {  // single column (the common case): no division at all
        q.col = 0;
        q.p = q.live ? u : 0;
    } else {
        q.col = q.live ? u / a.units_per_col : 0;
        q.p = q.live ? u - q.col * a.units_per_col : 0;
    }
    return q;
}
__device__ __forceinline__ void ols_walk_next(OlsWalk& w, int64_t run_len, int64_t nslots) {
    if (++w.j == run_len) {
        w.j = 0;
        w.base += nslots * run_len;
    }
}

template <typename R, int E, int T, bool CPLX>
__device__ __forceinline__ void ols_issue_loads(OlsRaw<R, E, CPLX>& raw, const OlsFusedArgs& a, OlsPos q, int t) {
    using TT = std::conditional_t<CPLX, cx<R>, R>;
    constexpr int64_t SZ = (int64_t)sizeof(TT);
    const TT* xc = static_cast<const TT*>(a.x) + q.col * a.ldx;
    const int64_t g0 = CPLX ? q.p : 2 * q.p;      // first block of the unit
    const int64_t start = g0 * a.L - (a.nb - 1);  // window start in x; negative for the leading blocks
    {
        const __amdgpu_buffer_rsrc_t r = io::make_rsrc(xc + start, q.live ? (a.nx - start) * SZ : 0);
        const int lead = __builtin_amdgcn_readfirstlane((int)(start < 0 ? -start : 0));
        io::load_window<TT, E, T>(raw.a, r, lead, t);
    }
    if constexpr (!CPLX) {
        const bool haveB = q.live && (g0 + 1) < a.nblocks;
        const int64_t startB = start + a.L;
        const __amdgpu_buffer_rsrc_t r = io::make_rsrc(xc + startB, haveB ? (a.nx - startB) * SZ : 0);
        const int lead = __builtin_amdgcn_readfirstlane((int)(startB < 0 ? -startB : 0));
        io::load_window<TT, E, T>(raw.b, r, lead, t);
    }
}

template <typename R, int E, int T, bool CPLX>
__device__ __forceinline__ void ols_store(const cx<R> (&v)[E], const OlsFusedArgs& a, OlsPos q, int t) {
    using TT = std::conditional_t<CPLX, cx<R>, R>;
    constexpr int64_t SZ = (int64_t)sizeof(TT);
    const int64_t off0 = (CPLX ? q.p : 2 * q.p) * a.L;  // first output of the unit
    const int lead = a.nb - 1;                          // K3: the first nb-1 samples of a block are aliased -> dropped
    TT* yc = static_cast<TT*>(a.y) + q.col * a.ldy;
    {
        const __amdgpu_buffer_rsrc_t w = io::make_rsrc(yc + off0 - lead, q.live ? (a.nout - off0 + lead) * SZ : 0);
        if constexpr (CPLX) io::store_window<TT, E, T>([&](int e) { return v[e]; }, w, lead, t);
        else io::store_window<TT, E, T>([&](int e) { return v[e].x; }, w, lead, t);
    }
    if constexpr (!CPLX) {
        const int64_t offB = off0 + a.L;  // past nout when the unit has no second block: num_records <= 0 drops it
        const __amdgpu_buffer_rsrc_t w = io::make_rsrc(yc + offB - lead, (q.live && offB < a.nout) ? (a.nout - offB + lead) * SZ : 0);
        io::store_window<TT, E, T>([&](int e) { return v[e].y; }, w, lead, t);
    }
}

// Staged store of a unit of REAL Float32 blocks ("a store path decoupled from ..."): after the inverse transform a thread holds
// outputs i = t + T e -- one 4-byte store per lane, 256 contiguous bytes per wave instruction, at an odd element offset (L = nfft - nb + 1).
// With the transform removed those stores alone take 1.89 ms per 2^30 samples (2.3 TB/s) against 0.82 ms for the loads and 5.2 TB/s for a float4
// write stream (profiles/r02o_ablation.json): the store PATTERN, not HBM, bounded the kernel.  Here the unit's 2 L valid outputs -- which are
// CONTIGUOUS in y: block a then block b -- go through the (now idle) exchange buffer: ds_write_b32 in thread order, one barrier, ds_read_b128,
// and every lane stores 16 bytes: 1 KiB contiguous per wave instruction.  Only units that lie wholly inside the output take this path.
template <typename R, int E, int T, int N>
__device__ __forceinline__ void ols_store_staged(const cx<R> (&v)[E], const OlsFusedArgs& a, OlsPos q, int t, float* S) {
    static_assert(sizeof(R) == 4, "staged stores are wired for Float32");
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int lead = a.nb - 1, L = (int)a.L, twoL = 2 * L;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int j = t + T * e - lead;
        if (j >= 0) {
            S[j] = v[e].x;
            S[L + j] = v[e].y;
        }
    }
    fft::wg_sync<T>();
    constexpr int NV = (2 * N + 4 * T - 1) / (4 * T);   // float4 vectors per thread (2 L <= 2 N floats)
    f4 r[NV];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const int idx = 4 * (t + T * c);
        r[c] = idx + 4 <= ((twoL + 3) & ~3) ? *reinterpret_cast<const f4*>(S + idx) : f4{0.f, 0.f, 0.f, 0.f};   // S is 16-byte aligned, idx a multiple of 4
    }
    fft::wg_sync<T>();   // the buffer is the next unit's first exchange
    const int64_t off0 = 2 * q.p * a.L;
    float* yc = static_cast<float*>(a.y) + q.col * a.ldy + off0;
    const __amdgpu_buffer_rsrc_t w = io::make_rsrc(yc, (int64_t)twoL * 4);
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const int idx = 4 * (t + T * c);
        if (idx + 4 <= twoL) {
            u4 d;
            d.x = __float_as_uint(r[c].x); d.y = __float_as_uint(r[c].y); d.z = __float_as_uint(r[c].z); d.w = __float_as_uint(r[c].w);
            __builtin_amdgcn_raw_buffer_store_b128(d, w, idx * 4, 0, MDSP_IO_AUX_STORE);
        } else if (idx < twoL) {   // the one vector that straddles the end of the unit (2 L is not a multiple of 4)
            io::Ld<float>::store(r[c].x, w, idx * 4);
            if (idx + 1 < twoL) io::Ld<float>::store(r[c].y, w, idx * 4 + 4);
            if (idx + 2 < twoL) io::Ld<float>::store(r[c].z, w, idx * 4 + 8);
        }
    }
}

// ---- LDS-DMA staging of the NEXT unit's samples (round 3) ----------------------------------------------------------------------------
// Without a prefetch a unit's wave parks ~2 us on its loads at the top of every iteration, and with four two-wave workgroups per CU only
// ~25 KiB per CU are outstanding on average: the read side of the kernel is capped by bytes in flight (bandwidth = outstanding / latency),
// not by HBM.  A register prefetch costs the VGPRs that pay for the fourth workgroup (round 2 measured it slower).  buffer_load ... lds costs
// none: the contiguous span of the next unit -- its two real blocks overlap by nb - 1 samples, so it is ONE run of L + N samples -- goes from
// HBM straight into a 2 N-sample LDS buffer while this unit is transformed, and the next iteration starts with ds_reads instead of a trip to
// memory.  Hand-issued (the compiler would make every later ds_read wait for a DMA it knows about); M0 holds the LDS address and is saved
// and restored inside the statement.
using io::dma_i4;
using io::dma_rsrc;
using io::dma256;
using io::dma64;
// A unit whose L + N samples lie wholly inside the column (both blocks exist, no zero padding in front or behind) is staged; the handful of
// edge units of a column keep the direct loads (hardware zero fill).
__device__ __forceinline__ bool ols_unit_interior(const OlsFusedArgs& a, OlsPos q, int N) {
    const int64_t g0 = 2 * q.p, start = g0 * a.L - (a.nb - 1);
    return q.live && (g0 + 1) < a.nblocks && start >= 0 && start + a.L + N <= a.nx;
}
// issue the DMA of unit q's span into `stage` (T threads = T / 64 waves share the granules)
template <int T> __device__ __forceinline__ void ols_dma_issue(const OlsFusedArgs& a, OlsPos q, int N, float* stage, int tid) {
    const float* xc = static_cast<const float*>(a.x) + q.col * a.ldx;
    const int64_t start = 2 * q.p * a.L - (a.nb - 1);
    const int span = (int)a.L + N;                                  // floats
    const dma_i4 r = dma_rsrc(xc + start, (long long)span * 4);
    const unsigned base = io::lds_byte_address(stage);
    const int wave = __builtin_amdgcn_readfirstlane(tid / 64), lane = tid & 63;
    constexpr int NW = T / 64;
    const int nfull = span / 256;
    for (int g = wave; g < nfull; g += NW) dma256(r, base + (unsigned)g * 1024u, g * 1024 + lane * 16);
    const int ntail = (span - nfull * 256 + 63) / 64;               // 64-dword granules behind the last full one
    for (int g = wave; g < ntail; g += NW) dma64(r, base + (unsigned)nfull * 1024u + (unsigned)g * 256u, nfull * 1024 + g * 256 + lane * 4);
}

template <typename R, int N, int E, int G, int TWMODE, int PADSHIFT, bool CPLX, int MINW, int NBUF, bool PREFETCH, bool HREG = true, int PERM = false, bool STAGE = false, bool XDMA = false,
          bool ROWS = false>
__global__ __launch_bounds__((N / E) * G, MINW) void ols_fused_kernel(OlsFusedArgs a) {
    static_assert(!ROWS || (CPLX && !HREG && !PERM), "the rows form: complex blocks, one spectrum row per block");
    using C = fft::Cfg<N, E>;
    constexpr int T = C::T;
    static_assert(T % 64 == 0, "a transform must own whole wavefronts (uniform descriptors)");
    constexpr int NTWA = C::NTW > 0 ? C::NTW : 1;
    constexpr int REGION = fft::wg_lds_elems<C, PADSHIFT, NBUF>();
    __shared__ __attribute__((aligned(16))) cx<R> lds_all[G * REGION];
    const int t = threadIdx.x % T;
    const int ti = fft::io_lane<C, PERM>(t);   // this thread owns elements ti + T*e of a unit (ti == t unless lane-permuted)
    const int slot = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / T));  // wave-uniform by construction (T % 64 == 0)
    cx<R>* lds = lds_all + slot * REGION;
    const cx<R>* table = static_cast<const cx<R>*>(a.table);

    cx<R> tw[NTWA];
    __shared__ __attribute__((aligned(16))) cx<R> twl[(TWMODE == fft::TW_LDS || TWMODE == fft::TW_HYB) ? fft::tw_lds_entries<C, TWMODE>() : 1];
    const cx<R>* twsrc = fft::wg_twiddle_setup<C, TWMODE, PERM>(tw, twl, t, slot, table);
    cx<R> Hr[HREG ? E : 1];
    if constexpr (HREG) {
#pragma unroll
        for (int e = 0; e < E; ++e) Hr[e] = static_cast<const cx<R>*>(a.H)[ti + T * e];
    }
    const __amdgpu_buffer_rsrc_t hrsrc = io::make_rsrc(a.H, (int64_t)N * (int64_t)sizeof(cx<R>));

    const int64_t nslots = (int64_t)gridDim.x * G;
    OlsWalk walk{a.u_begin + ((int64_t)blockIdx.x * G + slot) * a.run_len, 0};
    OlsPos cur = ols_pos(a, walk, a.niter > 0);
    OlsRaw<R, E, CPLX> raw;
    constexpr bool DMA = XDMA && !CPLX && sizeof(R) == 4 && G == 1 && !PERM && !PREFETCH && C::P > 1;
    __shared__ __attribute__((aligned(16))) float stage[DMA ? 2 * N : 4];   // the staged span: L + N <= 2 N samples
    [[maybe_unused]] bool cur_staged = false;
    if constexpr (DMA) {
        cur_staged = ols_unit_interior(a, cur, N);                           // wave-uniform
        if (cur_staged) ols_dma_issue<T>(a, cur, N, stage, threadIdx.x);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if constexpr (PREFETCH) ols_issue_loads<R, E, T, CPLX>(raw, a, cur, ti);
    for (int64_t it = 0; it < a.niter; ++it) {   // same trip count for every slot (barriers inside)
        ols_walk_next(walk, a.run_len, nslots);
        const OlsPos nxt = ols_pos(a, walk, it + 1 < a.niter);
        if constexpr (DMA) {
            // every wave waited for its share of this unit's DMA before it left the previous iteration (or the prologue): after the barrier
            // the whole span is in `stage`
            fft::wg_sync<T>();
            if (cur_staged) {
                const int L = (int)a.L;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    raw.a[e] = stage[ti + T * e];
                    raw.b[e] = stage[L + ti + T * e];
                }
            } else if (!MDSP_ABLATED(a, 1)) ols_issue_loads<R, E, T, CPLX>(raw, a, cur, ti);
        } else if constexpr (!PREFETCH) {
            if (!MDSP_ABLATED(a, 1)) {
                if (a.memprio & 1) __builtin_amdgcn_s_setprio(3);
                ols_issue_loads<R, E, T, CPLX>(raw, a, cur, ti);
                if (a.memprio & 1) __builtin_amdgcn_s_setprio(0);
            }
        }
        cx<R> v[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if constexpr (CPLX) v[e] = raw.a[e];
            else v[e] = {raw.a[e], raw.b[e]};
        }
        // next unit's samples start streaming from HBM while this unit is transformed
        if constexpr (PREFETCH) { if (!MDSP_ABLATED(a, 1)) ols_issue_loads<R, E, T, CPLX>(raw, a, nxt, ti); }
        if (!MDSP_ABLATED(a, 2)) {
        if constexpr (DMA) {
            // the forward transform's first pass, by hand: behind its exchange barrier every wave has read `stage`, so the NEXT unit's span
            // may start to overwrite it -- a whole unit (two transforms) ahead of its use
            constexpr int BUF0 = 0;
            cx<R>* region = lds + BUF0 * fft::lds_elems<C::N, PADSHIFT>();
            fft::pass_compute<C, -1, 0, TWMODE, PADSHIFT, PERM>(v, t, tw, twsrc, region);
            fft::wg_sync<T>();
            cur_staged = ols_unit_interior(a, nxt, N);
            if (cur_staged) ols_dma_issue<T>(a, nxt, N, stage, threadIdx.x);
            fft::pass_reload<C, PADSHIFT, 1, PERM>(v, t, region);
            if constexpr (NBUF == 1) fft::wg_sync<T>();
            fft::wg_fft<C, -1, TWMODE, PADSHIFT, NBUF, 0, 1, PERM>(v, t, tw, twsrc, lds);
        } else
        fft::wg_fft<C, -1, TWMODE, PADSHIFT, NBUF, 0, 0, PERM>(v, t, tw, twsrc, lds);
        // spectral multiply (K2): natural order in registers
        if constexpr (HREG) {
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = fft::cmul(v[e], Hr[e]);
        } else {  // spectrum streamed from L2 (16 KiB, always resident) instead of living in 2E VGPRs
            cx<R> hh[E];
            if constexpr (ROWS) {
                const __amdgpu_buffer_rsrc_t hrow = io::make_rsrc(static_cast<const cx<R>*>(a.H) + (cur.p % a.hrows) * N, (int64_t)N * (int64_t)sizeof(cx<R>));
                io::load_window<cx<R>, E, T>(hh, hrow, 0, ti);
            } else
            io::load_window<cx<R>, E, T>(hh, hrsrc, 0, ti);
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = fft::cmul(v[e], hh[e]);
        }
        // inverse transform (unnormalised, like plan_brfft / inv(p).p)
        fft::wg_fft<C, +1, TWMODE, PADSHIFT, NBUF, (C::P - 1) % NBUF, 0, PERM>(v, t, tw, twsrc, lds);
        if constexpr (ROWS) {   // conj(W^{k1 (ti + T e)}) = conj(W^{k1 ti} W^{k1 T e}): one table walk per thread and row, E uniform ones (scalar loads)
            const cx<R>*t0 = static_cast<const cx<R>*>(a.rt0), *t1 = static_cast<const cx<R>*>(a.rt1);
            const unsigned k1 = (unsigned)(cur.p % a.hrows), mask = (1u << a.rlogS) - 1u;
            const auto root = [&](unsigned m) { return fft::cmul(t0[m & mask], t1[m >> a.rlogS]); };
            const cx<R> w0 = root(k1 * (unsigned)ti);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const cx<R> w = fft::cmul(w0, root(k1 * (unsigned)(T * e)));
                v[e] = fft::cmul(v[e], cx<R>{w.x, -w.y});
            }
        }
        }
        // no barrier needed here: with one buffer wg_fft ends every exchange with a barrier, with two the 2(P-1)
        // exchanges of a unit alternate buffers so the next unit's first write is two barriers behind its readers
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next unit's span has landed (issued two transforms ago); BEFORE this
                                                                              // unit's stores, which the counter would otherwise make us wait for as well
        if (!MDSP_ABLATED(a, 4)) {
            if constexpr (STAGE && !CPLX && sizeof(R) == 4 && (G == 1 || T == 64) && NBUF == 1 && !PERM) {
                // wave-uniform (workgroup-uniform: one transform per workgroup, or one-wave transforms): both blocks exist and lie inside y
                const bool whole = cur.live && (2 * cur.p + 1) < a.nblocks && (2 * cur.p + 2) * a.L <= a.nout;
                if (whole) ols_store_staged<R, E, T, N>(v, a, cur, ti, reinterpret_cast<float*>(lds));
                else ols_store<R, E, T, CPLX>(v, a, cur, ti);
            } else {
                if (a.memprio & 2) __builtin_amdgcn_s_setprio(3);
                ols_store<R, E, T, CPLX>(v, a, cur, ti);
                if (a.memprio & 2) __builtin_amdgcn_s_setprio(0);
            }
        }
        cur = nxt;
    }
}

// ======================================================================================================
// Partitioned overlap-save for long filters (fused engine, requested nfft beyond the in-LDS transform sizes)
// ======================================================================================================
// optimalfftfiltlength (dspbase.jl:268-291) asks for nfft = 16384 ... 2^20 once the filter has more than ~1100 taps: one block no longer fits
// a workgroup's LDS.  The same linear convolution is evaluated with a UNIFORMLY PARTITIONED filter instead: h = [h_0 | h_1 | ... | h_{P-1}],
// B taps each, N = 2B-point transforms,
//     X_m = FFT(x[(m-1)B .. (m+1)B)),     Y_m = sum_p X_{m-p} H_p,     y[mB .. (m+1)B) = IFFT(Y_m)[B .. 2B)
// (H_p = FFT([h_p, 0...]) / N).  A slot walks CONSECUTIVE blocks and keeps the last P-1 input spectra in REGISTERS (a frequency-domain delay
// line): one forward and one inverse transform per block whatever P is, every sample read from HBM once (the 50 % window overlap and nothing
// else comes from L2).  Real signals: a slot walks TWO runs of blocks at once, run A in the real and run B in the imaginary part -- with
// real taps the two never mix (the packing of ols_fused_kernel, but across runs instead of neighbouring blocks, because a delay line shifts
// by one block per step).  Each run is preceded by P-1 warm-up blocks that only fill the delay line (negative blocks are all zero padding).
struct UpolsArgs {
    const void* x;
    void* y;
    const void* table;   // N forward roots
    const void* Hp;      // P spectra of N bins, 1/N folded in
    int64_t nx, nout, ldx, ldy;
    int64_t nblocks;     // per column: blocks at / past it do not exist for this launch (ceil(nout / B), or the end of a block range)
    int64_t g_begin;     // first block of the launch (0, or the start of a block range)
    int64_t x_lo;        // samples below it are not dereferenced and read as zero (0; block ranges: where the caller's slice starts -- always at
                         // least P*B - (nb - 1) taps of zeros in front of the samples that matter, see mdsp_ols_exec_range)
    int64_t run_len;     // blocks per run; slot s of S owns blocks [g_begin + s*R*run_len, g_begin + (s+1)*R*run_len), R = 2 runs (real) or 1 (complex)
    int ablate;          // MDSP_DEBUG_KNOBS builds: 1 no input, 2 no transforms, 4 no stores, 8 no spectra loads (results are garbage)
    int memprio;         // MDSP_OLS_PRIO bit 0: a block's loads (next half window, partition 0's spectrum) issued at raised wave priority
};

// XDMA (Float32 real signals): the half windows live in a two-slot LDS ring per run.  A block's window is [previous half | new half]; the new
// half of the NEXT block is moved by `buffer_load_dword ... lds` while this block is transformed (no VGPRs, no exposed HBM latency, every
// sample fetched once instead of twice).  The ring is wave-private: wave w moves exactly the 64-float granules its own lanes read, so the only
// ordering needed is the wave's own s_waitcnt.
// NLDS: partitions 1 .. NLDS live in LDS for the whole launch (a workgroup multiplies the SAME bins by the same spectra block after block, but
// P spectra per thread do not fit in registers next to the delay line): half spectra, B + 1 bins each -- the taps are real, so
// H_p[N - k] = conj H_p[k] -- read in ascending lane order for the lower bins and descending for the mirrored ones (both conflict-free).  What is
// left streams from L2: profiles/r03g_longfilt_ablate.json has the L2 traffic of the spectra (24 B per sample at P = 3 against 8 B of signal) as
// the largest single item of the kernel's memory time.
template <typename R, int N, int E, int P, int TWMODE, int PADSHIFT, bool CPLX, int MINW, bool XDMA = false, int NLDS = 0>
__global__ __launch_bounds__(N / E, MINW) void upols_fused_kernel(UpolsArgs a) {
    using C = fft::Cfg<N, E>;
    using TT = std::conditional_t<CPLX, cx<R>, R>;
    constexpr int T = C::T, B = N / 2, H = E / 2;
    constexpr int64_t SZ = (int64_t)sizeof(TT);
    static_assert(T % 64 == 0 && P >= 2 && P <= 4, "geometry");
    static_assert(!XDMA || (!CPLX && sizeof(R) == 4), "the DMA ring moves Float32 samples");
    constexpr int NTWA = C::NTW > 0 ? C::NTW : 1;
    constexpr int RUNS = CPLX ? 1 : 2;
    __shared__ __attribute__((aligned(16))) cx<R> lds[fft::wg_lds_elems<C, PADSHIFT, 1>()];
    __shared__ __attribute__((aligned(16))) cx<R> twl[(TWMODE == fft::TW_LDS || TWMODE == fft::TW_HYB) ? fft::tw_lds_entries<C, TWMODE>() : 1];
    __shared__ __attribute__((aligned(16))) float ring[XDMA ? RUNS * 2 * B : 1];   // [run][slot][B]
    static_assert(NLDS >= 0 && NLDS <= P - 1, "partitions 1 .. NLDS");
    constexpr int HLD = B + 2;                                                       // bins 0 .. B of a half spectrum (+1: 16-byte rows)
    __shared__ __attribute__((aligned(16))) cx<R> hl[NLDS > 0 ? NLDS * HLD : 1];
    const int t = threadIdx.x;
    if constexpr (NLDS > 0) {
        for (int i = t; i < NLDS * (B + 1); i += T) {
            const int q = i / (B + 1), k = i - q * (B + 1);
            hl[q * HLD + k] = static_cast<const cx<R>*>(a.Hp)[(int64_t)(q + 1) * N + k];
        }
        __syncthreads();
    }
    const cx<R>* table = static_cast<const cx<R>*>(a.table);
    cx<R> tw[NTWA];
    const cx<R>* twsrc = fft::wg_twiddle_setup<C, TWMODE>(tw, twl, t, 0, table);
    const __amdgpu_buffer_rsrc_t hrsrc = io::make_rsrc(a.Hp, (int64_t)P * N * (int64_t)sizeof(cx<R>));
    const int64_t col = blockIdx.y;
    const TT* xc = static_cast<const TT*>(a.x) + col * a.ldx;
    TT* yc = static_cast<TT*>(a.y) + col * a.ldy;
    const int64_t first = a.g_begin + (int64_t)blockIdx.x * RUNS * a.run_len;   // run A: [first, first + run_len), run B: the run_len blocks after it

    cx<R> zp[P - 1][E];   // delay line: zp[q] = spectrum of the block q + 1 steps back
#pragma unroll
    for (int q = 0; q < P - 1; ++q)
#pragma unroll
        for (int e = 0; e < E; ++e) zp[q][e] = {(R)0, (R)0};

    // window of block m of this column: x[(m-1)B .. (m+1)B), zero outside [x_lo, nx)
    auto load_block = [&](TT (&dst)[E], int64_t m, bool on) {
        const int64_t start = (m - 1) * B;
        const bool live = on && start + N > a.x_lo && start < a.nx;
        const __amdgpu_buffer_rsrc_t r = io::make_rsrc(xc + start, live ? (a.nx - start) * SZ : 0);
        const int lead = __builtin_amdgcn_readfirstlane((int)(live && start < a.x_lo ? a.x_lo - start : 0));
        io::load_window<TT, E, T>(dst, r, lead, t);
    };
    // ring: half h = x[h B, (h+1) B) of a run into `slot`; element i of the half belongs to thread i % T (wave-private granules)
    const unsigned ring_base = io::lds_byte_address(ring);
    const int wave64 = __builtin_amdgcn_readfirstlane(t & ~63);
    auto fill_half = [&](int run, int slot, int64_t h, bool on) {
        if constexpr (XDMA) {
            const int64_t start = h * B;
            const bool live = on && start + B > a.x_lo && start < a.nx;
            const bool interior = live && start >= a.x_lo && start + B <= a.nx;      // wave-uniform
            float* dst = ring + (run * 2 + slot) * B;
            if (interior) {
                const io::dma_i4 r = io::dma_rsrc(xc + start, (long long)B * 4);
                const unsigned base = ring_base + (unsigned)((run * 2 + slot) * B + wave64) * 4u;
#pragma unroll
                for (int e = 0; e < H; ++e) io::dma64(r, base + (unsigned)(T * e) * 4u, (t + T * e) * 4);
            } else {   // edges and switched-off runs: through registers (zero outside the signal)
                const __amdgpu_buffer_rsrc_t r = io::make_rsrc(xc + start, live ? (a.nx - start) * SZ : 0);
                const int lead = __builtin_amdgcn_readfirstlane((int)(live && start < a.x_lo ? a.x_lo - start : 0));
                float tmp[H];
#pragma unroll
                for (int e = 0; e < H; ++e) tmp[e] = io::Ld<float>::load(r, (t + T * e) < lead ? io::OOB : (t + T * e) * 4);
#pragma unroll
                for (int e = 0; e < H; ++e) dst[t + T * e] = tmp[e];
            }
        }
    };
    if constexpr (XDMA) {   // the first window of each run
        const int64_t m0 = first - (P - 1);
        const bool runA = first < a.nblocks, runB = (first + a.run_len) < a.nblocks;
        fill_half(0, 1, m0 - 1, runA);
        fill_half(0, 0, m0, runA);
        fill_half(1, 1, m0 + a.run_len - 1, runB);
        fill_half(1, 0, m0 + a.run_len, runB);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    int par = 0;   // ring slot of the NEW half of the current block
    for (int64_t k = -(P - 1); k < a.run_len; ++k) {   // same trip count for every workgroup (barriers inside)
        const int64_t mA = first + k, mB = first + a.run_len + k;
        // blocks in front of a run's first block only feed the delay line; blocks at / past nblocks produce nothing (and run B may be empty)
        const bool onA = first < a.nblocks && mA < a.nblocks, onB = !CPLX && (first + a.run_len) < a.nblocks && mB < a.nblocks;
        cx<R> v[E];
        if (MDSP_ABLATED(a, 1)) {
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = {(R)(t + e), (R)(k & 3)};
        } else if constexpr (XDMA) {
            const float* lo0 = ring + (0 * 2 + (par ^ 1)) * B;
            const float* hi0 = ring + (0 * 2 + par) * B;
            const float* lo1 = ring + (1 * 2 + (par ^ 1)) * B;
            const float* hi1 = ring + (1 * 2 + par) * B;
#pragma unroll
            for (int e = 0; e < H; ++e) {
                v[e] = {lo0[t + T * e], lo1[t + T * e]};
                v[e + H] = {hi0[t + T * e], hi1[t + T * e]};
            }
            // the halves just read as "previous" are dead: the next block's new halves land there while this block is transformed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (k + 1 < a.run_len) {
                fill_half(0, par ^ 1, mA + 1, first < a.nblocks && mA + 1 < a.nblocks);
                fill_half(1, par ^ 1, mB + 1, (first + a.run_len) < a.nblocks && mB + 1 < a.nblocks);
            }
            par ^= 1;
        } else {
            TT ra[E];
            load_block(ra, mA, onA);
            if constexpr (CPLX) {
#pragma unroll
                for (int e = 0; e < E; ++e) v[e] = ra[e];
            } else {
                TT rb[E];
                load_block(rb, mB, onB);
#pragma unroll
                for (int e = 0; e < E; ++e) v[e] = {ra[e], rb[e]};
            }
        }
        if (!MDSP_ABLATED(a, 2)) fft::wg_fft<C, -1, TWMODE, PADSHIFT, 1, 0>(v, t, tw, twsrc, lds);
        if (k >= 0) {   // wave-uniform
            cx<R> y[E];
            {
                cx<R> hh[E];
                if (MDSP_ABLATED(a, 8)) {
#pragma unroll
                    for (int e = 0; e < E; ++e) hh[e] = {(R)0.5, (R)(e + 1)};
                } else
                io::load_window<cx<R>, E, T>(hh, hrsrc, 0, t);
#pragma unroll
                for (int e = 0; e < E; ++e) y[e] = fft::cmul(v[e], hh[e]);
            }
#pragma unroll
            for (int q = 0; q < P - 1; ++q) {
                cx<R> hh[E];
                const __amdgpu_buffer_rsrc_t hq = io::make_rsrc(static_cast<const cx<R>*>(a.Hp) + (int64_t)(q + 1) * N, (int64_t)N * (int64_t)sizeof(cx<R>));
                if (MDSP_ABLATED(a, 8)) {
#pragma unroll
                    for (int e = 0; e < E; ++e) hh[e] = {(R)0.25, (R)(e + q)};
                } else if (q < NLDS) {
                    const cx<R>* up = hl + q * HLD + t;           // bins t + T e, e < E/2
                    const cx<R>* dn = hl + q * HLD + (B - t);     // bins N - (t + T e) = B - t - T (e - E/2), e >= E/2
#pragma unroll
                    for (int e = 0; e < H; ++e) {
                        hh[e] = up[T * e];
                        const cx<R> m = dn[-T * e];
                        hh[e + H] = {m.x, -m.y};
                    }
                } else
                io::load_window<cx<R>, E, T>(hh, hq, 0, t);
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const cx<R> pr = fft::cmul(zp[q][e], hh[e]);
                    y[e] = fft::cadd(y[e], pr);
                }
            }
            if (!MDSP_ABLATED(a, 2)) fft::wg_fft<C, +1, TWMODE, PADSHIFT, 1, 0>(y, t, tw, twsrc, lds);
            // the next block's halves have landed (issued two transforms ago); BEFORE this block's stores, which the counter would otherwise
            // make the next iteration wait for as well
            if constexpr (XDMA) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            // valid outputs: the upper half of the window, i.e. elements e >= E/2 of every thread (t + T*e >= B)
            if (!MDSP_ABLATED(a, 4) || y[0].x == (R)12345.678) {
                const int64_t o = mA * B;
                const __amdgpu_buffer_rsrc_t w = io::make_rsrc(yc + o, (onA && o < a.nout) ? (a.nout - o) * SZ : 0);
                int off = t * (int)SZ;
                asm volatile("" : "+v"(off));
#pragma unroll
                for (int e = 0; e < H; ++e) {
                    if constexpr (CPLX) io::Ld<TT>::store(y[e + H], w, off + T * e * (int)SZ);
                    else io::Ld<TT>::store(y[e + H].x, w, off + T * e * (int)SZ);
                }
            }
            if constexpr (!CPLX) {
                if (!MDSP_ABLATED(a, 4) || y[1].y == (R)12345.678) {
                    const int64_t o = mB * B;
                    const __amdgpu_buffer_rsrc_t w = io::make_rsrc(yc + o, (onB && o < a.nout) ? (a.nout - o) * SZ : 0);
                    int off = t * (int)SZ;
                    asm volatile("" : "+v"(off));
#pragma unroll
                    for (int e = 0; e < H; ++e) io::Ld<TT>::store(y[e + H].y, w, off + T * e * (int)SZ);
                }
            }
        } else if constexpr (XDMA) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        // shift the delay line
#pragma unroll
        for (int q = P - 2; q > 0; --q)
#pragma unroll
            for (int e = 0; e < E; ++e) zp[q][e] = zp[q - 1][e];
#pragma unroll
        for (int e = 0; e < E; ++e) zp[0][e] = v[e];
    }
}

// ---- round 3 form: every memory latency of a block taken off the transform -> multiply -> transform chain -------------------------------
// profiles/r03g_longfilt_ablate.json: the transforms alone take 0.72 ms of the round-2 kernel's 1.0 ms (2^28 samples, 5120 taps) and every
// memory operation ADDS its latency on top (input 0.17, the spectra 0.13, stores 0.08), because with 230 VGPRs only two workgroups share a CU.
//   * input: a block's window is [previous half | new half]; the previous half is carried in registers and the NEXT block's new half is
//     loaded one whole iteration ahead (E/2 values per run instead of E, every sample fetched once);
//   * partitions 1 .. P-1 multiply the DELAY LINE, which is known before the block's own transform: that part of the output spectrum is
//     accumulated first, from half spectra that live in LDS (NLDS of them; the taps are real, H_p[N-k] = conj H_p[k], so B + 1 bins each,
//     read in ascending lane order for the lower bins and descending for the mirrored ones) or stream from L2;
//   * partition 0 multiplies the block's own spectrum: H0REG keeps it in registers for the whole launch, otherwise its loads are issued
//     BEFORE the forward transform.
template <typename R, int N, int E, int P, int TWMODE, int PADSHIFT, bool CPLX, int MINW, int NLDS, bool H0REG>
__global__ __launch_bounds__(N / E, MINW) void upols2_fused_kernel(UpolsArgs a) {
    using C = fft::Cfg<N, E>;
    using TT = std::conditional_t<CPLX, cx<R>, R>;
    constexpr int T = C::T, B = N / 2, H = E / 2;
    constexpr int64_t SZ = (int64_t)sizeof(TT);
    static_assert(T % 64 == 0 && P >= 2 && P <= 4 && NLDS >= 0 && NLDS <= P - 1, "geometry");
    constexpr int NTWA = C::NTW > 0 ? C::NTW : 1;
    constexpr int RUNS = CPLX ? 1 : 2;
    constexpr int HLD = B + 2;                                                       // bins 0 .. B of a half spectrum (+1: 16-byte rows)
    __shared__ __attribute__((aligned(16))) cx<R> lds[fft::wg_lds_elems<C, PADSHIFT, 1>()];
    __shared__ __attribute__((aligned(16))) cx<R> twl[(TWMODE == fft::TW_LDS || TWMODE == fft::TW_HYB) ? fft::tw_lds_entries<C, TWMODE>() : 1];
    __shared__ __attribute__((aligned(16))) cx<R> hl[NLDS > 0 ? NLDS * HLD : 1];
    const int t = threadIdx.x;
    const cx<R>* Hg = static_cast<const cx<R>*>(a.Hp);
    if constexpr (NLDS > 0) {
        for (int i = t; i < NLDS * (B + 1); i += T) {
            const int q = i / (B + 1), k = i - q * (B + 1);
            hl[q * HLD + k] = Hg[(int64_t)(q + 1) * N + k];
        }
    }
    const cx<R>* table = static_cast<const cx<R>*>(a.table);
    cx<R> tw[NTWA];
    const cx<R>* twsrc = fft::wg_twiddle_setup<C, TWMODE>(tw, twl, t, 0, table);
    __syncthreads();
    const int64_t col = blockIdx.y;
    const TT* xc = static_cast<const TT*>(a.x) + col * a.ldx;
    TT* yc = static_cast<TT*>(a.y) + col * a.ldy;
    const int64_t first = a.g_begin + (int64_t)blockIdx.x * RUNS * a.run_len;   // run A: [first, first + run_len), run B: the run_len blocks after it
    const bool runA = first < a.nblocks, runB = !CPLX && (first + a.run_len) < a.nblocks;

    cx<R> h0[H0REG ? E : 1];
    if constexpr (H0REG) {
#pragma unroll
        for (int e = 0; e < E; ++e) h0[e] = Hg[t + T * e];
    }
    cx<R> zp[P - 1][E];   // delay line: zp[q] = spectrum of the block q + 1 steps back
#pragma unroll
    for (int q = 0; q < P - 1; ++q)
#pragma unroll
        for (int e = 0; e < E; ++e) zp[q][e] = {(R)0, (R)0};

    // half h of this column: x[h B, (h+1) B), zero outside [x_lo, nx) and for switched-off runs; thread t takes elements t + T e, e < E/2
    auto load_half = [&](TT (&dst)[H], int64_t h, bool on) {
        const int64_t start = h * B;
        const bool live = on && start + B > a.x_lo && start < a.nx;
        const __amdgpu_buffer_rsrc_t r = io::make_rsrc(xc + start, live ? (a.nx - start) * SZ : 0);
        const int lead = __builtin_amdgcn_readfirstlane((int)(live && start < a.x_lo ? a.x_lo - start : 0));
        io::load_window<TT, H, T>(dst, r, lead, t);
    };
    TT carryA[H], carryB[CPLX ? 1 : H], nxtA[H], nxtB[CPLX ? 1 : H];
    {
        const int64_t m0 = first - (P - 1);                      // the first (warm-up) block of run A; run B: run_len further on
        load_half(carryA, m0 - 1, runA && !MDSP_ABLATED(a, 1));
        load_half(nxtA, m0, runA && !MDSP_ABLATED(a, 1));
        if constexpr (!CPLX) {
            load_half(carryB, m0 + a.run_len - 1, runB && !MDSP_ABLATED(a, 1));
            load_half(nxtB, m0 + a.run_len, runB && !MDSP_ABLATED(a, 1));
        }
    }
    for (int64_t k = -(P - 1); k < a.run_len; ++k) {   // same trip count for every workgroup (barriers inside)
        const int64_t mA = first + k, mB = first + a.run_len + k;
        // blocks in front of a run's first block only feed the delay line; blocks at / past nblocks produce nothing (and run B may be empty)
        const bool onA = runA && mA < a.nblocks, onB = runB && mB < a.nblocks;
        cx<R> v[E];
#pragma unroll
        for (int e = 0; e < H; ++e) {
            if constexpr (CPLX) {
                v[e] = carryA[e];
                v[e + H] = nxtA[e];
            } else {
                v[e] = {carryA[e], carryB[e]};
                v[e + H] = {nxtA[e], nxtB[e]};
            }
            carryA[e] = nxtA[e];
            if constexpr (!CPLX) carryB[e] = nxtB[e];
        }
        // partition 0's spectrum first (it returns first: loads retire in order), then the next block's new halves: in flight for a whole block
        if (a.memprio & 1) __builtin_amdgcn_s_setprio(3);
        cx<R> hh0[H0REG ? 1 : E];
        if constexpr (!H0REG) {
            if (k >= 0 && !MDSP_ABLATED(a, 8)) io::load_window<cx<R>, E, T>(hh0, io::make_rsrc(Hg, (int64_t)N * (int64_t)sizeof(cx<R>)), 0, t);
        }
        if (k + 1 < a.run_len && !MDSP_ABLATED(a, 1)) {
            load_half(nxtA, mA + 1, runA && mA + 1 < a.nblocks);
            if constexpr (!CPLX) load_half(nxtB, mB + 1, runB && mB + 1 < a.nblocks);
        }
        if (a.memprio & 1) __builtin_amdgcn_s_setprio(0);
        cx<R> y[E];
        if (k >= 0) {   // wave-uniform: what the delay line contributes
#pragma unroll
            for (int q = P - 2; q >= 0; --q) {
                cx<R> hh[E];
                if (q < NLDS) {
                    const cx<R>* up = hl + q * HLD + t;           // bins t + T e, e < E/2
                    const cx<R>* dn = hl + q * HLD + (B - t);     // bins N - (t + T e) = B - t - T (e - E/2), e >= E/2
#pragma unroll
                    for (int e = 0; e < H; ++e) {
                        hh[e] = up[T * e];
                        const cx<R> m = dn[-T * e];
                        hh[e + H] = {m.x, -m.y};
                    }
                } else if (MDSP_ABLATED(a, 8)) {
#pragma unroll
                    for (int e = 0; e < E; ++e) hh[e] = {(R)0.25, (R)(e + q)};
                } else {
                    io::load_window<cx<R>, E, T>(hh, io::make_rsrc(Hg + (int64_t)(q + 1) * N, (int64_t)N * (int64_t)sizeof(cx<R>)), 0, t);
                }
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const cx<R> pr = fft::cmul(zp[q][e], hh[e]);
                    y[e] = q == P - 2 ? pr : fft::cadd(y[e], pr);
                }
            }
        }
        if (!MDSP_ABLATED(a, 2)) fft::wg_fft<C, -1, TWMODE, PADSHIFT, 1, 0>(v, t, tw, twsrc, lds);
        if (k >= 0) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                cx<R> hv;
                if constexpr (H0REG) hv = h0[e];
                else hv = MDSP_ABLATED(a, 8) ? cx<R>{(R)0.5, (R)(e + 1)} : hh0[e];
                y[e] = fft::cadd(y[e], fft::cmul(v[e], hv));
            }
        }
        // shift the delay line (before the inverse transform: v is dead afterwards)
#pragma unroll
        for (int q = P - 2; q > 0; --q)
#pragma unroll
            for (int e = 0; e < E; ++e) zp[q][e] = zp[q - 1][e];
#pragma unroll
        for (int e = 0; e < E; ++e) zp[0][e] = v[e];
        if (k >= 0) {
            if (!MDSP_ABLATED(a, 2)) fft::wg_fft<C, +1, TWMODE, PADSHIFT, 1, 0>(y, t, tw, twsrc, lds);
            // valid outputs: the upper half of the window, i.e. elements e >= E/2 of every thread (t + T*e >= B)
            if (!MDSP_ABLATED(a, 4) || y[0].x == (R)12345.678) {
                const int64_t o = mA * B;
                const __amdgpu_buffer_rsrc_t w = io::make_rsrc(yc + o, (onA && o < a.nout) ? (a.nout - o) * SZ : 0);
                int off = t * (int)SZ;
                asm volatile("" : "+v"(off));
#pragma unroll
                for (int e = 0; e < H; ++e) {
                    if constexpr (CPLX) io::Ld<TT>::store(y[e + H], w, off + T * e * (int)SZ);
                    else io::Ld<TT>::store(y[e + H].x, w, off + T * e * (int)SZ);
                }
            }
            if constexpr (!CPLX) {
                if (!MDSP_ABLATED(a, 4) || y[1].y == (R)12345.678) {
                    const int64_t o = mB * B;
                    const __amdgpu_buffer_rsrc_t w = io::make_rsrc(yc + o, (onB && o < a.nout) ? (a.nout - o) * SZ : 0);
                    int off = t * (int)SZ;
                    asm volatile("" : "+v"(off));
#pragma unroll
                    for (int e = 0; e < H; ++e) io::Ld<TT>::store(y[e + H].y, w, off + T * e * (int)SZ);
                }
            }
        }
    }
}

}  // namespace

// ======================================================================================================
// Plan object
// ======================================================================================================
// struct mdsp_ols_plan_s: ols_plan.h (shared with hostpath.hip)

namespace {

bool fused_supported(int dtype, int64_t nfft) {
    const bool dbl = dtype_is_double(dtype);
    switch (nfft) {
        case 256: case 512: case 1024: case 2048: case 4096: return true;
        case 8192: return !dbl;
        default: return false;
    }
}

// Execution geometry of the fused engine for a requested (reference) nfft:
//   * the transform sizes it holds in LDS run as they are;
//   * beyond them (filters of ~1100 taps and more, where optimalfftfiltlength returns 16384 ... 2^20) the SAME convolution is re-blocked:
//     one block of the largest in-LDS size while the filter covers at most half of it, else a uniformly partitioned filter
//     (upols_fused_kernel) with B = N/2 taps per partition and at most four partitions.
// Returns false when the fused engine cannot run the request (non power-of-two sizes, filters beyond 4 partitions).
bool fused_geometry(int dtype, int64_t nb, int64_t nfft_req, int64_t* exec_nfft, int* partitions) {
    *partitions = 1;
    *exec_nfft = nfft_req;
    if (fused_supported(dtype, nfft_req)) return true;
    const int64_t nmax = dtype_is_double(dtype) ? 4096 : 8192;
    if (nfft_req <= nmax || (nfft_req & (nfft_req - 1)) != 0) return false;   // small or mixed-radix sizes: not this engine
    if (nb - 1 <= nmax / 2) {
        *exec_nfft = nmax;
        return true;
    }
    for (int64_t n = 4096; n <= nmax; n *= 2) {
        const int64_t parts = cdiv(nb, n / 2);
        if (parts <= 4) {
            *exec_nfft = n;
            *partitions = (int)std::max<int64_t>(2, parts);
            return true;
        }
    }
    return false;
}

template <typename R> int upload_spectrum(mdsp_ols_plan_s* pl, const std::vector<zd>& Hfull, bool half) {
    const int64_t n = half ? pl->nfft / 2 + 1 : pl->nfft;
    std::vector<cx<R>> h((size_t)n);
    for (int64_t k = 0; k < n; ++k) h[(size_t)k] = {(R)Hfull[(size_t)k].real(), (R)Hfull[(size_t)k].imag()};
    MDSP_TRY(pl->H.reserve(sizeof(cx<R>) * (size_t)n));
    MDSP_HIP(hipMemcpy(pl->H.p, h.data(), sizeof(cx<R>) * (size_t)n, hipMemcpyHostToDevice));
    return MDSP_OK;
}

template <typename R> int upload_table(DevBuf& buf, int64_t n) {
    std::vector<cx<R>> w((size_t)n);
    for (int64_t k = 0; k < n; ++k) {
        const zd r = unit_root(k, n, -1);
        w[(size_t)k] = {(R)r.real(), (R)r.imag()};
    }
    MDSP_TRY(buf.reserve(sizeof(cx<R>) * (size_t)n));
    MDSP_HIP(hipMemcpy(buf.p, w.data(), sizeof(cx<R>) * (size_t)n, hipMemcpyHostToDevice));
    return MDSP_OK;
}

// ---- fused launch ---------------------------------------------------------------------------------------
template <typename R, int N, int E, int G, int TWMODE, int PADSHIFT, bool CPLX, int MINW = 2, int NBUF = 2, bool PREFETCH = true, bool HREG = true,
          int PERM = false, bool STAGE = false, bool XDMA = false, bool ROWS = false>
int launch_fused_variant(const OlsFusedArgs& a, hipStream_t s) {
    auto kern = ols_fused_kernel<R, N, E, G, TWMODE, PADSHIFT, CPLX, MINW, NBUF, PREFETCH, HREG, PERM, STAGE, XDMA, ROWS>;
    constexpr int threads = (N / E) * G;
    int per_cu = 0;
    MDSP_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, 0));
    if (per_cu < 1) per_cu = 1;
    if (tunables().wg_per_cu > 0) per_cu = tunables().wg_per_cu;
    const int64_t todo = a.nunits - a.u_begin;
    const int64_t want = cdiv(todo, G);
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(want, (int64_t)device_cu_count() * per_cu));
    OlsFusedArgs b = a;
    const int64_t runs = tunables().runs_per_slot;                     // runs per slot (1 = fully contiguous)
    const int64_t nslots = (int64_t)grid * G;
    b.run_len = std::max<int64_t>(1, cdiv(todo, nslots * runs));
    b.niter = cdiv(cdiv(todo, b.run_len), nslots) * b.run_len;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, s, b);
    MDSP_LAUNCH_CHECK();
    return MDSP_OK;
}

// The middle of the long-filter convolution (bigfft.hip run_ols_rows): `rows` rows of S complex points in place -- forward transform, the row's spectrum,
// inverse transform, inverse inter-pass twiddle -- on the single-workgroup transforms of this file (S = 8192 Float32, 4096 Float64).
int ols_rows_impl(int dbl, void* work, int64_t rows, int hrows, const void* Hrows, const void* table, const void* rt0, const void* rt1, int rlogS, hipStream_t st) {
    OlsFusedArgs a{};
    const int64_t S = dbl ? 4096 : 8192;
    a.x = work;
    a.y = work;
    a.table = table;
    a.H = Hrows;
    a.nx = a.nout = a.ldx = a.ldy = rows * S;
    a.L = S;
    a.nb = 1;
    a.nblocks = a.units_per_col = a.nunits = rows;
    a.u_begin = 0;
    a.memprio = tunables().ols_prio;
    a.hrows = hrows;
    a.rt0 = rt0;
    a.rt1 = rt1;
    a.rlogS = rlogS;
    //                                                 R   N   E  G TW PAD CPLX MINW NBUF PREF  HREG  PERM  STAGE  XDMA  ROWS
    if (dbl) return launch_fused_variant<double, 4096, 8, 1, 1, 4, true, 2, 2, false, false, false, false, false, true>(a, st);
    return launch_fused_variant<float, 8192, 16, 1, 1, 4, true, 2, 1, false, false, false, false, false, true>(a, st);
}

template <typename R, int N, bool CPLX> int launch_fused_n(const OlsFusedArgs& a, int variant, hipStream_t s) {
    // Geometry: E elements per thread so that a transform owns whole wavefronts (T = N/E >= 64); G transforms
    // per workgroup so that workgroups have 256 threads where possible.  `variant` selects tuning alternatives
    // (MDSP_OLS_VARIANT, swept by bench/tune.py); the non-default ones are only built for the headline shape.
    constexpr bool DBL = sizeof(R) == 8;
    // Elements per thread.  Float32, nfft >= 1024: 16 -- half as many waves per transform (one wave at 1024: no s_barrier at
    // all), measured 6-10 % faster than 8 at nfft = 2048 once the packed-FP32 butterflies removed the register spills
    // (profiles/r01i_tune_e16.json).  Float64 and small transforms: 8.
    constexpr int EMAX = (!DBL && N >= 1024) ? 16 : 8;
    constexpr int E = (N / 64 < EMAX) ? N / 64 : EMAX;
    constexpr int T = N / E;
    constexpr int G = slots_per_workgroup(T);
    constexpr int NBUF = (T <= 64 || E == 16) ? 1 : 2;
#ifndef MDSP_TW_F64
#define MDSP_TW_F64 1
#endif
    // Float64 twiddles: 1 = registers, 0 = global table (registers win even where they spill 84 B: ComplexF64 nfft 4096 85 vs 69 Gsamples/s)
    constexpr int TWREG = DBL ? MDSP_TW_F64 : 1;
    if constexpr (N == 2048 && !CPLX && !DBL) {
        switch (variant) {
            //                                    R  N   E  G  TW PAD CPLX MINW NBUF PREF HREG PERM   (TW: 0 global, 1 regs, 2 LDS)
            case 2: return launch_fused_variant<R, N, 8, 1, 1, 5, CPLX, 2, 2, true, true, false>(a, s);   // identity lanes, pad 5 (best E = 8 form)
#ifdef MDSP_DEBUG_KNOBS   // the variants that lost (HISTORY.md section 4.2): only in builds made with -DMDSP_DEBUG_KNOBS
            case 1: return launch_fused_variant<R, N, 8, 1, 1, 4, CPLX, 2, 2, true, true, false>(a, s);   // identity lanes, pad 4 (previous default)
            case 10: return launch_fused_variant<R, N, 8, 1, 1, 5, CPLX, 2, 2, true, true, true>(a, s);   // permuted lanes, pad 5
            case 11: return launch_fused_variant<R, N, 16, 2, 1, 5, CPLX, 2, 1, true, true, false>(a, s);  // E = 16: two waves per transform
            case 12: return launch_fused_variant<R, N, 16, 2, 1, 4, CPLX, 2, 1, true, true, false>(a, s);  // (= default)
            case 13: return launch_fused_variant<R, N, 16, 2, 1, 5, CPLX, 2, 2, true, true, false>(a, s);
            case 3: return launch_fused_variant<R, N, 8, 1, 1, 4, CPLX, 2, 2, true, true, true>(a, s);    // permuted lanes, pad 4
            case 4: return launch_fused_variant<R, N, 8, 1, 1, 3, CPLX, 2, 2, true, true, false>(a, s);
            case 5: return launch_fused_variant<R, N, 8, 1, 1, 5, CPLX, 3, 2, true, true, true>(a, s);
            case 6: return launch_fused_variant<R, N, 8, 1, 0, 5, CPLX, 2, 2, true, true, true>(a, s);    // global twiddles
            case 7: return launch_fused_variant<R, N, 8, 1, 1, 5, CPLX, 2, 1, true, true, true>(a, s);    // single LDS buffer
            case 8: return launch_fused_variant<R, N, 8, 1, 1, 5, CPLX, 2, 2, false, true, true>(a, s);   // no prefetch
            case 9: return launch_fused_variant<R, N, 16, 2, 1, 4, CPLX, 2, 1, true, false, false>(a, s);  // E = 16
            // register diets for three / four resident workgroups per CU: MINW caps the allocation at 168 / 128 VGPRs
            case 14: return launch_fused_variant<R, N, 16, 2, 1, 4, CPLX, 3, 1, true, true, false>(a, s);   // registers only, capped at 168
            case 15: return launch_fused_variant<R, N, 16, 2, 3, 4, CPLX, 3, 1, true, false, false>(a, s);  // hybrid twiddles + spectrum from L2
            case 16: return launch_fused_variant<R, N, 16, 2, 3, 4, CPLX, 3, 1, true, true, false>(a, s);   // hybrid twiddles, spectrum in registers
            case 17: return launch_fused_variant<R, N, 16, 2, 2, 4, CPLX, 4, 1, true, false, false>(a, s);  // LDS twiddles + spectrum from L2, capped at 128
            case 18: return launch_fused_variant<R, N, 16, 2, 2, 4, CPLX, 3, 1, true, false, false>(a, s);  // LDS twiddles + spectrum from L2, capped at 168
            case 19: return launch_fused_variant<R, N, 16, 2, 3, 4, CPLX, 3, 1, false, false, false>(a, s); // hybrid, L2 spectrum, no prefetch
            case 23: return launch_fused_variant<R, N, 16, 2, 3, 4, CPLX, 4, 1, false, false, false>(a, s); // same, capped at 128 VGPRs (four workgroups)
            case 24: return launch_fused_variant<R, N, 16, 2, 0, 4, CPLX, 4, 1, false, false, false>(a, s); // global twiddles, L2 spectrum, no prefetch, 128
            case 25: return launch_fused_variant<R, N, 16, 2, 3, 5, CPLX, 3, 1, false, false, false>(a, s); // 19 with pad 5
            case 26: return launch_fused_variant<R, N, 16, 2, 3, 4, CPLX, 3, 1, false, true, false>(a, s);  // hybrid, spectrum in registers, no prefetch
            case 27: return launch_fused_variant<R, N, 16, 2, 1, 4, CPLX, 3, 1, false, false, false>(a, s); // register twiddles, L2 spectrum, no prefetch
            case 28: return launch_fused_variant<R, N, 16, 2, 1, 4, CPLX, 2, 1, false, true, false>(a, s);  // default geometry minus the prefetch
            case 29: return launch_fused_variant<R, N, 16, 1, 3, 4, CPLX, 3, 1, false, true, false>(a, s);  // the default with ONE transform per (128-thread)
            // workgroup: the two transforms of a 256-thread workgroup share every s_barrier although they exchange nothing
            case 30: return launch_fused_variant<R, N, 16, 1, 1, 4, CPLX, 2, 1, false, true, false>(a, s);  // 28 with one transform per workgroup
            case 31: return launch_fused_variant<R, N, 16, 1, 3, 4, CPLX, 4, 1, false, false, false>(a, s); // 23 (<= 128 VGPRs, spectrum from L2) with one transform per workgroup
            case 32: return launch_fused_variant<R, N, 16, 1, 3, 4, CPLX, 3, 1, false, false, false>(a, s); // 19 with one transform per workgroup
            case 33: return launch_fused_variant<R, N, 16, 1, 1, 4, CPLX, 2, 1, true, true, false>(a, s);   // 12 (prefetch) with one transform per workgroup
            case 34: return launch_fused_variant<R, N, 16, 1, 1, 4, CPLX, 2, 1, false, true, false, true>(a, s);   // 30 + outputs staged through LDS, 16-byte stores
            case 35: return launch_fused_variant<R, N, 16, 1, 3, 4, CPLX, 3, 1, false, true, false, true>(a, s);   // 29 + staged stores
            case 36: return launch_fused_variant<R, N, 16, 1, 1, 4, CPLX, 2, 1, false, true, false, false, true>(a, s);   // 30 + the next unit's span staged in LDS by DMA
            case 37: return launch_fused_variant<R, N, 16, 1, 3, 4, CPLX, 3, 1, false, true, false, false, true>(a, s);   // 29 (hybrid twiddles, <= 168 VGPRs) + DMA staging
#endif
            // DEFAULT (= 30): register twiddles, filter spectrum in registers, no software prefetch, ONE transform per 128-thread workgroup
            // (191 VGPRs, four workgroups of two waves per CU).  12-round interleaved A/B on two boxes (profiles/r02l_ols_decoupled.json,
            // r02e_ols_ab.json): 1.85 / 1.78 ms vs 1.93 / 1.86 (29: hybrid twiddles, 3 waves per SIMD) vs 2.05 / 1.97 (26: the same kernel
            // with two transforms per workgroup) vs 2.17 (12: the round-1 default, prefetch).  Variant 2 is the best E = 8 form; the
            // lane-permuted schedule (10) has fewer LDS conflicts still, but its permuted global accesses cost more than the LDS cycles it
            // saves (3.8 vs 4.2 TB/s, profiles/r01e_tune_lanes.json).
            default: return launch_fused_variant<R, N, 16, 1, 1, 4, CPLX, 2, 1, false, true, false>(a, s);
        }
    }
    // Software prefetch of the next unit's samples: OFF by default.  Measured on MI355X (profiles/r02c_tune.json, 2^30 Float32, nfft 2048):
    // the same geometry without the prefetch is 14 % faster (1.85 vs 2.16 ms) -- the 32 registers it frees matter less than the issue
    // pattern: loads at the top of the iteration park the wave while its partner workgroup on the SIMD computes, which puts the two
    // resident workgroups in antiphase on their own.  MDSP_OLS_PREFETCH=1 restores the prefetching form (tools/bench_matrix.py sweeps both).
    if (tunables().ols_prefetch == 1) return launch_fused_variant<R, N, E, G, TWREG, 4, CPLX, 2, NBUF, true>(a, s);
    return launch_fused_variant<R, N, E, G, TWREG, 4, CPLX, 2, NBUF, false>(a, s);
}

template <typename R, bool CPLX> int launch_fused(int64_t nfft, const OlsFusedArgs& a, int variant, hipStream_t s) {
    switch (nfft) {
        case 256: return launch_fused_n<R, 256, CPLX>(a, variant, s);
        case 512: return launch_fused_n<R, 512, CPLX>(a, variant, s);
        case 1024: return launch_fused_n<R, 1024, CPLX>(a, variant, s);
        case 2048: return launch_fused_n<R, 2048, CPLX>(a, variant, s);
        case 4096: return launch_fused_n<R, 4096, CPLX>(a, variant, s);
        case 8192:
            if constexpr (sizeof(R) == 4) return launch_fused_n<R, 8192, CPLX>(a, variant, s);
        default: break;
    }
    MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "fused overlap-save does not support nfft=%lld", (long long)nfft);
}

// ---- partitioned launch ------------------------------------------------------------------------------------
// blocks [g0, g1) of every column (whole columns: g0 = 0, g1 = ceil(nout / B)); x_lo: first sample that may be dereferenced
struct UpolsRange {
    int64_t g0, g1, x_lo;
};
template <typename R, int N, int E, int P, bool CPLX, int MINW, bool XDMA, int NLDS = 0, int FORM = 1, bool H0REG = false>
int launch_upols_k(const mdsp_ols_plan_s* pl, const void* x, int64_t nx, int64_t ncols, int64_t ldx, void* y, int64_t nout, int64_t ldy, UpolsRange rg, hipStream_t s) {
    constexpr bool DBL = sizeof(R) == 8;
    constexpr int TW = DBL ? 1 : fft::TW_HYB;       // Float32: pass-1 twiddles from LDS (30 VGPRs less next to the delay line)
    void (*kern)(UpolsArgs) = nullptr;
    if constexpr (FORM == 2) kern = upols2_fused_kernel<R, N, E, P, TW, 4, CPLX, MINW, NLDS, H0REG>;
    else kern = upols_fused_kernel<R, N, E, P, TW, 4, CPLX, MINW, XDMA, NLDS>;
    constexpr int threads = N / E, RUNS = CPLX ? 1 : 2;
    UpolsArgs a;
    a.x = x; a.y = y; a.table = pl->table.p; a.Hp = pl->H.p;
    a.nx = nx; a.nout = nout; a.ldx = ldx; a.ldy = ldy;
    a.nblocks = rg.g1;
    a.g_begin = rg.g0;
    a.x_lo = rg.x_lo;
    a.ablate = MDSP_DBG(ablate);
    a.memprio = tunables().ols_prio;
    const int64_t nblk = rg.g1 - rg.g0;
    // resident workgroups per CU from the kernel's own resources (hipOccupancyMaxActiveBlocksPerMultiprocessor has answered half of what the
    // hardware admits for LDS-heavy kernels)
    // Queried once per instantiation and device (hipFuncGetAttributes on every launch was host latency on every chunk of the host
    // pipeline).  The register file (512 per SIMD lane, 8-register granule) and the LDS (160 KiB per CU) are gfx950's, as constants: the runtime's
    // device properties describe a workgroup's limits (65536 registers, 64 KiB), not the CU, and sizing the grid from them halved the occupancy of
    // the 256-VGPR partitioned kernels (5120 taps 1.06 -> 1.32 ms until found, round 4).
    static std::atomic<int> per_cu_cache[64];
    int dev = 0;
    MDSP_HIP(hipGetDevice(&dev));
    int per_cu = per_cu_cache[dev & 63].load(std::memory_order_acquire);
    if (per_cu == 0) {
        hipFuncAttributes fa;
        MDSP_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern)));
        const int waves = threads / 64, vg = std::max(1, fa.numRegs);
        const int by_regs = std::max(1, (512 / ((vg + 7) & ~7)) * 4 / waves);
        const int by_lds = (int)std::max<size_t>(1, (size_t)(160 * 1024) / std::max<size_t>(1, fa.sharedSizeBytes));
        per_cu = std::max(1, std::min({by_regs, by_lds, 2048 / threads}));
        per_cu_cache[dev & 63].store(per_cu, std::memory_order_release);
    }
    if (tunables().wg_per_cu > 0) per_cu = tunables().wg_per_cu;
    // persistent grid; runs of at least 16 (P - 1) blocks keep the warm-up blocks of a run below ~6 % of its work
    const int64_t resident = std::max<int64_t>(1, (int64_t)device_cu_count() * per_cu / std::max<int64_t>(1, ncols));
    const int64_t by_work = std::max<int64_t>(1, nblk / (RUNS * 16 * (P - 1)));
    const int64_t slots = std::min(resident, by_work);
    a.run_len = cdiv(nblk, slots * RUNS);
    const int64_t grid = cdiv(nblk, a.run_len * RUNS);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid, (unsigned)ncols), dim3(threads), 0, s, a);
    MDSP_LAUNCH_CHECK();
    return MDSP_OK;
}
// plan variants (MDSP_OLS_VARIANT when the plan is made; tools/bench_longfilt.py): 0 default; 1 the round-2 form (register loads, every spectrum
// from L2); 2 / 4 the round-2 form with the LDS ring of half windows; 3 the round-2 form with half spectra in LDS; 5 .. 8 the round-3 form
// (upols2_fused_kernel): 5 half spectra in LDS + partition 0 prefetched (the default up to three partitions), 6 everything from L2, 7 partition 0
// in registers and nothing in LDS, 8 half spectra in LDS + partition 0 in registers.  Measured: profiles/r03g_longfilt*.json
#define UPOLS_ARGS pl, x, nx, ncols, ldx, y, nout, ldy, rg, s
template <typename R, int N, int E, int P, bool CPLX>
int launch_upols_np(const mdsp_ols_plan_s* pl, const void* x, int64_t nx, int64_t ncols, int64_t ldx, void* y, int64_t nout, int64_t ldy, UpolsRange rg, hipStream_t s) {
    constexpr bool F32 = sizeof(R) == 4;
    const int v = pl->variant;
    if constexpr (F32 && N == 4096) {   // 36 KiB of exchange buffer per workgroup, two workgroups per CU: room for two half spectra (2 x 16 KiB) each
        constexpr int NL = P - 1 < 2 ? P - 1 : 2;
        if (v == 1) return launch_upols_k<R, N, E, P, CPLX, 2, false>(UPOLS_ARGS);
        if (v == 3) return launch_upols_k<R, N, E, P, CPLX, 2, false, NL>(UPOLS_ARGS);
        if (v == 5) return launch_upols_k<R, N, E, P, CPLX, 2, false, NL, 2, false>(UPOLS_ARGS);
        if (v == 6) return launch_upols_k<R, N, E, P, CPLX, 2, false, 0, 2, false>(UPOLS_ARGS);
        if (v == 7) return launch_upols_k<R, N, E, P, CPLX, 2, false, 0, 2, true>(UPOLS_ARGS);
        if (v == 8) return launch_upols_k<R, N, E, P, CPLX, 2, false, NL, 2, true>(UPOLS_ARGS);
        if constexpr (!CPLX) {
            if (v == 2 || (v == 0 && P == 4)) return launch_upols_k<R, N, E, P, CPLX, 2, true>(UPOLS_ARGS);   // four partitions: the round-3 form spills
        } else {
            if (v == 0 && P == 4) return launch_upols_k<R, N, E, P, CPLX, 2, false, NL>(UPOLS_ARGS);
        }
        return launch_upols_k<R, N, E, P, CPLX, 2, false, NL, 2, false>(UPOLS_ARGS);
    } else {   // 8192 points (one workgroup's exchange buffer is 70 KiB), Float64: the round-3 form spills there and measured slower
        if constexpr (F32 && !CPLX) {
            if (v == 4) return launch_upols_k<R, N, E, P, CPLX, 2, true>(UPOLS_ARGS);   // the ring (64 KiB) leaves one workgroup per CU
        }
        return launch_upols_k<R, N, E, P, CPLX, 2, false>(UPOLS_ARGS);
    }
}
#undef UPOLS_ARGS
template <typename R, int N, int E, bool CPLX>
int launch_upols_n(const mdsp_ols_plan_s* pl, const void* x, int64_t nx, int64_t ncols, int64_t ldx, void* y, int64_t nout, int64_t ldy, UpolsRange rg, hipStream_t s) {
    switch (pl->partitions) {
        case 2: return launch_upols_np<R, N, E, 2, CPLX>(pl, x, nx, ncols, ldx, y, nout, ldy, rg, s);
        case 3: return launch_upols_np<R, N, E, 3, CPLX>(pl, x, nx, ncols, ldx, y, nout, ldy, rg, s);
        case 4: return launch_upols_np<R, N, E, 4, CPLX>(pl, x, nx, ncols, ldx, y, nout, ldy, rg, s);
        default: break;
    }
    MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "partitioned overlap-save supports 2..4 partitions, got %d", pl->partitions);
}
template <typename R, bool CPLX>
int launch_upols(const mdsp_ols_plan_s* pl, const void* x, int64_t nx, int64_t ncols, int64_t ldx, void* y, int64_t nout, int64_t ldy, UpolsRange rg, hipStream_t s) {
    constexpr bool DBL = sizeof(R) == 8;
    if (pl->nfft == 4096) return launch_upols_n<R, 4096, DBL ? 8 : 16, CPLX>(pl, x, nx, ncols, ldx, y, nout, ldy, rg, s);
    if constexpr (!DBL) {
        if (pl->nfft == 8192) return launch_upols_n<R, 8192, 16, CPLX>(pl, x, nx, ncols, ldx, y, nout, ldy, rg, s);
    }
    MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "partitioned overlap-save does not support nfft=%lld", (long long)pl->nfft);
}

// ---- rocFFT engine --------------------------------------------------------------------------------------
template <typename R, bool CPLX>
int exec_rocfft(mdsp_ols_plan_s* pl, const void* x, int64_t nx, int64_t ncols, int64_t ldx, void* y, int64_t nout, int64_t ldy,
                hipStream_t s, int64_t u_begin = 0, int64_t u_end = -1) {
    using T = std::conditional_t<CPLX, cx<R>, R>;
    const int64_t nfft = pl->nfft, L = pl->L;
    const int64_t nblocks = cdiv(nout, L);
    const int64_t nunits = u_end >= 0 ? std::min(u_end, nblocks * ncols) : nblocks * ncols;   // unit = block (no pair packing in this engine)
    const int64_t nspec = CPLX ? nfft : nfft / 2 + 1;
    // chunk so that td + fd stay cache resident (~64 MiB)
    const int64_t per_unit = (int64_t)sizeof(T) * nfft + (CPLX ? 0 : (int64_t)sizeof(cx<R>) * nspec);
    const int64_t batch = std::max<int64_t>(1, std::min<int64_t>(nblocks * ncols, (int64_t(64) << 20) / per_unit));
    if (pl->batch != batch) {
        MDSP_TRY(pl->td.reserve((size_t)(sizeof(T) * nfft * batch)));
        if (!CPLX) MDSP_TRY(pl->fd.reserve((size_t)(sizeof(cx<R>) * nspec * batch)));
        const bool dbl = sizeof(R) == 8;
        if (CPLX) {
            MDSP_TRY(pl->fwd.create(FftKind::C2C_FWD, dbl, nfft, batch, true));
            MDSP_TRY(pl->inv.create(FftKind::C2C_INV, dbl, nfft, batch, true));
        } else {
            MDSP_TRY(pl->fwd.create(FftKind::R2C, dbl, nfft, batch, false));
            MDSP_TRY(pl->inv.create(FftKind::C2R, dbl, nfft, batch, false));
        }
        pl->batch = batch;
    }
    const int gx = (int)std::min<int64_t>(cdiv(nfft, 256), 8);
    for (int64_t u0 = u_begin; u0 < nunits; u0 += batch) {
        const int64_t cnt = std::min<int64_t>(batch, nunits - u0);
        hipLaunchKernelGGL(ols_segment_kernel<T>, dim3((unsigned)cnt, gx), dim3(256), 0, s, (const T*)x, pl->td.as<T>(), nx, ldx,
                           nblocks, L, (int)pl->nb, (int)nfft, u0, nunits);
        MDSP_LAUNCH_CHECK();
        if (CPLX) {
            MDSP_TRY(pl->fwd.exec(pl->td.p, nullptr, s));
            hipLaunchKernelGGL(ols_cmul_kernel<R>, dim3((unsigned)cnt, gx), dim3(256), 0, s, pl->td.as<cx<R>>(), pl->H.as<cx<R>>(),
                               (int)nspec, cnt);
            MDSP_LAUNCH_CHECK();
            MDSP_TRY(pl->inv.exec(pl->td.p, nullptr, s));
        } else {
            MDSP_TRY(pl->fwd.exec(pl->td.p, pl->fd.p, s));
            hipLaunchKernelGGL(ols_cmul_kernel<R>, dim3((unsigned)cnt, gx), dim3(256), 0, s, pl->fd.as<cx<R>>(), pl->H.as<cx<R>>(),
                               (int)nspec, cnt);
            MDSP_LAUNCH_CHECK();
            MDSP_TRY(pl->inv.exec(pl->fd.p, pl->td.p, s));
        }
        hipLaunchKernelGGL(ols_save_kernel<T>, dim3((unsigned)cnt, gx), dim3(256), 0, s, pl->td.as<T>(), (T*)y, nout, ldy, nblocks, L,
                           (int)pl->nb, (int)nfft, u0, nunits);
        MDSP_LAUNCH_CHECK();
    }
    return MDSP_OK;
}

}  // namespace

namespace mdsp {
int ols_rows(int dbl, void* work, int64_t rows, int hrows, const void* Hrows, const void* table, const void* rt0, const void* rt1, int rlogS, hipStream_t st) {
    return ols_rows_impl(dbl, work, rows, hrows, Hrows, table, rt0, rt1, rlogS, st);
}
int ols_rows_table(int dbl, DevBuf& buf) { return dbl ? upload_table<double>(buf, 4096) : upload_table<float>(buf, 8192); }
}  // namespace mdsp

extern "C" {

// What a plan for (nb, nfft, nx_hint, dtype, mode, engine) executes: the engine, the transform that runs, the partitions, whether the blocks go through the
// multi-pass engine (and in how many rows).  Pure host arithmetic, shared by mdsp_ols_plan_create and mdsp_ols_geometry_for.
struct OlsChoice {
    int eng = MDSP_ENGINE_ROCFFT, parts = 1, rows = 0;
    int64_t nfft = 0, exec_nfft = 0;
    bool big = false;
};
static int ols_choose(int64_t nb, int64_t nfft, int64_t nx_hint, int dtype, int mode, int engine, OlsChoice* c) {
    if (nb < 1) MDSP_FAIL(MDSP_ERR_ARGUMENT, "filter vector b must be non-empty");
    if (!dtype_valid(dtype)) MDSP_FAIL(MDSP_ERR_ARGUMENT, "invalid dtype %d", dtype);
    if (mode != MDSP_OLS_FILT && mode != MDSP_OLS_CONV) MDSP_FAIL(MDSP_ERR_ARGUMENT, "invalid mode %d", mode);
    if (nfft == 0) nfft = mdsp_optimal_fft_len(nb, std::max<int64_t>(nx_hint, 1));
    if (nfft < nb) MDSP_FAIL(MDSP_ERR_ARGUMENT, "nfft (%lld) must be >= length(b) (%lld)", (long long)nfft, (long long)nb);
    if (nfft > (int64_t(1) << 24)) MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "nfft %lld too large", (long long)nfft);
    int eng = engine;
    if (eng == MDSP_ENGINE_AUTO) eng = tunables().engine;
    int64_t exec_nfft = nfft;
    int parts = 1;
    const bool fused_ok = fused_geometry(dtype, nb, nfft, &exec_nfft, &parts);
    // filters beyond four partitions of the largest in-LDS transform: the same convolution in blocks on the multi-pass engine (bigfft.hip ols_size)
    const int64_t lds_max = dtype_is_double(dtype) ? 4096 : 8192;
    const int64_t big_n = (!fused_ok && eng != MDSP_ENGINE_ROCFFT && nb > 2 * lds_max) ? mdsp::big::ols_size(dtype, nb, nx_hint > 1 ? nx_hint + (mode == MDSP_OLS_CONV ? nb - 1 : 0) : 0) : 0;
    if (eng == MDSP_ENGINE_AUTO) eng = (fused_ok || big_n) ? MDSP_ENGINE_FUSED : MDSP_ENGINE_ROCFFT;
    if (eng == MDSP_ENGINE_FUSED && !fused_ok && big_n) {
        exec_nfft = big_n;
        parts = 1;
    } else if (eng == MDSP_ENGINE_FUSED && !fused_ok)
        MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "fused engine supports power-of-two nfft >= 256; got nfft=%lld, %lld taps", (long long)nfft, (long long)nb);
    if (eng != MDSP_ENGINE_FUSED && eng != MDSP_ENGINE_ROCFFT) MDSP_FAIL(MDSP_ERR_ARGUMENT, "invalid engine %d", engine);
    if (eng != MDSP_ENGINE_FUSED) {
        exec_nfft = nfft;
        parts = 1;
    }
    c->eng = eng;
    c->parts = parts;
    c->nfft = nfft;
    c->exec_nfft = exec_nfft;
    c->big = eng == MDSP_ENGINE_FUSED && !fused_ok;
    c->rows = c->big ? mdsp::big::ols_rows_r0(dtype, exec_nfft) : 0;
    return MDSP_OK;
}

int mdsp_ols_geometry_for(int64_t nb, int64_t nfft, int64_t nx_hint, int dtype, int mode, int engine, int64_t* exec_nfft, int64_t* exec_block_len, int* partitions,
                          int* engine_used, int* rows) {
    OlsChoice c;
    MDSP_TRY(ols_choose(nb, nfft, nx_hint, dtype, mode, engine, &c));
    if (exec_nfft) *exec_nfft = c.exec_nfft;
    if (exec_block_len) *exec_block_len = c.parts > 1 ? c.exec_nfft / 2 : c.exec_nfft - (nb - 1);
    if (partitions) *partitions = c.parts;
    if (engine_used) *engine_used = c.eng;
    if (rows) *rows = c.rows;
    return MDSP_OK;
}

int mdsp_ols_plan_create(mdsp_ols_plan* plan, const void* taps_host, int64_t nb, int64_t nfft, int64_t nx_hint, int dtype, int mode,
                         int engine) {
    if (!plan) MDSP_FAIL(MDSP_ERR_ARGUMENT, "plan is NULL");
    *plan = nullptr;
    if (!taps_host || nb < 1) MDSP_FAIL(MDSP_ERR_ARGUMENT, "filter vector b must be non-empty");
    OlsChoice ch;
    MDSP_TRY(ols_choose(nb, nfft, nx_hint, dtype, mode, engine, &ch));
    nfft = ch.nfft;
    const int eng = ch.eng, parts = ch.parts;
    const int64_t exec_nfft = ch.exec_nfft;

    auto pl = new mdsp_ols_plan_s();
    pl->dtype = dtype;
    pl->mode = mode;
    pl->engine = eng;
    pl->nb = nb;
    pl->ref_nfft = nfft;
    pl->ref_L = nfft - (nb - 1);
    pl->partitions = parts;
    pl->nfft = exec_nfft;
    pl->L = parts > 1 ? exec_nfft / 2 : exec_nfft - (nb - 1);
    pl->variant = tunables().ols_variant;
    pl->big = ch.big;
    nfft = exec_nfft;   // from here on: the transform size that executes

    // Filter spectrum in double on the host.  FILT: taps scaled by 1/nfft before the transform (filt.jl:499);
    // CONV: spectrum scaled by 1/nfft afterwards (dspbase.jl:516).  The scaling is applied in the plan's working
    // precision at the same place the reference applies it, the transform itself is evaluated in double.
    const bool cplx = dtype_is_complex(dtype), dbl = dtype_is_double(dtype);
    if (parts > 1) {   // P spectra of the zero-padded partitions h_p = h[pB .. (p+1)B), 1/N folded in (both modes: one scaling per partition)
        const int64_t B = nfft / 2;
        std::vector<zd> Hall((size_t)(parts * nfft));
        for (int p = 0; p < parts; ++p) {
            std::vector<zd> hpart((size_t)nfft, zd(0, 0));
            for (int64_t i = 0; i < B && p * B + i < nb; ++i) {
                const int64_t j = p * B + i;
                zd v;
                if (dtype == MDSP_F32) v = zd(((const float*)taps_host)[j], 0);
                else if (dtype == MDSP_F64) v = zd(((const double*)taps_host)[j], 0);
                else if (dtype == MDSP_C32) v = zd(((const float*)taps_host)[2 * j], ((const float*)taps_host)[2 * j + 1]);
                else v = zd(((const double*)taps_host)[2 * j], ((const double*)taps_host)[2 * j + 1]);
                hpart[(size_t)i] = v / (double)nfft;
            }
            const std::vector<zd> Hq = host_fft(hpart, -1);
            std::copy(Hq.begin(), Hq.end(), Hall.begin() + (size_t)p * (size_t)nfft);
        }
        int stp = MDSP_OK;
        if (dbl) {
            std::vector<cx<double>> h(Hall.size());
            for (size_t k = 0; k < Hall.size(); ++k) h[k] = {Hall[k].real(), Hall[k].imag()};
            stp = pl->H.reserve(sizeof(cx<double>) * h.size());
            if (stp == MDSP_OK && hipMemcpy(pl->H.p, h.data(), sizeof(cx<double>) * h.size(), hipMemcpyHostToDevice) != hipSuccess) stp = set_error(MDSP_ERR_DEVICE, "spectrum upload failed");
        } else {
            std::vector<cx<float>> h(Hall.size());
            for (size_t k = 0; k < Hall.size(); ++k) h[k] = {(float)Hall[k].real(), (float)Hall[k].imag()};
            stp = pl->H.reserve(sizeof(cx<float>) * h.size());
            if (stp == MDSP_OK && hipMemcpy(pl->H.p, h.data(), sizeof(cx<float>) * h.size(), hipMemcpyHostToDevice) != hipSuccess) stp = set_error(MDSP_ERR_DEVICE, "spectrum upload failed");
        }
        if (stp == MDSP_OK) stp = dbl ? upload_table<double>(pl->table, nfft) : upload_table<float>(pl->table, nfft);
        if (stp != MDSP_OK) {
            delete pl;
            return stp;
        }
        *plan = pl;
        return MDSP_OK;
    }
    std::vector<zd> hp((size_t)nfft, zd(0, 0));
    for (int64_t i = 0; i < nb; ++i) {
        zd v;
        if (dtype == MDSP_F32) v = zd(((const float*)taps_host)[i], 0);
        else if (dtype == MDSP_F64) v = zd(((const double*)taps_host)[i], 0);
        else if (dtype == MDSP_C32) v = zd(((const float*)taps_host)[2 * i], ((const float*)taps_host)[2 * i + 1]);
        else v = zd(((const double*)taps_host)[2 * i], ((const double*)taps_host)[2 * i + 1]);
        if (mode == MDSP_OLS_FILT) {
            if (dbl) v = v / (double)nfft;
            else v = zd((double)((float)v.real() / (float)nfft), (double)((float)v.imag() / (float)nfft));
        }
        hp[(size_t)i] = v;
    }
    std::vector<zd> Hf = host_fft(hp, -1);
    if (mode == MDSP_OLS_CONV) {
        const double sc = 1.0 / (double)nfft;
        for (auto& h : Hf) h *= sc;
    }
    pl->big_rows = ch.rows;
    if (pl->big_rows) {   // the rows form multiplies row k1 of the two-pass transform by H[k1 + R0 k2], k2 along the row
        const int64_t R0 = pl->big_rows, S = nfft / R0;
        std::vector<zd> Hr(Hf.size());
        for (int64_t k1 = 0; k1 < R0; ++k1)
            for (int64_t k2 = 0; k2 < S; ++k2) Hr[(size_t)(k1 * S + k2)] = Hf[(size_t)(k1 + R0 * k2)];
        Hf.swap(Hr);
    }
    int st = MDSP_OK;
    const bool half = (eng == MDSP_ENGINE_ROCFFT) && !cplx;
    st = dbl ? upload_spectrum<double>(pl, Hf, half) : upload_spectrum<float>(pl, Hf, half);
    if (st == MDSP_OK && eng == MDSP_ENGINE_FUSED && !pl->big) st = dbl ? upload_table<double>(pl->table, nfft) : upload_table<float>(pl->table, nfft);
    if (st != MDSP_OK) {
        delete pl;
        return st;
    }
    *plan = pl;
    return MDSP_OK;
}

int mdsp_ols_plan_destroy(mdsp_ols_plan plan) {
    delete plan;
    return MDSP_OK;
}

int mdsp_ols_plan_info(mdsp_ols_plan plan, int64_t* nfft, int64_t* block_len, int* engine_used) {
    if (!plan) MDSP_FAIL(MDSP_ERR_ARGUMENT, "plan is NULL");
    if (nfft) *nfft = plan->ref_nfft;          // the reference's geometry: what optimalfftfiltlength / the caller chose
    if (block_len) *block_len = plan->ref_L;
    if (engine_used) *engine_used = plan->engine;
    return MDSP_OK;
}

int mdsp_ols_plan_geometry(mdsp_ols_plan plan, int64_t* exec_nfft, int64_t* exec_block_len, int* partitions) {
    if (!plan) MDSP_FAIL(MDSP_ERR_ARGUMENT, "plan is NULL");
    if (exec_nfft) *exec_nfft = plan->nfft;
    if (exec_block_len) *exec_block_len = plan->L;
    if (partitions) *partitions = plan->partitions;
    return MDSP_OK;
}

// (Round 4's hand-allocated kernel of the headline shape -- one wavefront per four blocks, csrc/ols_w64_asm.s -- measured 5 - 7 % SLOWER than ols_fused_kernel on
// two boxes (profiles/r04_ols_asm_ablation.json) and was removed in round 5 together with its generator; HISTORY.md section 4.2 has the record.)

// blocks [g_begin, g_end) of every column's block grid (g_end < 0: all).  x / y may be "virtual" bases: only the elements those blocks
// touch are dereferenced (mdsp_ols_exec_range).
static int ols_exec_core(mdsp_ols_plan plan, const void* x_dev, int64_t nx, int64_t ncols, int64_t ldx, void* y_dev, int64_t nout, int64_t ldy,
                         int64_t g_begin, int64_t g_end, hipStream_t s, int64_t x_lo = 0) {
    const bool cplx = dtype_is_complex(plan->dtype), dbl = dtype_is_double(plan->dtype);
    const int64_t nblocks = cdiv(nout, plan->L);
    if (plan->partitions > 1) {   // long filters: uniformly partitioned overlap-save
        if (ncols > 65535) MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "more than 65535 columns per call");
        if (g_end < 0 || g_end > nblocks) g_end = nblocks;
        if (g_begin >= g_end) return MDSP_OK;
        const UpolsRange rg{g_begin, g_end, x_lo};
        if (cplx) return dbl ? launch_upols<double, true>(plan, x_dev, nx, ncols, ldx, y_dev, nout, ldy, rg, s) : launch_upols<float, true>(plan, x_dev, nx, ncols, ldx, y_dev, nout, ldy, rg, s);
        return dbl ? launch_upols<double, false>(plan, x_dev, nx, ncols, ldx, y_dev, nout, ldy, rg, s) : launch_upols<float, false>(plan, x_dev, nx, ncols, ldx, y_dev, nout, ldy, rg, s);
    }
    if (g_end < 0 || g_end > nblocks) g_end = nblocks;
    if (g_begin >= g_end) return MDSP_OK;
    if (plan->big) {   // one column at a time: a column's blocks fill the chip by themselves
        const size_t esz = dtype_size(plan->dtype);
        for (int64_t c = 0; c < ncols; ++c)
            MDSP_TRY(mdsp::big::ols(plan->bigeng, plan->dtype, plan->nfft, plan->big_rows, static_cast<const char*>(x_dev) + (size_t)(c * ldx) * esz, nx, plan->H.p, plan->nb,
                                    static_cast<char*>(y_dev) + (size_t)(c * ldy) * esz, nout, g_begin, g_end, s));
        return MDSP_OK;
    }
    if (plan->engine == MDSP_ENGINE_ROCFFT) {
        const int64_t ub = g_begin, ue = (g_begin == 0 && g_end == nblocks) ? -1 : g_end;
        if (cplx) return dbl ? exec_rocfft<double, true>(plan, x_dev, nx, ncols, ldx, y_dev, nout, ldy, s, ub, ue)
                             : exec_rocfft<float, true>(plan, x_dev, nx, ncols, ldx, y_dev, nout, ldy, s, ub, ue);
        return dbl ? exec_rocfft<double, false>(plan, x_dev, nx, ncols, ldx, y_dev, nout, ldy, s, ub, ue)
                   : exec_rocfft<float, false>(plan, x_dev, nx, ncols, ldx, y_dev, nout, ldy, s, ub, ue);
    }
    OlsFusedArgs a;
    a.x = x_dev;
    a.y = y_dev;
    a.table = plan->table.p;
    a.H = plan->H.p;
    a.nx = nx;
    a.nout = nout;
    a.ldx = ldx;
    a.ldy = ldy;
    a.L = plan->L;
    a.nb = (int)plan->nb;
    a.nblocks = g_end;                     // blocks past the range do not exist for this launch (second block of the last pair)
    a.units_per_col = cplx ? nblocks : cdiv(nblocks, 2);
    a.nunits = (g_end == nblocks) ? a.units_per_col * ncols : (cplx ? g_end : cdiv(g_end, 2));   // ranges: single column (checked by the caller)
    a.u_begin = cplx ? g_begin : g_begin / 2;
    a.run_len = 1;
    a.niter = 0;
    a.ablate = MDSP_DBG(ablate);
    a.memprio = tunables().ols_prio;
    if (cplx) return dbl ? launch_fused<double, true>(plan->nfft, a, plan->variant, s) : launch_fused<float, true>(plan->nfft, a, plan->variant, s);
    return dbl ? launch_fused<double, false>(plan->nfft, a, plan->variant, s) : launch_fused<float, false>(plan->nfft, a, plan->variant, s);
}

int mdsp_ols_exec(mdsp_ols_plan plan, const void* x_dev, int64_t nx, int64_t ncols, int64_t ldx, void* y_dev, int64_t nout, int64_t ldy,
                  void* stream) {
    if (!plan) MDSP_FAIL(MDSP_ERR_ARGUMENT, "plan is NULL");
    if (nx < 0 || ncols < 0 || nout < 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "negative size");
    if (nout > nx + plan->nb - 1) MDSP_FAIL(MDSP_ERR_ARGUMENT, "nout (%lld) exceeds nx+nb-1", (long long)nout);
    if (ncols > 1 && (ldx < nx || ldy < nout)) MDSP_FAIL(MDSP_ERR_ARGUMENT, "leading dimension smaller than the column length");
    if (nout == 0 || ncols == 0) return MDSP_OK;
    if (!x_dev && nx > 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "x is NULL");
    if (!y_dev) MDSP_FAIL(MDSP_ERR_ARGUMENT, "out is NULL");
    if (x_dev == y_dev) MDSP_FAIL(MDSP_ERR_ARGUMENT, "out may not alias x");
    return ols_exec_core(plan, x_dev, nx, ncols, ldx, y_dev, nout, ldy, 0, -1, as_stream(stream));
}

int mdsp_ols_exec_range(mdsp_ols_plan plan, const void* xs_dev, int64_t xs_first, int64_t xs_len, int64_t nx, void* ys_dev, int64_t first_block,
                        int64_t nblocks_range, int64_t nout, void* stream) {
    if (!plan) MDSP_FAIL(MDSP_ERR_ARGUMENT, "plan is NULL");
    if (nx < 0 || nout < 0 || xs_first < 0 || xs_len < 0 || first_block < 0 || nblocks_range < 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "negative size");
    if (nout > nx + plan->nb - 1) MDSP_FAIL(MDSP_ERR_ARGUMENT, "nout (%lld) exceeds nx+nb-1", (long long)nout);
    const int64_t L = plan->L, nb = plan->nb, nblocks = cdiv(nout, L);
    const int64_t g0 = first_block, g1 = std::min(nblocks, first_block + nblocks_range);
    if (g0 >= g1) return MDSP_OK;
    if (plan->partitions == 1 && !dtype_is_complex(plan->dtype) && (g0 & 1)) MDSP_FAIL(MDSP_ERR_ARGUMENT, "first_block must be even (two real blocks share a transform)");
    // samples the blocks read: [g0 L - (nb-1), g1 L) clipped to the signal
    const int64_t need_lo = std::max<int64_t>(0, g0 * L - (nb - 1)), need_hi = std::min(nx, g1 * L);
    if (need_hi > need_lo && (xs_first > need_lo || xs_first + xs_len < need_hi))
        MDSP_FAIL(MDSP_ERR_ARGUMENT, "the slice [%lld, %lld) does not cover the samples [%lld, %lld) that blocks [%lld, %lld) read", (long long)xs_first,
                  (long long)(xs_first + xs_len), (long long)need_lo, (long long)need_hi, (long long)g0, (long long)g1);
    if (!xs_dev && need_hi > need_lo) MDSP_FAIL(MDSP_ERR_ARGUMENT, "x is NULL");
    if (!ys_dev) MDSP_FAIL(MDSP_ERR_ARGUMENT, "out is NULL");
    const size_t esz = dtype_size(plan->dtype);
    // virtual bases of the whole column: x[0] and y[0]; only [need_lo, need_hi) and [g0 L, min(nout, g1 L)) are dereferenced
    const char* xv = static_cast<const char*>(xs_dev) - (ptrdiff_t)xs_first * (ptrdiff_t)esz;
    char* yv = static_cast<char*>(ys_dev) - (ptrdiff_t)(g0 * L) * (ptrdiff_t)esz;
    const int64_t nx_eff = std::min(nx, xs_first + xs_len);      // nothing past the slice is read (hardware zero fill past nx_eff is never reached)
    const int64_t nout_eff = std::min(nout, g1 * L);
    // partitioned plans warm their delay line up with the P blocks in front of g0: what the slice does not hold of them lies more than nb - 1
    // samples back, under the zero taps of the last partition, and is read as zero (x_lo)
    return ols_exec_core(plan, xv, nx_eff, 1, nx_eff, yv, nout_eff, nout_eff, g0, g1, as_stream(stream), xs_first);
}

int mdsp_ols_segment(mdsp_ols_plan plan, const void* x_dev, int64_t nx, int64_t first_block, int64_t nblocks, void* seg_dev, void* stream) {
    if (!plan) MDSP_FAIL(MDSP_ERR_ARGUMENT, "plan is NULL");
    if (first_block < 0 || nblocks < 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "negative block range");
    if (nblocks == 0) return MDSP_OK;
    hipStream_t s = as_stream(stream);
    const int64_t nfft = plan->ref_nfft;          // the REFERENCE's blocks (tmp1 of Filters/filt.jl:504-510), whatever geometry executes
    const int gx = (int)std::min<int64_t>(cdiv(nfft, 256), 8);
    const int64_t total = first_block + nblocks;  // blocks of a single column: unit index == block index
    const int64_t big = INT64_MAX / 4;
    for (int64_t b0 = 0; b0 < nblocks; b0 += 32768) {
        const int64_t cnt = std::min<int64_t>(32768, nblocks - b0);
#define SEG(TT)                                                                                                                      \
    hipLaunchKernelGGL(ols_segment_kernel<TT>, dim3((unsigned)cnt, gx), dim3(256), 0, s, (const TT*)x_dev, (TT*)seg_dev + b0 * nfft, nx, \
                       (int64_t)0, big, plan->ref_L, (int)plan->nb, (int)nfft, first_block + b0, total)
        switch (plan->dtype) {
            case MDSP_F32: SEG(float); break;
            case MDSP_F64: SEG(double); break;
            case MDSP_C32: SEG(cx<float>); break;
            default: SEG(cx<double>); break;
        }
#undef SEG
        MDSP_LAUNCH_CHECK();
    }
    return MDSP_OK;
}

}  // extern "C"
