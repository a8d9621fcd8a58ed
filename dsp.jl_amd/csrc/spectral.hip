// Framing, Welch PSD, STFT / spectrogram / periodogram (mdsp_frames, mdsp_welch_*, mdsp_stft_*).
//
// Reference loops being replaced (src/periodograms.jl):
//   K4  ArraySplit getindex :57-69     buf[i] = s[offset+i] * window[i], zero tail up to nfft
//   F3  mul!(outbuf, plan, sig) :754, :888        rfft (real) / fft (complex), out of place
//   K5  fft2pow! :142-172              out[i] = muladd(abs2(X[i]), m, out[i]),  m = 1/r or 2/r
//   K6  fft2oneortwosided! :234-244    raw STFT column, conjugate mirror for real -> two-sided
//
// Frame k (0-based) of a channel covers s[k*hop .. k*hop+n), hop = n - noverlap; K = (len-n) div hop + 1 frames
// (the trailing partial frame is dropped, :49-50).  The window is ALWAYS Float64 in the reference (:250): the
// product is formed in double and rounded once to the buffer eltype -- done the same way here.
//
// Engines:
//   FUSED  (power-of-two nfft in [256, 8192]):
//     Welch : persistent kernel; a transform slot packs TWO real frames into one complex FFT (z = w*(a + i b)),
//             and accumulates |Z[k]|^2 per bin in double in registers.  Since |A[k]|^2 + |B[k]|^2 =
//             (|Z[k]|^2 + |Z[N-k]|^2)/2, no untangling is needed: the fold happens once in the tiny finalize
//             kernel.  HBM traffic = the signal, read once (+ overlap re-reads served by L2).
//     STFT  : window -> FFT -> (|X|^2 * m | X) -> coalesced column store, one kernel.
//   ROCFFT : K4 kernel -> batched rocFFT -> K5/K6 kernel over cache-sized chunks; any nfft.
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include <memory>
#include <mutex>
#include <vector>

#include "common.h"
#include "devio.h"
#include "fft_lds.h"
#include "fft_wg.h"
#include "fft_w64.h"
#include "hostfft.h"
#include "rocfft_wrap.h"
#include "welch_plan.h"
#include "hostpipe.h"
#include "gx_kernels.h"   // (namespace mdsp::gx and the kernel template: ahead of the anonymous namespace that includes spectral_gx.h)

using namespace mdsp;
using mdsp::fft::cx;

namespace {

// Intermediates (windowed frames + spectra) of one rocFFT-engine chunk: small enough to stay in the 256 MiB Infinity Cache
// between the three kernels that touch them, large enough that each launch fills the GPU.  MDSP_ROCFFT_CHUNK_MIB overrides.
inline int64_t rocfft_chunk_bytes() {
    return (int64_t)tunables().rocfft_chunk_mib << 20;   // default 192; swept 32..1024 MiB on config 4: 192 is 14 % faster than 64 (tools/rocfft_chunk_sweep.sh)
}

template <typename T> struct real_of { using type = T; };
template <typename R> struct real_of<cx<R>> { using type = R; };

// Float64 window product, rounded once (periodograms.jl:66 `x.buf[i] = x.s[offset+i] * window[i]`)
__device__ __forceinline__ float win_mul(float s, double w) { return (float)((double)s * w); }
__device__ __forceinline__ double win_mul(double s, double w) { return s * w; }
__device__ __forceinline__ cx<float> win_mul(cx<float> s, double w) { return {(float)((double)s.x * w), (float)((double)s.y * w)}; }
__device__ __forceinline__ cx<double> win_mul(cx<double> s, double w) { return {s.x * w, s.y * w}; }

// ======================================================================================================
// rocFFT-engine kernels
// ======================================================================================================
// K4: frames [f0, f0+count) of channel `ch` (unit u = ch*K + f) -> fr[(u-u0)*nfft + i]
template <typename T>
__global__ __launch_bounds__(256) void frame_window_kernel(const T* __restrict__ s, T* __restrict__ fr, const double* __restrict__ win,
                                                           int64_t lds_, int64_t K, int64_t hop, int n, int nfft, int64_t u0, int64_t nunits) {
    const int64_t u = u0 + blockIdx.x;
    if (u >= nunits) return;
    const int64_t ch = u / K, f = u - ch * K;
    const T* src = s + ch * lds_ + f * hop;
    T* dst = fr + (int64_t)blockIdx.x * nfft;
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < nfft; i += gridDim.y * blockDim.x) {
        T v{};
        if (i < n) v = win ? win_mul(src[i], win[i]) : src[i];
        dst[i] = v;
    }
}

// K5 (Welch): partial[slice][ch][k] += sum over this chunk's frames of channel ch of |X[k]|^2   (double)
// grid: (bins/256, nslices, nch).  Frame u of the chunk belongs to slice (u % nslices); deterministic order.
template <typename R>
__global__ __launch_bounds__(256) void abs2_accum_kernel(const cx<R>* __restrict__ spec, double* __restrict__ partial, int nspec, int64_t K,
                                                         int64_t u0, int64_t cnt, int64_t nch) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nspec) return;
    const int64_t ch = blockIdx.z;
    // units of this chunk that belong to channel ch: [max(u0, ch*K), min(u0+cnt, (ch+1)*K))
    const int64_t lo = std::max<int64_t>(u0, ch * K), hi = std::min<int64_t>(u0 + cnt, (ch + 1) * K);
    double acc = 0;
    for (int64_t u = lo + blockIdx.y; u < hi; u += gridDim.y) {
        const cx<R> z = spec[(u - u0) * nspec + k];
        acc += (double)z.x * (double)z.x + (double)z.y * (double)z.y;
    }
    if (lo + (int64_t)blockIdx.y < hi) partial[((int64_t)blockIdx.y * nch + ch) * nspec + k] += acc;
}

// Slice reduction: reduced[ch][k] = sum_s partial[s][ch][k], fixed order (deterministic).  32 bins x 8 slice lanes per
// workgroup so the nslices-long sum is spread over threads instead of being one latency-bound loop per bin.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double* __restrict__ partial, double* __restrict__ reduced, int nslices,
                                                              int64_t nch, int nacc, int accumulate) {
    __shared__ double sm[8][33];
    const int bx = threadIdx.x & 31, sy = threadIdx.x >> 5;
    const int k = blockIdx.x * 32 + bx;
    const int64_t ch = blockIdx.y;
    double a = 0;
    if (k < nacc)
        for (int s = sy; s < nslices; s += 8) a += partial[((int64_t)s * nch + ch) * nacc + k];
    sm[sy][bx] = a;
    __syncthreads();
    if (sy == 0 && k < nacc) {
        double t = sm[0][bx];
#pragma unroll
        for (int i = 1; i < 8; ++i) t += sm[i][bx];
        reduced[ch * nacc + k] = accumulate ? reduced[ch * nacc + k] + t : t;   // accumulate: a further slice of the same stream
    }
}

// With hundreds of slices (a persistent grid writes one row per workgroup: 1024 rows of nfft doubles are 24-32 MiB) the single kernel above is
// ~100 workgroups walking 128 rows each with 256-byte reads -- 60-120 us behind a 0.3-1.4 ms kernel.  Two steps instead: groups of rows summed by
// (bins / 256) x groups workgroups with 2 KiB reads per wave, then the kernel above over the groups.  Fixed order either way.
constexpr int REDUCE_GROUPS = 64;
__global__ __launch_bounds__(256) void reduce_partials_groups_kernel(const double* __restrict__ partial, double* __restrict__ tmp, int nslices, int64_t nch, int nacc,
                                                                     int per_group) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int g = blockIdx.y;
    const int64_t ch = blockIdx.z;
    if (k >= nacc) return;
    const int s0 = g * per_group, s1 = min(nslices, s0 + per_group);
    double a = 0;
#pragma unroll 4
    for (int s = s0; s < s1; ++s) a += partial[((int64_t)s * nch + ch) * nacc + k];
    tmp[((int64_t)g * nch + ch) * nacc + k] = a;
}
inline int reduce_partials(mdsp_welch_plan_s* pl, const double* partial, double* reduced, int nslices, int64_t nch, int nacc, int accumulate, hipStream_t st) {
    if (nslices > 2 * REDUCE_GROUPS) {
        const int per_group = (int)cdiv(nslices, REDUCE_GROUPS), ng = (int)cdiv(nslices, per_group);
        MDSP_TRY(pl->redtmp.reserve(sizeof(double) * (size_t)ng * (size_t)nch * (size_t)nacc));
        hipLaunchKernelGGL(reduce_partials_groups_kernel, dim3((unsigned)cdiv(nacc, 256), (unsigned)ng, (unsigned)nch), dim3(256), 0, st, partial, pl->redtmp.as<double>(),
                           nslices, nch, nacc, per_group);
        MDSP_LAUNCH_CHECK();
        partial = pl->redtmp.as<double>();
        nslices = ng;
    }
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)cdiv(nacc, 32), (unsigned)nch), dim3(256), 0, st, partial, reduced, nslices, nch, nacc, accumulate);
    MDSP_LAUNCH_CHECK();
    return MDSP_OK;
}

// Welch finalize: psd[ch][j] = T( m_j * fold(sum over slices) )
//   MODE 0: one-sided from half spectrum  (acc has nspec = nfft/2+1 bins)       m = 1/r (DC, Nyquist if even) else 2/r
//   MODE 1: two-sided from full spectrum  (acc has nfft bins)                   m = 1/r
//   MODE 2: two-sided from half spectrum  (mirror, periodograms.jl:158-168)     m = 1/r
//   MODE 3: one-sided from PAIR-PACKED full spectrum: (A[k] + A[N-k])/2
//   MODE 4: two-sided from PAIR-PACKED full spectrum
template <typename R, int MODE>
__global__ __launch_bounds__(256) void welch_finalize_kernel(const double* __restrict__ partial, R* __restrict__ psd, int64_t ldp, int nslices,
                                                             int64_t nch, int nacc, int nfft, int nout, double r_total, const double* __restrict__ kdev,
                                                             double r_unit) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ch = blockIdx.y;
    if (j >= nout) return;
    if (kdev) {   // frame count summed over ranks on the device (mdsp_welch_allreduce): r = K fs sum(w^2) is formed here
        const double k = *kdev;
        if (!(k > 0)) {   // no frames anywhere: fill!(out, 0)
            psd[ch * ldp + j] = (R)0;
            return;
        }
        r_total = k * r_unit;
    }
    auto sum_bin = [&](int k) {
        double a = 0;
        for (int s = 0; s < nslices; ++s) a += partial[((int64_t)s * nch + ch) * nacc + k];
        return a;
    };
    double v;
    const double m1 = 1.0 / r_total, m2 = 2.0 / r_total;
    if (MODE == 0) {
        v = sum_bin(j) * ((j == 0 || (j == nout - 1 && (nfft % 2 == 0))) ? m1 : m2);
    } else if (MODE == 1) {
        v = sum_bin(j) * m1;
    } else if (MODE == 2) {
        const int k = j <= nfft / 2 ? j : nfft - j;
        v = sum_bin(k) * m1;
    } else {
        const int k2 = (nfft - j) % nfft;
        const double a = 0.5 * (sum_bin(j) + sum_bin(k2));
        if (MODE == 3) v = a * ((j == 0 || (j == nout - 1 && (nfft % 2 == 0))) ? m1 : m2);
        else v = a * m1;
    }
    psd[ch * ldp + j] = (R)v;
}

// K5/K6 (STFT): column store from a batched spectrum.  PSD: out real; else complex.
//   HALF: spectrum has nfft/2+1 bins (real input); two-sided output mirrors (conjugate for raw STFT).
template <typename R, bool PSD, bool HALF>
__global__ __launch_bounds__(256) void stft_store_kernel(const cx<R>* __restrict__ spec, void* __restrict__ out, int nspec, int nfft, int nout,
                                                         int64_t K, int64_t ldo, int64_t chs, int64_t u0, int64_t nunits, double r, int onesided,
                                                         int accumulate) {
    const int64_t u = u0 + blockIdx.x;
    if (u >= nunits) return;
    const int64_t ch = u / K, f = u - ch * K;
    const cx<R>* z = spec + (int64_t)blockIdx.x * nspec;
    const R m1 = (R)(1.0 / r), m2 = (R)(2.0 / r);
    for (int j = blockIdx.y * blockDim.x + threadIdx.x; j < nout; j += gridDim.y * blockDim.x) {
        int k = j;
        bool conj = false;
        if (HALF && j > nfft / 2) {
            k = nfft - j;
            conj = true;
        }
        const cx<R> v = z[k];
        if (PSD) {
            const R p = v.x * v.x + v.y * v.y;
            R m = m1;
            if (onesided && !(j == 0 || (j == nout - 1 && nfft % 2 == 0))) m = m2;
            R* o = static_cast<R*>(out) + ch * chs + f * ldo + j;
            *o = accumulate ? fma(p, m, *o) : p * m;                 // fft2pow!: out = muladd(abs2, m, out)
        } else {
            static_cast<cx<R>*>(out)[ch * chs + f * ldo + j] = conj ? cx<R>{v.x, -v.y} : v;
        }
    }
}

// ======================================================================================================
// Fused kernels
// ======================================================================================================
// LDS padding of the workgroup FFT: one element per 2^shift.  4 everywhere except the two headline geometries, where 5 measured
// faster (Welch nfft = 4096 E = 16: +5 %; overlap-save nfft = 2048: +3 %; the STFT nfft = 1024 kernel loses 7 % with 5).
template <typename R> constexpr int pad_default() { return 4; }

struct SpecArgs {
    const void* s;
    void* out;             // Welch: double partials [slot][ch][N];  STFT: output matrix
    const void* table;     // N forward roots
    const double* win;     // n doubles or nullptr
    int64_t len, lds_, K, hop;
    int64_t units_per_ch;  // frame pairs (real Welch) or frames
    int64_t nch;
    int64_t ldo, chs;      // STFT output strides
    int n, nout, onesided;
    int64_t run_len, niter;  // unit schedule: runs of run_len consecutive units per slot, niter iterations per slot
    int ablate;              // profiling aid (MDSP_ABLATE): 1 skip HBM loads, 2 skip transforms, 4 skip accumulate/stores
    int memprio;             // MDSP_SPEC_PRIO: 1 = a unit's loads (Welch) / loads and stores (STFT) are issued at raised wave priority
    int accumulate;          // STFT PSD mode: add to the output column instead of overwriting it (multitaper)
    int ntapers;             // > 0: multitaper PSD in one launch (stft_pair_kernel<MT>): win holds ntapers windows of n doubles, rinv their 1/r
    const double* rinv;
    double r;
};

// Load E window values for the thread (zero past n so that the zero tail needs no branch later)
template <int E, int T> __device__ __forceinline__ void load_window_regs(double (&w)[E], const double* win, int n, int t) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = t + T * e;
        w[e] = i < n ? (win ? win[i] : 1.0) : 0.0;
    }
}

// ---- Welch ------------------------------------------------------------------------------------------------
// SHIFT > 0 (complex signals; host guarantees n == N and hop == SHIFT * T): the samples a frame shares with its predecessor stay
// in registers, shifted by SHIFT elements per frame -- see stft_fused_kernel.
template <typename R, int N, int E, int G, int TWMODE, int PADSHIFT, bool CPLX, int MINW, int NBUF, bool PREFETCH, bool WIN64, int SHIFT = 0>
__global__ __launch_bounds__((N / E) * G, MINW) void welch_fused_kernel(SpecArgs a) {
    using C = fft::Cfg<N, E>;
    using TT = std::conditional_t<CPLX, cx<R>, R>;
    constexpr int T = C::T;
    static_assert(T % 64 == 0, "a transform must own whole wavefronts");
    constexpr int NTWA = C::NTW > 0 ? C::NTW : 1;
    constexpr int REGION = fft::wg_lds_elems<C, PADSHIFT, NBUF>();
    constexpr int64_t SZ = (int64_t)sizeof(TT);
    __shared__ __attribute__((aligned(16))) cx<R> lds_all[G * REGION];
    const int t = threadIdx.x % T;
    const int slot = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / T));  // wave-uniform (T % 64 == 0)
    cx<R>* lds = lds_all + slot * REGION;
    const cx<R>* table = static_cast<const cx<R>*>(a.table);
    const int64_t ch = blockIdx.y;

    cx<R> tw[NTWA];
    __shared__ __attribute__((aligned(16))) cx<R> twl[(TWMODE == fft::TW_LDS || TWMODE == fft::TW_HYB) ? fft::tw_lds_entries<C, TWMODE>() : 1];
    const cx<R>* twsrc = fft::wg_twiddle_setup<C, TWMODE>(tw, twl, t, slot, table);
    std::conditional_t<WIN64, double, R> w[E];
    {
        double wd[E];
        load_window_regs<E, T>(wd, a.win, a.n, t);
#pragma unroll
        for (int e = 0; e < E; ++e) w[e] = (std::conditional_t<WIN64, double, R>)wd[e];
    }
    double acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.0;

    const TT* sc = static_cast<const TT*>(a.s) + ch * a.lds_;
    // Unit schedule (wave-uniform): slot s of S walks runs of run_len consecutive units; run g of slot s starts at unit
    // (g*S + s)*run_len.  Consecutive frames share their n-hop overlap through L1/L2.
    const int64_t nslots = (int64_t)gridDim.x * G;
    int64_t wbase = ((int64_t)blockIdx.x * G + slot) * a.run_len, wj = 0;
    auto unit_cur = [&](bool more) { return (more && wbase + wj < a.units_per_ch) ? wbase + wj : a.units_per_ch; };  // else dead
    auto walk = [&]() {
        if (++wj == a.run_len) {
            wj = 0;
            wbase += nslots * a.run_len;
        }
    };
    const int64_t niter = a.niter;

    TT ra[E];
    TT rb[CPLX ? 1 : E];
    int64_t held = -2;   // SHIFT: frame whose raw samples ra[] holds
    auto issue = [&](int64_t u) {
        const bool live = u < a.units_per_ch;
        const int64_t f0 = CPLX ? u : 2 * u;
        const int64_t start = f0 * a.hop;
        // a frame never reads past start+n: the descriptor ends there (zero tail) or at the end of the signal
        const __amdgpu_buffer_rsrc_t r0 = io::make_rsrc(sc + start, live ? std::min<int64_t>(a.n, a.len - start) * SZ : 0);
        if constexpr (SHIFT > 0) {
            static_assert(CPLX && SHIFT < E, "frame-overlap reuse is wired for complex signals");
            if (live && u == held + 1) {   // wave-uniform: the next frame of the same run
#pragma unroll
                for (int e = 0; e < E - SHIFT; ++e) ra[e] = ra[e + SHIFT];
                io::load_window_tail<TT, E, T, E - SHIFT>(ra, r0, t);
                held = u;
                return;
            }
            held = live ? u : -2;
        }
        io::load_window<TT, E, T>(ra, r0, 0, t);
        if constexpr (!CPLX) {
            const bool haveB = live && (f0 + 1) < a.K;
            const int64_t startB = start + a.hop;
            const __amdgpu_buffer_rsrc_t r1 = io::make_rsrc(sc + startB, haveB ? std::min<int64_t>(a.n, a.len - startB) * SZ : 0);
            io::load_window<TT, E, T>(rb, r1, 0, t);
        }
    };
    int64_t u = unit_cur(niter > 0);
    if constexpr (PREFETCH) issue(u);
    for (int64_t it = 0; it < niter; ++it) {
        walk();
        const int64_t unext = unit_cur(it + 1 < niter);
        if constexpr (!PREFETCH) { if (!MDSP_ABLATED(a, 1)) issue(u); }
        cx<R> v[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if constexpr (CPLX) {
                if constexpr (WIN64) v[e] = win_mul(ra[e], (double)w[e]);
                else v[e] = {ra[e].x * w[e], ra[e].y * w[e]};
            } else {
                if constexpr (WIN64) v[e] = {win_mul(ra[e], (double)w[e]), win_mul(rb[e], (double)w[e])};
                else v[e] = {ra[e] * w[e], rb[e] * w[e]};
            }
        }
        if constexpr (PREFETCH) { if (!MDSP_ABLATED(a, 1)) issue(unext); }
        u = unext;
        if (!MDSP_ABLATED(a, 2))
        fft::wg_fft<C, -1, TWMODE, PADSHIFT, NBUF, 0>(v, t, tw, twsrc, lds);
        // an odd number of exchanges per iteration would re-enter on the buffer that was used last: fence it
        if constexpr (C::P > 1 && NBUF > 1 && ((C::P - 1) % NBUF) != 0) fft::wg_sync<T>();  // NBUF == 1: wg_fft already ends every exchange with a barrier
        // K5: |Z|^2 in the working precision (one rounding per term), accumulated over frames in double
        if (!MDSP_ABLATED(a, 4)) {
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] += (double)(v[e].x * v[e].x + v[e].y * v[e].y);
        }
    }
    // partial[(blockIdx.x*G + slot)][ch][k]
    double* part = static_cast<double*>(a.out) + (((int64_t)blockIdx.x * G + slot) * a.nch + ch) * N;
#pragma unroll
    for (int e = 0; e < E; ++e) part[t + T * e] = acc[e];
}

// ---- STFT / spectrogram -------------------------------------------------------------------------------------
// One frame per transform slot (real frames ride with a zero imaginary part).
// SHIFT > 0 (host guarantees n == N and hop == SHIFT * T): consecutive frames of a slot's run overlap by N - hop samples, and a
// thread's element e of frame f+1 is its element e + SHIFT of frame f -- the raw samples stay in registers, shift by SHIFT
// elements per frame, and only the SHIFT new ones are loaded: every sample is read from memory once per run instead of
// N / hop times (config 4, 75 % overlap: the L2 absorbed only part of the re-reads, PMC fetch 2.3x the input).
template <typename R, int N, int E, int G, int TWMODE, int PADSHIFT, bool CPLX, bool PSD, int MINW, int NBUF, bool PREFETCH, int SHIFT = 0>
__global__ __launch_bounds__((N / E) * G, MINW) void stft_fused_kernel(SpecArgs a) {
    using C = fft::Cfg<N, E>;
    using TT = std::conditional_t<CPLX, cx<R>, R>;
    constexpr int T = C::T;
    static_assert(T % 64 == 0, "a transform must own whole wavefronts");
    constexpr int NTWA = C::NTW > 0 ? C::NTW : 1;
    constexpr int REGION = fft::wg_lds_elems<C, PADSHIFT, NBUF>();
    constexpr int64_t SZ = (int64_t)sizeof(TT);
    __shared__ __attribute__((aligned(16))) cx<R> lds_all[G * REGION];
    const int t = threadIdx.x % T;
    const int slot = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / T));  // wave-uniform (T % 64 == 0)
    cx<R>* lds = lds_all + slot * REGION;
    const cx<R>* table = static_cast<const cx<R>*>(a.table);
    const int64_t ch = blockIdx.y;

    cx<R> tw[NTWA];
    __shared__ __attribute__((aligned(16))) cx<R> twl[(TWMODE == fft::TW_LDS || TWMODE == fft::TW_HYB) ? fft::tw_lds_entries<C, TWMODE>() : 1];
    const cx<R>* twsrc = fft::wg_twiddle_setup<C, TWMODE>(tw, twl, t, slot, table);
    // window in the working precision: Float32 frames are multiplied by the Float32-rounded window (one rounding more than the
    // reference's Float64 product rounded once, periodograms.jl:62 -- 6e-8 relative, far inside the Float32 FFT's own error),
    // which halves the window registers and drops four conversions per sample from the VALU stream
    R w[E];
    {
        double wd[E];
        load_window_regs<E, T>(wd, a.win, a.n, t);
#pragma unroll
        for (int e = 0; e < E; ++e) w[e] = (R)wd[e];
    }
    const bool havewin = a.win != nullptr;

    const TT* sc = static_cast<const TT*>(a.s) + ch * a.lds_;
    const int64_t nslots = (int64_t)gridDim.x * G;
    int64_t wbase = ((int64_t)blockIdx.x * G + slot) * a.run_len, wj = 0;
    auto unit_cur = [&](bool more) { return (more && wbase + wj < a.K) ? wbase + wj : a.K; };
    auto walk = [&]() {
        if (++wj == a.run_len) {
            wj = 0;
            wbase += nslots * a.run_len;
        }
    };
    const int64_t niter = a.niter;
    const R m1 = (R)(1.0 / a.r), m2 = (R)(2.0 / a.r);

    TT ra[E];
    int64_t held = -2;   // frame whose raw samples ra[] holds
    auto issue = [&](int64_t f) {
        const bool live = f < a.K;
        const int64_t start = f * a.hop;
        const __amdgpu_buffer_rsrc_t r0 = io::make_rsrc(sc + start, live ? std::min<int64_t>(a.n, a.len - start) * SZ : 0);
        if constexpr (SHIFT > 0) {
            static_assert(SHIFT < E, "a frame shift leaves samples to reuse");
            if (live && f == held + 1) {   // wave-uniform: the next frame of the same run
#pragma unroll
                for (int e = 0; e < E - SHIFT; ++e) ra[e] = ra[e + SHIFT];
                io::load_window_tail<TT, E, T, E - SHIFT>(ra, r0, t);
                held = f;
                return;
            }
            held = live ? f : -2;
        }
        io::load_window<TT, E, T>(ra, r0, 0, t);
    };
    int64_t fcur = unit_cur(niter > 0);
    if constexpr (PREFETCH) issue(fcur);
    for (int64_t it = 0; it < niter; ++it) {
        const int64_t f = fcur;
        walk();
        fcur = unit_cur(it + 1 < niter);
        if constexpr (!PREFETCH) {
            if (a.memprio & 1) __builtin_amdgcn_s_setprio(3);
            issue(f);
            if (a.memprio & 1) __builtin_amdgcn_s_setprio(0);
        }
        cx<R> v[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if constexpr (CPLX) v[e] = havewin ? cx<R>{ra[e].x * w[e], ra[e].y * w[e]} : ra[e];
            else v[e] = {havewin ? ra[e] * w[e] : ra[e], (R)0};
        }
        if constexpr (PREFETCH) issue(fcur);
        fft::wg_fft<C, -1, TWMODE, PADSHIFT, NBUF, 0>(v, t, tw, twsrc, lds);
        if constexpr (C::P > 1 && NBUF > 1 && ((C::P - 1) % NBUF) != 0) fft::wg_sync<T>();  // NBUF == 1: wg_fft already ends every exchange with a barrier
        // column store: bins k = t + T*e < nout, contiguous across lanes
        const bool live = f < a.K;
        if (a.memprio & 2) __builtin_amdgcn_s_setprio(3);
        if constexpr (PSD) {
            R* col = static_cast<R*>(a.out) + ch * a.chs + f * a.ldo;
            const __amdgpu_buffer_rsrc_t wr = io::make_rsrc(col, live ? (int64_t)a.nout * (int64_t)sizeof(R) : 0);
            R old[E];
            if (a.accumulate) io::load_window<R, E, T>(old, wr, 0, t);   // wave-uniform branch; the column is this slot's alone
            io::store_window<R, E, T>(
                [&](int e) {
                    const int k = t + T * e;
                    R m = m1;
                    if (a.onesided && !(k == 0 || (k == a.nout - 1 && N % 2 == 0))) m = m2;
                    const R p = (v[e].x * v[e].x + v[e].y * v[e].y) * m;
                    return a.accumulate ? p + old[e] : p;
                },
                wr, 0, t);
        } else {
            cx<R>* col = static_cast<cx<R>*>(a.out) + ch * a.chs + f * a.ldo;
            const __amdgpu_buffer_rsrc_t wr = io::make_rsrc(col, live ? (int64_t)a.nout * (int64_t)sizeof(cx<R>) : 0);
            io::store_window<cx<R>, E, T>([&](int e) { return v[e]; }, wr, 0, t);
        }
        if (a.memprio & 2) __builtin_amdgcn_s_setprio(0);
    }
}

// ---- STFT / spectrogram of REAL signals: two frames per complex transform ----------------------------------------------
// z = w (a + i b) with a, b two consecutive frames; after the forward transform the two spectra are untangled with the
// mirror bin,  A[k] = (Z[k] + conj(Z[N-k])) / 2,  B[k] = (Z[k] - conj(Z[N-k])) / (2i),  for the nfft/2+1 non-redundant
// bins.  Z[N-k] lives in another thread (bin N-k = (T-t) + T (E-1-e)), so the spectrum takes one more trip through LDS
// -- one exchange instead of a second transform's P-1 exchanges and all of its butterflies.
template <typename R, int N, int E, int G, int TWMODE, int PADSHIFT, bool PSD, int MINW, int NBUF, bool MT = false>
__global__ __launch_bounds__((N / E) * G, MINW) void stft_pair_kernel(SpecArgs a) {
    using C = fft::Cfg<N, E>;
    constexpr int T = C::T, H = E / 2;
    static_assert(T % 64 == 0 && E % 2 == 0, "a transform must own whole wavefronts");
    constexpr int NTWA = C::NTW > 0 ? C::NTW : 1;
    constexpr int REGION = fft::wg_lds_elems<C, PADSHIFT, NBUF>();
    constexpr int64_t SZ = (int64_t)sizeof(R);
    __shared__ __attribute__((aligned(16))) cx<R> lds_all[G * REGION];
    const int t = threadIdx.x % T;
    const int slot = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / T));
    cx<R>* lds = lds_all + slot * REGION;
    const cx<R>* table = static_cast<const cx<R>*>(a.table);
    const int64_t ch = blockIdx.y;

    cx<R> tw[NTWA];
    __shared__ __attribute__((aligned(16))) cx<R> twl[(TWMODE == fft::TW_LDS || TWMODE == fft::TW_HYB) ? fft::tw_lds_entries<C, TWMODE>() : 1];
    const cx<R>* twsrc = fft::wg_twiddle_setup<C, TWMODE>(tw, twl, t, slot, table);
    R w[E];
    {
        double wd[E];
        load_window_regs<E, T>(wd, a.win, a.n, t);
#pragma unroll
        for (int e = 0; e < E; ++e) w[e] = (R)wd[e];
    }
    const R* sc = static_cast<const R*>(a.s) + ch * a.lds_;
    const int64_t nslots = (int64_t)gridDim.x * G;
    int64_t wbase = ((int64_t)blockIdx.x * G + slot) * a.run_len, wj = 0;
    auto unit_cur = [&](bool more) { return (more && wbase + wj < a.units_per_ch) ? wbase + wj : a.units_per_ch; };
    auto walk = [&]() {
        if (++wj == a.run_len) {
            wj = 0;
            wbase += nslots * a.run_len;
        }
    };
    const int64_t niter = a.niter;
    const R m1 = (R)(1.0 / a.r), m2 = (R)(2.0 / a.r);
    using OT = std::conditional_t<PSD, R, cx<R>>;

    R ra[E], rb[E];
    auto issue = [&](int64_t u) {   // frames 2u and 2u+1 (the second one may not exist)
        const int64_t fa = 2 * u, fb = 2 * u + 1;
        const int64_t sa = fa * a.hop, sb = fb * a.hop;
        const __amdgpu_buffer_rsrc_t r0 = io::make_rsrc(sc + sa, fa < a.K ? std::min<int64_t>(a.n, a.len - sa) * SZ : 0);
        const __amdgpu_buffer_rsrc_t r1 = io::make_rsrc(sc + sb, fb < a.K ? std::min<int64_t>(a.n, a.len - sb) * SZ : 0);
        io::load_window<R, E, T>(ra, r0, 0, t);
        io::load_window<R, E, T>(rb, r1, 0, t);
    };
    int64_t ucur = unit_cur(niter > 0);
    issue(ucur);
    for (int64_t it = 0; it < niter; ++it) {
        const int64_t u = ucur;
        walk();
        ucur = unit_cur(it + 1 < niter);
        if constexpr (MT && PSD) {
            // multitaper PSD (mt_pgram!, multitaper.jl:239-243): the frame pair stays in registers while every taper is applied,
            // transformed, untangled and accumulated; one store per bin at the end
            static_assert(PSD, "multitaper accumulates powers");
            R pa[H], pb[H], pna = (R)0, pnb = (R)0;
#pragma unroll
            for (int e = 0; e < H; ++e) pa[e] = pb[e] = (R)0;
            const int mbase = fft::lds_pad<PADSHIFT>((N - t) & (N - 1));
            for (int k = 0; k < a.ntapers; ++k) {
                const double* wk = a.win + (int64_t)k * a.n;
                cx<R> v[E];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int i = t + T * e;
                    const R wv = i < a.n ? (R)wk[i] : (R)0;
                    v[e] = {ra[e] * wv, rb[e] * wv};
                }
                fft::wg_fft<C, -1, TWMODE, PADSHIFT, NBUF, 0>(v, t, tw, twsrc, lds);
                if constexpr (C::P > 1 && NBUF > 1) fft::wg_sync<T>();
                {
                    const int base = fft::lds_pad<PADSHIFT>(t);
#pragma unroll
                    for (int e = 0; e < E; ++e) lds[base + fft::lds_padc<PADSHIFT>(T * e)] = v[e];
                }
                fft::wg_sync<T>();
                const double rk = a.rinv[k];
                const R k1 = (R)rk, k2 = (R)(2.0 * rk);
                auto power = [&](cx<R> z, cx<R> p, R m, R& qa, R& qb) {
                    const cx<R> A = {(z.x + p.x) * (R)0.5, (z.y - p.y) * (R)0.5};
                    const cx<R> B = {(z.y + p.y) * (R)0.5, (p.x - z.x) * (R)0.5};
                    qa += (A.x * A.x + A.y * A.y) * m;
                    qb += (B.x * B.x + B.y * B.y) * m;
                };
#pragma unroll
                for (int e = 0; e < H; ++e) {
                    const cx<R> p = e == 0 ? lds[mbase] : lds[fft::lds_pad<PADSHIFT>(N - t - T * e)];
                    const bool dc = (t + T * e) == 0;
                    power(v[e], p, (a.onesided && !dc) ? k2 : k1, pa[e], pb[e]);
                }
                if (t == 0) power(v[H], v[H], k1, pna, pnb);       // Nyquist (nfft even): weight 1/r in both layouts
                if constexpr (C::P > 1) fft::wg_sync<T>();
            }
            issue(ucur);
            const int64_t fa = 2 * u, fb = 2 * u + 1;
            R* colA = static_cast<R*>(a.out) + ch * a.chs + fa * a.ldo;
            R* colB = static_cast<R*>(a.out) + ch * a.chs + fb * a.ldo;
            const __amdgpu_buffer_rsrc_t wa = io::make_rsrc(colA, fa < a.K ? (int64_t)a.nout * (int64_t)sizeof(R) : 0);
            const __amdgpu_buffer_rsrc_t wb = io::make_rsrc(colB, fb < a.K ? (int64_t)a.nout * (int64_t)sizeof(R) : 0);
#pragma unroll
            for (int e = 0; e < H; ++e) {
                const int kb = t + T * e;
                io::Ld<R>::store(pa[e], wa, kb * (int)sizeof(R));
                io::Ld<R>::store(pb[e], wb, kb * (int)sizeof(R));
                if (!a.onesided && kb != 0) {
                    io::Ld<R>::store(pa[e], wa, (N - kb) * (int)sizeof(R));
                    io::Ld<R>::store(pb[e], wb, (N - kb) * (int)sizeof(R));
                }
            }
            if (t == 0) {
                io::Ld<R>::store(pna, wa, (N / 2) * (int)sizeof(R));
                io::Ld<R>::store(pnb, wb, (N / 2) * (int)sizeof(R));
            }
            continue;
        }
        cx<R> v[E];
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = {ra[e] * w[e], rb[e] * w[e]};   // w == 1 beyond... load_window_regs gives 1 without a window, 0 past n
        issue(ucur);
        fft::wg_fft<C, -1, TWMODE, PADSHIFT, NBUF, 0>(v, t, tw, twsrc, lds);
        // mirror exchange through the (now idle) first LDS buffer: natural order, so both the writes and the descending reads
        // are contiguous across lanes
        if constexpr (C::P > 1 && NBUF > 1) fft::wg_sync<T>();   // the last gather of the transform may still be reading buffer (P-2) % NBUF
        {
            const int base = fft::lds_pad<PADSHIFT>(t);
#pragma unroll
            for (int e = 0; e < E; ++e) lds[base + fft::lds_padc<PADSHIFT>(T * e)] = v[e];
        }
        fft::wg_sync<T>();
        const int64_t fa = 2 * u, fb = 2 * u + 1;
        const bool liveA = fa < a.K, liveB = fb < a.K;
        OT* colA = static_cast<OT*>(a.out) + ch * a.chs + fa * a.ldo;
        OT* colB = static_cast<OT*>(a.out) + ch * a.chs + fb * a.ldo;
        const __amdgpu_buffer_rsrc_t wa = io::make_rsrc(colA, liveA ? (int64_t)a.nout * (int64_t)sizeof(OT) : 0);
        const __amdgpu_buffer_rsrc_t wb = io::make_rsrc(colB, liveB ? (int64_t)a.nout * (int64_t)sizeof(OT) : 0);
        auto emit = [&](int k, cx<R> z, cx<R> p) {   // bin k <= N/2 of both frames from Z[k] = z and Z[N-k] = p
            const cx<R> A = {(z.x + p.x) * (R)0.5, (z.y - p.y) * (R)0.5};
            const cx<R> B = {(z.y + p.y) * (R)0.5, (p.x - z.x) * (R)0.5};
            const int off = k * (int)sizeof(OT);
            if constexpr (PSD) {
                R m = m1;
                if (a.onesided && !(k == 0 || (k == N / 2 && N % 2 == 0))) m = m2;
                R pa = (A.x * A.x + A.y * A.y) * m, pb = (B.x * B.x + B.y * B.y) * m;
                if (a.accumulate) {   // multitaper: fft2pow! adds this taper's power to the column (wave-uniform branch)
                    pa += io::Ld<R>::load(wa, off);
                    pb += io::Ld<R>::load(wb, off);
                }
                io::Ld<R>::store(pa, wa, off);
                io::Ld<R>::store(pb, wb, off);
                if (!a.onesided && k != 0 && k != N / 2) {   // real signal, two-sided output: the mirrored bin carries the same power
                    const int offm = (N - k) * (int)sizeof(OT);
                    io::Ld<R>::store(pa, wa, offm);
                    io::Ld<R>::store(pb, wb, offm);
                }
            } else {
                io::Ld<cx<R>>::store(A, wa, off);
                io::Ld<cx<R>>::store(B, wb, off);
                if (!a.onesided && k != 0 && k != N / 2) {   // fft2oneortwosided!: out[N-k] = conj(out[k])
                    const int offm = (N - k) * (int)sizeof(OT);
                    io::Ld<cx<R>>::store(cx<R>{A.x, -A.y}, wa, offm);
                    io::Ld<cx<R>>::store(cx<R>{B.x, -B.y}, wb, offm);
                }
            }
        };
        // bins k = t + T e, e < E/2, are <= N/2 - 1 + ... (k < N/2); their mirrors N - k sit at index (N - t) - T e
        {
            const int mbase = fft::lds_pad<PADSHIFT>((N - t) & (N - 1));   // t == 0: bin 0 mirrors itself (index 0)
#pragma unroll
            for (int e = 0; e < H; ++e) {
                cx<R> p;
                if (e == 0) p = lds[mbase];
                else p = lds[fft::lds_pad<PADSHIFT>(N - t - T * e)];
                emit(t + T * e, v[e], p);
            }
            if (t == 0) emit(N / 2, v[H], v[H]);   // the Nyquist bin is thread 0's element E/2 and its own mirror
        }
        if constexpr (C::P > 1) fft::wg_sync<T>();   // the next transform's first scatter reuses this buffer
    }
}

#include "spectral_gen.h"
#include "spectral_gx.h"

bool fused_size_ok(int dtype, int64_t nfft) {
    const bool dbl = dtype_is_double(dtype);
    switch (nfft) {
        case 256: case 512: case 1024: case 2048: case 4096: return true;
        case 8192: return !dbl;
        default: return false;
    }
}

// round 6: the run-time-schedule kernel (spectral_gx.h) takes every 7-smooth size it plans that has neither a register-resident power-of-two kernel
// nor a compile-time schedule -- in front of the round-2 LDS kernel, the multi-pass engine and the rocFFT pipeline.  kind: 0 Welch sums, 1 columns.
// Measured against both (profiles/r06_gx_vs_r5.json: Float32 / Float64 / ComplexF32 / ComplexF64, 1125 .. 65536 points):
//   Welch          : 2 - 7 x either at every size but the powers of two from 32768 (the multi-pass engine's two register stages: a tie)
//   real columns   : 1.1 - 6 x, a tie with rocFFT at R0 >= 5 (12500, 40000); 16384 = 2 x 8192 loses 8 % to the multi-pass engine
//   complex columns: wins from 4097 points while one workgroup holds the transform and up to R0 = 4 (8400, 20000); rocFFT is faster below 4097
//                    (1.4 - 1.7 against 0.9 - 1.3 TB/s) and at R0 >= 5 (40000: 0.75 / 0.91 against 0.67 / 0.59)
// nfft = R0 x S in two kernels (spectral_ctrows.hip), R0 up to 32: from R0 = 6.  Measured (profiles/r06_ctrows.json, TB/s): 98304 = 6 x 16384 0.70 against 0.61
// on the fused column step, 100000 = 8 x 12500 0.66 / 0.49, 131072 0.74 / 0.54; a tie at R0 = 5 (81920: 0.69 / 0.69); the fused step wins below (65536 = 4 x 16384
// 0.74 / 0.81, 32768 0.76 / 1.01).  Beyond eight rows there is no fused step: 150000 .. 500000 points 0.50 - 0.67 against the multi-pass engine's 0.08 - 0.18, 2^19 =
// 32 x 16384 0.73 against 0.52 on the engine's rows form.  MDSP_GX=8: wherever a split exists (A/B), MDSP_GX=-1: never.
int ctrows_r0(int dtype, int64_t nfft) {
    const int m = tunables().gx;
    if (m == 0 || m == -1 || m == 4 || m == 5 || m == 6 || ctbig_ok(dtype, nfft) || (m == 3 && dtype_is_double(dtype))) return 0;
    const int any = ctrows_split(dtype, nfft, 2);
    if (any == 0) return 0;
    if (m == 8) return any;
    const int fused = ctcols_split(dtype, nfft);
    return (fused == 0 || fused >= (dtype_is_double(dtype) ? 5 : 6)) ? any : 0;   // (Float64, r06s71 / r06s72: 48000 = 5 x 9600 0.60 against 0.51 fused, 57600 0.62 / 0.45; a tie at R0 = 4)
}
bool use_gx(int dtype, int64_t nfft, bool direct, int kind) {
    const int m = tunables().gx;
    if (m == 0) return false;
    if (kind == 0 && m != 4 && m != 5 && ctbig_preferred(dtype, nfft)) return true;
    if (kind == 0 && m != 4 && !fused_size_ok(dtype, nfft) && !gen_ct_size(dtype, nfft, direct) && ((m != 5 && ctbig_ok(dtype, nfft)) || ctcols_split(dtype, nfft) > 0 || ctrows_r0(dtype, nfft) > 0))
        return true;   // Welch sums on a compile-time schedule (one workgroup, or R0 x S rows): whatever the run-time-schedule kernel plans
    if (kind == 1 && m != 4 && m != 5 && !fused_size_ok(dtype, nfft) && !gen_ct_size(dtype, nfft, direct) && ctbig_cols_ok(dtype, nfft))
        return true;   // columns on a single-workgroup compile-time schedule (spectral_ctbig_cols.hip), 16384 points included
    if (!gx_size_ok(dtype, nfft)) return false;
    if (m >= 2) return true;
    const bool pow2 = (nfft & (nfft - 1)) == 0;
    if (pow2 && nfft >= (kind == 0 ? 32768 : 16384) && big::size_ok(dtype, nfft) && !(kind == 0 && ctcols_split(dtype, nfft) > 0)) return false;   // (Welch, Float32: 4 x / 8 x 8192 on the compile-time rows)
    return !fused_size_ok(dtype, nfft) && !gen_ct_size(dtype, nfft, direct);
}
// ... and where AUTO prefers it to the rocFFT pipeline (engine = FUSED takes it wherever use_gx says so)
bool gx_wins(int dtype, int64_t nfft, int kind) {
    if (kind == 0 || !dtype_is_complex(dtype)) return true;
    if (tunables().gx != 4 && tunables().gx != 5 && ctbig_cols_ok(dtype, nfft)) return true;   // single-workgroup compile-time columns: 1.6 - 2.9 TB/s against rocFFT's 0.3 - 1.4 (r06s58)
    if (nfft <= 4096) return false;
    return gx_split_r0(dtype, nfft) <= 4;
}
// the multi-pass engine takes what no single-workgroup kernel does
bool use_big(int dtype, int64_t nfft, bool direct, int kind) {
    return !use_gx(dtype, nfft, direct, kind) && !fused_size_ok(dtype, nfft) && !gen_size_ok(dtype, nfft) && !gen_ct_size(dtype, nfft, direct) && big::size_ok(dtype, nfft);
}

// kind: 0 = Welch (sums of |X|^2), 1 = STFT / spectrogram / periodogram columns
int resolve_engine(int engine, int dtype, int64_t nfft, int* out, int kind = 1) {
    int eng = engine;
    if (eng == MDSP_ENGINE_AUTO) eng = tunables().engine;
    // fused: the register-resident power-of-two sizes, and the mixed-radix LDS kernel for the other 7-smooth sizes nextfastfft returns
    const bool direct = kind == 0 || dtype_is_complex(dtype);   // the last pass is consumed from registers (spectral_gen.h)
    const bool big_ok = use_big(dtype, nfft, direct, kind);   // round 5: the multi-pass engine (bigfft.hip) for everything above the one-workgroup sizes
    const bool gx_ok = use_gx(dtype, nfft, direct, kind);
    const bool fused_ok = gx_ok || fused_size_ok(dtype, nfft) || gen_size_ok(dtype, nfft) || gen_ct_size(dtype, nfft, direct) || big_ok;
    // AUTO takes the mixed-radix kernel where it measured faster than the rocFFT pipeline (profiles/r02g_mixed.json, 2^27 samples): Welch and
    // real-signal columns up to 4096 points (1.4-3x), complex columns above (1.6x); elsewhere the two are within 20 % and rocFFT is kept.
    // Round 3: the sizes with a compile-time schedule (spectral_gen.h, Float32 / ComplexF32) beat the rocFFT pipeline 2-9x in every mode
    // (profiles/r03d_mixed_ct.json) and are always taken.
    // ... taken by AUTO where it measured faster than the rocFFT pipeline (profiles/r05_big_vs_rocfft.json, 2^26 samples): powers of two from 16384 on
    // (two-stage register passes: Welch 1.5 - 2.2x, columns 1.1 - 1.4x; the 8192-point Float64 split loses 0.7 - 0.8x), other 7-smooth sizes from
    // 50000 points on for Welch and real-signal columns (1.1 - 1.7x at 50000 / 100000 / 125000 / 200000; complex columns lose 0.7 - 0.9x there, and
    // below 50000 the generic passes' small tiles lose to rocFFT at most sizes: 8400 .. 10000 0.7x, 20000 0.5x, 40000 1.0x)
    const bool pow2 = (nfft & (nfft - 1)) == 0;
    const bool big_wins = big_ok && (pow2 ? nfft >= 16384 : (nfft >= 50000 && !(kind == 1 && dtype_is_complex(dtype))));
    const bool gen_wins = (gx_ok && gx_wins(dtype, nfft, kind)) || big_wins || fused_size_ok(dtype, nfft) || gen_ct_size(dtype, nfft, direct) ||
                          (gen_size_ok(dtype, nfft) && ((nfft <= 4096) == (kind == 0 || !dtype_is_complex(dtype))));
    if (eng == MDSP_ENGINE_AUTO) eng = gen_wins ? MDSP_ENGINE_FUSED : MDSP_ENGINE_ROCFFT;
    if (eng == MDSP_ENGINE_FUSED && !fused_ok)
        MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "fused engine supports nfft = 2^a 3^b 5^c 7^d (up to %d, or splitting into 2..4 factors of at most 512); got %lld",
                  dtype_is_double(dtype) ? 4096 : 8192, (long long)nfft);
    if (eng != MDSP_ENGINE_FUSED && eng != MDSP_ENGINE_ROCFFT) MDSP_FAIL(MDSP_ERR_ARGUMENT, "invalid engine %d", engine);
    *out = eng;
    return MDSP_OK;
}

// the window as the lean compile-time schedules read it (spectral_gen.h ct_pass0_lean): nfft values in the working precision, ones without a window, zero tail
template <typename R> __global__ __launch_bounds__(256) void lean_window_kernel(const double* __restrict__ win, R* __restrict__ out, int n, int nfft) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i < nfft) out[i] = i < n ? (win ? (R)win[i] : (R)1) : (R)0;
}
template <typename R> int lean_window(DevBuf& buf, const double* win, int n, int64_t nfft, hipStream_t st) {
    MDSP_TRY(buf.reserve(sizeof(R) * (size_t)nfft));
    hipLaunchKernelGGL(lean_window_kernel<R>, dim3((unsigned)cdiv(nfft, 256)), dim3(256), 0, st, win, buf.as<R>(), n, (int)nfft);
    MDSP_LAUNCH_CHECK();
    return MDSP_OK;
}
constexpr bool lean_window_needed(int64_t nfft, bool dbl) { return dbl && MDSP_F64_LEAN && nfft >= 4800; }   // (wherever gen_ct_f64_tw2l may set flag 4096)

template <typename R> int upload_roots(DevBuf& buf, int64_t n) {
    std::vector<cx<R>> w((size_t)n);
    for (int64_t k = 0; k < n; ++k) {
        const zd r = unit_root(k, n, -1);
        w[(size_t)k] = {(R)r.real(), (R)r.imag()};
    }
    MDSP_TRY(buf.reserve(sizeof(cx<R>) * (size_t)n));
    MDSP_HIP(hipMemcpy(buf.p, w.data(), sizeof(cx<R>) * (size_t)n, hipMemcpyHostToDevice));
    return MDSP_OK;
}

int check_split(int64_t n, int64_t noverlap, int64_t nfft) {
    if (n < 1) MDSP_FAIL(MDSP_ERR_DOMAIN, "n (%lld) must be positive", (long long)n);
    if (!(0 <= noverlap && noverlap < n))
        MDSP_FAIL(MDSP_ERR_DOMAIN, "noverlap must be between zero and n (noverlap=%lld, n=%lld)", (long long)noverlap, (long long)n);
    if (!(nfft >= n)) MDSP_FAIL(MDSP_ERR_DOMAIN, "nfft must be >= n (nfft=%lld, n=%lld)", (long long)nfft, (long long)n);
    return MDSP_OK;
}

// geometry shared by the fused launchers
template <typename R, int N> struct Geo {
    static constexpr bool DBL = sizeof(R) == 8;
    // 8 elements per thread in general; nfft = 1024 in Float32 takes 16 so that a transform is ONE wavefront (no s_barrier)
#ifndef MDSP_GEO_E1024
#define MDSP_GEO_E1024 16
#endif
    static constexpr int EMAX = (N == 1024 && !DBL) ? MDSP_GEO_E1024 : 8;
    static constexpr int E = (N / 64 < EMAX) ? N / 64 : EMAX;
    static constexpr int T = N / E;
    static constexpr int G = slots_per_workgroup(T);
    static constexpr int NBUF = T <= 64 ? 1 : 2;
#ifndef MDSP_TW_F64
#define MDSP_TW_F64 1
#endif
    static constexpr int TWREG = DBL ? MDSP_TW_F64 : 1;   // Float64 twiddles: 1 = registers, 0 = global table
};

// runs of consecutive units per slot (default: one run = fully contiguous), identical trip count for every slot
void set_schedule(SpecArgs& a, int64_t nunits, int64_t nslots) {
    a.ablate = MDSP_DBG(ablate);
    a.memprio = tunables().spec_prio;
    const int64_t runs = tunables().runs_per_slot;
    a.run_len = std::max<int64_t>(1, cdiv(nunits, nslots * runs));
    a.niter = cdiv(cdiv(nunits, a.run_len), nslots) * a.run_len;
}

template <typename K> int grid_for(K kern, int threads, int64_t work_wgs, int64_t nch, int* grid) {
    int per_cu = 0;
    MDSP_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, 0));
    if (per_cu < 1) per_cu = 1;
    if (tunables().wg_per_cu > 0) per_cu = tunables().wg_per_cu;
    const int64_t resident = (int64_t)device_cu_count() * per_cu;
    const int64_t per_ch = std::max<int64_t>(1, resident / std::max<int64_t>(1, nch));
    *grid = (int)std::max<int64_t>(1, std::min<int64_t>(work_wgs, per_ch));
    return MDSP_OK;
}

}  // namespace

// ======================================================================================================
// mdsp_frames (K4 only)
// ======================================================================================================
extern "C" int mdsp_frames(const void* s_dev, int64_t len, int dtype, int64_t n, int64_t noverlap, int64_t nfft, const double* window_host,
                           int64_t first, int64_t count, void* frames_dev, void* stream) {
    if (!dtype_valid(dtype)) MDSP_FAIL(MDSP_ERR_ARGUMENT, "invalid dtype %d", dtype);
    MDSP_TRY(check_split(n, noverlap, nfft));
    const int64_t K = mdsp_frame_count(len, n, noverlap);
    if (first < 0 || count < 0 || first + count > K) MDSP_FAIL(MDSP_ERR_ARGUMENT, "frame range [%lld,%lld) outside [0,%lld)", (long long)first, (long long)(first + count), (long long)K);
    if (count == 0) return MDSP_OK;
    hipStream_t st = as_stream(stream);
    DevBuf win;
    if (window_host) {
        MDSP_TRY(win.reserve(sizeof(double) * (size_t)n));
        MDSP_HIP(hipMemcpyAsync(win.p, window_host, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
    }
    const int64_t hop = n - noverlap;
    const int gx = (int)std::min<int64_t>(cdiv(nfft, 256), 8);
    for (int64_t c0 = 0; c0 < count; c0 += 32768) {
        const int64_t cnt = std::min<int64_t>(32768, count - c0);
#define FR(TT)                                                                                                                          \
    hipLaunchKernelGGL(frame_window_kernel<TT>, dim3((unsigned)cnt, gx), dim3(256), 0, st, (const TT*)s_dev, (TT*)frames_dev + c0 * nfft, \
                       window_host ? win.as<double>() : nullptr, (int64_t)0, K, hop, (int)n, (int)nfft, first + c0, first + count)
        switch (dtype) {
            case MDSP_F32: FR(float); break;
            case MDSP_F64: FR(double); break;
            case MDSP_C32: FR(cx<float>); break;
            default: FR(cx<double>); break;
        }
#undef FR
        MDSP_LAUNCH_CHECK();
    }
    if (window_host) MDSP_HIP(hipStreamSynchronize(st));  // `win` is freed on return
    return MDSP_OK;
}

// ======================================================================================================
// Welch
// ======================================================================================================
// struct mdsp_welch_plan_s: welch_plan.h (shared with comm.hip / hostpath.hip)

namespace {

template <typename R, bool CPLX>
int welch_accumulate_rocfft(mdsp_welch_plan_s* pl, const void* s, int64_t len, int64_t nch, int64_t lds_, hipStream_t st) {
    using TT = std::conditional_t<CPLX, cx<R>, R>;
    const int64_t nfft = pl->nfft, n = pl->n, hop = pl->n - pl->noverlap;
    const int64_t K = mdsp_frame_count(len, n, pl->noverlap);
    const int nspec = (int)(CPLX ? nfft : nfft / 2 + 1);
    const int64_t nunits = K * nch;
    const int nslices = 32;
    pl->acc_nslices = nslices;
    pl->acc_nacc = nspec;
    pl->acc_mode = CPLX ? 1 : (pl->onesided ? 0 : 2);
    if (pl->acc_fresh) {
        MDSP_TRY(pl->partial.reserve(sizeof(double) * (size_t)nslices * (size_t)nch * (size_t)nspec));
        MDSP_HIP(hipMemsetAsync(pl->partial.p, 0, sizeof(double) * (size_t)nslices * (size_t)nch * (size_t)nspec, st));
        pl->acc_fresh = false;
    }
    if (nunits > 0) {
        const int64_t per_unit = (int64_t)sizeof(TT) * nfft + (int64_t)sizeof(cx<R>) * nspec;
        const int64_t batch = std::max<int64_t>(1, std::min<int64_t>(nunits, rocfft_chunk_bytes() / per_unit));
        if (pl->batch != batch) {
            MDSP_TRY(pl->fr.reserve((size_t)(sizeof(TT) * nfft * batch)));
            MDSP_TRY(pl->spec.reserve((size_t)(sizeof(cx<R>) * nspec * batch)));
            MDSP_TRY(pl->fwd.create(CPLX ? FftKind::C2C_FWD : FftKind::R2C, sizeof(R) == 8, nfft, batch, false));
            pl->batch = batch;
        }
        const int gx = (int)std::min<int64_t>(cdiv(nfft, 256), 8);
        for (int64_t u0 = 0; u0 < nunits; u0 += batch) {
            const int64_t cnt = std::min<int64_t>(batch, nunits - u0);
            hipLaunchKernelGGL(frame_window_kernel<TT>, dim3((unsigned)cnt, gx), dim3(256), 0, st, (const TT*)s, pl->fr.as<TT>(),
                               pl->have_win ? pl->win.as<double>() : nullptr, lds_, K, hop, (int)n, (int)nfft, u0, nunits);
            MDSP_LAUNCH_CHECK();
            MDSP_TRY(pl->fwd.exec(pl->fr.p, pl->spec.p, st));
            hipLaunchKernelGGL(abs2_accum_kernel<R>, dim3((unsigned)cdiv(nspec, 256), nslices, (unsigned)nch), dim3(256), 0, st,
                               pl->spec.as<cx<R>>(), pl->partial.as<double>(), nspec, K, u0, cnt, nch);
            MDSP_LAUNCH_CHECK();
        }
    }
    return MDSP_OK;
}

// psd[ch][j] = T( m_j * fold(accumulated |X|^2 sums) ), m from r_total = K_total * fs * sum(w^2)  (periodograms.jl:751, :142-172)
template <typename R> int welch_finalize(mdsp_welch_plan_s* pl, int64_t K_total, int64_t nch, void* psd, int64_t ldp, hipStream_t st) {
    const int nout = (int)pl->nout;
    const double* kdev = (K_total == 0 && pl->frames_on_device) ? pl->kdev.as<double>() : nullptr;
    if (K_total <= 0 && !kdev) {  // fill!(out, 0); no frames (0 * r would be a division by zero in m)
        for (int64_t c = 0; c < nch; ++c) MDSP_HIP(hipMemsetAsync((R*)psd + c * ldp, 0, sizeof(R) * (size_t)nout, st));
        return MDSP_OK;
    }
    const double r_total = (double)K_total * pl->r;
    const dim3 grid((unsigned)cdiv(nout, 256), (unsigned)nch);
    const double* acc = pl->acc_ptr();
    const int ns = pl->acc_nslices, na = pl->acc_nacc, nfft = (int)pl->nfft;
#define MDSP_FIN(MODE) hipLaunchKernelGGL((welch_finalize_kernel<R, MODE>), grid, dim3(256), 0, st, acc, (R*)psd, ldp, ns, nch, na, nfft, nout, r_total, kdev, pl->r)
    switch (pl->acc_mode) {
        case 0: MDSP_FIN(0); break;
        case 1: MDSP_FIN(1); break;
        case 2: MDSP_FIN(2); break;
        case 3: MDSP_FIN(3); break;
        default: MDSP_FIN(4); break;
    }
#undef MDSP_FIN
    MDSP_LAUNCH_CHECK();
    return MDSP_OK;
}

// ---- Welch, real input, n == nfft, 50 % overlap (the default `noverlap = n >> 1` and BASELINE config 3) -----------
// With hop = N/2 the second half of frame a IS the first half of frame b, and the second half of frame b IS the first
// half of the next unit's frame a.  A slot that walks consecutive units therefore loads every sample exactly ONCE:
// E loads per thread per unit instead of 2E, and 3E/2 instead of 2E prefetch registers.
// PREF = false: no software prefetch -- the unit's two new halves are loaded at the top of its iteration and are dead before the transform
// starts, which frees 16 of the 24 sample registers; latency is covered by a third resident workgroup per CU instead (measured on the
// overlap-save kernel: +11 % over the prefetching two-workgroup form, profiles/r02b_tune.json).
template <typename R, int N, int E, int G, int TWMODE, int PADSHIFT, int MINW, int NBUF, int PERM = false, bool PREF = true>
__global__ __launch_bounds__((N / E) * G, MINW) void welch_half_kernel(SpecArgs a) {
    using C = fft::Cfg<N, E>;
    constexpr int T = C::T;
    constexpr int H = E / 2;
    static_assert(T % 64 == 0 && E % 2 == 0, "geometry");
    constexpr int NTWA = C::NTW > 0 ? C::NTW : 1;
    constexpr int REGION = fft::wg_lds_elems<C, PADSHIFT, NBUF>();
    constexpr int64_t SZ = (int64_t)sizeof(R);
    __shared__ __attribute__((aligned(16))) cx<R> lds_all[G * REGION];
    const int traw = threadIdx.x % T;
    const int t = fft::io_lane<C, PERM>(traw);     // sample / window ownership before the transform: t + T*e (== traw unless lane-permuted)
    const int tout = fft::out_lane<C, PERM>(traw);  // bin ownership after it (== t except with the wave-private exchange, PERM == 2)
    const int slot = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / T));
    cx<R>* lds = lds_all + slot * REGION;
    const cx<R>* table = static_cast<const cx<R>*>(a.table);
    const int64_t ch = blockIdx.y;

    cx<R> tw[NTWA];
    __shared__ __attribute__((aligned(16))) cx<R> twl[(TWMODE == fft::TW_LDS || TWMODE == fft::TW_HYB) ? fft::tw_lds_entries<C, TWMODE>() : 1];
    const cx<R>* twsrc = fft::wg_twiddle_setup<C, TWMODE, PERM>(tw, twl, traw, slot, table);
    R w[E];
    {
        double wd[E];
        load_window_regs<E, T>(wd, a.win, a.n, t);
#pragma unroll
        for (int e = 0; e < E; ++e) w[e] = (R)wd[e];
    }
    // Power accumulators.  Float32: (re^2, im^2) pairs fed by one packed FMA per bin, folded into the slot's Float64
    // partial row every FLUSH units (the reference accumulates in Float32 over ALL frames, periodograms.jl:757; a
    // run of <= 2 FLUSH same-sign terms per lane keeps this far tighter).  Float64: plain double accumulators.
    constexpr bool PAIR = sizeof(R) == 4;
    constexpr bool SCALAR_ACC = MINW >= 3;   // tighter register budget: one Float32 per bin (two scalar FMAs) instead of a pair
    constexpr int FLUSH = 128;
    cx<R> accp[PAIR ? E : 1];
    double acc[PAIR ? 1 : E];
    if constexpr (PAIR) {
#pragma unroll
        for (int e = 0; e < E; ++e) accp[e] = {(R)0, (R)0};
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = 0.0;
    }
    double* part = static_cast<double*>(a.out) + (((int64_t)blockIdx.x * G + slot) * a.nch + ch) * N;
    const __amdgpu_buffer_rsrc_t prs = io::make_rsrc(part, (int64_t)N * 8);
    int since = 0;
    bool first = true;
    auto flush = [&]() {   // wave-uniform control flow; per-lane state is one laundered byte offset
        int off = tout * 8;
        asm volatile("" : "+v"(off));
        if (first) {
#pragma unroll
            for (int e = 0; e < E; ++e) io::Ld<double>::store((double)accp[e].x + (double)accp[e].y, prs, off + T * e * 8);
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const double s = io::Ld<double>::load(prs, off + T * e * 8) + ((double)accp[e].x + (double)accp[e].y);
                io::Ld<double>::store(s, prs, off + T * e * 8);
            }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) accp[e] = {(R)0, (R)0};
        first = false;
        since = 0;
    };

    const R* sc = static_cast<const R*>(a.s) + ch * a.lds_;
    const int64_t nslots = (int64_t)gridDim.x * G;
    int64_t wbase = ((int64_t)blockIdx.x * G + slot) * a.run_len, wj = 0;
    auto unit_cur = [&](bool more) { return (more && wbase + wj < a.units_per_ch) ? wbase + wj : a.units_per_ch; };
    auto walk = [&]() {
        if (++wj == a.run_len) {
            wj = 0;
            wbase += nslots * a.run_len;
        }
    };
    const int64_t niter = a.niter;
    // half-frame h of unit u starts at sample u*N + h*(N/2), h = 0 (a lo), 1 (a hi = b lo), 2 (b hi)
    auto load_half = [&](R (&dst)[H], int64_t u, int h, bool on) {
        const int64_t pos = u * N + (int64_t)h * (N / 2);
        const __amdgpu_buffer_rsrc_t r = io::make_rsrc(sc + pos, on ? std::min<int64_t>(N / 2, a.len - pos) * SZ : 0);
        io::load_window<R, H, T>(dst, r, 0, t);
    };
    R lo[H], ahi[H], bhi[H];
    int64_t u = unit_cur(niter > 0);
    bool have_lo = false;   // PREF = false: lo[] already holds the first half of unit u (carried over from the previous unit)
    if constexpr (PREF) {
        const bool live = u < a.units_per_ch;
        load_half(lo, u, 0, live);
        load_half(ahi, u, 1, live);
        load_half(bhi, u, 2, live && (2 * u + 1) < a.K);
    }
    for (int64_t it = 0; it < niter; ++it) {
        walk();
        const int64_t unext = unit_cur(it + 1 < niter);
        const bool haveB = (2 * u + 1) < a.K;   // u live implied (K >= 1) -- a dead unit has all-zero halves anyway
        if constexpr (!PREF) {
            const bool live = u < a.units_per_ch;
            if (!have_lo) load_half(lo, u, 0, live);
            load_half(ahi, u, 1, live);
            load_half(bhi, u, 2, live && haveB);
        }
        cx<R> v[E];
        if (haveB) {
#if defined(__HIP_DEVICE_COMPILE__)
            if constexpr (PAIR) {   // one packed multiply per bin: (lo, ahi) w_e and (ahi, bhi) w_{e+H}, the window half picked by op_sel
#pragma unroll
                for (int e = 0; e < H; ++e) {
                    const cx<R> wp = {w[e], w[e + H]};
                    v[e] = fft::pk_mul_blo(cx<R>{lo[e], ahi[e]}, wp);
                    v[e + H] = fft::pk_mul_bhi(cx<R>{ahi[e], bhi[e]}, wp);
                }
            } else
#endif
            {
#pragma unroll
                for (int e = 0; e < H; ++e) {
                    v[e] = {lo[e] * w[e], ahi[e] * w[e]};
                    v[e + H] = {ahi[e] * w[e + H], bhi[e] * w[e + H]};
                }
            }
        } else {  // odd frame count: the last unit has no second frame
#pragma unroll
            for (int e = 0; e < H; ++e) {
                v[e] = {lo[e] * w[e], (R)0};
                v[e + H] = {ahi[e] * w[e + H], (R)0};
            }
        }
        // next unit: reuse this unit's last half when it is the next frame's first half
        if constexpr (PREF) {
            const bool nlive = unext < a.units_per_ch;
            if (unext == u + 1 && nlive) {
#pragma unroll
                for (int e = 0; e < H; ++e) lo[e] = bhi[e];
            } else {
                load_half(lo, unext, 0, nlive);
            }
            load_half(ahi, unext, 1, nlive);
            load_half(bhi, unext, 2, nlive && (2 * unext + 1) < a.K);
        } else {
            have_lo = unext == u + 1 && unext < a.units_per_ch;   // wave-uniform
            if (have_lo) {
#pragma unroll
                for (int e = 0; e < H; ++e) lo[e] = bhi[e];
            }
        }
        u = unext;
        if (!MDSP_ABLATED(a, 2)) {
            // wave-private last exchange: the workgroup barrier that protects buffer 0 sits HERE, a whole transform after its readers
            // passed it -- nobody waits at it in steady state
            if constexpr (PERM == 2) __syncthreads();
            fft::wg_fft<C, -1, TWMODE, PADSHIFT, NBUF, 0, 0, PERM>(v, traw, tw, twsrc, lds);
            if constexpr (C::P > 1 && NBUF > 1 && ((C::P - 1) % NBUF) != 0) fft::wg_sync<T>();  // NBUF == 1: wg_fft already ends every exchange with a barrier
        }
        if constexpr (PAIR) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if constexpr (SCALAR_ACC) accp[e].x = v[e].x * v[e].x + (v[e].y * v[e].y + accp[e].x);
                else accp[e] = fft::lanefma(v[e], v[e], accp[e]);
            }
            if (++since == FLUSH) flush();
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] += (double)(v[e].x * v[e].x + v[e].y * v[e].y);
        }
    }
    if constexpr (PAIR) {
        flush();
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e) part[tout + T * e] = acc[e];
    }
}

// ---- the same path, Float32, 16 elements per thread, with the instruction diet of round 3 ---------------------------------------------
// The round-2 loop issued ~420 VALU instructions per thread and unit for 335 of arithmetic.  What went:
//   * the samples of a unit live in PAIRS: Q[e] = (frame a, frame b)[t + T e] for the first half-frame, F[e] for the second -- exactly the
//     (re, im) operands of the packed first stage, so no v_mov builds them.  Frame b's first half IS frame a's second half: it is loaded a
//     second time (an L1/L2 hit, a VMEM slot instead of VALU work), predicated on the unit HAVING a frame b -- the odd last frame needs no
//     branch and no zeroing in the loop;
//   * the half-frame a unit hands to its successor (b's second half = the next a's first half) is not copied: the pair that holds it BECOMES
//     the successor's Q, and the successor packs z = b + i a instead of a + i b (|Z[k]|^2 + |Z[N-k]|^2 is symmetric in a <-> b, and k <-> N-k is
//     folded by the finalize kernel anyway), so the roles of the two pair arrays and of their halves alternate from unit to unit (loop unrolled x2);
//   * the window rides in the first butterfly stage (fft::bfly16_win: 8 packed operations less), and the butterflies' constant roots come
//     from SGPR pairs (fft_lds.h, MDSP_FFT_SGPR_CONST).
//   * DEEP: the samples of the next TWO units are in flight instead of one.  The phase profile of the one-deep form (profiles/r03b_welch_phases.txt)
//     showed a wave parked ~1000 of a unit's ~5250 clocks in front of the first butterfly stage, waiting for loads it had issued one unit
//     (~3300 clocks, 1.5 us) earlier: with two workgroups per CU and one unit (16 KiB) in flight per workgroup -- and only between issue
//     and arrival -- a CU keeps ~20 KiB outstanding, and Little's law (bytes in flight = bandwidth x latency, ~2 us under load) caps the
//     kernel near 3 TB/s whatever the arithmetic does.  Two register sets alternate; the half-frame hand-over is replaced by a third
//     (L2-hit) load so that a unit's pairs do not depend on its predecessor's registers.
template <int N, int PADSHIFT, int NBUF, bool DEEP = false>
__global__ __launch_bounds__(N / 16, 2) void welch_half3_kernel(SpecArgs a) {
    using R = float;
    constexpr int E = 16, H = 8;
    using C = fft::Cfg<N, E>;
    constexpr int T = C::T;
    static_assert(T % 64 == 0 && T >= 128, "one transform per workgroup");
    constexpr int NTWA = C::NTW > 0 ? C::NTW : 1;
    constexpr int REGION = fft::wg_lds_elems<C, PADSHIFT, NBUF>();
    __shared__ __attribute__((aligned(16))) cx<R> lds[REGION];
    const int t = threadIdx.x;
    const cx<R>* table = static_cast<const cx<R>*>(a.table);
    const int64_t ch = blockIdx.y;

    cx<R> tw[NTWA];
    fft::load_twiddles<C, R, 1, fft::TW_REG, false>(tw, t, table);
    cx<R> wp[H];   // {w[t + T e], w[t + T (e + 8)]}
    {
        double wd[E];
        load_window_regs<E, T>(wd, a.win, a.n, t);
#pragma unroll
        for (int e = 0; e < H; ++e) wp[e] = {(R)wd[e], (R)wd[e + H]};
    }
    constexpr int FLUSH = 128;
    cx<R> accp[E];
#pragma unroll
    for (int e = 0; e < E; ++e) accp[e] = {(R)0, (R)0};
    double* part = static_cast<double*>(a.out) + ((int64_t)blockIdx.x * a.nch + ch) * N;
    const __amdgpu_buffer_rsrc_t prs = io::make_rsrc(part, (int64_t)N * 8);
    int since = 0;
    bool first = true;
    auto flush = [&]() {
        int off = t * 8;
        asm volatile("" : "+v"(off));
        if (first) {
#pragma unroll
            for (int e = 0; e < E; ++e) io::Ld<double>::store((double)accp[e].x + (double)accp[e].y, prs, off + T * e * 8);
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const double s = io::Ld<double>::load(prs, off + T * e * 8) + ((double)accp[e].x + (double)accp[e].y);
                io::Ld<double>::store(s, prs, off + T * e * 8);
            }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) accp[e] = {(R)0, (R)0};
        first = false;
        since = 0;
    };

    const R* sc = static_cast<const R*>(a.s) + ch * a.lds_;
    const int64_t nslots = (int64_t)gridDim.x;
    int64_t wbase = (int64_t)blockIdx.x * a.run_len, wj = 0;
    auto unit_cur = [&](bool more) { return (more && wbase + wj < a.units_per_ch) ? wbase + wj : a.units_per_ch; };
    auto walk = [&]() {
        if (++wj == a.run_len) {
            wj = 0;
            wbase += nslots * a.run_len;
        }
    };
    // half-frame h of unit u starts at sample u N + h N/2: h = 0 (a lo), 1 (a hi = b lo), 2 (b hi); `on` false -> all zeros
    auto half_rsrc = [&](int64_t u, int h, bool on) {
        const int64_t pos = u * N + (int64_t)h * (N / 2);
        return io::make_rsrc(sc + pos, on ? std::min<int64_t>(N / 2, a.len - pos) * 4 : 0);
    };
    // component selectors: XA = true -> frame a rides in .x (z = a + i b), false -> in .y (z = b + i a)
    auto ld8 = [&](cx<R> (&dst)[H], bool to_x, const __amdgpu_buffer_rsrc_t r) {
        int off = t * 4;
        asm volatile("" : "+v"(off));
#pragma unroll
        for (int e = 0; e < H; ++e) {
            const R s = io::Ld<R>::load(r, off + T * e * 4);
            if (to_x) dst[e].x = s;
            else dst[e].y = s;
        }
    };
    // all four half-frame streams of unit u: Q = (a lo | b lo), F = (a hi | b hi); `carry`: a lo is already in Q (handed over by the predecessor)
    auto load_unit = [&](cx<R> (&Q)[H], cx<R> (&F)[H], bool xa, int64_t u, bool carry) {
        const bool live = u < a.units_per_ch, haveB = live && (2 * u + 1) < a.K;
        if (a.memprio) __builtin_amdgcn_s_setprio(3);
        if (!carry) ld8(Q, xa, half_rsrc(u, 0, live));
        ld8(Q, !xa, half_rsrc(u, 1, haveB));
        ld8(F, xa, half_rsrc(u, 1, live));
        ld8(F, !xa, half_rsrc(u, 2, haveB));
        if (a.memprio) __builtin_amdgcn_s_setprio(0);
    };
    cx<R> P1[H], P2[H];
    int64_t u = unit_cur(a.niter > 0);
    load_unit(P1, P2, true, u, false);
#ifdef MDSP_WELCH_PROF
    unsigned long long prof_sum[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, prof_last = 0, prof_units = 0, prof_t0 = 0;
#define MDSP_STAMP(k)                                                                   \
    do {                                                                                \
        unsigned long long now_;                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_) : : "memory"); \
        prof_sum[k] += now_ - prof_last;                                                \
        prof_last = now_;                                                               \
    } while (0)
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(prof_last) : : "memory");
    prof_t0 = prof_last;
#endif
    // everything of a unit after its first pass (whose results are on their way to LDS) and after the prefetch has been issued
    auto tail = [&]() {
        cx<R> v[E];
#ifdef MDSP_WELCH_PROF
        // phase timeline of one unit (debug builds: build.py --tag prof --cflags -DMDSP_WELCH_PROF): shader-clock stamps at the phase boundaries,
        // summed per wave; s_memtime needs lgkmcnt(0), which every boundary here waits for anyway (barriers, first use of the reloaded operands)
        static_assert(C::P == 3 && NBUF == 1, "the phase profile is wired for three passes and one LDS buffer");
        MDSP_STAMP(1);   // pass 0: butterflies + LDS writes done, next unit's loads issued
        fft::wg_sync<T>();
        MDSP_STAMP(2);   // barrier 1
        fft::pass_reload<C, PADSHIFT, 1, 0>(v, t, lds);
        MDSP_STAMP(3);   // operands of pass 1 back from LDS
        fft::wg_sync<T>();
        MDSP_STAMP(4);   // barrier 2
        fft::pass_compute<C, -1, 1, fft::TW_REG, PADSHIFT, 0>(v, t, tw, table, lds);
        MDSP_STAMP(5);   // pass 1 + LDS writes
        fft::wg_sync<T>();
        MDSP_STAMP(6);   // barrier 3
        fft::pass_reload<C, PADSHIFT, 2, 0>(v, t, lds);
        MDSP_STAMP(7);   // operands of pass 2
        fft::wg_sync<T>();
        MDSP_STAMP(8);   // barrier 4
        fft::pass_compute<C, -1, 2, fft::TW_REG, PADSHIFT, 0>(v, t, tw, table, lds);
#pragma unroll
        for (int e = 0; e < E; ++e) accp[e] = fft::lanefma(v[e], v[e], accp[e]);
        asm volatile("" :: "v"(accp[0].x), "v"(accp[E - 1].y));
        MDSP_STAMP(9);   // pass 2 + |Z|^2
        if (++since == FLUSH) flush();
        MDSP_STAMP(0);   // (flush); also the start of the next unit
        ++prof_units;
#else
        fft::wg_sync<T>();
        fft::pass_reload<C, PADSHIFT, 1, 0>(v, t, lds);
        if constexpr (NBUF == 1) fft::wg_sync<T>();
        fft::wg_fft<C, -1, fft::TW_REG, PADSHIFT, NBUF, 0, 1, 0>(v, t, tw, table, lds);
        if constexpr (C::P > 1 && NBUF > 1 && ((C::P - 1) % NBUF) != 0) fft::wg_sync<T>();
#pragma unroll
        for (int e = 0; e < E; ++e) accp[e] = fft::lanefma(v[e], v[e], accp[e]);
        if (++since == FLUSH) flush();
#endif
    };
    if constexpr (!DEEP) {
        auto unit = [&](cx<R> (&Q)[H], cx<R> (&F)[H], bool xa, bool more) {
            walk();
            const int64_t unext = unit_cur(more);
#ifdef MDSP_WELCH_PROF
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            MDSP_STAMP(10);   // this unit's samples have arrived
            {
                cx<R> v0[16];
                fft::bfly16_win<-1>(Q, F, wp, v0);
                asm volatile("" :: "v"(v0[0].x), "v"(v0[15].y));
                MDSP_STAMP(11);   // first pass: arithmetic
                fft::pass0_scatter<C, PADSHIFT>(v0, t, lds);
                MDSP_STAMP(12);   // first pass: LDS writes retired
            }
#else
            fft::pass0_windowed<C, PADSHIFT>(Q, F, wp, t, lds);   // consumes Q and F
#endif
            // the successor: F's frame-b-second-half component is its frame a's first half when it follows directly (roles and halves swap)
            const bool carry = unext == u + 1 && unext < a.units_per_ch;   // wave-uniform
            load_unit(F, Q, !xa, unext, carry);
            u = unext;
            tail();
        };
        for (int64_t it = 0; it < a.niter; it += 2) {   // same trip count for every slot (barriers inside)
            unit(P1, P2, true, it + 1 < a.niter);
            if (it + 1 < a.niter) unit(P2, P1, false, it + 2 < a.niter);
        }
    } else {
        // two register sets, both with frame a in .x; a set is refilled with the unit TWO ahead as soon as the first stage has consumed it
        cx<R> P3[H], P4[H];
        int64_t taken = 1;   // (P1, P2) hold the slot's first unit
        auto take = [&]() {
            walk();
            return unit_cur(taken++ < a.niter);
        };
        load_unit(P3, P4, true, take(), false);
        auto unit_deep = [&](cx<R> (&Q)[H], cx<R> (&F)[H]) {
#ifdef MDSP_WELCH_PROF
            asm volatile("s_waitcnt vmcnt(32)" ::: "memory");   // this unit's 32 loads are the older half of what is in flight
            MDSP_STAMP(10);
            {
                cx<R> v0[16];
                fft::bfly16_win<-1>(Q, F, wp, v0);
                asm volatile("" :: "v"(v0[0].x), "v"(v0[15].y));
                MDSP_STAMP(11);
                fft::pass0_scatter<C, PADSHIFT>(v0, t, lds);
                MDSP_STAMP(12);
            }
#else
            fft::pass0_windowed<C, PADSHIFT>(Q, F, wp, t, lds);
#endif
            load_unit(Q, F, true, take(), false);
            tail();
        };
        for (int64_t it = 0; it < a.niter; it += 2) {
            unit_deep(P1, P2);
            if (it + 1 < a.niter) unit_deep(P3, P4);
        }
    }
    flush();
#ifdef MDSP_WELCH_PROF
    if (a.rinv && (threadIdx.x & 63) == 0) {   // one record per wave: 10 phase sums, units, total clocks  (a.rinv: the profile buffer in these builds)
        unsigned long long* rec = (unsigned long long*)a.rinv + ((size_t)blockIdx.x * (T / 64) + threadIdx.x / 64) * 16;
        for (int k = 0; k < 13; ++k) rec[k] = prof_sum[k];
        rec[13] = prof_units;
        rec[14] = prof_last - prof_t0;
    }
#undef MDSP_STAMP
#endif
}

// ---- the same path with the samples staged in LDS by DMA (round 3) ----------------------------------------------------------------------
// Phase profile of welch_half3_kernel (profiles/r03c_welch_phases.txt): of a unit's ~5300 clocks a wave spends ~700 ISSUING its 32
// buffer_load_dword (a wave instruction moves 256 bytes, and the four waves of a workgroup reach the load burst together) and ~800 waiting for
// them, however early they were issued (two units ahead changes nothing) -- 28 % of the unit in a load path built from 4-byte-per-lane
// instructions.  Here a unit's two new half-frames (2 x 8 KiB) arrive by buffer_load_dwordx4 ... lds: four DMA instructions per wave and unit
// instead of 32 loads, no VGPRs, issued two units ahead.  Half-frame k of the channel lives in ring slot k mod 5 (5 x 8 KiB next to the 33 KiB
// exchange buffer: two workgroups per CU still fit); unit u reads half-frames 2u, 2u+1, 2u+2 as the (frame a, frame b) pairs of the first stage
// and, once every wave is past the first exchange barrier (all reads of the two oldest slots done), its waves refill those two slots with the
// half-frames of unit u+2.  Every wave issues the same number of DMA instructions per unit (a zero-size descriptor for half-frames that do not
// exist), so "all but the newest batch have landed" is the constant s_waitcnt vmcnt(4), placed in front of an exchange barrier that exists
// anyway: the pipeline adds no barrier.  One contiguous run of units per slot (the default schedule); other schedules take welch_half3_kernel.
template <int N, int PADSHIFT>
__global__ __launch_bounds__(N / 16, 2) void welch_half4_kernel(SpecArgs a) {
    using R = float;
    constexpr int E = 16, H = 8, NBUF = 1, NSLOT = 5;
    using C = fft::Cfg<N, E>;
    constexpr int T = C::T, NW = T / 64;
    constexpr int HALF = N / 2;                      // samples per half-frame
    constexpr int GRAN = HALF / 256;                 // 1 KiB DMA granules per half-frame
    static_assert(T % 64 == 0 && T >= 128 && C::P == 3 && GRAN % NW == 0, "geometry");
    constexpr int GPW = GRAN / NW;                   // granules per wave and half-frame
    constexpr int NTWA = C::NTW > 0 ? C::NTW : 1;
    constexpr int REGION = fft::wg_lds_elems<C, PADSHIFT, NBUF>();
    __shared__ __attribute__((aligned(16))) cx<R> lds[REGION];
    __shared__ __attribute__((aligned(16))) R ring[NSLOT * HALF];
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t / 64), lane = t & 63;
    const cx<R>* table = static_cast<const cx<R>*>(a.table);
    const int64_t ch = blockIdx.y;

    cx<R> tw[NTWA];
    fft::load_twiddles<C, R, 1, fft::TW_REG, false>(tw, t, table);
    cx<R> wp[H];
    {
        double wd[E];
        load_window_regs<E, T>(wd, a.win, a.n, t);
#pragma unroll
        for (int e = 0; e < H; ++e) wp[e] = {(R)wd[e], (R)wd[e + H]};
    }
    constexpr int FLUSH = 128;
    cx<R> accp[E];
#pragma unroll
    for (int e = 0; e < E; ++e) accp[e] = {(R)0, (R)0};
    double* part = static_cast<double*>(a.out) + ((int64_t)blockIdx.x * a.nch + ch) * N;
    const __amdgpu_buffer_rsrc_t prs = io::make_rsrc(part, (int64_t)N * 8);
    int since = 0;
    bool first = true;
    auto flush = [&]() {
        int off = t * 8;
        asm volatile("" : "+v"(off));
        if (first) {
#pragma unroll
            for (int e = 0; e < E; ++e) io::Ld<double>::store((double)accp[e].x + (double)accp[e].y, prs, off + T * e * 8);
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const double s = io::Ld<double>::load(prs, off + T * e * 8) + ((double)accp[e].x + (double)accp[e].y);
                io::Ld<double>::store(s, prs, off + T * e * 8);
            }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) accp[e] = {(R)0, (R)0};
        first = false;
        since = 0;
    };

    const R* sc = static_cast<const R*>(a.s) + ch * a.lds_;
    // this slot's units: u0 .. u0 + cnt - 1 (consecutive; run_len covers the slot's whole share), dead iterations behind them
    const int64_t u0 = (int64_t)blockIdx.x * a.run_len;
    const int64_t uend = std::min<int64_t>(u0 + a.run_len, a.units_per_ch);           // one past this slot's last live unit
    // half-frame k (of the channel) is wanted by this slot iff it belongs to a frame of one of its units: k in [2 u0, klast]
    const int64_t klast = uend > u0 ? 2 * (uend - 1) + (((2 * (uend - 1) + 1) < a.K) ? 2 : 1) : -1;
    const unsigned ring0 = io::lds_byte_address(ring);
    // this wave's share of half-frame k -> ring slot k mod NSLOT (always GPW instructions: an unwanted half-frame moves nothing)
    auto dma_half = [&](int64_t k, int slot) {
        const bool want = k >= 2 * u0 && k <= klast;
        const io::dma_i4 r = io::dma_rsrc(sc + k * HALF, want ? (long long)HALF * 4 : 0);
#pragma unroll
        for (int g = 0; g < GPW; ++g) {
            const int gr = wave * GPW + g;
            io::dma256(r, ring0 + (unsigned)slot * (unsigned)(HALF * 4) + (unsigned)gr * 1024u, gr * 1024 + lane * 16);
        }
    };
    // prologue: unit u0 (three half-frames) and unit u0 + 1 (two more) -- slots 0 .. 4; then only the newest batch (GPW x 2) may be outstanding
    dma_half(2 * u0, 0); dma_half(2 * u0 + 1, 1); dma_half(2 * u0 + 2, 2);
    dma_half(2 * u0 + 3, 3); dma_half(2 * u0 + 4, 4);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * GPW) : "memory");
    __syncthreads();
    int s0 = 0;                                      // ring slot of half-frame 2 u (wave-uniform)
    int64_t u = u0;
    for (int64_t it = 0; it < a.niter; ++it, ++u) {  // same trip count for every slot (barriers inside)
        const bool live = u < uend, haveB = live && (2 * u + 1) < a.K;
        const int s1 = s0 + 1 >= NSLOT ? s0 + 1 - NSLOT : s0 + 1, s2 = s0 + 2 >= NSLOT ? s0 + 2 - NSLOT : s0 + 2;
        cx<R> Q[H], F[H];                            // Q = (a lo | b lo), F = (a hi | b hi)
        if (live) {
            const R* h0 = ring + s0 * HALF + t;
            const R* h1 = ring + s1 * HALF + t;
            const R* h2 = ring + s2 * HALF + t;
#pragma unroll
            for (int e = 0; e < H; ++e) {
                const R m = h1[T * e];
                Q[e] = {h0[T * e], m};
                F[e] = {m, h2[T * e]};
            }
            if (!haveB) {                            // the odd last frame of the channel: no frame b
#pragma unroll
                for (int e = 0; e < H; ++e) {
                    Q[e].y = (R)0;
                    F[e].y = (R)0;
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < H; ++e) Q[e] = F[e] = {(R)0, (R)0};
        }
        fft::pass0_windowed<C, PADSHIFT>(Q, F, wp, t, lds);
        cx<R> v[E];
        fft::wg_sync<T>();                           // exchange barrier 1: every wave has read its half-frames
        // refill the two oldest slots with the half-frames of unit u + 2
        dma_half(2 * u + 5, s0);
        dma_half(2 * u + 6, s1);
        fft::pass_reload<C, PADSHIFT, 1, 0>(v, t, lds);
        fft::wg_sync<T>();
        fft::pass_compute<C, -1, 1, fft::TW_REG, PADSHIFT, 0>(v, t, tw, table, lds);
        fft::wg_sync<T>();
        fft::pass_reload<C, PADSHIFT, 2, 0>(v, t, lds);
        // all but the batch just issued has landed (this wave's share); behind the barrier the next unit's half-frames are complete
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * GPW) : "memory");
        fft::wg_sync<T>();
        fft::pass_compute<C, -1, 2, fft::TW_REG, PADSHIFT, 0>(v, t, tw, table, lds);
#pragma unroll
        for (int e = 0; e < E; ++e) accp[e] = fft::lanefma(v[e], v[e], accp[e]);
        if (++since == FLUSH) flush();
        s0 = s2;
    }
    flush();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing of this workgroup may still be writing its LDS when it exits
}

template <int N, int PADSHIFT> int welch_run_half4(mdsp_welch_plan_s* pl, SpecArgs& a, hipStream_t st, int* nslices) {
    auto kern = welch_half4_kernel<N, PADSHIFT>;
    constexpr int threads = N / 16;
    int grid = 1;
    MDSP_TRY(grid_for(kern, threads, a.units_per_ch, a.nch, &grid));
    MDSP_TRY(pl->partial.reserve(sizeof(double) * (size_t)grid * (size_t)a.nch * N));
    a.out = pl->partial.p;
    set_schedule(a, a.units_per_ch, (int64_t)grid);
    if (a.run_len * (int64_t)grid < a.units_per_ch) return -1000;   // several runs per slot (MDSP_RUNS_PER_SLOT): the caller takes welch_half3_kernel
    hipLaunchKernelGGL(kern, dim3(grid, (unsigned)a.nch), dim3(threads), 0, st, a);
    MDSP_LAUNCH_CHECK();
    *nslices = grid;
    return MDSP_OK;
}

template <int N, int PADSHIFT, int NBUF, bool DEEP = false> int welch_run_half3(mdsp_welch_plan_s* pl, SpecArgs& a, hipStream_t st, int* nslices) {
    auto kern = welch_half3_kernel<N, PADSHIFT, NBUF, DEEP>;
    constexpr int threads = N / 16;
    int grid = 1;
    MDSP_TRY(grid_for(kern, threads, a.units_per_ch, a.nch, &grid));
    MDSP_TRY(pl->partial.reserve(sizeof(double) * (size_t)grid * (size_t)a.nch * N));
    a.out = pl->partial.p;
    set_schedule(a, a.units_per_ch, (int64_t)grid);
#ifdef MDSP_WELCH_PROF
    static DevBuf profbuf;
    const size_t nrec = (size_t)grid * (threads / 64);
    MDSP_TRY(profbuf.reserve(nrec * 16 * 8));
    a.rinv = a.nch == 1 ? static_cast<const double*>(profbuf.p) : nullptr;
    hipEvent_t e0, e1;
    MDSP_HIP(hipEventCreate(&e0)); MDSP_HIP(hipEventCreate(&e1));
    MDSP_HIP(hipEventRecord(e0, st));
#endif
    hipLaunchKernelGGL(kern, dim3(grid, (unsigned)a.nch), dim3(threads), 0, st, a);
    MDSP_LAUNCH_CHECK();
#ifdef MDSP_WELCH_PROF
    MDSP_HIP(hipEventRecord(e1, st));
    MDSP_HIP(hipStreamSynchronize(st));
    float ms = 0;
    MDSP_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (a.rinv) {   // debug build only: one line per launch
        std::vector<unsigned long long> h(nrec * 16);
        MDSP_HIP(hipMemcpy(h.data(), profbuf.p, nrec * 16 * 8, hipMemcpyDeviceToHost));
        double sum[13] = {0}, units = 0, tot = 0, totmax = 0;
        for (size_t r = 0; r < nrec; ++r) {
            for (int k = 0; k < 13; ++k) sum[k] += (double)h[r * 16 + k];
            units += (double)h[r * 16 + 13];
            tot += (double)h[r * 16 + 14];
            totmax = std::max(totmax, (double)h[r * 16 + 14]);
        }
        static const char* nm[13] = {"flush+loop", "load-issue", "barrier1", "reload1", "barrier2", "pass1+ldsW", "barrier3", "reload2", "barrier4", "pass2+acc",
                                     "memwait", "pass0-valu", "pass0-ldsW"};
        fprintf(stderr, "WELCHPROF grid %d x %d waves, %.4f ms, %.0f units/wave, clocks/unit %.0f (max-wave total %.0f clocks -> %.3f GHz):", grid, threads / 64, ms,
                units / nrec, tot / units, totmax, totmax / (ms * 1e6));
        for (int k : {10, 11, 12, 1, 2, 3, 4, 5, 6, 7, 8, 9, 0}) fprintf(stderr, " %s %.0f", nm[k], sum[k] / units);
        fprintf(stderr, "\n");
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
#endif
    *nslices = grid;
    return MDSP_OK;
}

#include "welch_w64.h"   // one wavefront per transform (round 4): welch_w64_kernel, welch_run_w64

template <typename R, int N, int E, int G, int TWMODE, int PADSHIFT, int MINW, int NBUF, int PERM = false, bool PREF = true>
int welch_run_half(mdsp_welch_plan_s* pl, SpecArgs& a, hipStream_t st, int* nslices) {
    auto kern = welch_half_kernel<R, N, E, G, TWMODE, PADSHIFT, MINW, NBUF, PERM, PREF>;
    constexpr int threads = (N / E) * G;
    int grid = 1;
    MDSP_TRY(grid_for(kern, threads, cdiv(a.units_per_ch, G), a.nch, &grid));
    MDSP_TRY(pl->partial.reserve(sizeof(double) * (size_t)grid * G * (size_t)a.nch * N));
    a.out = pl->partial.p;
    set_schedule(a, a.units_per_ch, (int64_t)grid * G);
    hipLaunchKernelGGL(kern, dim3(grid, (unsigned)a.nch), dim3(threads), 0, st, a);
    MDSP_LAUNCH_CHECK();
    *nslices = grid * G;
    return MDSP_OK;
}

template <typename R, int N, int E, int G, int TWMODE, int PADSHIFT, bool CPLX, int MINW, int NBUF, bool PREFETCH, bool WIN64, int SHIFT = 0>
int welch_run_variant(mdsp_welch_plan_s* pl, SpecArgs& a, hipStream_t st, int* nslices) {
    auto kern = welch_fused_kernel<R, N, E, G, TWMODE, PADSHIFT, CPLX, MINW, NBUF, PREFETCH, WIN64, SHIFT>;
    constexpr int threads = (N / E) * G;
    int grid = 1;
    MDSP_TRY(grid_for(kern, threads, cdiv(a.units_per_ch, G), a.nch, &grid));
    MDSP_TRY(pl->partial.reserve(sizeof(double) * (size_t)grid * G * (size_t)a.nch * N));
    a.out = pl->partial.p;
    set_schedule(a, a.units_per_ch, (int64_t)grid * G);
    hipLaunchKernelGGL(kern, dim3(grid, (unsigned)a.nch), dim3(threads), 0, st, a);
    MDSP_LAUNCH_CHECK();
    *nslices = grid * G;
    return MDSP_OK;
}

template <typename R, int N, bool CPLX>
int welch_launch_n(mdsp_welch_plan_s* pl, SpecArgs& a, hipStream_t st) {
    using Gm = Geo<R, N>;
    int nslices = 0, rc = MDSP_OK;
    const bool half_ok = !CPLX && a.n == N && 2 * a.hop == N && !MDSP_DBG(welch_nohalf);
    if constexpr (!CPLX && N >= 256) {
        if (half_ok && !(N == 4096 && sizeof(R) == 4 && pl->variant >= 1 && pl->variant <= 9)) {
            constexpr int EH = (N >= 2048 && sizeof(R) == 4) ? 16 : Gm::E;   // Float32, nfft >= 1024: 16 elements per thread (Geo does 1024)
            constexpr int GH = slots_per_workgroup(N / EH);
            constexpr int NB = (N / EH) <= 64 ? 1 : 2;
            constexpr int NBH = (EH == 16) ? 1 : NB;
            bool done = false;
            if constexpr (N == 4096 && sizeof(R) == 4) {  // tuning alternatives of the headline shape (MDSP_WELCH_VARIANT)
                done = true;
                //                                              R  N  E   G  TW PAD MINW NBUF
                // product builds: 18 (the round-2 form: identity lanes, pad 5), 30 (welch_half3_kernel, the default below the hand-allocated kernel's stream
                // length) and 43 (mdsp_welch_w64c_asm, the default); every other variant lost its A/B (HISTORY.md section 4.3) and is built with -DMDSP_DEBUG_KNOBS only
                if (pl->variant == 18) rc = welch_run_half<R, N, EH, GH, 1, 5, 2, 1, false>(pl, a, st, &nslices);   // identity lanes, pad 5
                else if (pl->variant == 30) rc = welch_run_half3<N, 5, 1>(pl, a, st, &nslices);   // round 3: paired samples, branch-free, window in the first stage
                else if (pl->variant == 43) {   // ... the same, hand-allocated, shared half-frame carried (csrc/welch_w64c_asm.s): reduces into pl->reduced itself
                    bool handled = false;
                    rc = w64::welch_run_w64asm(pl, a, st, &handled);
                    if (rc == MDSP_OK && handled) goto reduced_done;
                }
#ifdef MDSP_DEBUG_KNOBS
                else if (pl->variant == 10) rc = welch_run_half<R, N, EH, GH, 1, 4, 2, 1>(pl, a, st, &nslices);        // identity lanes, pad 4 (previous default)
                else if (pl->variant == 11) rc = welch_run_half<R, N, EH, GH, 1, 4, 2, 2>(pl, a, st, &nslices);
                else if (pl->variant == 19) rc = welch_run_half<R, N, EH, GH, 1, 5, 2, 2, true>(pl, a, st, &nslices);    // permuted, two LDS buffers
                else if (pl->variant == 20) rc = welch_run_half<R, N, EH, GH, 1, 5, 2, 1, true>(pl, a, st, &nslices);    // permuted lanes, pad 5
                else if (pl->variant == 21) rc = welch_run_half<R, N, EH, GH, 1, 5, 2, 2, false>(pl, a, st, &nslices);   // identity lanes, pad 5, two LDS buffers
                else if (pl->variant == 22) rc = welch_run_half<R, N, EH, GH, 1, 4, 2, 2, 2>(pl, a, st, &nslices);       // wave-private last exchange (one real barrier per transform)
                else if (pl->variant == 23) rc = welch_run_half<R, N, EH, GH, 1, 5, 2, 1, 0, false>(pl, a, st, &nslices);   // registers, no prefetch (still two workgroups)
                else if (pl->variant == 24) rc = welch_run_half<R, N, EH, GH, 3, 5, 3, 1, 0, false>(pl, a, st, &nslices);   // hybrid twiddles, scalar accumulators, <= 168 VGPRs
                else if (pl->variant == 25) rc = welch_run_half<R, N, EH, GH, 3, 4, 3, 1, 0, false>(pl, a, st, &nslices);   // same, pad 4
                else if (pl->variant == 26) rc = welch_run_half<R, N, EH, GH, 2, 5, 3, 1, 0, false>(pl, a, st, &nslices);   // LDS twiddles
                else if (pl->variant == 27) rc = welch_run_half<R, N, EH, GH, 3, 5, 3, 1, 0, true>(pl, a, st, &nslices);    // hybrid + prefetch (= 16 with pad 5)
                else if (pl->variant == 12) rc = welch_run_half<R, N, EH, GH, 2, 4, 2, 1>(pl, a, st, &nslices);
                else if (pl->variant == 13) rc = welch_run_half<R, N, 8, 1, 1, 4, 2, 2>(pl, a, st, &nslices);
                else if (pl->variant == 14) rc = welch_run_half<R, N, 8, 1, 1, 4, 4, 2>(pl, a, st, &nslices);
                else if (pl->variant == 15) rc = welch_run_half<R, N, 8, 1, 1, 4, 4, 1>(pl, a, st, &nslices);
                else if (pl->variant == 16) rc = welch_run_half<R, N, EH, GH, 3, 4, 3, 1>(pl, a, st, &nslices);
                else if (pl->variant == 17) rc = welch_run_half<R, N, EH, GH, 3, 4, 2, 1>(pl, a, st, &nslices);
                else if (pl->variant == 31) rc = welch_run_half3<N, 5, 2>(pl, a, st, &nslices);   // ... with two LDS buffers (one barrier per exchange)
                else if (pl->variant == 32) rc = welch_run_half3<N, 4, 1>(pl, a, st, &nslices);   // ... pad 4
                else if (pl->variant == 33) rc = welch_run_half3<N, 5, 1, true>(pl, a, st, &nslices);   // ... two units in flight (two register sets)
                else if (pl->variant == 34) rc = welch_run_half3<N, 4, 1, true>(pl, a, st, &nslices);
                else if (pl->variant == 41) rc = w64::welch_run_w64b(pl, a, st, &nslices);  // ... two waves per SIMD: two-level twiddles, direct loads
                else if (pl->variant == 40) rc = w64::welch_run_w64(pl, a, st, &nslices);   // round 4: one wavefront per transform, 64 x 64, one exchange
                else if (pl->variant == 35 || pl->variant == 36) {   // half-frames staged in LDS by DMA, two units ahead (pad 5 / pad 4)
                    rc = pl->variant == 35 ? welch_run_half4<N, 5>(pl, a, st, &nslices) : welch_run_half4<N, 4>(pl, a, st, &nslices);
                    if (rc == -1000) rc = welch_run_half3<N, 5, 1>(pl, a, st, &nslices);
                }
#endif
                else done = false;
            }
            if (!done) {
                if constexpr (N == 4096 && sizeof(R) == 4) {
                    // Round 3 default: welch_half3_kernel (paired samples, branch-free, window in the first stage: 17 % fewer vector instructions).
                    // The kernel runs AT the 1400 W package power cap (profiles/r03e_power_probe.json), so what the diet buys is energy: 1.4-2.3 %
                    // less time in sustained runs, nothing measurable in short bursts (profiles/r03a_tune_new.json).  MDSP_WELCH_VARIANT=18 is the
                    // round-2 kernel (identity lanes, pad 5); several runs per slot (MDSP_RUNS_PER_SLOT) work in both.
                    // Round 4: streams long enough to give every wave of the chip a unit take the hand-allocated one-wavefront-per-transform kernel
                    // (csrc/welch_w64c_asm.s = variant 43, the form that carries the shared half-frame: 1.00 against 1.17 ms on an all-zero stream,
                    // 1.22 against 1.34 under the power cap, HBM traffic 1.01 x algorithmic; profiles/r04_welch_carry_power.json; variant 42 is the
                    // first form, which re-reads that half-frame); MDSP_WELCH_VARIANT=30 keeps welch_half3_kernel.
                    if (a.K / 2 >= (int64_t)device_cu_count() * 8 && tunables().runs_per_slot == 1) {
                        bool handled = false;
                        rc = w64::welch_run_w64asm(pl, a, st, &handled);
                        if (rc == MDSP_OK && handled) goto reduced_done;
                        if (rc != MDSP_OK) return rc;
                    }
                    rc = welch_run_half3<N, 5, 1>(pl, a, st, &nslices);
                }
                else
                    rc = welch_run_half<R, N, EH, GH, Gm::TWREG, pad_default<R>(), 2, NBH>(pl, a, st, &nslices);
            }
            goto finalize;
        }
    }
    if constexpr (N == 4096 && !CPLX && sizeof(R) == 4) {
#ifdef MDSP_DEBUG_KNOBS
        switch (pl->variant) {  // tuning alternatives (MDSP_WELCH_VARIANT), built for the headline shape only
            //                                  R  N   E  G TW PAD CPLX MINW NBUF PREF WIN64      (TW: 0 global, 1 regs, 2 LDS)
            case 1: rc = welch_run_variant<R, N, 16, 1, 1, 4, CPLX, 2, 2, true, false>(pl, a, st, &nslices); break;
            case 2: rc = welch_run_variant<R, N, 16, 1, 1, 4, CPLX, 2, 2, false, false>(pl, a, st, &nslices); break;
            case 3: rc = welch_run_variant<R, N, 16, 1, 1, 4, CPLX, 2, 1, true, false>(pl, a, st, &nslices); break;
            case 4: rc = welch_run_variant<R, N, 16, 1, 1, 4, CPLX, 2, 1, false, false>(pl, a, st, &nslices); break;
            case 5: rc = welch_run_variant<R, N, 16, 1, 2, 4, CPLX, 2, 1, true, false>(pl, a, st, &nslices); break;
            case 6: rc = welch_run_variant<R, N, 16, 1, 2, 4, CPLX, 2, 1, false, false>(pl, a, st, &nslices); break;
            case 7: rc = welch_run_variant<R, N, 16, 1, 1, 5, CPLX, 2, 2, false, false>(pl, a, st, &nslices); break;
            case 8: rc = welch_run_variant<R, N, 8, 1, 1, 4, CPLX, 4, 2, false, false>(pl, a, st, &nslices); break;
            case 9: rc = welch_run_variant<R, N, 8, 1, 1, 4, CPLX, 2, 2, false, false>(pl, a, st, &nslices); break;
            case 10: rc = welch_run_variant<R, N, 8, 1, 1, 4, CPLX, 2, 2, true, false>(pl, a, st, &nslices); break;
            default: rc = welch_run_variant<R, N, 16, 1, 1, 4, CPLX, 2, 1, true, false>(pl, a, st, &nslices); break;
        }
#else
        rc = welch_run_variant<R, N, 16, 1, 1, 4, CPLX, 2, 1, true, false>(pl, a, st, &nslices);
#endif
    } else {
        bool done = false;
        if constexpr (CPLX && sizeof(R) == 4 && Gm::E > 4) {   // complex Float32 frames advancing by whole elements: overlap stays in registers
            constexpr int T = N / Gm::E;
            const int shift = (!MDSP_DBG(stft_noshift) && a.n == N && a.hop % T == 0 && a.hop / T < Gm::E) ? (int)(a.hop / T) : 0;
            done = shift == 1 || shift == 2 || shift == 4;
            if (shift == 1) rc = welch_run_variant<R, N, Gm::E, Gm::G, Gm::TWREG, pad_default<R>(), CPLX, 2, Gm::NBUF, true, false, 1>(pl, a, st, &nslices);
            else if (shift == 2) rc = welch_run_variant<R, N, Gm::E, Gm::G, Gm::TWREG, pad_default<R>(), CPLX, 2, Gm::NBUF, true, false, 2>(pl, a, st, &nslices);
            else if (shift == 4) rc = welch_run_variant<R, N, Gm::E, Gm::G, Gm::TWREG, pad_default<R>(), CPLX, 2, Gm::NBUF, true, false, 4>(pl, a, st, &nslices);
        }
        if (!done) rc = welch_run_variant<R, N, Gm::E, Gm::G, Gm::TWREG, pad_default<R>(), CPLX, 2, Gm::NBUF, true, sizeof(R) == 8>(pl, a, st, &nslices);
    }
finalize:
    if (rc != MDSP_OK) return rc;
    // fold the slots' partial rows into the plan's Float64 accumulator (added to what earlier slices of the stream left there)
    MDSP_TRY(pl->reduced.reserve(sizeof(double) * (size_t)a.nch * N));
    MDSP_TRY(reduce_partials(pl, pl->partial.as<double>(), pl->reduced.as<double>(), nslices, a.nch, N, pl->acc_fresh ? 0 : 1, st));
reduced_done:
    if (rc != MDSP_OK) return rc;
    pl->acc_fresh = false;
    pl->acc_nslices = 1;
    pl->acc_nacc = N;
    pl->acc_mode = CPLX ? 1 : (pl->onesided ? 3 : 4);
    return MDSP_OK;
}

template <typename R, bool CPLX>
int welch_accumulate_fused(mdsp_welch_plan_s* pl, const void* s, int64_t len, int64_t nch, int64_t lds_, hipStream_t st) {
    const int64_t K = mdsp_frame_count(len, pl->n, pl->noverlap);
    if (K == 0) {
        if (pl->acc_fresh) {   // nothing to add; make the accumulator exist (zeros) so that a later finalize / all-reduce is defined
            MDSP_TRY(pl->reduced.reserve(sizeof(double) * (size_t)nch * (size_t)pl->nfft));
            MDSP_HIP(hipMemsetAsync(pl->reduced.p, 0, sizeof(double) * (size_t)nch * (size_t)pl->nfft, st));
            pl->acc_fresh = false;
            pl->acc_nslices = 1;
            pl->acc_nacc = (int)pl->nfft;
            pl->acc_mode = CPLX ? 1 : (pl->onesided ? 3 : 4);
        }
        return MDSP_OK;
    }
    if (use_gx(pl->dtype, pl->nfft, true, 0)) {   // run-time-schedule kernel (spectral_gx.h): partial rows per group of workgroups, same Float64 accumulator protocol
        GxArgs g{};
        g.s = s; g.lds_ = lds_; g.K = K; g.hop = pl->n - pl->noverlap; g.nch = nch;
        g.n = (int)pl->n; g.nfft = (int)pl->nfft; g.nout = (int)pl->nout; g.onesided = pl->onesided; g.r = pl->r;
        int64_t ngroups = 0;
        // R0 x a row size with a COMPILE-TIME schedule (16384 = 2 x 8192, 12500 = 5 x 2500, 20000 = 4 x 5000 ...): the same decomposition on the kernels of
        // spectral_gen.h (spectral_ctcols.hip) -- about half the vector instructions per point of the run-time schedule
        if (tunables().gx != 4 && tunables().gx != 5 && ctbig_ok(pl->dtype, pl->nfft))   // one workgroup, compile-time schedule (spectral_ctbig.hip): 8400 .. 12500 points
            MDSP_TRY(ctbig_welch(pl->ctcols, pl->dtype, s, lds_, K, pl->n - pl->noverlap, nch, (int)pl->n, pl->nfft, pl->have_win ? pl->win.as<double>() : nullptr, st, &ngroups,
                                 &pl->partial));
        else if (const int r0rows = ctrows_r0(pl->dtype, pl->nfft); r0rows > 0) {   // nfft = R0 x S in two kernels (spectral_ctrows.hip): straight into the accumulator
            MDSP_TRY(pl->reduced.reserve(sizeof(double) * (size_t)nch * (size_t)pl->nfft));
            MDSP_TRY(ctrows_welch(pl->ctrows, pl->dtype, r0rows, s, lds_, K, pl->n - pl->noverlap, nch, (int)pl->n, pl->nfft, pl->have_win ? pl->win.as<double>() : nullptr,
                                  pl->reduced.as<double>(), pl->acc_fresh, st));
            pl->acc_fresh = false;
            pl->acc_nslices = 1;
            pl->acc_nacc = (int)pl->nfft;
            pl->acc_mode = CPLX ? 1 : (pl->onesided ? 3 : 4);
            return MDSP_OK;
        } else if (tunables().gx != 4 && ctcols_split(pl->dtype, pl->nfft) > 0)
            MDSP_TRY(ctcols_welch(pl->ctcols, pl->dtype, s, lds_, K, pl->n - pl->noverlap, nch, (int)pl->n, pl->nfft, pl->have_win ? pl->win.as<double>() : nullptr, st, &ngroups,
                                  &pl->partial));
        else
        MDSP_TRY((gx_launch<R, CPLX, 0>(pl->gx, g, pl->have_win ? pl->win.as<double>() : nullptr, pl->dtype, st, &ngroups, &pl->partial)));
        const int N = (int)pl->nfft;
        MDSP_TRY(pl->reduced.reserve(sizeof(double) * (size_t)nch * (size_t)N));
        MDSP_TRY(reduce_partials(pl, pl->partial.as<double>(), pl->reduced.as<double>(), (int)ngroups, nch, N, pl->acc_fresh ? 0 : 1, st));
        pl->acc_fresh = false;
        pl->acc_nslices = 1;
        pl->acc_nacc = N;
        pl->acc_mode = CPLX ? 1 : (pl->onesided ? 3 : 4);
        return MDSP_OK;
    }
    if (use_big(pl->dtype, pl->nfft, true, 0)) {   // nfft above the one-workgroup sizes: channel by channel through the multi-pass engine, same accumulator protocol
        using TT = std::conditional_t<CPLX, cx<R>, R>;
        const int64_t N = pl->nfft;
        MDSP_TRY(pl->reduced.reserve(sizeof(double) * (size_t)nch * (size_t)N));
        for (int64_t c = 0; c < nch; ++c)
            MDSP_TRY(big::welch(pl->big, pl->dtype, pl->n, N, static_cast<const TT*>(s) + c * lds_, K, pl->n - pl->noverlap,
                                pl->have_win ? pl->win.as<double>() : nullptr, pl->reduced.as<double>() + c * N, pl->acc_fresh, st));
        pl->acc_fresh = false;
        pl->acc_nslices = 1;
        pl->acc_nacc = (int)N;
        pl->acc_mode = CPLX ? 1 : (pl->onesided ? 3 : 4);
        return MDSP_OK;
    }
    SpecArgs a{};
    a.s = s;
    a.table = pl->table.p;
    a.win = pl->have_win ? pl->win.as<double>() : nullptr;
    a.len = len;
    a.lds_ = lds_;
    a.K = K;
    a.hop = pl->n - pl->noverlap;
    a.units_per_ch = CPLX ? K : cdiv(K, 2);
    a.nch = nch;
    a.n = (int)pl->n;
    a.nout = (int)pl->nout;
    a.onesided = pl->onesided;
    a.r = pl->r;
    if (!fused_size_ok(pl->dtype, pl->nfft)) {   // mixed-radix sizes: everything through LDS (spectral_gen.h), same Float64 accumulator protocol
        GenArgs g{};
        g.s = s; g.roots = pl->table.p; g.win = a.win;
        g.len = len; g.lds_ = lds_; g.K = K; g.hop = a.hop; g.nch = nch; g.units_per_ch = a.units_per_ch;
        g.n = a.n; g.N = (int)pl->nfft; g.nout = a.nout; g.onesided = a.onesided; g.r = a.r;
        if (lean_window_needed(pl->nfft, sizeof(R) == 8)) {
            if (!pl->winr_ready) MDSP_TRY(lean_window<R>(pl->winr, a.win, a.n, pl->nfft, st));
            pl->winr_ready = true;
            g.winr = pl->winr.p;
        }
        int64_t nslots = 0;
        MDSP_TRY((gen_launch<R, CPLX, 0>(g, nch, st, &nslots, &pl->partial)));
        const int N = (int)pl->nfft;
        MDSP_TRY(pl->reduced.reserve(sizeof(double) * (size_t)nch * (size_t)N));
        MDSP_TRY(reduce_partials(pl, pl->partial.as<double>(), pl->reduced.as<double>(), (int)nslots, nch, N, pl->acc_fresh ? 0 : 1, st));
        pl->acc_fresh = false;
        pl->acc_nslices = 1;
        pl->acc_nacc = N;
        pl->acc_mode = CPLX ? 1 : (pl->onesided ? 3 : 4);
        return MDSP_OK;
    }
    switch (pl->nfft) {
        case 256: return welch_launch_n<R, 256, CPLX>(pl, a, st);
        case 512: return welch_launch_n<R, 512, CPLX>(pl, a, st);
        case 1024: return welch_launch_n<R, 1024, CPLX>(pl, a, st);
        case 2048: return welch_launch_n<R, 2048, CPLX>(pl, a, st);
        case 4096: return welch_launch_n<R, 4096, CPLX>(pl, a, st);
        case 8192:
            if constexpr (sizeof(R) == 4) return welch_launch_n<R, 8192, CPLX>(pl, a, st);
        default: break;
    }
    MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "fused Welch does not support nfft=%lld", (long long)pl->nfft);
}

}  // namespace

namespace mdsp {
int gx_run(int id, const GxArgs& a, unsigned grid_x, unsigned grid_y, int threads, size_t lds_bytes, hipStream_t st) {
    switch (id) {
#define MDSP_GX_CASE(ID) case ID: return gx_run_##ID(a, grid_x, grid_y, threads, lds_bytes, st);
        MDSP_GX_CASE(0) MDSP_GX_CASE(1) MDSP_GX_CASE(2) MDSP_GX_CASE(3) MDSP_GX_CASE(5) MDSP_GX_CASE(6) MDSP_GX_CASE(7) MDSP_GX_CASE(8)
        MDSP_GX_CASE(10) MDSP_GX_CASE(11) MDSP_GX_CASE(12) MDSP_GX_CASE(14) MDSP_GX_CASE(15) MDSP_GX_CASE(16) MDSP_GX_CASE(17) MDSP_GX_CASE(19)
#undef MDSP_GX_CASE
        default: MDSP_FAIL(MDSP_ERR_ASSERTION, "gx kernel id %d", id);
    }
}
// Welch in the rows form of the multi-pass engine (bigfft.hip run_welch_rows): `nch` rows of S = pl->nfft complex points, K frames `hop` elements apart
// (the rows of K transforms of nch x S points, column pass done) -> pl->reduced[row][bin] (+)= |FFT_S(frame)|^2 on the single-workgroup kernels.
// pl: a complex, two-sided, window-free FUSED plan of S points (n = nfft = S).
int welch_rows_accumulate(mdsp_welch_plan_s* pl, const void* work, int64_t K, int64_t hop, int64_t nch, hipStream_t st) {
    if (K <= 0 || nch <= 0) return MDSP_OK;
    SpecArgs a{};
    a.s = work;
    a.table = pl->table.p;
    a.win = nullptr;
    a.len = (K - 1) * hop + pl->n;
    a.lds_ = pl->n;
    a.K = K;
    a.hop = hop;
    a.units_per_ch = K;
    a.nch = nch;
    a.n = (int)pl->n;
    a.nout = (int)pl->nout;
    a.onesided = 0;
    a.r = 1.0;
    if (pl->dtype == MDSP_C32 && pl->nfft == 8192) return welch_launch_n<float, 8192, true>(pl, a, st);
    if (pl->dtype == MDSP_C64 && pl->nfft == 4096) return welch_launch_n<double, 4096, true>(pl, a, st);
    MDSP_FAIL(MDSP_ERR_ASSERTION, "rows of %lld points of dtype %d", (long long)pl->nfft, pl->dtype);
}
}  // namespace mdsp

extern "C" {

int mdsp_welch_plan_create(mdsp_welch_plan* plan, int64_t n, int64_t noverlap, int64_t nfft, const double* window_host, double r, int onesided,
                           int dtype, int engine) {
    if (!plan) MDSP_FAIL(MDSP_ERR_ARGUMENT, "plan is NULL");
    *plan = nullptr;
    if (!dtype_valid(dtype)) MDSP_FAIL(MDSP_ERR_ARGUMENT, "invalid dtype %d", dtype);
    if (onesided && dtype_is_complex(dtype)) MDSP_FAIL(MDSP_ERR_ARGUMENT, "cannot compute one-sided FFT of a complex signal");
    if (!(nfft >= n)) MDSP_FAIL(MDSP_ERR_DOMAIN, "nfft must be >= n (nfft=%lld, n=%lld)", (long long)nfft, (long long)n);
    MDSP_TRY(check_split(n, noverlap, nfft));
    if (!(r > 0)) MDSP_FAIL(MDSP_ERR_ARGUMENT, "normalisation r must be positive");
    int eng;
    MDSP_TRY(resolve_engine(engine, dtype, nfft, &eng, 0));
    auto pl = new mdsp_welch_plan_s();
    pl->dtype = dtype;
    pl->engine = eng;
    pl->onesided = onesided ? 1 : 0;
    pl->n = n;
    pl->noverlap = noverlap;
    pl->nfft = nfft;
    pl->nout = onesided ? nfft / 2 + 1 : nfft;
    pl->r = r;
    pl->variant = tunables().welch_variant;
    int st = MDSP_OK;
    if (window_host) {
        pl->have_win = true;
        st = pl->win.reserve(sizeof(double) * (size_t)n);
        if (st == MDSP_OK && hipMemcpy(pl->win.p, window_host, sizeof(double) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess)
            st = set_error(MDSP_ERR_DEVICE, "window upload failed");
    }
    if (st == MDSP_OK && eng == MDSP_ENGINE_FUSED && !use_big(dtype, nfft, true, 0) && !use_gx(dtype, nfft, true, 0))   // (the multi-pass engine and the run-time-schedule kernel build their own, much shorter, tables)
        st = dtype_is_double(dtype) ? upload_roots<double>(pl->table, nfft) : upload_roots<float>(pl->table, nfft);
    if (st != MDSP_OK) {
        delete pl;
        return st;
    }
    *plan = pl;
    return MDSP_OK;
}

int mdsp_welch_plan_destroy(mdsp_welch_plan plan) {
    delete plan;
    return MDSP_OK;
}

int mdsp_welch_plan_info(mdsp_welch_plan plan, int64_t* nout, int* engine_used) {
    if (!plan) MDSP_FAIL(MDSP_ERR_ARGUMENT, "plan is NULL");
    if (nout) *nout = plan->nout;
    if (engine_used) *engine_used = plan->engine;
    return MDSP_OK;
}

static int welch_check_args(mdsp_welch_plan plan, const void* s_dev, int64_t len, int64_t nch, int64_t lds_) {
    if (!plan) MDSP_FAIL(MDSP_ERR_ARGUMENT, "plan is NULL");
    if (len < 0 || nch < 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "negative size");
    if (!s_dev && len > 0 && nch > 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "s is NULL");
    if (nch > 1 && lds_ < len) MDSP_FAIL(MDSP_ERR_DIMENSION, "leading dimension smaller than the column length");
    if (nch > 65535) MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "more than 65535 channels per call");
    return MDSP_OK;
}

int mdsp_welch_reset(mdsp_welch_plan plan) {
    if (!plan) MDSP_FAIL(MDSP_ERR_ARGUMENT, "plan is NULL");
    plan->acc_fresh = true;
    plan->acc_frames = 0;
    plan->acc_nch = 0;
    plan->frames_on_device = false;
    plan->sums_global = false;
    return MDSP_OK;
}

int mdsp_welch_accumulate(mdsp_welch_plan plan, const void* s_dev, int64_t len, int64_t nch, int64_t lds_, void* stream) {
    MDSP_TRY(welch_check_args(plan, s_dev, len, nch, lds_));
    if (nch == 0) return MDSP_OK;
    if (!plan->acc_fresh && plan->acc_nch != nch)
        MDSP_FAIL(MDSP_ERR_DIMENSION, "accumulating %lld channels into sums of %lld channels (mdsp_welch_reset first)", (long long)nch, (long long)plan->acc_nch);
    if (plan->sums_global)   // a later all-reduce would add the earlier global totals once per rank
        MDSP_FAIL(MDSP_ERR_ARGUMENT, "these sums are totals over ranks (mdsp_welch_allreduce ran): mdsp_welch_reset before accumulating again");
    hipStream_t st = as_stream(stream);
    const bool cplx = dtype_is_complex(plan->dtype), dbl = dtype_is_double(plan->dtype);
    int rc;
    if (plan->engine == MDSP_ENGINE_ROCFFT) {
        if (cplx) rc = dbl ? welch_accumulate_rocfft<double, true>(plan, s_dev, len, nch, lds_, st) : welch_accumulate_rocfft<float, true>(plan, s_dev, len, nch, lds_, st);
        else rc = dbl ? welch_accumulate_rocfft<double, false>(plan, s_dev, len, nch, lds_, st) : welch_accumulate_rocfft<float, false>(plan, s_dev, len, nch, lds_, st);
    } else {
        if (cplx) rc = dbl ? welch_accumulate_fused<double, true>(plan, s_dev, len, nch, lds_, st) : welch_accumulate_fused<float, true>(plan, s_dev, len, nch, lds_, st);
        else rc = dbl ? welch_accumulate_fused<double, false>(plan, s_dev, len, nch, lds_, st) : welch_accumulate_fused<float, false>(plan, s_dev, len, nch, lds_, st);
    }
    if (rc != MDSP_OK) return rc;
    plan->acc_nch = nch;
    plan->acc_frames += mdsp_frame_count(len, plan->n, plan->noverlap);
    plan->frames_on_device = false;   // an all-reduced count (mdsp_welch_allreduce) does not know about these frames: finalize takes frames_total from its caller then
    return MDSP_OK;
}

int mdsp_welch_frames_accumulated(mdsp_welch_plan plan, int64_t* frames_per_channel) {
    if (!plan || !frames_per_channel) MDSP_FAIL(MDSP_ERR_ARGUMENT, "NULL argument");
    *frames_per_channel = plan->acc_frames;
    if (plan->frames_on_device) {   // after mdsp_welch_allreduce: the total over ranks, read back (this query, unlike the collective, synchronises)
        double k = 0;
        MDSP_HIP(hipStreamSynchronize(plan->count_stream));   // the set / all-reduce of the count were queued there (possibly a non-blocking stream)
        MDSP_HIP(hipMemcpy(&k, plan->kdev.p, sizeof(double), hipMemcpyDeviceToHost));
        *frames_per_channel = (int64_t)k;
    }
    return MDSP_OK;
}

int mdsp_welch_accumulator(mdsp_welch_plan plan, void** acc_dev, int64_t* count) {
    if (!plan || !acc_dev || !count) MDSP_FAIL(MDSP_ERR_ARGUMENT, "NULL argument");
    if (plan->acc_fresh) MDSP_FAIL(MDSP_ERR_ARGUMENT, "nothing accumulated since the last mdsp_welch_reset");
    *acc_dev = plan->acc_ptr();
    *count = (int64_t)plan->acc_nslices * plan->acc_nch * plan->acc_nacc;
    return MDSP_OK;
}

int mdsp_welch_finalize(mdsp_welch_plan plan, int64_t frames_total, void* psd_dev, int64_t ldp, void* stream) {
    if (!plan) MDSP_FAIL(MDSP_ERR_ARGUMENT, "plan is NULL");
    if (!psd_dev) MDSP_FAIL(MDSP_ERR_ARGUMENT, "out is NULL");
    if (plan->acc_fresh) MDSP_FAIL(MDSP_ERR_ARGUMENT, "nothing accumulated since the last mdsp_welch_reset");
    if (plan->acc_nch > 1 && ldp < plan->nout) MDSP_FAIL(MDSP_ERR_DIMENSION, "leading dimension smaller than the column length");
    const int64_t K = frames_total > 0 ? frames_total : (plan->frames_on_device ? 0 : plan->acc_frames);   // 0 with frames_on_device: the kernel reads the all-reduced count
    return dtype_is_double(plan->dtype) ? welch_finalize<double>(plan, K, plan->acc_nch, psd_dev, ldp, as_stream(stream))
                                        : welch_finalize<float>(plan, K, plan->acc_nch, psd_dev, ldp, as_stream(stream));
}

int mdsp_welch_exec(mdsp_welch_plan plan, const void* s_dev, int64_t len, int64_t nch, int64_t lds_, void* psd_dev, int64_t ldp, void* stream) {
    MDSP_TRY(welch_check_args(plan, s_dev, len, nch, lds_));
    if (nch == 0) return MDSP_OK;
    if (!psd_dev) MDSP_FAIL(MDSP_ERR_ARGUMENT, "out is NULL");
    if (nch > 1 && ldp < plan->nout) MDSP_FAIL(MDSP_ERR_DIMENSION, "leading dimension smaller than the column length");
    MDSP_TRY(mdsp_welch_reset(plan));
    MDSP_TRY(mdsp_welch_accumulate(plan, s_dev, len, nch, lds_, stream));
    return mdsp_welch_finalize(plan, 0, psd_dev, ldp, stream);
}

}  // extern "C"

// channel sum (local part of the cross-channel Welch mean); `scale` != 1: the mean in the same launch, rounded exactly as sum-then-scale would be
template <typename R>
__global__ __launch_bounds__(256) void channel_sum_kernel(const R* __restrict__ psd, R* __restrict__ out, int64_t nout, int64_t nch, int64_t ldp, double scale) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nout) return;
    double a = 0;
    for (int64_t c = 0; c < nch; ++c) a += (double)psd[c * ldp + j];
    const R sum = (R)a;
    out[j] = scale == 1.0 ? sum : (R)((double)sum * scale);
}

namespace mdsp {
int channel_sum_scaled(const void* psd_dev, int64_t nout, int64_t nch, int64_t ldp, int real_dtype, void* sum_dev, double scale, void* stream) {
    if (real_dtype != MDSP_F32 && real_dtype != MDSP_F64) MDSP_FAIL(MDSP_ERR_ARGUMENT, "real dtype expected");
    if (nout <= 0) return MDSP_OK;
    const dim3 g((unsigned)cdiv(nout, 256));
    if (real_dtype == MDSP_F32)
        hipLaunchKernelGGL(channel_sum_kernel<float>, g, dim3(256), 0, as_stream(stream), (const float*)psd_dev, (float*)sum_dev, nout, nch, ldp, scale);
    else
        hipLaunchKernelGGL(channel_sum_kernel<double>, g, dim3(256), 0, as_stream(stream), (const double*)psd_dev, (double*)sum_dev, nout, nch, ldp, scale);
    MDSP_LAUNCH_CHECK();
    return MDSP_OK;
}
}  // namespace mdsp

extern "C" int mdsp_channel_sum(const void* psd_dev, int64_t nout, int64_t nch, int64_t ldp, int real_dtype, void* sum_dev, void* stream) {
    return mdsp::channel_sum_scaled(psd_dev, nout, nch, ldp, real_dtype, sum_dev, 1.0, stream);
}

// ======================================================================================================
// STFT / spectrogram
// ======================================================================================================
struct mdsp_stft_plan_s {
    int dtype = MDSP_F32, engine = MDSP_ENGINE_ROCFFT, onesided = 1, psd_only = 0;
    int64_t n = 0, noverlap = 0, nfft = 0, nout = 0;
    double r = 1;
    bool have_win = false;
    DevBuf win, table;
    const double* win_ptr = nullptr;   // window used by exec (the plan's own, or one taper of a multitaper plan)
    int accumulate = 0;                // PSD mode: out += |X|^2 m instead of out = (the taper loop of mt_pgram!, multitaper.jl:240-243)
    int mt_ntapers = 0;                // > 0: all tapers in one launch where the kernel supports it (real input, fused engine)
    const double* mt_rinv = nullptr;
    RocPlan fwd;
    DevBuf fr, spec;
    int64_t batch = 0;
    big::EngineHolder big;             // nfft above the single-workgroup kernels: the multi-pass engine (bigfft.hip)
    mdsp::GxPlan gx;                   // 7-smooth sizes without a compile-time schedule: the run-time-schedule kernel (spectral_gx.h)
    mdsp::DevBuf winr;                 // lean compile-time schedules (CtSched flag 4096): the window of THIS launch in the working precision (multitaper plans change it per taper)
    mdsp::CtColsPlan ctbig;            // columns on the single-workgroup schedules of ctbig_sizes.h (spectral_ctbig_cols.hip)
};

namespace {

template <typename R, bool CPLX>
int stft_exec_rocfft(mdsp_stft_plan_s* pl, const void* s, int64_t len, int64_t nch, int64_t lds_, void* out, int64_t ldo, int64_t chs, hipStream_t st) {
    using TT = std::conditional_t<CPLX, cx<R>, R>;
    const int64_t nfft = pl->nfft, n = pl->n, hop = pl->n - pl->noverlap;
    const int64_t K = mdsp_frame_count(len, n, pl->noverlap);
    const int nspec = (int)(CPLX ? nfft : nfft / 2 + 1);
    const int64_t nunits = K * nch;
    if (nunits == 0) return MDSP_OK;
    const int64_t per_unit = (int64_t)sizeof(TT) * nfft + (int64_t)sizeof(cx<R>) * nspec;
    const int64_t batch = std::max<int64_t>(1, std::min<int64_t>(nunits, rocfft_chunk_bytes() / per_unit));
    if (pl->batch != batch) {
        MDSP_TRY(pl->fr.reserve((size_t)(sizeof(TT) * nfft * batch)));
        MDSP_TRY(pl->spec.reserve((size_t)(sizeof(cx<R>) * nspec * batch)));
        MDSP_TRY(pl->fwd.create(CPLX ? FftKind::C2C_FWD : FftKind::R2C, sizeof(R) == 8, nfft, batch, false));
        pl->batch = batch;
    }
    const int gx = (int)std::min<int64_t>(cdiv(nfft, 256), 8);
    const int nout = (int)pl->nout;
    const bool direct = !pl->psd_only && nout == nspec && ldo == nout && (nch == 1 || chs == K * (int64_t)nout) && !MDSP_DBG(stft_nodirect);
    for (int64_t u0 = 0; u0 < nunits; u0 += batch) {
        const int64_t cnt = std::min<int64_t>(batch, nunits - u0);
        hipLaunchKernelGGL(frame_window_kernel<TT>, dim3((unsigned)cnt, gx), dim3(256), 0, st, (const TT*)s, pl->fr.as<TT>(),
                           pl->have_win ? pl->win_ptr : nullptr, lds_, K, hop, (int)n, (int)nfft, u0, nunits);
        MDSP_LAUNCH_CHECK();
        // raw STFT whose output matrix has exactly the batched transform's layout (contiguous columns of nspec bins, channels
        // back to back): full batches are transformed straight into it, one pass over the spectra less
        if (direct && cnt == batch) {
            MDSP_TRY(pl->fwd.exec(pl->fr.p, static_cast<cx<R>*>(out) + u0 * nout, st));
            continue;
        }
        MDSP_TRY(pl->fwd.exec(pl->fr.p, pl->spec.p, st));
        const dim3 g((unsigned)cnt, gx);
        if (pl->psd_only)
            hipLaunchKernelGGL((stft_store_kernel<R, true, !CPLX>), g, dim3(256), 0, st, pl->spec.as<cx<R>>(), out, nspec, (int)nfft, nout, K, ldo, chs, u0,
                               nunits, pl->r, pl->onesided, pl->accumulate);
        else
            hipLaunchKernelGGL((stft_store_kernel<R, false, !CPLX>), g, dim3(256), 0, st, pl->spec.as<cx<R>>(), out, nspec, (int)nfft, nout, K, ldo, chs, u0,
                               nunits, pl->r, pl->onesided, 0);
        MDSP_LAUNCH_CHECK();
    }
    return MDSP_OK;
}

template <typename R, int N, bool CPLX> int stft_launch_n(mdsp_stft_plan_s* pl, SpecArgs& a, hipStream_t st) {
    using Gm = Geo<R, N>;
    constexpr int E = Gm::E, G = Gm::G, NBUF = Gm::NBUF;
    constexpr bool TWREG = Gm::TWREG;
    constexpr int threads = (N / E) * G;
    const int64_t work = cdiv(a.K, G);
    int grid = 1;
    auto run = [&](auto kern) -> int {
        MDSP_TRY(grid_for(kern, threads, work, a.nch, &grid));
        set_schedule(a, a.K, (int64_t)grid * G);
        hipLaunchKernelGGL(kern, dim3(grid, (unsigned)a.nch), dim3(threads), 0, st, a);
        MDSP_LAUNCH_CHECK();
        return MDSP_OK;
    };
    if constexpr (N == 1024 && sizeof(R) == 4 && CPLX) {   // tuning alternatives of the config-4 shape (MDSP_STFT_VARIANT)
        // default for this shape: one wavefront per transform (E = 16, T = 64, four transforms per workgroup, no s_barrier
        // at all) -- measured 7-11 % faster than the two-wave E = 8 geometry (variant 9).  Variants 2-4: global twiddles.
        const int variant = tunables().stft_variant;
        if (variant >= 1 && variant <= 4) {
            constexpr int E2 = 16, G2 = 4, T2 = 64;
            const int64_t work2 = cdiv(a.K, G2);
            auto run2 = [&](auto kern) -> int {
                MDSP_TRY(grid_for(kern, T2 * G2, work2, a.nch, &grid));
                set_schedule(a, a.K, (int64_t)grid * G2);
                hipLaunchKernelGGL(kern, dim3(grid, (unsigned)a.nch), dim3(T2 * G2), 0, st, a);
                MDSP_LAUNCH_CHECK();
                return MDSP_OK;
            };
            if (variant == 1) {
                // register reuse of the overlapping samples when a frame advances by a whole number of elements per thread
                const int shift = (!MDSP_DBG(stft_noshift) && a.n == N && a.hop % T2 == 0) ? (int)(a.hop / T2) : 0;
#define MDSP_STFT_V1(S) (pl->psd_only ? run2(stft_fused_kernel<R, N, E2, G2, 1, 4, CPLX, true, 2, 1, true, S>) : run2(stft_fused_kernel<R, N, E2, G2, 1, 4, CPLX, false, 2, 1, true, S>))
                switch (shift) {
                    case 1: return MDSP_STFT_V1(1);
                    case 2: return MDSP_STFT_V1(2);
                    case 4: return MDSP_STFT_V1(4);
                    case 8: return MDSP_STFT_V1(8);
                    default: return MDSP_STFT_V1(0);
                }
#undef MDSP_STFT_V1
            }
            if (variant == 2) return pl->psd_only ? run2(stft_fused_kernel<R, N, E2, G2, 0, 4, CPLX, true, 2, 1, true>) : run2(stft_fused_kernel<R, N, E2, G2, 0, 4, CPLX, false, 2, 1, true>);
            if (variant == 3) return pl->psd_only ? run2(stft_fused_kernel<R, N, E2, G2, 0, 5, CPLX, true, 2, 1, true>) : run2(stft_fused_kernel<R, N, E2, G2, 0, 5, CPLX, false, 2, 1, true>);
            return pl->psd_only ? run2(stft_fused_kernel<R, N, E2, G2, 0, 4, CPLX, true, 3, 1, true>) : run2(stft_fused_kernel<R, N, E2, G2, 0, 4, CPLX, false, 3, 1, true>);
        }
    }
    if constexpr (!CPLX) {
        // real signals: two frames per transform (stft_pair_kernel); MDSP_STFT_NOPAIR=1 keeps the one-frame-per-transform kernel
        const bool nopair = MDSP_DBG(stft_nopair);
        if (!nopair && a.n <= N) {
            const int64_t units = cdiv(a.K, 2);
            a.units_per_ch = units;
            auto runp = [&](auto kern) -> int {
                MDSP_TRY(grid_for(kern, threads, cdiv(units, G), a.nch, &grid));
                set_schedule(a, units, (int64_t)grid * G);
                hipLaunchKernelGGL(kern, dim3(grid, (unsigned)a.nch), dim3(threads), 0, st, a);
                MDSP_LAUNCH_CHECK();
                return MDSP_OK;
            };
            if (pl->psd_only && a.ntapers > 0) return runp(stft_pair_kernel<R, N, E, G, TWREG, pad_default<R>(), true, 2, NBUF, true>);
            if (pl->psd_only) return runp(stft_pair_kernel<R, N, E, G, TWREG, pad_default<R>(), true, 2, NBUF>);
            return runp(stft_pair_kernel<R, N, E, G, TWREG, pad_default<R>(), false, 2, NBUF>);
        }
    }
    if constexpr (CPLX && sizeof(R) == 4) {   // complex Float32 frames that advance by whole elements: overlap stays in registers
        constexpr int T = N / E;
        const int shift = (!MDSP_DBG(stft_noshift) && a.n == N && a.hop % T == 0 && a.hop / T < E) ? (int)(a.hop / T) : 0;
#define MDSP_STFT_GEN(S) \
    (pl->psd_only ? run(stft_fused_kernel<R, N, E, G, TWREG, pad_default<R>(), CPLX, true, 2, NBUF, true, S>) \
                  : run(stft_fused_kernel<R, N, E, G, TWREG, pad_default<R>(), CPLX, false, 2, NBUF, true, S>))
        if constexpr (E > 4) {
            if (shift == 1) return MDSP_STFT_GEN(1);
            if (shift == 2) return MDSP_STFT_GEN(2);
            if (shift == 4) return MDSP_STFT_GEN(4);
        }
#undef MDSP_STFT_GEN
    }
    if (pl->psd_only) return run(stft_fused_kernel<R, N, E, G, TWREG, pad_default<R>(), CPLX, true, 2, NBUF, true>);
    return run(stft_fused_kernel<R, N, E, G, TWREG, pad_default<R>(), CPLX, false, 2, NBUF, true>);
}

template <typename R, bool CPLX>
int stft_exec_fused(mdsp_stft_plan_s* pl, const void* s, int64_t len, int64_t nch, int64_t lds_, void* out, int64_t ldo, int64_t chs, hipStream_t st) {
    const int64_t K = mdsp_frame_count(len, pl->n, pl->noverlap);
    if (K == 0) return MDSP_OK;
    if (use_gx(pl->dtype, pl->nfft, CPLX, 1)) {   // run-time-schedule kernel (spectral_gx.h); multitaper plans come here once per taper (accumulate)
        if (tunables().gx != 4 && tunables().gx != 5 && ctbig_cols_ok(pl->dtype, pl->nfft)) {   // ... or a single-workgroup compile-time schedule (spectral_ctbig_cols.hip)
            CtBigColsArgs c{};
            c.s = s; c.out = out; c.win = pl->have_win ? pl->win_ptr : nullptr; c.len = len; c.lds_ = lds_; c.K = K; c.hop = pl->n - pl->noverlap; c.nch = nch; c.ldo = ldo; c.chs = chs;
            c.n = (int)pl->n; c.nfft = pl->nfft; c.nout = (int)pl->nout; c.onesided = pl->onesided; c.psd = pl->psd_only; c.accumulate = pl->accumulate; c.r = pl->r;
            return ctbig_stft(pl->ctbig, pl->dtype, c, st);
        }
        GxArgs g{};
        g.s = s; g.out = out; g.lds_ = lds_; g.K = K; g.hop = pl->n - pl->noverlap; g.nch = nch; g.ldo = ldo; g.chs = chs;
        g.n = (int)pl->n; g.nfft = (int)pl->nfft; g.nout = (int)pl->nout; g.onesided = pl->onesided; g.psd = pl->psd_only; g.accumulate = pl->accumulate; g.r = pl->r;
        int64_t ngroups = 0;
        return gx_launch<R, CPLX, 1>(pl->gx, g, pl->have_win ? pl->win_ptr : nullptr, pl->dtype, st, &ngroups, nullptr);
    }
    if (use_big(pl->dtype, pl->nfft, CPLX, 1)) {   // nfft above the one-workgroup sizes (bigfft.hip); multitaper plans come here once per taper (accumulate)
        using TT = std::conditional_t<CPLX, cx<R>, R>;
        const size_t osz = pl->psd_only ? sizeof(R) : sizeof(cx<R>);
        for (int64_t c = 0; c < nch; ++c)
            MDSP_TRY(big::stft(pl->big, pl->dtype, pl->n, pl->nfft, static_cast<const TT*>(s) + c * lds_, K, pl->n - pl->noverlap, pl->have_win ? pl->win_ptr : nullptr,
                               static_cast<char*>(out) + (size_t)(c * chs) * osz, ldo, pl->nout, pl->onesided, pl->psd_only, pl->accumulate, pl->r, st));
        return MDSP_OK;
    }
    SpecArgs a{};
    a.s = s;
    a.out = out;
    a.table = pl->table.p;
    a.win = pl->have_win ? pl->win_ptr : nullptr;
    a.accumulate = pl->accumulate;
    a.ntapers = pl->mt_ntapers;
    a.rinv = pl->mt_rinv;
    a.len = len;
    a.lds_ = lds_;
    a.K = K;
    a.hop = pl->n - pl->noverlap;
    a.units_per_ch = K;
    a.nch = nch;
    a.ldo = ldo;
    a.chs = chs;
    a.n = (int)pl->n;
    a.nout = (int)pl->nout;
    a.onesided = pl->onesided;
    a.r = pl->r;
    if (!fused_size_ok(pl->dtype, pl->nfft)) {   // mixed-radix sizes (spectral_gen.h); multitaper plans come here once per taper (accumulate)
        GenArgs g{};
        g.s = s; g.out = out; g.roots = pl->table.p; g.win = a.win;
        g.len = len; g.lds_ = lds_; g.K = K; g.hop = a.hop; g.nch = nch; g.ldo = ldo; g.chs = chs;
        g.units_per_ch = CPLX ? K : cdiv(K, 2);
        g.n = a.n; g.N = (int)pl->nfft; g.nout = a.nout; g.onesided = a.onesided; g.psd = pl->psd_only; g.accumulate = pl->accumulate; g.r = a.r;
        if (CPLX && lean_window_needed(pl->nfft, sizeof(R) == 8)) {   // (per launch: the window pointer of a multitaper plan changes between tapers)
            MDSP_TRY(lean_window<R>(pl->winr, a.win, a.n, pl->nfft, st));
            g.winr = pl->winr.p;
        }
        int64_t nslots = 0;
        return gen_launch<R, CPLX, 1>(g, nch, st, &nslots, nullptr);
    }
    switch (pl->nfft) {
        case 256: return stft_launch_n<R, 256, CPLX>(pl, a, st);
        case 512: return stft_launch_n<R, 512, CPLX>(pl, a, st);
        case 1024: return stft_launch_n<R, 1024, CPLX>(pl, a, st);
        case 2048: return stft_launch_n<R, 2048, CPLX>(pl, a, st);
        case 4096: return stft_launch_n<R, 4096, CPLX>(pl, a, st);
        case 8192:
            if constexpr (sizeof(R) == 4) return stft_launch_n<R, 8192, CPLX>(pl, a, st);
        default: break;
    }
    MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "fused STFT does not support nfft=%lld", (long long)pl->nfft);
}

}  // namespace

extern "C" {

int mdsp_stft_plan_create(mdsp_stft_plan* plan, int64_t n, int64_t noverlap, int64_t nfft, const double* window_host, double r, int onesided,
                          int psd_only, int dtype, int engine) {
    if (!plan) MDSP_FAIL(MDSP_ERR_ARGUMENT, "plan is NULL");
    *plan = nullptr;
    if (!dtype_valid(dtype)) MDSP_FAIL(MDSP_ERR_ARGUMENT, "invalid dtype %d", dtype);
    if (onesided && dtype_is_complex(dtype)) MDSP_FAIL(MDSP_ERR_ARGUMENT, "cannot compute one-sided FFT of a complex signal");
    MDSP_TRY(check_split(n, noverlap, nfft));
    if (psd_only && !(r > 0)) MDSP_FAIL(MDSP_ERR_ARGUMENT, "normalisation r must be positive");
    int eng;
    MDSP_TRY(resolve_engine(engine, dtype, nfft, &eng));
    auto pl = new mdsp_stft_plan_s();
    pl->dtype = dtype;
    pl->engine = eng;
    pl->onesided = onesided ? 1 : 0;
    pl->psd_only = psd_only ? 1 : 0;
    pl->n = n;
    pl->noverlap = noverlap;
    pl->nfft = nfft;
    pl->nout = onesided ? nfft / 2 + 1 : nfft;
    pl->r = r;
    int st = MDSP_OK;
    if (window_host) {
        pl->have_win = true;
        st = pl->win.reserve(sizeof(double) * (size_t)n);
        if (st == MDSP_OK && hipMemcpy(pl->win.p, window_host, sizeof(double) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess)
            st = set_error(MDSP_ERR_DEVICE, "window upload failed");
        pl->win_ptr = pl->win.as<double>();
    }
    if (st == MDSP_OK && eng == MDSP_ENGINE_FUSED && !use_big(dtype, nfft, dtype_is_complex(dtype), 1) && !use_gx(dtype, nfft, dtype_is_complex(dtype), 1))
        st = dtype_is_double(dtype) ? upload_roots<double>(pl->table, nfft) : upload_roots<float>(pl->table, nfft);
    if (st != MDSP_OK) {
        delete pl;
        return st;
    }
    *plan = pl;
    return MDSP_OK;
}

int mdsp_stft_plan_destroy(mdsp_stft_plan plan) {
    delete plan;
    return MDSP_OK;
}

int mdsp_stft_plan_info(mdsp_stft_plan plan, int64_t* nout, int* engine_used) {
    if (!plan) MDSP_FAIL(MDSP_ERR_ARGUMENT, "plan is NULL");
    if (nout) *nout = plan->nout;
    if (engine_used) *engine_used = plan->engine;
    return MDSP_OK;
}

int mdsp_stft_exec(mdsp_stft_plan plan, const void* s_dev, int64_t len, int64_t nch, int64_t lds_, void* out_dev, int64_t ldo, int64_t chs,
                   void* stream) {
    if (!plan) MDSP_FAIL(MDSP_ERR_ARGUMENT, "plan is NULL");
    if (len < 0 || nch < 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "negative size");
    const int64_t K = mdsp_frame_count(len, plan->n, plan->noverlap);
    if (nch == 0 || K == 0) return MDSP_OK;
    if (!out_dev || !s_dev) MDSP_FAIL(MDSP_ERR_ARGUMENT, "NULL buffer");
    if (ldo < plan->nout) MDSP_FAIL(MDSP_ERR_DIMENSION, "column stride smaller than the column length");
    if (nch > 1 && (lds_ < len || chs < ldo * (K - 1) + plan->nout)) MDSP_FAIL(MDSP_ERR_DIMENSION, "channel stride too small");
    if (nch > 65535) MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "more than 65535 channels per call");
    hipStream_t st = as_stream(stream);
    const bool cplx = dtype_is_complex(plan->dtype), dbl = dtype_is_double(plan->dtype);
    if (plan->engine == MDSP_ENGINE_ROCFFT) {
        if (cplx) return dbl ? stft_exec_rocfft<double, true>(plan, s_dev, len, nch, lds_, out_dev, ldo, chs, st)
                             : stft_exec_rocfft<float, true>(plan, s_dev, len, nch, lds_, out_dev, ldo, chs, st);
        return dbl ? stft_exec_rocfft<double, false>(plan, s_dev, len, nch, lds_, out_dev, ldo, chs, st)
                   : stft_exec_rocfft<float, false>(plan, s_dev, len, nch, lds_, out_dev, ldo, chs, st);
    }
    if (cplx) return dbl ? stft_exec_fused<double, true>(plan, s_dev, len, nch, lds_, out_dev, ldo, chs, st)
                         : stft_exec_fused<float, true>(plan, s_dev, len, nch, lds_, out_dev, ldo, chs, st);
    return dbl ? stft_exec_fused<double, false>(plan, s_dev, len, nch, lds_, out_dev, ldo, chs, st)
               : stft_exec_fused<float, false>(plan, s_dev, len, nch, lds_, out_dev, ldo, chs, st);
}

// stft / spectrogram of host arrays (periodograms.jl:872-897 takes a host vector and returns a host matrix): channel by channel, runs of whole
// frames.  Chunk c of a channel holds frames [k0, k1): their samples [k0 hop, (k1-1) hop + n) go up, their columns come down -- every frame is the
// same frame the device-resident call transforms, so the two agree bit for bit.  The output is 2-8x the input (40 B per sample at config 4), which is
// why the chunk is sized by its OUTPUT and why the pipeline keeps the D2H stream busy while the next chunk uploads (hostpipe.h).
int mdsp_stft_exec_host(mdsp_stft_plan plan, const void* s_host, int64_t len, int64_t nch, int64_t lds_, void* out_host, int64_t ldo, int64_t chs,
                        int flags) {
    if (!plan) MDSP_FAIL(MDSP_ERR_ARGUMENT, "plan is NULL");
    if (len < 0 || nch < 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "negative size");
    if (plan->accumulate || plan->mt_ntapers > 0) MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "multitaper plans have no host-array entry");
    const int64_t n = plan->n, hop = plan->n - plan->noverlap, nout = plan->nout;
    const int64_t K = mdsp_frame_count(len, n, plan->noverlap);
    if (nch == 0 || K == 0) return MDSP_OK;
    if (!out_host || !s_host) MDSP_FAIL(MDSP_ERR_ARGUMENT, "NULL buffer");
    if (ldo < nout) MDSP_FAIL(MDSP_ERR_DIMENSION, "column stride smaller than the column length");
    if (nch > 1 && (lds_ < len || chs < ldo * (K - 1) + nout)) MDSP_FAIL(MDSP_ERR_DIMENSION, "channel stride too small");
    const bool pinned = (flags & MDSP_HOST_PINNED) != 0;
    const size_t esz = dtype_size(plan->dtype);
    const size_t osz = plan->psd_only ? dtype_size(dtype_real_of(plan->dtype)) : dtype_size(dtype_complex_of(plan->dtype));
    // frames per chunk: EVEN -- real signals ride two consecutive frames (2j, 2j+1) per transform, and a chunk that started at an odd frame would
    // pair them differently from the device-resident call (same values up to rounding, not bit for bit)
    int64_t fpc = std::max<int64_t>(2, ((int64_t)tunables().host_chunk_mib << 20) / (int64_t)(nout * (int64_t)osz));
    fpc &= ~int64_t(1);
    const size_t in_cap = (size_t)((fpc - 1) * hop + n) * esz, out_cap = (size_t)fpc * (size_t)nout * osz;
    hostpipe::Session ss(in_cap, out_cap, pinned);
    int rc = ss.status();
    for (int64_t c = 0; c < nch && rc == MDSP_OK; ++c) {
        const char* sc = static_cast<const char*>(s_host) + (size_t)(c * lds_) * esz;
        char* oc = static_cast<char*>(out_host) + (size_t)(c * chs) * osz;
        for (int64_t k0 = 0; k0 < K && rc == MDSP_OK; k0 += fpc) {
            hostpipe::Lane* ln = nullptr;
            if ((rc = ss.acquire(&ln)) != MDSP_OK) break;
            const int64_t k1 = std::min(K, k0 + fpc), cl = (k1 - k0 - 1) * hop + n;
            if ((rc = ss.upload(ln, sc + (size_t)(k0 * hop) * esz, (size_t)cl * esz, (size_t)cl * esz, 1)) != MDSP_OK) break;
            if ((rc = mdsp_stft_exec(plan, ln->din.p, cl, 1, cl, ln->dout.p, nout, nout * (k1 - k0), ss.kstream())) != MDSP_OK) break;
            rc = ss.download(ln, oc + (size_t)(k0 * ldo) * osz, (size_t)ldo * osz, (size_t)nout * osz, (size_t)(k1 - k0), 0, (size_t)nout * osz);
        }
    }
    return ss.finish(rc);
}

}  // extern "C"

// ======================================================================================================
// Multitaper (src/multitaper.jl): MTConfig, mt_pgram!, mt_spectrogram!, mt_cross_power_spectra!, mt_coherence!
// ======================================================================================================
// The tapers are ntapers windows applied to the SAME frame; mt_pgram! (multitaper.jl:225-245) is the loop
//     output .= 0;  for taper: fft_input = window[:, taper] .* signal; fft; fft2pow!(output, fft_output, nfft, r[taper], onesided)
// which is the STFT/PSD pipeline above run once per taper with the accumulate flag set from the second taper on.
struct mdsp_mt_plan_s {
    mdsp_stft_plan_s st;          // engine, tables, rocFFT scratch -- window pointer / r / psd_only are re-pointed per taper
    int64_t ntapers = 0;
    DevBuf wins;                  // (n, ntapers) Float64, column-major
    std::vector<double> r;        // inverse normalisation per taper (multitaper.jl:15-17)
    DevBuf wts;                   // 2 / r  (normalization_weights, :499), Float64
    DevBuf rinv;                  // 1 / r, Float64 (in-kernel taper loop)
    DevBuf demeaned;              // scratch for demean = true (:566-570)
    DevBuf finds;                 // frequency indices of the last cross-spectra call
};

namespace {

// demeaned[c, :] = signal[c, :] - mean(signal[c, :])   (real signals only: check_onesided_real, :417-422)
template <typename R> __global__ __launch_bounds__(256) void demean_kernel(const R* __restrict__ s, R* __restrict__ out, int64_t n, int64_t lds_) {
    __shared__ double red[256];
    const int64_t ch = blockIdx.x;
    const R* sc = s + ch * lds_;
    double a = 0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) a += (double)sc[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    const R m = (R)(red[0] / (double)n);
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) out[ch * n + i] = sc[i] - m;
}

// cs_inner! (multitaper.jl:602-616) on x_mt[f, taper, ch] with the DC / Nyquist 1/sqrt(2) of :577-580 folded in:
//   out[l, m, fi] = sum_k w_k x[f, k, l] conj(x[f, k, m])
template <typename R>
__global__ __launch_bounds__(256) void cross_spectra_kernel(const cx<R>* __restrict__ x, const double* __restrict__ w, const int64_t* __restrict__ finds,
                                                            cx<R>* __restrict__ out, int64_t nfreq, int64_t ntapers, int64_t nch, int64_t nfi, int nfft_even) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nch * nch * nfi) return;
    const int64_t l = idx % nch, m = (idx / nch) % nch, fi = idx / (nch * nch);
    const int64_t f = finds[fi];
    const bool edge = f == 0 || (nfft_even && f == nfreq - 1);
    R ax = 0, ay = 0;
    for (int64_t k = 0; k < ntapers; ++k) {
        cx<R> a = x[f + nfreq * (k + ntapers * l)], b = x[f + nfreq * (k + ntapers * m)];
        if (edge) {   // x_mt[1, :, :] ./= sqrt(2) (and the Nyquist row for even nfft)
            a = {a.x / (R)1.4142135623730951, a.y / (R)1.4142135623730951};
            b = {b.x / (R)1.4142135623730951, b.y / (R)1.4142135623730951};
        }
        const R wk = (R)w[k];
        const cx<R> p = fft::cmulc(a, b);   // a * conj(b)
        ax += wk * p.x;
        ay += wk * p.y;
    }
    out[idx] = {ax, ay};
}

// coherence_from_cs! (multitaper.jl:704-723)
template <typename R>
__global__ __launch_bounds__(256) void coherence_kernel(const cx<R>* __restrict__ cs, R* __restrict__ out, int64_t nch, int64_t nf) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nch * nch * nf) return;
    const int64_t c1 = idx % nch, c2 = (idx / nch) % nch, f = idx / (nch * nch);
    if (c1 == c2) {
        out[idx] = (R)1;
        return;
    }
    const int64_t hi = c1 > c2 ? c1 : c2, lo = c1 > c2 ? c2 : c1;      // the lower triangle is computed, then mirrored
    const cx<R> v = cs[hi + nch * (lo + nch * f)];
    const cx<R> d1 = cs[hi + nch * (hi + nch * f)], d2 = cs[lo + nch * (lo + nch * f)];
    out[idx] = sqrt(v.x * v.x + v.y * v.y) / sqrt(d1.x * d2.x - d1.y * d2.y);
}

int stft_dispatch(mdsp_stft_plan_s* plan, const void* s_dev, int64_t len, int64_t nch, int64_t lds_, void* out_dev, int64_t ldo, int64_t chs, hipStream_t st) {
    const bool cplx = dtype_is_complex(plan->dtype), dbl = dtype_is_double(plan->dtype);
    if (plan->engine == MDSP_ENGINE_ROCFFT) {
        if (cplx) return dbl ? stft_exec_rocfft<double, true>(plan, s_dev, len, nch, lds_, out_dev, ldo, chs, st)
                             : stft_exec_rocfft<float, true>(plan, s_dev, len, nch, lds_, out_dev, ldo, chs, st);
        return dbl ? stft_exec_rocfft<double, false>(plan, s_dev, len, nch, lds_, out_dev, ldo, chs, st)
                   : stft_exec_rocfft<float, false>(plan, s_dev, len, nch, lds_, out_dev, ldo, chs, st);
    }
    if (cplx) return dbl ? stft_exec_fused<double, true>(plan, s_dev, len, nch, lds_, out_dev, ldo, chs, st)
                         : stft_exec_fused<float, true>(plan, s_dev, len, nch, lds_, out_dev, ldo, chs, st);
    return dbl ? stft_exec_fused<double, false>(plan, s_dev, len, nch, lds_, out_dev, ldo, chs, st)
               : stft_exec_fused<float, false>(plan, s_dev, len, nch, lds_, out_dev, ldo, chs, st);
}

}  // namespace

extern "C" {

int mdsp_mt_plan_create(mdsp_mt_plan* plan, int64_t n, int64_t nfft, const double* tapers_host, int64_t ntapers, const double* r_host, int onesided,
                        int dtype, int engine) {
    if (!plan) MDSP_FAIL(MDSP_ERR_ARGUMENT, "plan is NULL");
    *plan = nullptr;
    if (!dtype_valid(dtype)) MDSP_FAIL(MDSP_ERR_ARGUMENT, "invalid dtype %d", dtype);
    if (onesided && dtype_is_complex(dtype)) MDSP_FAIL(MDSP_ERR_ARGUMENT, "cannot compute one-sided FFT of a complex signal");   // :46-47, :115-117
    if (n <= 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "`n_samples` must be positive");                                                     // :21
    if (nfft < n) MDSP_FAIL(MDSP_ERR_ARGUMENT, "Must have `nfft >= n_samples`");                                                  // :22
    if (ntapers <= 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "`ntapers` must be positive");                                                 // :23
    if (!tapers_host || !r_host) MDSP_FAIL(MDSP_ERR_ARGUMENT, "tapers / r are NULL");
    for (int64_t k = 0; k < ntapers; ++k)
        if (!(r_host[k] > 0)) MDSP_FAIL(MDSP_ERR_ARGUMENT, "normalisation r must be positive");
    int eng;
    MDSP_TRY(resolve_engine(engine, dtype, nfft, &eng));
    std::unique_ptr<mdsp_mt_plan_s> pl(new mdsp_mt_plan_s());
    mdsp_stft_plan_s& st = pl->st;
    st.dtype = dtype;
    st.engine = eng;
    st.onesided = onesided ? 1 : 0;
    st.n = n;
    st.noverlap = 0;
    st.nfft = nfft;
    st.nout = onesided ? nfft / 2 + 1 : nfft;
    st.have_win = true;
    pl->ntapers = ntapers;
    pl->r.assign(r_host, r_host + ntapers);
    MDSP_TRY(pl->wins.reserve(sizeof(double) * (size_t)(n * ntapers)));
    MDSP_HIP(hipMemcpy(pl->wins.p, tapers_host, sizeof(double) * (size_t)(n * ntapers), hipMemcpyHostToDevice));
    std::vector<double> w((size_t)ntapers);
    for (int64_t k = 0; k < ntapers; ++k) w[(size_t)k] = 2.0 / r_host[k];
    MDSP_TRY(pl->wts.reserve(sizeof(double) * (size_t)ntapers));
    MDSP_HIP(hipMemcpy(pl->wts.p, w.data(), sizeof(double) * (size_t)ntapers, hipMemcpyHostToDevice));
    for (int64_t k = 0; k < ntapers; ++k) w[(size_t)k] = 1.0 / r_host[k];
    MDSP_TRY(pl->rinv.reserve(sizeof(double) * (size_t)ntapers));
    MDSP_HIP(hipMemcpy(pl->rinv.p, w.data(), sizeof(double) * (size_t)ntapers, hipMemcpyHostToDevice));
    if (eng == MDSP_ENGINE_FUSED) MDSP_TRY(dtype_is_double(dtype) ? upload_roots<double>(st.table, nfft) : upload_roots<float>(st.table, nfft));
    *plan = pl.release();
    return MDSP_OK;
}

int mdsp_mt_plan_destroy(mdsp_mt_plan plan) {
    delete plan;
    return MDSP_OK;
}

int mdsp_mt_plan_info(mdsp_mt_plan plan, int64_t* nout, int64_t* ntapers, int* engine_used) {
    if (!plan) MDSP_FAIL(MDSP_ERR_ARGUMENT, "plan is NULL");
    if (nout) *nout = plan->st.nout;
    if (ntapers) *ntapers = plan->ntapers;
    if (engine_used) *engine_used = plan->st.engine;
    return MDSP_OK;
}

int mdsp_mt_psd_exec(mdsp_mt_plan plan, const void* s_dev, int64_t len, int64_t noverlap, int64_t nch, int64_t lds_, void* out_dev, int64_t ldo,
                     int64_t chs, void* stream) {
    if (!plan) MDSP_FAIL(MDSP_ERR_ARGUMENT, "plan is NULL");
    if (len < 0 || nch < 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "negative size");
    mdsp_stft_plan_s& st = plan->st;
    if (noverlap < 0 || noverlap >= st.n) MDSP_FAIL(MDSP_ERR_ARGUMENT, "Need `samples_per_window > n_overlap_samples`");   // multitaper.jl:264-266
    const int64_t K = mdsp_frame_count(len, st.n, noverlap);
    if (nch == 0 || K == 0) return MDSP_OK;
    if (!out_dev || !s_dev) MDSP_FAIL(MDSP_ERR_ARGUMENT, "NULL buffer");
    if (ldo < st.nout) MDSP_FAIL(MDSP_ERR_DIMENSION, "column stride smaller than the column length");
    if (nch > 1 && (lds_ < len || chs < ldo * (K - 1) + st.nout)) MDSP_FAIL(MDSP_ERR_DIMENSION, "channel stride too small");
    if (nch > 65535) MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "more than 65535 channels per call");
    st.noverlap = noverlap;
    st.psd_only = 1;
    // real signals on the fused engine: every taper inside one launch (the frame pair stays in registers); otherwise one
    // pass per taper with the accumulate flag
    if (st.engine == MDSP_ENGINE_FUSED && fused_size_ok(st.dtype, st.nfft) && !dtype_is_complex(st.dtype) && !MDSP_DBG(mt_passes) && !MDSP_DBG(stft_nopair)) {
        st.win_ptr = plan->wins.as<double>();
        st.r = plan->r[0];
        st.accumulate = 0;
        st.mt_ntapers = (int)plan->ntapers;
        st.mt_rinv = plan->rinv.as<double>();
        const int rc = stft_dispatch(&st, s_dev, len, nch, lds_, out_dev, ldo, chs, as_stream(stream));
        st.mt_ntapers = 0;
        st.mt_rinv = nullptr;
        return rc;
    }
    for (int64_t k = 0; k < plan->ntapers; ++k) {
        st.win_ptr = plan->wins.as<double>() + k * st.n;
        st.r = plan->r[(size_t)k];
        st.accumulate = k > 0;
        const int rc = stft_dispatch(&st, s_dev, len, nch, lds_, out_dev, ldo, chs, as_stream(stream));
        if (rc != MDSP_OK) return rc;
    }
    st.accumulate = 0;
    return MDSP_OK;
}

int mdsp_mt_spectra_exec(mdsp_mt_plan plan, const void* s_dev, int64_t nch, int64_t lds_, int demean, void* xmt_dev, void* stream) {
    if (!plan) MDSP_FAIL(MDSP_ERR_ARGUMENT, "plan is NULL");
    mdsp_stft_plan_s& st = plan->st;
    if (dtype_is_complex(st.dtype) || !st.onesided)   // check_onesided_real, multitaper.jl:417-422
        MDSP_FAIL(MDSP_ERR_ARGUMENT, "Only real data is supported (with the default choice of `onesided=true`) for this operation.");
    if (nch <= 0) return MDSP_OK;
    if (!s_dev || !xmt_dev) MDSP_FAIL(MDSP_ERR_ARGUMENT, "NULL buffer");
    if (nch > 65535) MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "more than 65535 channels per call");
    if (nch > 1 && lds_ < st.n) MDSP_FAIL(MDSP_ERR_DIMENSION, "channel stride too small");
    hipStream_t sm = as_stream(stream);
    const void* src = s_dev;
    int64_t ld = lds_;
    if (demean) {   // mean!(mean_per_channel, signal); demeaned_signal .= signal .- mean_per_channel   (:566-570)
        const size_t esz = dtype_size(st.dtype);
        MDSP_TRY(plan->demeaned.reserve(esz * (size_t)(st.n * nch)));
        if (dtype_is_double(st.dtype))
            hipLaunchKernelGGL(demean_kernel<double>, dim3((unsigned)nch), dim3(256), 0, sm, (const double*)s_dev, plan->demeaned.as<double>(), st.n, lds_);
        else
            hipLaunchKernelGGL(demean_kernel<float>, dim3((unsigned)nch), dim3(256), 0, sm, (const float*)s_dev, plan->demeaned.as<float>(), st.n, lds_);
        MDSP_LAUNCH_CHECK();
        src = plan->demeaned.p;
        ld = st.n;
    }
    // x_mt[:, taper, k] = rfft(window[:, taper] .* signal[k, :])   (mt_fft_tapered_multichannel!, :596-600)
    st.noverlap = 0;
    st.psd_only = 0;
    st.accumulate = 0;
    const size_t csz = dtype_is_double(st.dtype) ? 16 : 8;
    for (int64_t k = 0; k < plan->ntapers; ++k) {
        st.win_ptr = plan->wins.as<double>() + k * st.n;
        char* dst = static_cast<char*>(xmt_dev) + csz * (size_t)(k * st.nout);
        const int rc = stft_dispatch(&st, src, st.n, nch, ld, dst, st.nout, st.nout * plan->ntapers, sm);
        if (rc != MDSP_OK) return rc;
    }
    return MDSP_OK;
}

int mdsp_mt_cross_spectra(mdsp_mt_plan plan, const void* xmt_dev, int64_t nch, const int64_t* freq_inds_host, int64_t nfi, void* out_dev, void* stream) {
    if (!plan) MDSP_FAIL(MDSP_ERR_ARGUMENT, "plan is NULL");
    mdsp_stft_plan_s& st = plan->st;
    if (dtype_is_complex(st.dtype) || !st.onesided)
        MDSP_FAIL(MDSP_ERR_ARGUMENT, "Only real data is supported (with the default choice of `onesided=true`) for this operation.");
    if (nch <= 0 || nfi <= 0) return MDSP_OK;
    if (!xmt_dev || !out_dev || !freq_inds_host) MDSP_FAIL(MDSP_ERR_ARGUMENT, "NULL buffer");
    for (int64_t i = 0; i < nfi; ++i)
        if (freq_inds_host[i] < 0 || freq_inds_host[i] >= st.nout) MDSP_FAIL(MDSP_ERR_ARGUMENT, "frequency index out of range");   // @boundscheck :605-606
    hipStream_t sm = as_stream(stream);
    MDSP_TRY(plan->finds.reserve(sizeof(int64_t) * (size_t)nfi));
    MDSP_HIP(hipMemcpyAsync(plan->finds.p, freq_inds_host, sizeof(int64_t) * (size_t)nfi, hipMemcpyHostToDevice, sm));
    MDSP_HIP(hipStreamSynchronize(sm));   // the host index vector may go away after the call
    const int64_t total = nch * nch * nfi;
    const dim3 g((unsigned)cdiv(total, 256));
    if (dtype_is_double(st.dtype))
        hipLaunchKernelGGL(cross_spectra_kernel<double>, g, dim3(256), 0, sm, (const cx<double>*)xmt_dev, plan->wts.as<double>(), plan->finds.as<int64_t>(),
                           (cx<double>*)out_dev, st.nout, plan->ntapers, nch, nfi, (int)(st.nfft % 2 == 0));
    else
        hipLaunchKernelGGL(cross_spectra_kernel<float>, g, dim3(256), 0, sm, (const cx<float>*)xmt_dev, plan->wts.as<double>(), plan->finds.as<int64_t>(),
                           (cx<float>*)out_dev, st.nout, plan->ntapers, nch, nfi, (int)(st.nfft % 2 == 0));
    MDSP_LAUNCH_CHECK();
    return MDSP_OK;
}

int mdsp_coherence_from_cs(const void* cs_dev, int64_t nch, int64_t nf, int real_dtype, void* out_dev, void* stream) {
    if (real_dtype != MDSP_F32 && real_dtype != MDSP_F64) MDSP_FAIL(MDSP_ERR_ARGUMENT, "real dtype expected");
    if (nch <= 0 || nf <= 0) return MDSP_OK;
    if (!cs_dev || !out_dev) MDSP_FAIL(MDSP_ERR_ARGUMENT, "NULL buffer");
    const dim3 g((unsigned)cdiv(nch * nch * nf, 256));
    if (real_dtype == MDSP_F64)
        hipLaunchKernelGGL(coherence_kernel<double>, g, dim3(256), 0, as_stream(stream), (const cx<double>*)cs_dev, (double*)out_dev, nch, nf);
    else
        hipLaunchKernelGGL(coherence_kernel<float>, g, dim3(256), 0, as_stream(stream), (const cx<float>*)cs_dev, (float*)out_dev, nch, nf);
    MDSP_LAUNCH_CHECK();
    return MDSP_OK;
}

}  // extern "C"

// ======================================================================================================
// hilbert(x) (src/util.jl:31-87): analytic signal along the first dimension
//   X = rfft(x);  X[2 : N/2 + isodd(N)] *= 2;  X[N/2 + 2 : N] = 0;  out = bfft(X) / N
// Whole-column transforms of arbitrary length: batched rocFFT (R2C, then an in-place inverse C2C).
// ======================================================================================================
namespace {

template <typename R>
__global__ __launch_bounds__(256) void hilbert_spectrum_kernel(const cx<R>* __restrict__ half, cx<R>* __restrict__ full, int64_t n, int64_t nspec, int64_t ncols) {
    const int64_t col = blockIdx.x;
    const cx<R>* h = half + col * nspec;
    cx<R>* f = full + col * n;
    const int64_t last2 = n / 2 + (n & 1);   // 1-based indices 2 .. N/2 + isodd(N) are doubled
    for (int64_t i = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.y * blockDim.x) {
        cx<R> v = {(R)0, (R)0};
        if (i < nspec) {
            v = h[i];
            if (i >= 1 && i < last2) v = {v.x * (R)2, v.y * (R)2};
        }
        f[i] = v;
    }
}

template <typename R>
__global__ __launch_bounds__(256) void hilbert_scale_kernel(const cx<R>* __restrict__ in, cx<R>* __restrict__ out, int64_t n, int64_t ldo, double inv_n) {
    const int64_t col = blockIdx.x;
    const R s = (R)inv_n;
    for (int64_t i = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.y * blockDim.x) {
        const cx<R> v = in[col * n + i];
        out[col * ldo + i] = {v.x * s, v.y * s};
    }
}

// Transforms and scratch of the most recent (length, batch) per precision: plan creation for a 2^20-point column costs ~15 ms,
// the transform itself tens of microseconds.  A call holds the entry's lock; a call on another stream first waits for the
// previous user's stream.
struct HilbertCache {
    std::mutex mu;
    int64_t n = 0, batch = 0;
    RocPlan fwd, inv;
    DevBuf xin, half, full;
    hipStream_t last = nullptr;
    bool used = false;
};
HilbertCache& hilbert_cache(bool dbl) {
    static HilbertCache c[2];
    return c[dbl ? 1 : 0];
}

template <typename R> int hilbert_run(const void* x, int64_t n, int64_t ncols, int64_t ldx, void* out, int64_t ldo, hipStream_t st) {
    const int64_t nspec = n / 2 + 1;
    // columns are processed in batches whose intermediates stay within ~256 MiB
    const int64_t per_col = (int64_t)sizeof(R) * n + (int64_t)sizeof(cx<R>) * (nspec + n);
    const int64_t batch = std::max<int64_t>(1, std::min<int64_t>(ncols, (int64_t(256) << 20) / per_col));
    HilbertCache& hc = hilbert_cache(sizeof(R) == 8);
    std::lock_guard<std::mutex> lk(hc.mu);
    if (hc.used && hc.last != st) MDSP_HIP(hipStreamSynchronize(hc.last));
    hc.last = st;
    hc.used = true;
    DevBuf& xin = hc.xin;
    DevBuf& half = hc.half;
    DevBuf& full = hc.full;
    RocPlan& fwd = hc.fwd;
    RocPlan& inv = hc.inv;
    const bool packed = ldx == n;
    if (hc.n != n || hc.batch != batch) {
        hc.n = hc.batch = 0;
        MDSP_HIP(hipStreamSynchronize(st));   // an earlier call on this stream may still use the buffers about to be replaced
        MDSP_TRY(half.reserve(sizeof(cx<R>) * (size_t)(nspec * batch)));
        MDSP_TRY(full.reserve(sizeof(cx<R>) * (size_t)(n * batch)));
        MDSP_TRY(fwd.create(FftKind::R2C, sizeof(R) == 8, n, batch, false));
        MDSP_TRY(inv.create(FftKind::C2C_INV, sizeof(R) == 8, n, batch, true));
        hc.n = n;
        hc.batch = batch;
    }
    if (!packed) MDSP_TRY(xin.reserve(sizeof(R) * (size_t)(n * batch)));
    const unsigned gx = (unsigned)std::min<int64_t>(cdiv(n, 256), 4096);
    for (int64_t c0 = 0; c0 < ncols; c0 += batch) {
        const int64_t cnt = std::min<int64_t>(batch, ncols - c0);
        const R* src = static_cast<const R*>(x) + c0 * ldx;
        if (!packed) {
            MDSP_HIP(hipMemcpy2DAsync(xin.p, sizeof(R) * (size_t)n, src, sizeof(R) * (size_t)ldx, sizeof(R) * (size_t)n, (size_t)cnt, hipMemcpyDeviceToDevice, st));
            src = xin.as<R>();
        }
        if (cnt < batch) {   // the plans are built for `batch` columns: clear what the tail batch does not fill
            MDSP_HIP(hipMemsetAsync(full.p, 0, sizeof(cx<R>) * (size_t)(n * batch), st));
            if (packed) {     // the tail batch must not read past the caller's buffer
                MDSP_TRY(xin.reserve(sizeof(R) * (size_t)(n * batch)));
                MDSP_HIP(hipMemsetAsync(xin.p, 0, sizeof(R) * (size_t)(n * batch), st));
                MDSP_HIP(hipMemcpyAsync(xin.p, src, sizeof(R) * (size_t)(n * cnt), hipMemcpyDeviceToDevice, st));
                src = xin.as<R>();
            }
        }
        MDSP_TRY(fwd.exec(const_cast<R*>(src), half.p, st));
        hipLaunchKernelGGL(hilbert_spectrum_kernel<R>, dim3((unsigned)cnt, gx), dim3(256), 0, st, half.as<cx<R>>(), full.as<cx<R>>(), n, nspec, cnt);
        MDSP_LAUNCH_CHECK();
        MDSP_TRY(inv.exec(full.p, full.p, st));
        hipLaunchKernelGGL(hilbert_scale_kernel<R>, dim3((unsigned)cnt, gx), dim3(256), 0, st, full.as<cx<R>>(), static_cast<cx<R>*>(out) + c0 * ldo, n, ldo,
                           1.0 / (double)n);
        MDSP_LAUNCH_CHECK();
    }
    return MDSP_OK;   // stream-ordered: the cached scratch is next touched by launches on this stream, or after the wait above
}

}  // namespace

extern "C" int mdsp_hilbert(const void* x_dev, int64_t n, int64_t ncols, int64_t ldx, int real_dtype, void* out_dev, int64_t ldo, void* stream) {
    if (real_dtype != MDSP_F32 && real_dtype != MDSP_F64) MDSP_FAIL(MDSP_ERR_ARGUMENT, "hilbert is defined for real signals");
    if (n < 0 || ncols < 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "negative size");
    if (n == 0 || ncols == 0) return MDSP_OK;
    if (!x_dev || !out_dev) MDSP_FAIL(MDSP_ERR_ARGUMENT, "NULL buffer");
    if (ldx < n || ldo < n) MDSP_FAIL(MDSP_ERR_DIMENSION, "column stride smaller than the column length");
    if (ncols > 65535) MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "more than 65535 columns per call");
    MDSP_TRY(rocfft_ensure_setup());
    return real_dtype == MDSP_F64 ? hilbert_run<double>(x_dev, n, ncols, ldx, out_dev, ldo, as_stream(stream))
                                  : hilbert_run<float>(x_dev, n, ncols, ldx, out_dev, ldo, as_stream(stream));
}
