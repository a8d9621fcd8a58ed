// Mixed-radix fused spectral kernel (included by spectral.hip inside its anonymous namespace): Welch / STFT / spectrogram / periodogram for
// the transform sizes nextfastfft (util.jl:134) produces besides powers of two -- 2^a 3^b 5^c 7^d, e.g. 1000, 1536, 3000 -- which DSP.jl
// uses by default (nfft = nextfastfft(n): periodograms.jl:393, :560, :872).  One persistent kernel:
//
//   window the frame(s) into LDS straight from the signal  ->  mixed-radix Stockham passes LDS <-> LDS (fft_lds.h "mixed-radix transforms")
//   ->  consume the natural-order spectrum from LDS:  |Z|^2 into per-slot Float64 sums (Welch)  or  the output column (STFT / PSD).
//
// Real signals ride two frames per transform (z = w (a + i b)); Welch needs no untangling (|A|^2 + |B|^2 = (|Z[k]|^2 + |Z[N-k]|^2) / 2, folded by
// welch_finalize_kernel modes 3 / 4), the STFT modes untangle A[k] = (Z[k] + conj Z[N-k]) / 2, B[k] = (Z[k] - conj Z[N-k]) / (2i) while reading
// the spectrum back from LDS.  HBM traffic = the signal once (+ frame overlap from L2) and the output once: the algorithmic bytes of SURVEY 8d,
// against ~9x that for the K4 -> rocFFT -> K5/K6 pipeline these sizes used to take.
#pragma once

#ifndef MDSP_GEN_LEAN_FLUSH
#define MDSP_GEN_LEAN_FLUSH 64   // units between flushes of the lean form's Float32 sums (a power of two): as the run-time-schedule kernel (gx_kernels.h)
#endif
struct GenArgs {
    const void* s;
    void* out;             // Welch: double partials [slot][ch][N];  STFT: output matrices
    const void* roots;     // N forward roots
    const double* win;     // n doubles or nullptr
    int64_t len, lds_, K, hop, nch, ldo, chs;
    int64_t units_per_ch;  // frames (complex signals) or frame pairs (real signals)
    int64_t per_slot;      // consecutive units per transform slot (same trip count for every slot)
    int n, N, nout, onesided, psd, accumulate;
    int P, T;              // passes, threads per transform slot (256 / T slots per workgroup)
    int roots_in_lds;      // the root table is staged into LDS behind the transform buffers
    int radix[MDSP_GEN_MAXP], ns[MDSP_GEN_MAXP];
    unsigned divm[MDSP_GEN_MAXP];
    double r;
    const void* winr;      // lean compile-time schedules (CtSched flag 4096): the window in the working precision, N values (ones without a window, zero tail)
};

// bins per thread in the Welch accumulator: 16 up to N = 4096 with T = 256, 32 beyond (template parameter EMAX)

// 7-smooth size the mixed-radix kernel takes: not one of the register-resident power-of-two sizes, two padded LDS buffers per slot
inline bool gen_size_ok(int dtype, int64_t nfft) {
    if (nfft < 2 || nfft > (dtype_is_double(dtype) ? 4096 : 8192)) return false;
    int radix[MDSP_GEN_MAXP], ns[MDSP_GEN_MAXP];
    return fft::gen_schedule((int)nfft, radix, ns) > 0;
}

template <typename R, bool CPLX, int MODE, int GEN_EMAX>   // MODE 0: Welch sums, 1: STFT columns (raw or PSD)
__global__ __launch_bounds__(256) void gen_spectral_kernel(GenArgs a) {
    using TT = std::conditional_t<CPLX, cx<R>, R>;
    extern __shared__ __attribute__((aligned(16))) unsigned char gen_smem[];
    const int T = a.T, N = a.N;
    const int slot = threadIdx.x / T, t = threadIdx.x - slot * T, G = blockDim.x / T;
    // A transform's waves synchronise with s_barrier, which spans the workgroup: one transform per workgroup when T >= 128 (the launcher's
    // choice), and one-wave transforms (T == 64: several per workgroup) need only a scheduling fence -- a wave's DS operations execute in order.
    const bool one_wave = T <= 64;
    auto sync = [&]() {
        if (one_wave) __builtin_amdgcn_wave_barrier();
        else __syncthreads();
    };
    const int region = fft::gen_lds_elems(N);
    cx<R>* bufA = reinterpret_cast<cx<R>*>(gen_smem) + (size_t)slot * 2 * region;
    cx<R>* bufB = bufA + region;
    const cx<R>* roots = static_cast<const cx<R>*>(a.roots);
    if (a.roots_in_lds) {   // the root table (N entries) staged once per workgroup: twiddle fetches become LDS reads instead of scattered L1 / L2 hits
        cx<R>* twl = reinterpret_cast<cx<R>*>(gen_smem) + (size_t)G * 2 * region;
        for (int i = threadIdx.x; i < N; i += blockDim.x) fft::st2(twl + i, roots[i]);
        roots = twl;
        __syncthreads();
    }
    const int64_t ch = blockIdx.y;
    const TT* sc = static_cast<const TT*>(a.s) + ch * a.lds_;
    const int64_t gslot = (int64_t)blockIdx.x * G + slot;
    const int64_t u0 = gslot * a.per_slot;

    double acc[MODE == 0 ? GEN_EMAX : 1];
    if constexpr (MODE == 0) {
#pragma unroll
        for (int i = 0; i < GEN_EMAX; ++i) acc[i] = 0.0;
    }
    for (int64_t it = 0; it < a.per_slot; ++it) {
        const int64_t u = u0 + it;
        const bool live = u < a.units_per_ch;
        const int64_t f0 = CPLX ? u : 2 * u;
        const bool haveB = !CPLX && live && (f0 + 1) < a.K;
        // K4 (periodograms.jl:57-69): frame * window in Float64, rounded once, zero tail; frames lie wholly inside the signal by construction
        const TT* fa = sc + f0 * a.hop;
        constexpr int LU = 8;   // loads of LU elements per thread are in flight together
        for (int i0 = t; i0 < N; i0 += LU * T) {
            TT ra[LU], rb[CPLX ? 1 : LU];
            std::conditional_t<sizeof(R) == 4, float, double> w[LU];   // Float32 signals: window rounded to Float32 first (as welch_half_kernel / stft_pair_kernel do)
#pragma unroll
            for (int u = 0; u < LU; ++u) {
                const int i = i0 + u * T;
                const bool on = live && i < a.n;
                ra[u] = on ? fa[i] : TT{};
                if constexpr (!CPLX) rb[u] = (on && haveB) ? fa[i + a.hop] : TT{};
                w[u] = (on && a.win) ? (std::conditional_t<sizeof(R) == 4, float, double>)a.win[i] : 1;
            }
#pragma unroll
            for (int u = 0; u < LU; ++u) {
                const int i = i0 + u * T;
                if (i < N) {
                    cx<R> z;
                    if constexpr (CPLX) z = {ra[u].x * w[u], ra[u].y * w[u]};
                    else z = {ra[u] * w[u], rb[u] * w[u]};
                    fft::st2(bufA + fft::gen_pad(i), z);
                }
            }
        }
        sync();
        cx<R>*src = bufA, *dst = bufB;
        for (int p = 0; p < a.P; ++p) {
            fft::gen_pass_dispatch(a.radix[p], src, dst, roots, N, a.ns[p], a.divm[p], t, T);
            sync();
            cx<R>* tmp = src;
            src = dst;
            dst = tmp;
        }
        if constexpr (MODE == 0) {   // K5: |Z|^2 in the working precision (one rounding per term), accumulated over frames in double
#pragma unroll
            for (int i = 0; i < GEN_EMAX; ++i) {
                const int k = t + T * i;
                if (k < N && live) {
                    const cx<R> z = fft::ld2(src + fft::gen_pad(k));
                    acc[i] += (double)(z.x * z.x + z.y * z.y);
                }
            }
        } else if (live) {
            const R m1 = (R)(1.0 / a.r), m2 = (R)(2.0 / a.r);
            const int64_t o0 = ch * a.chs + f0 * a.ldo;
            for (int j = t; j < a.nout; j += T) {
                if constexpr (CPLX) {   // two-sided only (a complex signal has no one-sided form, periodograms.jl:876)
                    const cx<R> z = fft::ld2(src + fft::gen_pad(j));
                    if (a.psd) {
                        R* o = static_cast<R*>(a.out) + o0 + j;
                        const R pw = z.x * z.x + z.y * z.y;
                        *o = a.accumulate ? fma(pw, m1, *o) : pw * m1;       // fft2pow!: out = muladd(abs2, m, out)
                    } else static_cast<cx<R>*>(a.out)[o0 + j] = z;
                } else {
                    const bool mirror = j > N / 2;                              // real -> two-sided: X[N-k] = conj(X[k]) (fft2oneortwosided!, :234-244)
                    const int k = mirror ? N - j : j;
                    const cx<R> zk = fft::ld2(src + fft::gen_pad(k)), zm = fft::ld2(src + fft::gen_pad(k == 0 ? 0 : N - k));
                    cx<R> A = {(R)0.5 * (zk.x + zm.x), (R)0.5 * (zk.y - zm.y)};   // (Z[k] + conj Z[N-k]) / 2
                    cx<R> B = {(R)0.5 * (zk.y + zm.y), (R)0.5 * (zm.x - zk.x)};   // (Z[k] - conj Z[N-k]) / (2i)
                    if (a.psd) {
                        R m = m1;
                        if (a.onesided && !(j == 0 || (j == a.nout - 1 && N % 2 == 0))) m = m2;
                        R* o = static_cast<R*>(a.out) + o0 + j;
                        const R pa = A.x * A.x + A.y * A.y;
                        *o = a.accumulate ? fma(pa, m, *o) : pa * m;
                        if (haveB) {
                            const R pb = B.x * B.x + B.y * B.y;
                            o[a.ldo] = a.accumulate ? fma(pb, m, o[a.ldo]) : pb * m;
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
        }
        sync();   // the spectrum buffer may be the one the next frame is windowed into
    }
    if constexpr (MODE == 0) {
        double* part = static_cast<double*>(a.out) + (gslot * a.nch + ch) * (int64_t)N;
#pragma unroll
        for (int i = 0; i < GEN_EMAX; ++i) {
            const int k = t + T * i;
            if (k < N) part[k] = acc[i];
        }
    }
}

// ---- compile-time schedules for the common nextfastfft sizes (round 3) ------------------------------------------------------------------
// gen_spectral_kernel above pays for its generality: radices, strides and the butterfly count per pass are run-time values, so every LDS
// access carries index arithmetic (padding, a multiply-shift division), every butterfly fetches its twiddles from a table, and nothing
// unrolls -- ~8x the VALU instructions per point of the power-of-two kernels, 0.05 of the HBM roof at nfft = 1000 / 1536 / 3000
// (profiles/r02g_mixed.json), a 6-7x cliff next to 1024 / 2048 / 4096 for what is the DEFAULT nfft of periodogram / welch_pgram / stft
// (nfft = nextfastfft(n), periodograms.jl:393, :560, :872).  gen_ct_kernel is the same algorithm with the schedule as template constants:
//   * passes fully unrolled, M(p) = ceil(N / (R_p T)) butterflies per thread and pass; operand reads are  t*8 + immediate  (one address VGPR),
//     the scatter index needs one constant division per butterfly;
//   * a thread runs the SAME butterflies for every frame, so its twiddles are loop invariants: they live in registers for the whole launch
//     (30-40 complex values), no table reads at all;
//   * no LDS padding: the FIRST pass has an odd radix (3 or 5), whose scatter stride spreads the 16 lanes of a ds_write_b64 group over all
//     32 banks by itself, and every read is contiguous by lane.
#include "ct_sched.h"   // struct CtSched<N, T, flags, radices...>, MDSP_GEN_CT_SIZES, MDSP_GEN_CT_WIDE_SIZES (host-testable)

// two-level twiddle tables in LDS (CtSched::TW2L): lo[i] = W^i, i < 128; hi[i] = W^{128 i}
template <typename R> struct CtTw {
    const cx<R>*lo, *hi;
};
// twiddle q of the butterfly whose index inside its group is k, pass p: W_N^{q k N / (Ns R)} (< N: no reduction needed)
template <typename S, int p, typename R>
__device__ __forceinline__ cx<R> ct_tw(const cx<R> (&tw)[S::NTW], CtTw<R> t2, int m, int q, unsigned k) {
    if constexpr (!S::TW2L) return tw[S::twoff(p) + m * (S::radix(p) - 1) + (q - 1)];
    else {
        const unsigned e = (unsigned)q * k * (unsigned)(S::N / (S::ns(p) * S::radix(p)));
        return fft::cmul(fft::ld2(t2.hi + (e >> 7)), fft::ld2(t2.lo + (e & 127u)));
    }
}

// The twiddles of one butterfly applied to its operands 1 .. R-1.  Table form with CtSched::TWD (round 6): W^{q e1}, q = C a + b, is the product of
// W^{b e1} (b < C) and W^{C a e1} -- about 2 sqrt(R) table values (two LDS reads and a product each) instead of R - 1, the same number of complex
// products per butterfly: a radix-32 pass reads 20 table entries instead of 62 (the single-workgroup kernels of 8193 .. 16384 points run two waves per
// SIMD at LDS 26 - 36 % and VALU 41 - 46 % busy: they wait on LDS latency, profiles/r06_ctbig_lean.json).
constexpr int ct_twd_c(int r) { return r <= 6 ? r : r <= 9 ? 3 : r <= 16 ? 4 : r <= 25 ? 5 : 6; }
template <typename S, int p, typename R>
__device__ __forceinline__ void ct_apply_twiddles(cx<R> (&v)[S::radix(p)], const cx<R> (&tw)[S::NTW], CtTw<R> t2, int m, unsigned k) {
    constexpr int Rdx = S::radix(p);
    if constexpr (S::TW2L && S::TWD && (Rdx > 6)) {
        constexpr int C = ct_twd_c(Rdx), A = (Rdx + C - 1) / C;
        const unsigned e1 = k * (unsigned)(S::N / (S::ns(p) * Rdx));
        auto table = [&](unsigned e) __attribute__((always_inline)) { return fft::cmul(fft::ld2(t2.hi + (e >> 7)), fft::ld2(t2.lo + (e & 127u))); };
        cx<R> lo_[C], hi_[A];
#pragma unroll
        for (int b = 1; b < C; ++b) lo_[b] = table((unsigned)b * e1);
#pragma unroll
        for (int a = 1; a < A; ++a) hi_[a] = table((unsigned)(C * a) * e1);
#pragma unroll
        for (int q = 1; q < Rdx; ++q) {
            const int a = q / C, b = q % C;
            const cx<R> w = a == 0 ? lo_[b] : b == 0 ? hi_[a] : fft::cmul(hi_[a], lo_[b]);
            v[q] = fft::cmul(v[q], w);
        }
    } else {
#pragma unroll
        for (int q = 1; q < Rdx; ++q) v[q] = fft::cmul(v[q], ct_tw<S, p>(tw, t2, m, q, k));
    }
}

template <typename S, int p, typename R> __device__ __forceinline__ void ct_load_twiddles(cx<R> (&tw)[S::NTW], const cx<R>* roots, int t) {
    if constexpr (p < S::P && !S::TW2L) {
        if constexpr (p > 0) {
            constexpr int Rdx = S::radix(p), Ns = S::ns(p), stride = S::N / (Ns * Rdx);
#pragma unroll
            for (int m = 0; m < S::M(p); ++m) {
                const unsigned j = (unsigned)(t + S::T * m), k = j % (unsigned)Ns;
#pragma unroll
                for (int q = 1; q < Rdx; ++q) tw[S::twoff(p) + m * (Rdx - 1) + (q - 1)] = roots[((unsigned)q * k * (unsigned)stride) % (unsigned)S::N];
            }
        }
        ct_load_twiddles<S, p + 1>(tw, roots, t);
    }
}

// passes p .. END-1, ping-ponging between the two buffers (a barrier behind each); returns the buffer the last of them wrote
template <typename S, int p, int END = S::P, typename R>
__device__ __forceinline__ const cx<R>* ct_passes(const cx<R>* in, cx<R>* out, const cx<R> (&tw)[S::NTW], int t, CtTw<R> t2 = CtTw<R>{nullptr, nullptr}) {
    if constexpr (p >= END) return in;
    else {
        constexpr int Rdx = S::radix(p), Ns = S::ns(p), nbf = S::nbf(p), M = S::M(p);
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const unsigned j = (unsigned)(t + S::T * m);
            if ((m + 1) * S::T <= nbf || j < (unsigned)nbf) {   // a compile-time `true` for every trip but a partial last one
                cx<R> v[Rdx];
                const unsigned jb = S::gin(p) ? j + j / (unsigned)(S::gin(p) ? S::gin(p) : 1) : j;
#pragma unroll
                for (int q = 0; q < Rdx; ++q) v[q] = fft::ld2(in + jb + S::rstride(p) * q);
                const unsigned hi = j / (unsigned)Ns, k = j - hi * (unsigned)Ns;
                if constexpr (p > 0) {
                    ct_apply_twiddles<S, p>(v, tw, t2, m, k);
                }
                fft::gen_bfly<Rdx>(v);
                cx<R>* o = out + hi * (unsigned)(Ns * Rdx + (S::padded(p) ? 1 : 0)) + k;
#pragma unroll
                for (int q = 0; q < Rdx; ++q) fft::st2(o + Ns * q, v[q]);
            }
        }
        __syncthreads();
        return ct_passes<S, p + 1, END>(out, const_cast<cx<R>*>(in), tw, t, t2);
    }
}

// the middle passes on ONE buffer (CtSched::INPLACE): every butterfly of the pass is read into registers, a barrier, then twiddles, butterflies and
// the scatter, a barrier
template <typename S, int p, int END, typename R> __device__ __forceinline__ void ct_passes_inplace(cx<R>* buf, const cx<R> (&tw)[S::NTW], int t, CtTw<R> t2 = CtTw<R>{nullptr, nullptr}) {
    if constexpr (p < END) {
        constexpr int Rdx = S::radix(p), Ns = S::ns(p), nbf = S::nbf(p), M = S::M(p);
        cx<R> v[M][Rdx];
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const unsigned j = (unsigned)(t + S::T * m);
            if ((m + 1) * S::T <= nbf || j < (unsigned)nbf) {
                const unsigned jb = S::gin(p) ? j + j / (unsigned)(S::gin(p) ? S::gin(p) : 1) : j;
#pragma unroll
                for (int q = 0; q < Rdx; ++q) v[m][q] = fft::ld2(buf + jb + S::rstride(p) * q);
            }
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const unsigned j = (unsigned)(t + S::T * m);
            if ((m + 1) * S::T <= nbf || j < (unsigned)nbf) {
                const unsigned hi = j / (unsigned)Ns, k = j - hi * (unsigned)Ns;
                ct_apply_twiddles<S, p>(v[m], tw, t2, m, k);
                fft::gen_bfly<Rdx>(v[m]);
                cx<R>* o = buf + hi * (unsigned)(Ns * Rdx + (S::padded(p) ? 1 : 0)) + k;
#pragma unroll
                for (int q = 0; q < Rdx; ++q) fft::st2(o + Ns * q, v[m][q]);
            }
        }
        __syncthreads();
        ct_passes_inplace<S, p + 1, END>(buf, tw, t, t2);
    }
}

// Pass 0 fed straight from the signal: butterfly j of the first pass reads points j + nbf q, which lanes j = t, t + 1, ... load as contiguous runs --
// no windowed copy of the frame through LDS.  w0[m R + q] is the window at those points (0 past n: the zero tail), loop-invariant per thread.
// (round 4: through buffer descriptors -- one per frame, `n` elements long, empty for a frame that does not exist -- so a load is one VGPR offset
// plus a constant and the zero tail / missing frames come from the bounds check: the 64-bit address arithmetic, compares and selects of the
// pointer form were a fifth of the kernel's vector instructions)
template <typename S, bool CPLX, typename TT>
__device__ __forceinline__ void ct_pass0_loads(const TT* fa, int64_t hop, bool live, bool haveB, int n, TT (&ra)[S::M(0)][S::radix(0)],
                                               TT (&rb)[CPLX ? 1 : S::M(0)][CPLX ? 1 : S::radix(0)], int t) {
    constexpr int Rdx = S::radix(0), nbf = S::nbf(0), M = S::M(0), SZ = (int)sizeof(TT);
    const __amdgpu_buffer_rsrc_t da = io::make_rsrc(fa, live ? (long long)n * SZ : 0);
    int off = t * SZ;
    asm volatile("" : "+v"(off));   // one offset VGPR; the rest of every address is a constant
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
        for (int q = 0; q < Rdx; ++q) ra[m][q] = io::Ld<TT>::load(da, off + (S::T * m + nbf * q) * SZ);   // lanes past the last butterfly read points nobody uses
    if constexpr (!CPLX) {
        const __amdgpu_buffer_rsrc_t db = io::make_rsrc(fa + hop, haveB ? (long long)n * SZ : 0);
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int q = 0; q < Rdx; ++q) rb[m][q] = io::Ld<TT>::load(db, off + (S::T * m + nbf * q) * SZ);
    }
}
template <typename S, typename R, bool CPLX, typename TT>
__device__ __forceinline__ void ct_pass0_compute(const TT (&ra)[S::M(0)][S::radix(0)], const TT (&rb)[CPLX ? 1 : S::M(0)][CPLX ? 1 : S::radix(0)],
                                                 const R (&w0)[S::M(0) * S::radix(0)], cx<R>* out, int t) {
    constexpr int Rdx = S::radix(0), nbf = S::nbf(0), M = S::M(0);
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int j = t + S::T * m;
        if ((m + 1) * S::T <= nbf || j < nbf) {
            cx<R> v[Rdx];
#pragma unroll
            for (int q = 0; q < Rdx; ++q) {
                const R w = w0[m * Rdx + q];
                if constexpr (CPLX) v[q] = {ra[m][q].x * w, ra[m][q].y * w};
                else v[q] = {ra[m][q] * w, rb[m][q] * w};
            }
            fft::gen_bfly<Rdx>(v);
            cx<R>* o = out + (unsigned)j * (unsigned)(Rdx + (S::padded(0) ? 1 : 0));   // Ns = 1: hi = j, k = 0
#pragma unroll
            for (int q = 0; q < Rdx; ++q) fft::st2(o + q, v[q]);
        }
    }
}
// Lean form (CtSched flag 4096, 16 points per thread at up to 1024 threads = 128 registers): the window is not kept in registers -- it is loaded beside the
// samples, from an N-value table in the working precision that stays in L2 -- one butterfly row (m) at a time.
template <typename S, typename R, bool CPLX, typename TT>
__device__ __forceinline__ void ct_pass0_lean(const TT* fa, int64_t hop, bool live, bool haveB, int n, const R* winr, cx<R>* out, int t) {
    constexpr int Rdx = S::radix(0), nbf = S::nbf(0), M = S::M(0), SZ = (int)sizeof(TT), WZ = (int)sizeof(R);
    const __amdgpu_buffer_rsrc_t da = io::make_rsrc(fa, live ? (long long)n * SZ : 0);
    const __amdgpu_buffer_rsrc_t db = io::make_rsrc(fa + (CPLX ? 0 : hop), (!CPLX && haveB) ? (long long)n * SZ : 0);
    const __amdgpu_buffer_rsrc_t dw = io::make_rsrc(winr, (long long)S::N * WZ);
    int off = t * SZ, woff = t * WZ;
    asm volatile("" : "+v"(off), "+v"(woff));
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int j = t + S::T * m;
        TT ra[Rdx], rb[CPLX ? 1 : Rdx];
        R rw[Rdx];
#pragma unroll
        for (int q = 0; q < Rdx; ++q) ra[q] = io::Ld<TT>::load(da, off + (S::T * m + nbf * q) * SZ);
        if constexpr (!CPLX) {
#pragma unroll
            for (int q = 0; q < Rdx; ++q) rb[q] = io::Ld<TT>::load(db, off + (S::T * m + nbf * q) * SZ);
        }
#pragma unroll
        for (int q = 0; q < Rdx; ++q) rw[q] = io::Ld<R>::load(dw, woff + (S::T * m + nbf * q) * WZ);
        if ((m + 1) * S::T <= nbf || j < nbf) {
            cx<R> v[Rdx];
#pragma unroll
            for (int q = 0; q < Rdx; ++q) {
                if constexpr (CPLX) v[q] = {ra[q].x * rw[q], ra[q].y * rw[q]};
                else v[q] = {ra[q] * rw[q], rb[q] * rw[q]};
            }
            fft::gen_bfly<Rdx>(v);
            cx<R>* o = out + (unsigned)j * (unsigned)(Rdx + (S::padded(0) ? 1 : 0));
#pragma unroll
            for (int q = 0; q < Rdx; ++q) fft::st2(o + q, v[q]);
        }
    }
}
template <typename S, typename R, bool CPLX, typename TT>
__device__ __forceinline__ void ct_pass0_global(const TT* fa, int64_t hop, bool live, bool haveB, int n, const R (&w0)[S::M(0) * S::radix(0)], cx<R>* out, int t) {
    TT ra[S::M(0)][S::radix(0)], rb[CPLX ? 1 : S::M(0)][CPLX ? 1 : S::radix(0)];
    ct_pass0_loads<S, CPLX>(fa, hop, live, haveB, n, ra, rb, t);
    ct_pass0_compute<S, R, CPLX>(ra, rb, w0, out, t);
}

// The last pass with its results left in registers: butterfly j produces the bins j + (N / R) q in natural order -- lanes j = t, t + 1, ... own
// contiguous bins, so |Z|^2 sums and complex columns are consumed (and stored, coalesced) without another trip through LDS.
template <typename S, typename R, typename F>
__device__ __forceinline__ void ct_last_pass_regs(const cx<R>* in, const cx<R> (&tw)[S::NTW], int t, CtTw<R> t2, F&& consume) {
    constexpr int p = S::P - 1, Rdx = S::radix(p), nbf = S::nbf(p), M = S::M(p);
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const int j = t + S::T * m;
        if ((m + 1) * S::T <= nbf || j < nbf) {
            cx<R> v[Rdx];
            const unsigned jb = S::gin(p) ? (unsigned)j + (unsigned)j / (unsigned)(S::gin(p) ? S::gin(p) : 1) : (unsigned)j;
#pragma unroll
            for (int q = 0; q < Rdx; ++q) v[q] = fft::ld2(in + jb + S::rstride(p) * q);
            const unsigned k = (unsigned)j % (unsigned)S::ns(p);
            ct_apply_twiddles<S, p>(v, tw, t2, m, k);
            fft::gen_bfly<Rdx>(v);
#pragma unroll
            for (int q = 0; q < Rdx; ++q) consume(m, q, j + nbf * q, v[q]);
        }
    }
}

template <typename R, bool CPLX, int MODE, typename S>   // MODE 0: Welch sums, 1: STFT columns (raw or PSD); one transform per workgroup
__global__ __launch_bounds__(S::T, (MODE == 1 && sizeof(R) == 4) ? S::MINW_REAL : (MODE == 0 && sizeof(R) == 4) ? S::MINW_WELCH : 1) void gen_ct_kernel(GenArgs a) {
    using TT = std::conditional_t<CPLX, cx<R>, R>;
    constexpr int N = S::N, T = S::T;
    constexpr int PL = S::P - 1, RL = S::radix(PL), ML = S::M(PL), NBL = S::nbf(PL);   // the last pass
    constexpr int W0 = S::M(0) * S::radix(0);
    // Welch sums and complex columns take the last pass's results from registers; real-signal columns need the mirror bin N - k of another
    // thread (A[k] = (Z[k] + conj Z[N-k]) / 2), so their last pass goes through LDS once more
    constexpr bool DIRECT = MODE == 0 || CPLX;
    constexpr bool INPL = S::INPLACE && (DIRECT || S::INPLACE_ALL);   // (INPLACE_ALL, round 6: real-signal columns on one buffer too -- the last pass writes the natural-order
                                                                       // spectrum over its own operands, behind the barrier that follows its reads)
    __shared__ __attribute__((aligned(16))) cx<R> buf[INPL ? S::NP : 2 * S::NP];
    cx<R>*bufA = buf, *bufB = INPL ? buf : buf + S::NP;
    const int t = threadIdx.x;
    const int64_t ch = blockIdx.y;
    const TT* sc = static_cast<const TT*>(a.s) + ch * a.lds_;
    const int64_t gslot = blockIdx.x;
    const int64_t u0 = gslot * a.per_slot;
    cx<R> tw[S::NTW];
    ct_load_twiddles<S, 0>(tw, static_cast<const cx<R>*>(a.roots), t);
    __shared__ __attribute__((aligned(16))) cx<R> twlo[S::TW2L ? S::TWS : 1], twhi[S::TW2L ? S::NTWHI : 1];
    const CtTw<R> t2{twlo, twhi};
    if constexpr (S::TW2L) {   // (read behind the first barrier of the frame loop)
        const cx<R>* g = static_cast<const cx<R>*>(a.roots);
        for (int i = t; i < S::TWS; i += T) fft::st2(twlo + i, g[i]);
        for (int i = t; i < S::NTWHI; i += T) fft::st2(twhi + i, g[(unsigned)i * S::TWS]);
    }
    constexpr bool LDSIN = !DIRECT && S::LDSIN;
    // lean form: no per-thread window registers, sums in the working precision flushed to the Float64 partials every MDSP_GEN_LEAN_FLUSH units
    constexpr bool LEAN = S::LEANW && DIRECT && !LDSIN && !S::PREFETCH;   // (window)
    constexpr bool LEANA = S::LEANA && MODE == 0;                              // (sums)
    R w0[LEAN ? 1 : W0];   // window at the points of this thread's first-pass butterflies (Float32 signals: rounded to Float32 first, as the other fused kernels do)
    if constexpr (!LEAN) {
#pragma unroll
        for (int m = 0; m < S::M(0); ++m)
#pragma unroll
            for (int q = 0; q < S::radix(0); ++q) {
                const int i = t + T * m + S::nbf(0) * q;
                w0[m * S::radix(0) + q] = i < a.n ? (a.win ? (R)a.win[i] : (R)1) : (R)0;
            }
    }
    R wl[LDSIN ? S::BINS : 1];   // LDS-input form: the window at i = t + T q
    if constexpr (LDSIN) {
#pragma unroll
        for (int q = 0; q < S::BINS; ++q) {
            const int i = t + T * q;
            wl[q] = i < a.n ? (a.win ? (R)a.win[i] : (R)1) : (R)0;
        }
    }
    std::conditional_t<LEANA, R, double> acc[MODE == 0 ? ML * RL : 1];
    if constexpr (MODE == 0) {
#pragma unroll
        for (int i = 0; i < ML * RL; ++i) acc[i] = 0;
    }
    constexpr int NTOUCH = S::TOUCH ? (N * (int)sizeof(TT) * (CPLX ? 2 : 3) / 2 / 128 + T - 1) / T : 1;   // 128-byte lines of a unit's span per thread
    [[maybe_unused]] float touched[NTOUCH];
    bool flushed = MODE == 0 && a.accumulate != 0;   // (Welch sums: a.accumulate = the partial row holds the sums of an earlier launch, spectral_ctrows.hip)
    auto flush = [&]() __attribute__((always_inline)) {   // the sums of up to MDSP_GEN_LEAN_FLUSH units into this workgroup's own Float64 partial row
        if constexpr (MODE == 0) {
            double* part = static_cast<double*>(a.out) + (gslot * a.nch + ch) * (int64_t)N;
#pragma unroll
            for (int m = 0; m < ML; ++m) {
                const int j = t + T * m;
                if ((m + 1) * T <= NBL || j < NBL) {
#pragma unroll
                    for (int q = 0; q < RL; ++q) {
                        double* o = part + j + NBL * q;
                        *o = flushed ? *o + (double)acc[m * RL + q] : (double)acc[m * RL + q];
                        if constexpr (LEANA) acc[m * RL + q] = 0;
                    }
                }
            }
            flushed = true;
        }
    };
    const R m1 = (R)(1.0 / a.r), m2 = (R)(2.0 / a.r);
    constexpr bool PREF = S::PREFETCH && !LDSIN;
    TT pra[PREF ? S::M(0) : 1][PREF ? S::radix(0) : 1], prb[(PREF && !CPLX) ? S::M(0) : 1][(PREF && !CPLX) ? S::radix(0) : 1];
    auto unit_loads = [&](int64_t it) __attribute__((always_inline)) {
        if constexpr (PREF) {
            const int64_t u = u0 + it;
            const bool live = it < a.per_slot && u < a.units_per_ch;
            const int64_t f0 = live ? (CPLX ? u : 2 * u) : 0;
            const bool haveB = !CPLX && live && (f0 + 1) < a.K;
            ct_pass0_loads<S, CPLX>(sc + f0 * a.hop, a.hop, live, haveB, a.n, pra, prb, t);
            asm volatile("" ::: "memory");   // the loads are issued HERE, in front of the passes that follow
        }
    };
    unit_loads(0);
    for (int64_t it = 0; it < a.per_slot; ++it) {
        const int64_t u = u0 + it;
        const bool live = u < a.units_per_ch;
        const int64_t f0 = CPLX ? u : 2 * u;
        const bool haveB = !CPLX && live && (f0 + 1) < a.K;
        int tl = t;   // lean form: a per-unit copy the compiler cannot see through -- the passes' LDS addresses are recomputed per unit instead of living in (spilled) registers
        if constexpr (LEAN || LEANA) asm volatile("" : "+v"(tl));
        // K4 (periodograms.jl:57-69) fused into the first pass: frame * window, zero tail, straight from the signal
        if constexpr (PREF) {
            ct_pass0_compute<S, R, CPLX>(pra, prb, w0, bufA, tl);
            unit_loads(it + 1);
        } else if constexpr (LEAN) {
            ct_pass0_lean<S, R, CPLX>(sc + f0 * a.hop, a.hop, live, haveB, a.n, static_cast<const R*>(a.winr), bufA, tl);
        }
        else if constexpr (!LDSIN) ct_pass0_global<S, R, CPLX>(sc + f0 * a.hop, a.hop, live, haveB, a.n, w0, bufA, tl);
        else {   // ... or windowed into LDS first (bufB), the first pass then runs LDS -> LDS like the others
            constexpr int BINS = S::BINS;
            const TT* fa = sc + f0 * a.hop;
            TT ra[BINS], rb[CPLX ? 1 : BINS];
#pragma unroll
            for (int q = 0; q < BINS; ++q) {
                const int i = t + T * q;
                const bool on = live && i < a.n;
                ra[q] = on ? fa[i] : TT{};
                if constexpr (!CPLX) rb[q] = (on && haveB) ? fa[i + a.hop] : TT{};
            }
#pragma unroll
            for (int q = 0; q < BINS; ++q) {
                const int i = t + T * q;
                if (BINS * T == N || i < N) {
                    cx<R> z;
                    if constexpr (CPLX) z = {ra[q].x * wl[q], ra[q].y * wl[q]};
                    else z = {ra[q] * wl[q], rb[q] * wl[q]};
                    fft::st2(bufB + i, z);
                }
            }
            __syncthreads();
            (void)ct_passes<S, 0, 1>(bufB, bufA, tw, tl, t2);   // pass 0: bufB -> bufA (ends with a barrier)
        }
        if constexpr (!LDSIN) __syncthreads();
        if constexpr (S::TOUCH && !LDSIN && !S::PREFETCH) {
            // the NEXT unit's samples pulled into the L2 while this unit runs its passes: one dword per 128-byte line, consumed (by nothing) at the end of the
            // unit -- the workgroup fills the LDS, so nothing else overlaps a unit's loads; behind this they come from the L2 instead of HBM
            const int64_t un = u + 1;
            const bool nlive = it + 1 < a.per_slot && un < a.units_per_ch;
            const int64_t fn = CPLX ? un : 2 * un;
            const long long nspan = nlive ? ((long long)a.n + ((!CPLX && fn + 1 < a.K) ? a.hop : 0)) * (long long)sizeof(TT) : 0;
            const __amdgpu_buffer_rsrc_t dn = io::make_rsrc(sc + fn * a.hop, nspan);
#pragma unroll
            for (int i = 0; i < NTOUCH; ++i) touched[i] = io::Ld<float>::load(dn, (tl + T * i) * 128);   // (past the span: the descriptor's zero)
        }
        const int64_t o0 = ch * a.chs + f0 * a.ldo;
        if constexpr (DIRECT) {
            const cx<R>* src = bufA;
            if constexpr (INPL) ct_passes_inplace<S, 1, S::P - 1>(bufA, tw, tl, t2);
            else src = ct_passes<S, 1, S::P - 1>(bufA, bufB, tw, tl, t2);
            ct_last_pass_regs<S>(src, tw, t, t2, [&](int m, int q, int k, cx<R> z) {
                if constexpr (MODE == 0) {   // K5: |Z|^2 in the working precision (one rounding per term), accumulated over frames in double
                    if (live) acc[m * RL + q] += (std::conditional_t<LEANA, R, double>)(z.x * z.x + z.y * z.y);
                } else if (live && k < a.nout) {   // complex signal: two-sided columns (periodograms.jl:876)
                    if (a.psd) {
                        R* o = static_cast<R*>(a.out) + o0 + k;
                        const R pw = z.x * z.x + z.y * z.y;
                        *o = a.accumulate ? fma(pw, m1, *o) : pw * m1;       // fft2pow!: out = muladd(abs2, m, out)
                    } else static_cast<cx<R>*>(a.out)[o0 + k] = z;
                }
            });
        } else {
            const cx<R>* src = bufA;
            if constexpr (INPL) ct_passes_inplace<S, 1, S::P>(bufA, tw, tl, t2);   // (every pass in place, the last one included: ends with a barrier)
            else src = ct_passes<S, 1>(bufA, bufB, tw, tl, t2);   // ends with a barrier; natural-order spectrum in LDS
            if (live) {
                for (int j = t; j < a.nout; j += T) {
                    const bool mirror = j > N / 2;                              // real -> two-sided: X[N-k] = conj(X[k]) (fft2oneortwosided!, :234-244)
                    const int k = mirror ? N - j : j;
                    const cx<R> zk = fft::ld2(src + k), zm = fft::ld2(src + (k == 0 ? 0 : N - k));
                    cx<R> A = {(R)0.5 * (zk.x + zm.x), (R)0.5 * (zk.y - zm.y)};   // (Z[k] + conj Z[N-k]) / 2
                    cx<R> B = {(R)0.5 * (zk.y + zm.y), (R)0.5 * (zm.x - zk.x)};   // (Z[k] - conj Z[N-k]) / (2i)
                    if (a.psd) {
                        R m = m1;
                        if (a.onesided && !(j == 0 || (j == a.nout - 1 && N % 2 == 0))) m = m2;
                        R* o = static_cast<R*>(a.out) + o0 + j;
                        const R pa = A.x * A.x + A.y * A.y;
                        *o = a.accumulate ? fma(pa, m, *o) : pa * m;
                        if (haveB) {
                            const R pb = B.x * B.x + B.y * B.y;
                            o[a.ldo] = a.accumulate ? fma(pb, m, o[a.ldo]) : pb * m;
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
        }
        __syncthreads();   // the buffer the last pass read may be the one the next frame's first pass writes
        if constexpr (S::TOUCH && !LDSIN && !S::PREFETCH) {
#pragma unroll
            for (int i = 0; i < NTOUCH; ++i) asm volatile("" ::"v"(touched[i]));
        }
        if constexpr (LEANA) {
            if ((it & (MDSP_GEN_LEAN_FLUSH - 1)) == MDSP_GEN_LEAN_FLUSH - 1) flush();
        }
    }
    flush();
}

template <typename R, bool CPLX, int MODE, typename S>
int gen_ct_launch(GenArgs& a, int64_t nch, hipStream_t st, int64_t* nslots, DevBuf* partial) {
    auto kern = gen_ct_kernel<R, CPLX, MODE, S>;
    hipFuncAttributes fa{};
    MDSP_HIP(hipFuncGetAttributes(&fa, (const void*)kern));
    const int regs = std::max(8, (fa.numRegs + 7) / 8 * 8), waves = S::T / 64;
    const size_t lds_bytes = sizeof(cx<R>) * ((S::INPLACE && (MODE == 0 || CPLX || S::INPLACE_ALL)) ? 1 : 2) * (size_t)S::NP;
    int per_cu = std::min<int>({32 / waves, (512 / regs) * 4 / waves, (int)((size_t)160 * 1024 / lds_bytes)});
    if (per_cu < 1) per_cu = 1;
    if (tunables().wg_per_cu > 0) per_cu = tunables().wg_per_cu;
    const int64_t resident = std::max<int64_t>(1, (int64_t)device_cu_count() * per_cu / std::max<int64_t>(1, nch));
    const int64_t wgs = std::max<int64_t>(1, std::min<int64_t>(a.units_per_ch, resident));
    *nslots = wgs;
    a.per_slot = cdiv(a.units_per_ch, wgs);
    if (MODE == 0) {
        MDSP_TRY(partial->reserve(sizeof(double) * (size_t)wgs * (size_t)nch * (size_t)S::N));
        a.out = partial->p;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs, (unsigned)nch), dim3(S::T), 0, st, a);
    MDSP_LAUNCH_CHECK();
    return MDSP_OK;
}
inline bool gen_ct_size(int dtype, int64_t nfft, bool direct) {
    if (!MDSP_GEN_CT || (dtype_is_double(dtype) && nfft > std::min<int64_t>(tunables().gen_ct_f64_max, direct ? GEN_CT_F64_MAX : GEN_CT_F64_REAL_COLUMNS_MAX))) return false;
    switch (nfft) {
#define MDSP_X(N, ...) case N:
        MDSP_GEN_CT_SIZES(MDSP_X)
#undef MDSP_X
        return true;
        default: return false;
    }
}
// Float64 schedules that take their twiddles from the two LDS tables (CtSched flag 2048) instead of registers: where it measured faster
// (profiles/r05_f64_twiddles.json, 2^27 samples): Welch at 4800 (0.73 -> 1.21 TB/s), 5120, 6144 and 8000 (0.62 -> 0.83); ComplexF64 columns at 4800, 6000
// (1.69 -> 1.85), 6400 (1.85 -> 2.46) and 8000 (1.68 -> 2.07).  Not at 5000 (columns 3.05 -> 2.38) and not for Welch at 6000 / 6400 (-5 ... -7 %: those
// keep 48 - 82 spilled registers either way -- the single-buffer pass holds all of a pass's butterflies in registers).
// Round 6: with the window loaded beside the samples (flag 4096, ct_pass0_lean) and the twiddles derived from ~2 sqrt(R) table values (16384) NONE of the
// Float64 schedules from 4800 points spills any more (5000: 56 -> 0 spilled registers, 6000: 193 -> 0, 6400: 142 -> 0, 8000: 57 -> 0): the register-
// consumed modes take table twiddles at every size from 4800 points.  -DMDSP_F64_LEAN=0 keeps round 5's choice.
#ifndef MDSP_F64_LEAN
#define MDSP_F64_LEAN 1
#endif
#ifndef MDSP_F64_TOUCH
#define MDSP_F64_TOUCH 0   // (32768 = the next unit touched into the L2: measured for Float64 / ComplexF64 Welch sums at 5120 .. 8000 points, r06s46: within +-2 %, not taken)
#endif
// Measured (profiles/r06_f64_lean.json, 2^25 samples, TB/s): Welch 6000 0.57 -> 1.01, 6400 0.63 -> 1.13, 8000 0.70 -> 1.07, 5120 1.10 -> 1.20, 6144 1.23 -> 1.31
// (4800 and 5000 within 4 %: round 5's form stays); ComplexF64 columns 4800 2.1 -> 2.9, 6000 1.7 -> 2.0, 8000 2.0 -> 2.2 (5000, 5120, 6144 lose: round 5's form).
constexpr int gen_ct_f64_tw2l(int N, int mode, bool cplx = false) {
    if (N < 4800) return 0;
    if (MDSP_F64_LEAN && mode == 0 && N > 5000) return 2048 | 4096 | 16384 | MDSP_F64_TOUCH;
    if (MDSP_F64_LEAN && mode == 1 && cplx && (N == 4800 || N == 6000 || N == 6400 || N == 8000)) return 2048 | 4096 | 16384;
    if (N == 5000) return 0;
    if (mode == 0 && (N == 6000 || N == 6400)) return 0;
    return 2048;
}
#ifndef MDSP_GEN_F32_WELCH_FLAGS
#define MDSP_GEN_F32_WELCH_FLAGS 0   // (A/B: 8192 = Float32 sums flushed every 64 units for the 23 nextfastfft sizes up to 8000 points as well)
#endif
template <typename R, bool CPLX, int MODE>
bool gen_ct_dispatch(GenArgs& a, int64_t nch, hipStream_t st, int64_t* nslots, DevBuf* partial, int* rc) {
    if constexpr (MDSP_GEN_CT && sizeof(R) == 4) {
        if (tunables().gen_wide) {
            switch (a.N) {
#define MDSP_X(N, T, F, ...)                                                                               \
    case N:                                                                                                \
        if constexpr (gen_ct_wide_mode(F, MODE, CPLX)) {                                                   \
            *rc = gen_ct_launch<R, CPLX, MODE, CtSched<N, T, gen_ct_touch(F, MODE) | (MODE == 0 ? MDSP_GEN_F32_WELCH_FLAGS : 0), __VA_ARGS__>>(a, nch, st, nslots, partial); \
            return true;                                                                                   \
        }                                                                                                  \
        break;
                MDSP_GEN_CT_WIDE_SIZES(MDSP_X)
#undef MDSP_X
                default: break;
            }
        }
    }
    if constexpr (MDSP_GEN_CT) {
        switch (a.N) {
#define MDSP_X(N, T, F, ...)                                                                                       \
    case N:                                                                                                        \
        if constexpr (sizeof(R) == 4) {                                                                            \
            *rc = gen_ct_launch<R, CPLX, MODE, CtSched<N, T, gen_ct_flags(F, MODE, CPLX) | (MODE == 0 ? MDSP_GEN_F32_WELCH_FLAGS : 0), __VA_ARGS__>>(a, nch, st, nslots, partial); \
            return true;                                                                                           \
        } else if constexpr (N <= GEN_CT_F64_TWO_BUF) {                                                            \
            *rc = gen_ct_launch<R, CPLX, MODE, CtSched<N, T, (F) & ~(1536 | 32768), __VA_ARGS__>>(a, nch, st, nslots, partial); \
            return true;                                                                                           \
        } else if constexpr (MODE == 0 || CPLX) {                                                                  \
            *rc = gen_ct_launch<R, CPLX, MODE, CtSched<N, T, ((F) & ~(1536 | 32768)) | 16 | gen_ct_f64_tw2l(N, MODE, CPLX), __VA_ARGS__>>(a, nch, st, nslots, partial); \
            return true;                                                                                           \
        } else if constexpr (N <= GEN_CT_F64_REAL_COLUMNS_TWO_BUF) {                                               \
            *rc = gen_ct_launch<R, CPLX, MODE, CtSched<N, T, ((F) & ~(1536 | 32768)) | gen_ct_f64_tw2l(N, MODE), __VA_ARGS__>>(a, nch, st, nslots, partial); \
            return true;                                                                                           \
        } else {   /* real-signal columns beyond: ONE buffer, the last pass leaves the spectrum in place (flag 65536, round 6) */ \
            *rc = gen_ct_launch<R, CPLX, MODE, CtSched<N, T, ((F) & ~(1536 | 32768)) | 16 | 65536 | 2048, __VA_ARGS__>>(a, nch, st, nslots, partial); \
            return true;                                                                                           \
        }                                                                                                          \
        break;
            MDSP_GEN_CT_SIZES(MDSP_X)
#undef MDSP_X
            default: break;
        }
    }
    return false;
}

// geometry + schedule shared by the two launchers; returns the slot count through *nslots
template <typename R, bool CPLX, int MODE, int EMAX>
int gen_launch_e(GenArgs& a, int64_t nch, hipStream_t st, int64_t* nslots, DevBuf* partial) {
    a.P = fft::gen_schedule(a.N, a.radix, a.ns);
    if (a.P <= 0) MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "nfft=%d has a prime factor other than 2, 3, 5, 7", a.N);
    for (int p = 0; p < a.P; ++p) a.divm[p] = (unsigned)(((1u << 24) + a.ns[p] - 1) / a.ns[p]);
    // threads per transform: enough that a thread owns at most GEN_EMAX bins and most lanes have a butterfly in the widest-radix pass
    // Passes are latency chains (LDS read -> twiddle -> butterfly -> LDS write): as many threads per transform as it has butterflies in its
    // widest pass, so a thread walks one or two of them per pass
    // Threads per transform: a thread walks two butterflies per trip (fft_lds.h gen_pass), so T ~ N / 8 keeps every lane busy in the radix-4 /
    // radix-5 passes without a second trip; the Welch form also needs N / T <= EMAX bins per thread.
    int T = a.N <= 512 ? 64 : (a.N <= 1024 ? 128 : 256);
    while (T < 256 && a.N > T * EMAX) T *= 2;
    a.T = T;
    const int G = T <= 64 ? 4 : 1;   // several one-wave transforms per workgroup; otherwise one transform per workgroup (no barrier coupling)
    const int threads = T * G;
    size_t lds_bytes = (size_t)G * 2 * (size_t)fft::gen_lds_elems(a.N) * sizeof(cx<R>);
    a.roots_in_lds = lds_bytes + (size_t)a.N * sizeof(cx<R>) <= 80 * 1024 ? 1 : 0;   // keeps two workgroups per CU
    if (a.roots_in_lds) lds_bytes += (size_t)a.N * sizeof(cx<R>);
    auto kern = gen_spectral_kernel<R, CPLX, MODE, EMAX>;
    if (lds_bytes > 160 * 1024) MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "nfft=%d does not fit the LDS in this precision", a.N);
    if (lds_bytes > 48 * 1024) MDSP_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    // resident workgroups per CU from the kernel's own resources (the occupancy query answered HALF of what the hardware admits for these
    // kernels: one workgroup per CU for the Welch form at 242 VGPRs, two for the STFT form at 119 -- profiles/r02h_gen_pmc.json)
    hipFuncAttributes fa{};
    MDSP_HIP(hipFuncGetAttributes(&fa, (const void*)kern));
    const int regs = std::max(8, (fa.numRegs + 7) / 8 * 8);
    int per_cu = std::min<int>({32 / (threads / 64), (512 / regs) * 4 / (threads / 64), (int)((size_t)160 * 1024 / std::max<size_t>(lds_bytes, 1))});
    if (per_cu < 1) per_cu = 1;
    if (tunables().wg_per_cu > 0) per_cu = tunables().wg_per_cu;
    const int64_t resident = std::max<int64_t>(1, (int64_t)device_cu_count() * per_cu / std::max<int64_t>(1, nch));
    const int64_t wgs = std::max<int64_t>(1, std::min<int64_t>(cdiv(a.units_per_ch, G), resident));
    *nslots = wgs * G;
    a.per_slot = cdiv(a.units_per_ch, *nslots);
    if (MODE == 0) {
        MDSP_TRY(partial->reserve(sizeof(double) * (size_t)(*nslots) * (size_t)nch * (size_t)a.N));
        a.out = partial->p;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs, (unsigned)nch), dim3(threads), lds_bytes, st, a);
    MDSP_LAUNCH_CHECK();
    return MDSP_OK;
}
template <typename R, bool CPLX, int MODE>
int gen_launch(GenArgs& a, int64_t nch, hipStream_t st, int64_t* nslots, DevBuf* partial) {
    int rc = MDSP_OK;
    if (gen_ct_dispatch<R, CPLX, MODE>(a, nch, st, nslots, partial, &rc)) return rc;
    if (MODE == 0 && a.N > 4096) return gen_launch_e<R, CPLX, MODE, 32>(a, nch, st, nslots, partial);
    return gen_launch_e<R, CPLX, MODE, 16>(a, nch, st, nslots, partial);
}
