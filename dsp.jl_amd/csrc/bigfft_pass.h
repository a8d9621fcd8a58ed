// Large transforms (nfft above the one-workgroup sizes): the pass of a multi-pass ("four-step") FFT whose every pass is fused with its
// neighbours' work -- window and framing in the first, inter-pass twiddles in the store of each, |Z|^2 sums or output columns in the last.
//
// Why: periodogram / welch_pgram / spectrogram / stft default to n = length(s) >> 3, nfft = nextfastfft(n) (periodograms.jl:560, :647, :828,
// :872) -- every signal longer than 65 536 samples called with defaults asks for a transform that no workgroup holds.  Until round 5 those went
// to frame_window_kernel -> rocFFT -> abs2_accum_kernel (0.2-0.3 TB/s of the 4 B/sample the call needs).
//
// Decomposition (decimation in time, in-place index): N = R_0 R_1 ... R_{P-1}, P = 2..4, every R_p <= 512.  With S_p = R_{p+1} ... R_{P-1},
// N_p = R_p S_p, input index n = sum n_p S_p and output index k = sum k_p M_p (M_p = R_0 ... R_{p-1}):
//
//     pass p:  A_p[k_0 .. k_p ; n_{p+1} ..] = W_{N_p}^{k_p n'} * sum_{n_p} A_{p-1}[k_0 .. k_{p-1} ; n_p, n'] W_{R_p}^{n_p k_p},   n' = sum_{q>p} n_q S_q
//
// stored where its operands were (k_p takes the place of n_p), so one buffer of N elements per transform serves every pass and a workgroup
// only ever touches its own tile: R_p rows x B columns, columns = B consecutive n' (rows of B elements, S_p apart).  The last pass has S = 1:
// its "columns" are whole contiguous runs of R_{P-1} elements, taken B at a time with consecutive k_0, so that its results -- natural index
// k = k_0 + R_0 (...) + M_{P-1} k_{P-1} -- leave as rows of B consecutive bins.
//
// Inside a workgroup the B sub-transforms of length R_p run LDS -> LDS (Stockham, radices 2..16 from gen_bfly, fft_lds.h) on a tile kept as
// [row][column] with B + 1 elements per row: every access has the 16 (8) lanes of a butterfly on consecutive columns, and the transposed
// accesses of the last pass (lanes along a row) fall on distinct banks because the row pitch is odd.
//
// Plain C++ (MDSP_HD): tests/cpu_harness/bigfft_emul.cpp runs the same phases thread by thread on the host against a Float64 DFT.
#pragma once

#include <cstdint>

#include "fft_lds.h"

namespace mdsp {
namespace big {

using fft::cx;

constexpr int TPB = 256;       // threads per workgroup
constexpr int MAXP = 4;        // passes of the large transform
constexpr int MAXSUB = 8;      // passes of a sub-transform
constexpr int RMAX = 512;      // longest sub-transform
template <typename R> constexpr int cols() { return sizeof(R) == 4 ? 16 : 8; }             // B: columns per tile (128-byte rows either way)
template <typename R> constexpr int elems() { return RMAX * cols<R>() / TPB; }             // elements per thread at most (32 / 16); sub-transforms up to 256 points need half

struct Pass {
    int64_t N;            // transform length
    int64_t Sp, Np;       // columns per prefix block, R_p S_p
    int64_t tpp, ntiles;  // tiles per prefix block (last pass: per value of `rest`), tiles per transform
    int64_t Q;            // last pass: runs per k_0 = N / (R_0 R_p)
    int Rp, R0, last;
    int B;                // columns per tile: cols<R>() for the generic phases, 256 / TJ for the two-stage form
    int fRA, fTJ;         // two-stage form: R_p = fRA x fTJ (0: generic phases)
    unsigned divR;        // ceil(2^24 / Rp): idx div Rp == (idx * divR) >> 24 for idx < 8192
    int nd;               // last pass: digits of `rest` (k_{P-2} fastest .. k_1), their radices and natural weights M_q
    int dR[MAXP];
    int64_t dM[MAXP];
    int nsub, radix[MAXSUB], ns[MAXSUB];
    unsigned divm[MAXSUB];
    int logS, nT1;        // twiddles behind this pass (not the last): W_{N_p}^m = T1[m >> logS] * T0[m & (2^logS - 1)]
    const void *T0, *T1, *roots;   // roots: R_p forward roots of the sub-transform
};

struct Tile {
    int64_t base, row_stride, col_stride, c0, nat;
    int ncols;   // columns of the tile that exist
};

template <typename R> MDSP_HD Tile tile_of(const Pass& p, int64_t tile) {
    constexpr int B = cols<R>();
    Tile t;
    if (!p.last) {
        const int64_t prefix = tile / p.tpp, c0 = (tile - prefix * p.tpp) * B;
        t.base = prefix * p.Np + c0;
        t.row_stride = p.Sp;
        t.col_stride = 1;
        t.c0 = c0;
        t.nat = 0;
        t.ncols = (int)((p.Sp - c0) < B ? (p.Sp - c0) : B);
    } else {
        const int64_t rest = tile / p.tpp, k00 = (tile - rest * p.tpp) * B;
        t.base = (k00 * p.Q + rest) * p.Rp;
        t.row_stride = 1;
        t.col_stride = p.Q * p.Rp;
        t.c0 = k00;
        int64_t rem = rest, nat = 0;
        for (int d = 0; d < p.nd; ++d) {
            const int64_t dig = rem % p.dR[d];
            rem /= p.dR[d];
            nat += dig * p.dM[d];
        }
        t.nat = nat;
        t.ncols = (int)((p.R0 - k00) < B ? (p.R0 - k00) : B);
    }
    return t;
}

// ---- phase 1: the tile into registers, then into LDS as [row][B + 1] ---------------------------------------------------------------------
// A thread's element e (E of them, E * TPB >= R_p * B):  not the last pass: column b = tid mod B, row r = tid div B + (TPB / B) e -- lanes along
// the columns, rows of B consecutive elements;  the last pass: idx = tid + TPB e, b = idx div R_p, r = idx mod R_p -- lanes along a run.
// The two halves are separate so that a kernel can have the NEXT tile's loads in flight while it transforms this one.
template <typename R> MDSP_HD bool elem_of(const Pass& p, int tid, int e, int& r, int& b) {
    constexpr int B = cols<R>();
    if (!p.last) {
        b = tid & (B - 1);
        r = tid / B + (TPB / B) * e;
        return r < p.Rp;
    }
    const int idx = tid + TPB * e;
    b = (int)(((unsigned long long)(unsigned)idx * p.divR) >> 24);
    r = idx - b * p.Rp;
    return idx < p.Rp * B;
}
// get(pos) -> cx<R>: element `pos` of the transform's input (the work buffer, or the windowed frames)
template <typename R, int E, typename F> MDSP_HD void load_regs(const Pass& p, const Tile& t, int tid, cx<R> (&v)[E], F&& get) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        int r, b;
        const bool in = elem_of<R>(p, tid, e, r, b);
        v[e] = {(R)0, (R)0};
        if (in && b < t.ncols) v[e] = get(t.base + (int64_t)r * t.row_stride + (int64_t)b * t.col_stride);
    }
}
template <typename R, int E> MDSP_HD void regs_to_lds(const Pass& p, int tid, const cx<R> (&v)[E], cx<R>* lds) {
    constexpr int Bp = cols<R>() + 1;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        int r, b;
        if (elem_of<R>(p, tid, e, r, b)) fft::st2(lds + r * Bp + b, v[e]);
    }
}

// ---- twiddles behind a pass: W_{N_p}^{(c0 + b) r} = W^{c0 r} W^{b r} ------------------------------------------------------------------------
// W^{b r} belongs to the thread (its b and rows never change): E registers, filled once per kernel.  W^{c0 r} belongs to the tile: R_p values,
// one table walk per THREAD and tile, shared through LDS -- instead of two scattered table reads per ELEMENT (which kept the L1 busier than the
// data did: profiles/r05_bigfft_first.txt).
template <typename R> MDSP_HD cx<R> big_root(const Pass& p, unsigned m) {   // W_{N_p}^m, m < N_p
    const cx<R>*T0 = static_cast<const cx<R>*>(p.T0), *T1 = static_cast<const cx<R>*>(p.T1);
    cx<R> w = T0[m & ((1u << p.logS) - 1u)];
    if (p.nT1 > 1) w = fft::cmul(w, T1[m >> p.logS]);
    return w;
}
template <typename R, int E> MDSP_HD void load_twb(const Pass& p, int tid, cx<R> (&twb)[E]) {
    constexpr int B = cols<R>();
    const unsigned b = (unsigned)(tid & (B - 1));
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const unsigned r = (unsigned)(tid / B + (TPB / B) * e);
        twb[e] = big_root<R>(p, r < (unsigned)p.Rp ? b * r : 0u);
    }
}
template <typename R> MDSP_HD void fill_twc(const Pass& p, const Tile& t, int tid, cx<R>* twc) {
    for (int r = tid; r < p.Rp; r += TPB) fft::st2(twc + r, big_root<R>(p, (unsigned)t.c0 * (unsigned)r));   // c0 r < S_p R_p = N_p < 2^31
}

// ---- phase 2: one pass of the B sub-transforms, LDS -> LDS -----------------------------------------------------------------------------
template <int RDX, typename R>
MDSP_HD void sub_pass(const cx<R>* in, cx<R>* out, const cx<R>* roots, int Rp, int Ns, unsigned divm, int tid) {
    constexpr int B = cols<R>(), Bp = B + 1, TJ = TPB / B;
    const int nbf = Rp / RDX, stride = Rp / (Ns * RDX);
    const int b = tid & (B - 1), tj = tid / B;
    for (int j = tj; j < nbf; j += TJ) {
        cx<R> v[RDX];
#pragma unroll
        for (int q = 0; q < RDX; ++q) v[q] = fft::ld2(in + (j + nbf * q) * Bp + b);
        const int hi = Ns == 1 ? j : (int)(((unsigned long long)(unsigned)j * divm) >> 24);
        const int k = j - hi * Ns;
        if (Ns > 1) {
            const int idx = k * stride;   // q * idx < Rp for q < RDX
#pragma unroll
            for (int q = 1; q < RDX; ++q) v[q] = fft::cmul(v[q], fft::ld2(roots + q * idx));
        }
        fft::gen_bfly<RDX>(v);
        const int base = hi * Ns * RDX + k;
#pragma unroll
        for (int q = 0; q < RDX; ++q) fft::st2(out + (base + Ns * q) * Bp + b, v[q]);
    }
}
template <typename R> MDSP_HD void phase_sub(const Pass& p, int sp, int tid, const cx<R>* in, cx<R>* out, const cx<R>* roots) {
    const int Ns = p.ns[sp];
    const unsigned dm = p.divm[sp];
    switch (p.radix[sp]) {
        case 16: sub_pass<16>(in, out, roots, p.Rp, Ns, dm, tid); break;
        case 8: sub_pass<8>(in, out, roots, p.Rp, Ns, dm, tid); break;
        case 4: sub_pass<4>(in, out, roots, p.Rp, Ns, dm, tid); break;
        case 2: sub_pass<2>(in, out, roots, p.Rp, Ns, dm, tid); break;
        case 3: sub_pass<3>(in, out, roots, p.Rp, Ns, dm, tid); break;
        case 5: sub_pass<5>(in, out, roots, p.Rp, Ns, dm, tid); break;
        case 7: sub_pass<7>(in, out, roots, p.Rp, Ns, dm, tid); break;
        case 6: sub_pass<6>(in, out, roots, p.Rp, Ns, dm, tid); break;
        case 10: sub_pass<10>(in, out, roots, p.Rp, Ns, dm, tid); break;
        case 12: sub_pass<12>(in, out, roots, p.Rp, Ns, dm, tid); break;
        case 15: sub_pass<15>(in, out, roots, p.Rp, Ns, dm, tid); break;
        default: break;
    }
}

// ---- phase 3: results out of LDS, lanes along the columns ------------------------------------------------------------------------------
// not the last pass:  put(e, pos, z)  with the twiddle W_{N_p}^{(c0 + b) r} = twc[r] twb[e] applied, pos = the element's own place in the work buffer
// the last pass:      put(e, k, z)    with k the natural index of the bin
// (`e` is a compile-time constant for the caller: the register accumulators of the Welch form)
template <typename R, int E, typename F>
MDSP_HD void phase_store(const Pass& p, const Tile& t, int tid, const cx<R>* lds, const cx<R>* twc, const cx<R> (&twb)[E], F&& put) {
    constexpr int B = cols<R>(), Bp = B + 1, TJ = TPB / B;
    const int b = tid & (B - 1), tj = tid / B;
    if (b >= t.ncols) return;
    const int64_t rs = p.last ? p.N / p.Rp : t.row_stride;
    const int64_t o0 = p.last ? t.c0 + b + t.nat : t.base + b;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int r = tj + TJ * e;
        if (r < p.Rp) {
            cx<R> z = fft::ld2(lds + r * Bp + b);
            if (!p.last) z = fft::cmul(z, fft::cmul(fft::ld2(twc + r), twb[e]));
            put(e, o0 + (int64_t)r * rs, z);
        }
#if defined(__HIP_DEVICE_COMPILE__)
        if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // four elements in flight at a time (their registers come on top of accumulators and prefetched samples)
#endif
    }
}

// ================================================================================================ two-stage sub-transforms in registers
// The generic phases above move every element through LDS four times (tile in, two or three sub-passes, tile out) with a barrier behind each
// step, and a workgroup spent two thirds of its time waiting (profiles/r05_bigfft_first.txt).  For R_p = RA x TJ -- 256 = 16 x 16, 128 = 8 x 16,
// 64 = 8 x 8, 32 = 4 x 8 -- the Stockham pair needs ONE exchange, as in the single-workgroup kernels (fft_lds.h):
//   * a column's TJ threads (B = 256 / TJ columns per tile) each hold rows tj + TJ q, q < RA, straight from their coalesced loads: those are the
//     operands of butterfly tj of the first stage (radix RA) -- no staging;
//   * its results go to LDS rows tj RA + q; after the one barrier, thread tj < RA reads rows tj + RA q, q < TJ, multiplies by W_{R_p}^{q tj}
//     (loop-invariant per thread) and runs butterfly tj of the second stage (radix TJ): it now holds rows tj + RA q' in natural order, which
//     leave for memory from registers with the inter-pass twiddle applied.
// The last pass reads its runs coalesced along the run, so its elements take one extra trip through LDS to reach the owner of their row.
// RA > TJ (128 = 16 x 8, Float32): a thread runs NS2 = RA / TJ second-stage butterflies, tj + TJ s -- every thread busy in both stages and tiles of
// 256 / 8 = 32 columns; the 8 x 16 form of the same pass keeps half of its threads for the second stage and has 16-column tiles (the slowest column
// pass of every table in profiles/r05_big_ols.json).  NO: a thread's results, NS2 x TJ.
template <int RA, int TJ> struct FastGeo {
    static constexpr int Rp = RA * TJ, B = TPB / TJ, Bp = B + 1;
    static constexpr int NS2 = RA > TJ ? RA / TJ : 1, NO = NS2 * TJ;
    static_assert(TPB % TJ == 0 && (RA <= TJ || RA % TJ == 0), "stage 2 runs on the first RA threads of a column, or every thread runs RA / TJ butterflies");
};

template <typename R> MDSP_HD Tile tile_of_b(const Pass& p, int64_t tile, int B) {
    Tile t;
    if (!p.last) {
        const int64_t prefix = tile / p.tpp, c0 = (tile - prefix * p.tpp) * B;
        t.base = prefix * p.Np + c0;
        t.row_stride = p.Sp;
        t.col_stride = 1;
        t.c0 = c0;
        t.nat = 0;
        t.ncols = (int)((p.Sp - c0) < B ? (p.Sp - c0) : B);
    } else {
        const int64_t rest = tile / p.tpp, k00 = (tile - rest * p.tpp) * B;
        t.base = (k00 * p.Q + rest) * p.Rp;
        t.row_stride = 1;
        t.col_stride = p.Q * p.Rp;
        t.c0 = k00;
        int64_t rem = rest, nat = 0;
        for (int d = 0; d < p.nd; ++d) {
            const int64_t dig = rem % p.dR[d];
            rem /= p.dR[d];
            nat += dig * p.dM[d];
        }
        t.nat = nat;
        t.ncols = (int)((p.R0 - k00) < B ? (p.R0 - k00) : B);
    }
    return t;
}

// a thread's RA input elements.  Not the last pass: (row tj + TJ q, column b);  the last pass: element idx = tid + TPB q of the tile counted
// along the runs (column idx div R_p, row idx mod R_p)
template <typename R, int RA, int TJ, typename F> MDSP_HD void fast_load(const Pass& p, const Tile& t, int tid, cx<R> (&v)[RA], F&& get) {
    using G = FastGeo<RA, TJ>;
#pragma unroll
    for (int q = 0; q < RA; ++q) {
        int r, b;
        if (!p.last) {
            b = tid % G::B;
            r = tid / G::B + TJ * q;
        } else {
            const int idx = tid + TPB * q;
            b = idx / G::Rp;
            r = idx % G::Rp;
        }
        v[q] = {(R)0, (R)0};
        if (b < t.ncols) v[q] = get(t.base + (int64_t)r * t.row_stride + (int64_t)b * t.col_stride);
    }
}
// the last pass only: elements to the owners of their rows through `stage` (a barrier between the two halves)
template <typename R, int RA, int TJ> MDSP_HD void fast_stage_put(int tid, const cx<R> (&v)[RA], cx<R>* stage) {
    using G = FastGeo<RA, TJ>;
#pragma unroll
    for (int q = 0; q < RA; ++q) {
        const int idx = tid + TPB * q;
        fft::st2(stage + (idx % G::Rp) * G::Bp + idx / G::Rp, v[q]);
    }
}
template <typename R, int RA, int TJ> MDSP_HD void fast_stage_get(int tid, cx<R> (&v)[RA], const cx<R>* stage) {
    using G = FastGeo<RA, TJ>;
    const int b = tid % G::B, tj = tid / G::B;
#pragma unroll
    for (int q = 0; q < RA; ++q) v[q] = fft::ld2(stage + (tj + TJ * q) * G::Bp + b);
}
// stage 1: butterfly tj (radix RA) on the thread's own rows; results to LDS rows tj RA + q
template <typename R, int RA, int TJ> MDSP_HD void fast_stage1(int tid, cx<R> (&v)[RA], cx<R>* lds) {
    using G = FastGeo<RA, TJ>;
    const int b = tid % G::B, tj = tid / G::B;
    fft::gen_bfly<RA>(v);
#pragma unroll
    for (int q = 0; q < RA; ++q) fft::st2(lds + (tj * RA + q) * G::Bp + b, v[q]);
}
// the thread's loop-invariant second-stage twiddles W_{R_p}^{q tj2}, q = 1 .. TJ - 1 (entry 0 of every butterfly unused)
template <typename R, int RA, int TJ> MDSP_HD void fast_roots(const Pass& p, int tid, cx<R> (&rt)[FastGeo<RA, TJ>::NO]) {
    using G = FastGeo<RA, TJ>;
    const int tj = tid / G::B;
    const cx<R>* roots = static_cast<const cx<R>*>(p.roots);
#pragma unroll
    for (int s = 0; s < G::NS2; ++s) {
        const int tj2 = tj + TJ * s;
#pragma unroll
        for (int q = 0; q < TJ; ++q) rt[s * TJ + q] = roots[tj2 < RA ? (q * tj2) % G::Rp : 0];
    }
}
// stage 2 (butterflies tj2 = tj + TJ s < RA): rows tj2 + RA q from LDS, twiddles, radix TJ: y[s TJ + q'] = row tj2 + RA q' of the sub-transform
template <typename R, int RA, int TJ> MDSP_HD void fast_stage2(int tid, const cx<R>* lds, const cx<R> (&rt)[FastGeo<RA, TJ>::NO], cx<R> (&y)[FastGeo<RA, TJ>::NO]) {
    using G = FastGeo<RA, TJ>;
    const int b = tid % G::B, tj = tid / G::B;
#pragma unroll
    for (int s = 0; s < G::NS2; ++s) {
        const int tj2 = tj + TJ * s;
        if (tj2 >= RA) return;
        cx<R> z[TJ];
#pragma unroll
        for (int q = 0; q < TJ; ++q) z[q] = fft::ld2(lds + (tj2 + RA * q) * G::Bp + b);
#pragma unroll
        for (int q = 1; q < TJ; ++q) z[q] = fft::cmul(z[q], rt[s * TJ + q]);
        fft::gen_bfly<TJ>(z);
#pragma unroll
        for (int q = 0; q < TJ; ++q) y[s * TJ + q] = z[q];
    }
}
// W_{N_p}^{b r} for the thread's output rows r = tj2 + RA e
template <typename R, int RA, int TJ> MDSP_HD void fast_twb(const Pass& p, int tid, cx<R> (&twb)[FastGeo<RA, TJ>::NO]) {
    using G = FastGeo<RA, TJ>;
    const unsigned b = (unsigned)(tid % G::B), tj = (unsigned)(tid / G::B);
#pragma unroll
    for (int s = 0; s < G::NS2; ++s) {
        const unsigned tj2 = tj + (unsigned)(TJ * s);
#pragma unroll
        for (int e = 0; e < TJ; ++e) twb[s * TJ + e] = big_root<R>(p, tj2 < (unsigned)RA ? b * (tj2 + RA * e) : 0u);
    }
}
// results out of registers: put(e, pos or k, z) as phase_store (e < NO: s TJ + e')
template <typename R, int RA, int TJ, typename F>
MDSP_HD void fast_store(const Pass& p, const Tile& t, int tid, const cx<R> (&y)[FastGeo<RA, TJ>::NO], const cx<R>* twc, const cx<R> (&twb)[FastGeo<RA, TJ>::NO], F&& put) {
    using G = FastGeo<RA, TJ>;
    const int b = tid % G::B, tj = tid / G::B;
    if (b >= t.ncols) return;
    const int64_t rs = p.last ? p.N / p.Rp : t.row_stride;
    const int64_t o0 = p.last ? t.c0 + b + t.nat : t.base + b;
#pragma unroll
    for (int s = 0; s < G::NS2; ++s) {
        const int tj2 = tj + TJ * s;
        if (tj2 >= RA) return;
#pragma unroll
        for (int e = 0; e < TJ; ++e) {
            const int r = tj2 + RA * e;
            cx<R> z = y[s * TJ + e];
            if (!p.last) z = fft::cmul(z, fft::cmul(fft::ld2(twc + r), twb[s * TJ + e]));
            put(s * TJ + e, o0 + (int64_t)r * rs, z);
        }
    }
}

}  // namespace big
}  // namespace mdsp
