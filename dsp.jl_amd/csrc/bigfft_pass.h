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
template <typename R> constexpr int elems() { return RMAX * cols<R>() / TPB; }             // elements per thread at most (32 / 16)

struct Pass {
    int64_t N;            // transform length
    int64_t Sp, Np;       // columns per prefix block, R_p S_p
    int64_t tpp, ntiles;  // tiles per prefix block (last pass: per value of `rest`), tiles per transform
    int64_t Q;            // last pass: runs per k_0 = N / (R_0 R_p)
    int Rp, R0, last;
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

// ---- phase 1: the tile into LDS as [row][B + 1] ----------------------------------------------------------------------------------------
// get(pos) -> cx<R>: element `pos` of the transform's input (the work buffer, or the windowed frames)
template <typename R, typename F> MDSP_HD void phase_load(const Pass& p, const Tile& t, int tid, cx<R>* lds, F&& get) {
    constexpr int B = cols<R>(), Bp = B + 1, U = 8;   // U loads in flight per thread and trip (the loop is NOT unrolled further: 32 x (value + 64-bit address) would not fit)
    const int total = p.Rp * B;
#pragma unroll 1
    for (int e0 = tid; e0 < total; e0 += U * TPB) {
        cx<R> v[U];
        int at[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = e0 + TPB * u;
            int r, b;
            if (!p.last) {   // lanes along the columns: rows of B consecutive elements
                b = idx & (B - 1);
                r = idx / B;
            } else {         // lanes along a run
                b = (int)(((unsigned long long)(unsigned)idx * p.divR) >> 24);
                r = idx - b * p.Rp;
            }
            at[u] = idx < total ? r * Bp + b : -1;
            v[u] = {(R)0, (R)0};
            if (idx < total && b < t.ncols) v[u] = get(t.base + (int64_t)r * t.row_stride + (int64_t)b * t.col_stride);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (at[u] >= 0) fft::st2(lds + at[u], v[u]);
    }
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
// not the last pass:  put(e, pos, z)  with the twiddle W_{N_p}^{(c0 + b) r} applied, pos = the element's own place in the work buffer
// the last pass:      put(e, k, z)    with k the natural index of the bin
// ALL: the loop over a thread's elements fully unrolled, so that `e` is a constant for the caller (register accumulators of the Welch form)
template <typename R, bool ALL, typename F> MDSP_HD void phase_store(const Pass& p, const Tile& t, int tid, const cx<R>* lds, F&& put) {
    constexpr int B = cols<R>(), Bp = B + 1, E = elems<R>(), TJ = TPB / B, U = ALL ? E : 4;
    const int b = tid & (B - 1), tj = tid / B;
    if (b >= t.ncols) return;
    const cx<R>*T0 = static_cast<const cx<R>*>(p.T0), *T1 = static_cast<const cx<R>*>(p.T1);
    const unsigned mask = (1u << p.logS) - 1u;
    const int64_t rs = p.last ? p.N / p.Rp : t.row_stride;
    const int64_t o0 = p.last ? t.c0 + b + t.nat : t.base + b;
#pragma unroll 1
    for (int e0 = 0; e0 < E && tj + TJ * e0 < p.Rp; e0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u, r = tj + TJ * e;
            if (r < p.Rp) {
                cx<R> z = fft::ld2(lds + r * Bp + b);
                if (!p.last) {
                    const unsigned m = (unsigned)(t.c0 + b) * (unsigned)r;   // < N_p < 2^31
                    if (m != 0u) {
                        cx<R> w = T0[m & mask];
                        if (p.nT1 > 1) w = fft::cmul(w, T1[m >> p.logS]);
                        z = fft::cmul(z, w);
                    }
                }
                put(ALL ? u : e, o0 + (int64_t)r * rs, z);
            }
#if defined(__HIP_DEVICE_COMPILE__)
            if (ALL && (u & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // four LDS reads in flight, not all 32 (their registers come on top of the accumulators)
#endif
        }
    }
}

}  // namespace big
}  // namespace mdsp
