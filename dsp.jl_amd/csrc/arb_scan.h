// Exact parallel evaluation of the FIRArbitrary phase recurrence (update!, stream_filt.jl:567-577)
//
//     acc <- fl(acc + delta);  if acc >= Nphi:  (q, acc) = divrem(acc, Nphi);  xIdx += q
//
// which the reference runs serially, one output at a time, and whose state decides every index on this path (which
// input samples and which filter phase feed output k, how many outputs a call produces, the state it leaves behind).
// The serial host loop (arb_trajectory in fir.hip) costs about 1 ns per output -- two orders of magnitude more than the
// filtering kernel -- so long streams evaluate the SAME recurrence in parallel on the device, bit for bit:
//
//  * Grid.  After one update acc is a multiple of g = ulp(delta): fl(acc + delta) >= delta is a multiple of its own
//    ulp >= g, and q Nphi is a multiple of g.  With A = acc / g, D = delta / g in [2^52, 2^53) and N = Nphi / g the
//    update is integer arithmetic:  S = A + D;  S' = RNE of S to a multiple of 2^j, j = bitlength(S) - 53 (0 when
//    S < 2^53);  q = floor(S' / N);  A <- S' - q N.  This is exactly what the floating-point operations compute
//    (the divrem remainder is exact), so it reproduces the reference's doubles.
//  * Unwrapped position.  U = A + N (xIdx - xIdx_0) advances by D + eps_k, eps_k = S' - S the rounding of step k:
//    U_k = U_0 + k D + E_k.  The unrounded part is closed form (128-bit integer arithmetic); E_k is small.
//  * eps_k depends on (i) the binade of S = (U_k mod N) + D, a function of where U_k mod N lies between the thresholds
//    {0, 2^53 - D, 2^54 - D, ...}, and (ii) the low j+1 bits of S, i.e. of E_k modulo R = 2^(J+1) (J = the largest j).
//    So with the binades read off the unrounded trajectory, E_k follows a finite-state recurrence on E mod R: each block
//    of BLK steps is a table  e -> sum of eps over the block, for the R entry residues,  tables compose associatively,
//    and a multi-level scan gives E at every block start.
//  * Prediction.  E_k is very nearly linear in k (the rounding pattern is quasi-periodic), so the tables of block b are
//    simulated around the unrounded position plus a predicted offset c_b (slope measured on a serial pilot; a second
//    pass uses the first pass's own E): the simulated candidates are then a few units from the true position.
//  * Verification.  Reading the binades off the simulated candidates is valid while the true position is on the same
//    side of every threshold as they are: each block records the minimum distance of its candidates to a threshold,
//    and the finalize pass rejects the block if that distance is not larger than |E_b - c_b| plus the drift possible
//    inside a block -- unless 0 <= E_b - c_b < R, when one candidate started exactly at the true position and the
//    block is exact unconditionally.  A rejected first pass (probability ~ 2^-40 per output) is repeated around its own
//    E, which certifies it block by block if it was right; a run rejected twice falls back to the serial host loop,
//    so the result is exact either way.
//
// Everything here is plain integer code shared by the device kernels (fir.hip) and a host driver used by the CPU tests.
#pragma once

#include <cfloat>
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define ARBSCAN_HD __host__ __device__ inline
#else
#define ARBSCAN_HD inline
#endif

namespace mdsp {
namespace arbscan {

constexpr int BLK = 32;        // updates per scan block
constexpr int ANCH = 16;       // outputs per anchor (ARB_BLK in fir.hip): two anchors per block, the second from finalize's own replay
constexpr int RMAX = 16;       // table entries: residues of E modulo 2^(J+1), J <= 3
constexpr int FAN = 64;        // tables composed per thread in one scan level

struct Grid {
    uint64_t D, N;             // delta and Nphi in units of g
    uint64_t qd, rd;           // D = qd N + rd
    int J, R;                  // rounding bits that can occur, table size 2^(J+1)
    int gexp;                  // g = 2^gexp
    uint64_t thr[4];           // binade thresholds for A: S = A + D reaches 2^(53+i) at A = thr[i]   (those inside [0, N))
    int nthr;
};

ARBSCAN_HD int bitlen64(uint64_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return v ? 64 - __clzll((long long)v) : 0;
#else
    return v ? 64 - __builtin_clzll(v) : 0;
#endif
}

// false: the recurrence does not fit the integer model (use the serial host loop)
inline bool make_grid(double delta, int64_t nphi, Grid& G) {
    if (!(delta >= DBL_MIN) || !std::isfinite(delta) || nphi < 1) return false;
    int e = 0;
    const double m = std::frexp(delta, &e);   // delta = m 2^e, m in [0.5, 1)
    G.D = (uint64_t)std::ldexp(m, 53);         // in [2^52, 2^53)
    G.gexp = e - 53;
    const int s = 53 - e;                      // N = nphi 2^s
    if (s < 0 || s > 61) return false;
    if (((uint64_t)nphi >> (61 - s)) != 0) return false;   // N < 2^61
    G.N = (uint64_t)nphi << s;
    G.J = bitlen64(G.N + G.D) - 53;
    if (G.J < 0) G.J = 0;
    if (G.J > 3) return false;                 // rate >= ~7: more than three rounding bits
    G.R = 1 << (G.J + 1);
    G.qd = G.D / G.N;
    G.rd = G.D % G.N;
    if (G.qd > 8) return false;                // rate < ~1/8: few outputs per input, the serial loop is cheap there
    G.nthr = 0;
    for (int i = 0; i < G.J; ++i) {
        const uint64_t t = (1ull << (53 + i)) - G.D;
        if (t < G.N) G.thr[G.nthr++] = t;
    }
    return true;
}

// acc on the grid?  (true after any update; the caller's initial phase may not be)
inline bool to_grid(const Grid& G, double acc, uint64_t& A) {
    if (!(acc >= 0.0)) return false;
    const double a = std::ldexp(acc, -G.gexp);
    if (!(a < 18446744073709551616.0) || a != std::floor(a)) return false;
    A = (uint64_t)a;
    return A < G.N;
}

ARBSCAN_HD double from_grid(const Grid& G, uint64_t A) { return ldexp((double)A, G.gexp); }   // exact: A is a representable multiple

// one update on the grid; returns eps = S' - S
ARBSCAN_HD int64_t step(const Grid& G, uint64_t& A, int64_t& W) {
    uint64_t S = A + G.D;
    const uint64_t S0 = S;
    if (S >> 53) {
        const int j = bitlen64(S) - 53;
        const uint64_t low = S & ((1ull << j) - 1), half = 1ull << (j - 1);
        S -= low;
        if (low > half || (low == half && ((S >> j) & 1))) S += 1ull << j;   // round to nearest, ties to even
    }
    const int64_t eps = (int64_t)(S - S0);
    while (S >= G.N) {
        S -= G.N;
        ++W;
    }
    A = S;
    return eps;
}

// one update of the unrounded trajectory
ARBSCAN_HD void base_step(const Grid& G, uint64_t& A0, int64_t& W0) {
    A0 += G.rd;
    W0 += (int64_t)G.qd;
    if (A0 >= G.N) {
        A0 -= G.N;
        ++W0;
    }
}

// circular distance of A to the nearest binade threshold / the wrap point
ARBSCAN_HD uint64_t threshold_distance(const Grid& G, uint64_t A) {
    uint64_t d = A < G.N - A ? A : G.N - A;
    for (int i = 0; i < G.nthr; ++i) {
        const uint64_t t = G.thr[i];
        const uint64_t di = A > t ? A - t : t - A;
        d = di < d ? di : d;
    }
    return d;
}

// (hi:lo) / n with hi < n: quotient and remainder by shift-subtract
ARBSCAN_HD void divmod128(uint64_t hi, uint64_t lo, uint64_t n, uint64_t& q, uint64_t& r) {
    uint64_t rem = hi, quo = 0;
    for (int i = 63; i >= 0; --i) {
        const uint64_t top = rem >> 63;
        rem = (rem << 1) | ((lo >> i) & 1);
        quo <<= 1;
        if (top || rem >= n) {
            rem -= n;
            quo |= 1;
        }
    }
    q = quo;
    r = rem;
}

ARBSCAN_HD void mul64(uint64_t a, uint64_t b, uint64_t& hi, uint64_t& lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    hi = __umul64hi(a, b);
    lo = a * b;
#else
    const unsigned __int128 p = (unsigned __int128)a * b;
    hi = (uint64_t)(p >> 64);
    lo = (uint64_t)p;
#endif
}

// unrounded state after n updates from (As, 0):  As + n D = W N + A
ARBSCAN_HD void base_at(const Grid& G, uint64_t As, uint64_t n, uint64_t& A, int64_t& W) {
    uint64_t hi, lo;
    mul64(n, G.rd, hi, lo);
    const uint64_t lo2 = lo + As;
    hi += lo2 < lo ? 1 : 0;
    uint64_t q, r;
    divmod128(hi, lo2, G.N, q, r);
    A = r;
    W = (int64_t)(n * G.qd + q);
}

ARBSCAN_HD uint64_t add_mod(const Grid& G, uint64_t A, int64_t c) {   // (A + c) mod N, |c| << 2^62
    int64_t v = (int64_t)A + c;
    const int64_t n = (int64_t)G.N;
    if (v >= n || v < 0) {
        v %= n;
        if (v < 0) v += n;
    }
    return (uint64_t)v;
}

// Offset the tables of block b are built around: the predicted E (a multiple of R), so that the true position stays
// within a few units of the simulated candidates however far E has drifted (E is very nearly linear in k: the
// rounding pattern is quasi-periodic).  pass 0: slope from a serial pilot; pass 1: the E of a previous scan.
ARBSCAN_HD int64_t predicted_offset(const Grid& G, double sigma, int64_t k) {
    const double c = sigma * (double)k / (double)G.R;
    return (int64_t)llrint(c) * (int64_t)G.R;
}
ARBSCAN_HD int64_t round_offset(const Grid& G, int64_t E) { return (E >> (G.J + 1)) << (G.J + 1); }

struct ScanArgs {
    Grid G;
    uint64_t As;               // grid state at output k0 (the first output of device block 0)
    int64_t xs;                // xIdx there
    int64_t k0;                // its output index (a multiple of BLK)
    int64_t xlen;
    int64_t nb;                // blocks
    double sigma;
    int pass;
    // per block
    int32_t* t0;               // tables, R entries per block: entry residue e = E mod R -> sum of eps over the block
    uint32_t* mind;            // clamped threshold distance of the block's candidates
    int64_t* cb;               // offset the block's candidates were built around
    int64_t* E;                // in (pass 1): E of the previous scan; out: E at the block start
    uint64_t* baseA;           // unrounded state at the block start, at [2 b] (aliases the block's first anchor slot until finalize)
    int64_t* baseW;
    // finalize
    int64_t* tab_x;            // anchors, indexed by k / ANCH
    double* tab_acc;
    int64_t* result;           // [0] status bits (1: ambiguous block, 2: end found), [1] nout, [2] xIdx_end, [3] bits of phi_acc_end
};

// K1: tables of block b.  RC = the table size as a compile-time constant (device: candidates live in registers), 0 = read it
// from the grid (host emulation).
template <int RC = 0> ARBSCAN_HD void scan_tables_body(const ScanArgs& a, int64_t b) {
    const Grid& G = a.G;
    constexpr int CAP = RC ? RC : RMAX;
    const int R = RC ? RC : G.R;
    uint64_t A0;
    int64_t W0;
    base_at(G, a.As, (uint64_t)b * BLK, A0, W0);
    a.baseA[2 * b] = A0;
    a.baseW[2 * b] = W0;
    const int64_t c = a.pass == 0 ? predicted_offset(G, a.sigma, b * BLK) : round_offset(G, a.E[b]);
    a.cb[b] = c;
    uint64_t Ac[CAP], Ab = add_mod(G, A0, c);
    int64_t Wd = 0;
    int32_t T[CAP];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int e = 0; e < R; ++e) {
        const uint64_t v = Ab + (uint64_t)e;
        Ac[e] = v >= G.N ? v - G.N : v;
        T[e] = 0;
    }
    uint64_t md = ~0ull;
    for (int k = 0; k < BLK; ++k) {
        const uint64_t d = threshold_distance(G, Ab);
        md = d < md ? d : md;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int e = 0; e < R; ++e) T[e] += (int32_t)step(G, Ac[e], Wd);
        base_step(G, Ab, Wd);
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int e = 0; e < R; ++e) a.t0[b * R + e] = T[e];
    a.mind[b] = md > 0xffffffffull ? 0xffffffffu : (uint32_t)md;
}

// K2 up: out[g] = in[FAN g] then in[FAN g + 1] then ...   (tables are R consecutive entries; wide sums above level 0)
template <int RC, typename TIn> ARBSCAN_HD void scan_compose_body(const Grid& G, const TIn* in, int64_t n, int64_t* out, int64_t g) {
    constexpr int CAP = RC ? RC : RMAX;
    const int R = RC ? RC : G.R;
    int64_t acc[CAP];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int e = 0; e < R; ++e) acc[e] = 0;
    const int64_t lo = g * FAN, hi = lo + FAN < n ? lo + FAN : n;
    for (int64_t t = lo; t < hi; ++t) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int e = 0; e < R; ++e) acc[e] += (int64_t)in[t * R + ((e + acc[e]) & (R - 1))];
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int e = 0; e < R; ++e) out[g * R + e] = acc[e];
}

// K2 top: serial over the coarsest level
ARBSCAN_HD void scan_top_body(const Grid& G, const int64_t* in, int64_t n, int64_t* Eout) {
    int64_t E = 0;
    for (int64_t t = 0; t < n; ++t) {
        Eout[t] = E;
        E += in[t * G.R + (E & (G.R - 1))];
    }
}

// K2 down: E at the starts of the fine elements of group g from the group's own start
template <typename TIn> ARBSCAN_HD void scan_expand_body(const Grid& G, const TIn* in, int64_t n, const int64_t* Ecoarse, int64_t* Efine, int64_t g) {
    int64_t E = Ecoarse[g];
    const int64_t lo = g * FAN, hi = lo + FAN < n ? lo + FAN : n;
    for (int64_t t = lo; t < hi; ++t) {
        Efine[t] = E;
        E += (int64_t)in[t * G.R + (E & (G.R - 1))];
    }
}

// K3: exact anchor of block b; the block that contains the end of the stream replays its updates and reports the end state
ARBSCAN_HD void scan_finalize_body(const ScanArgs& a, int64_t b) {
    const Grid& G = a.G;
    const int64_t E = a.E[b], dE = E - a.cb[b];
    const uint64_t absd = (uint64_t)(dE < 0 ? -dE : dE);
    const uint64_t margin = absd + (uint64_t)(BLK + 2) * (1ull << G.J) + 2 * (uint64_t)G.R;
    int64_t status = 0;
    // 0 <= E - c < R: the candidate of residue E - c started AT the true position, the block is exact whatever the
    // thresholds (exactly representable rates, where nothing ever rounds and positions sit on the wrap point; every
    // block of a second pass whose first pass was right)
    if (!(dE >= 0 && dE < G.R) && (uint64_t)a.mind[b] <= margin) status |= 1;
    // exact position: unrounded + E  (may wrap: E is not small compared with the distance to the wrap point)
    int64_t v = (int64_t)a.baseA[2 * b] + E, W = a.baseW[2 * b];
    const int64_t n = (int64_t)G.N;
    while (v >= n) {
        v -= n;
        ++W;
    }
    while (v < 0) {
        v += n;
        --W;
    }
    uint64_t A = (uint64_t)v;
    const int64_t kb = a.k0 + b * BLK;
    int64_t xi = a.xs + W;
    a.tab_x[kb / ANCH] = xi;
    a.tab_acc[kb / ANCH] = from_grid(G, A);
    if (xi <= a.xlen) {
        for (int i = 1; i <= BLK; ++i) {
            int64_t Wn = 0;
            step(G, A, Wn);
            xi += Wn;
            if (i == ANCH) {   // the block's second anchor (state of output kb + ANCH); harmless if that output does not exist
                a.tab_x[kb / ANCH + 1] = xi;
                a.tab_acc[kb / ANCH + 1] = from_grid(G, A);
            }
            if (xi > a.xlen) {
                status |= 2;
                a.result[1] = kb + i;
                a.result[2] = xi;
                const double accd = from_grid(G, A);
                int64_t bits;
                __builtin_memcpy(&bits, &accd, 8);
                a.result[3] = bits;
                break;
            }
        }
    }
    if (status) {
#if defined(__HIP_DEVICE_COMPILE__)
        atomicOr((unsigned long long*)&a.result[0], (unsigned long long)status);
#else
        a.result[0] |= status;
#endif
    }
}

}  // namespace arbscan
}  // namespace mdsp
