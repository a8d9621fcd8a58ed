// Run-time schedules of the single-workgroup in-place spectral kernel (spectral_gx.h, round 6): ANY 7-smooth transform size up to EMAX x 512 points
// (16384 Float32 / 8192 Float64) as 2 .. 6 Stockham passes on ONE LDS buffer, radices 2 .. 32 chosen per size at plan time.  Plain C++ (no HIP):
// tests/cpu_harness/gx_emul.cpp plans and runs every size class on the host with these formulas.
//
// Index arithmetic (the compile-time schedules' -- ct_sched.h -- with run-time numbers): pass p has nbf = N / R butterflies; butterfly j reads the
// points  j + nbf q  (q < R) of the natural-order input of the pass, multiplies by W_N^{q k stride} (k = j mod Ns, stride = N / (Ns R)) and writes
// (j div Ns) Ns R + k + Ns q.  A thread runs butterflies  t + T m,  m < M = ceil(nbf / T),  and holds all of them in registers between the
// barrier behind the reads and the writes, so M R <= EMAX.  Group padding: one element behind every output group (Ns R elements) of a pass whose
// group stride aliases the LDS banks (ct_sched.h "Round 4"); the reader adds  j div Ns  -- which it needs anyway -- to its index.
#pragma once

#include <cstdint>
#include <initializer_list>

#ifndef MDSP_GX_MAXP
#define MDSP_GX_MAXP 6
#endif

namespace mdsp {
namespace gx {

struct Sched {
    int N = 0, T = 0, P = 0, emax = 0;
    int radix[MDSP_GX_MAXP], ns[MDSP_GX_MAXP], nbf[MDSP_GX_MAXP], M[MDSP_GX_MAXP];
    int stride[MDSP_GX_MAXP];    // twiddle exponent step: N / (ns radix)
    int gin[MDSP_GX_MAXP];       // 1: the layout pass p reads carries one pad element per ns[p] elements
    int rstride[MDSP_GX_MAXP];   // distance of a butterfly's operands in that layout
    int gstride[MDSP_GX_MAXP];   // distance of the output groups of pass p: ns radix (+ 1 when padded)
    unsigned divm[MDSP_GX_MAXP]; // ceil(2^32 / ns): j div ns == mulhi(j, divm) for ns > 1 (exact while j ns < 2^32)
    int np = 0;                  // elements of an LDS buffer
    int nbuf = 1;                // LDS buffers: 2 = passes ping-pong (one barrier per pass), 1 = in place (two)
    int wgs = 1;                 // workgroups of this schedule a CU holds (registers and LDS)
    double cost = 0;
};

// radices with a register butterfly (fft_lds.h gen_bfly)
inline bool radix_ok(int r, int emax) {
    if (r > emax) return false;
    switch (r) {
        case 2: case 3: case 4: case 5: case 6: case 7: case 8: case 9: case 10: case 12: case 14: case 15: case 16:
        case 18: case 20: case 21: case 24: case 25: case 27: case 28: case 30: case 32: return true;
        default: return false;
    }
}

inline bool smooth7(int64_t n) {
    if (n < 1) return false;
    for (int p : {2, 3, 5, 7})
        while (n % p == 0) n /= p;
    return n == 1;
}

inline double ilog2f(int r) {
    double l = 0;
    for (int v = 1; v < r; v *= 2) l += 1;   // ceil(log2 r): the odd radices cost about their next power of two
    return l;
}

// fill the derived fields for the radices in s.radix[0 .. P-1]; false if a pass does not fit a thread's registers
inline bool finish(Sched& s) {
    int acc = 1, extra = 0;
    for (int p = 0; p < s.P; ++p) {
        const int R = s.radix[p];
        s.ns[p] = acc;
        s.nbf[p] = s.N / R;
        s.M[p] = (s.nbf[p] + s.T - 1) / s.T;
        if (s.M[p] * R > s.emax) return false;
        s.stride[p] = s.N / (acc * R);
        s.divm[p] = acc > 1 ? (unsigned)((((uint64_t)1 << 32) + (uint64_t)acc - 1) / (uint64_t)acc) : 0u;
        const int G = acc * R;
        const bool padded = p < s.P - 1 && G % 4 == 0;
        s.gstride[p] = G + (padded ? 1 : 0);
        if (padded && s.N / G > extra) extra = s.N / G;
        s.gin[p] = 0;
        s.rstride[p] = s.nbf[p];
        if (p > 0 && s.gstride[p - 1] != s.ns[p]) {   // the previous pass padded its groups (of ns[p] elements)
            s.gin[p] = 1;
            s.rstride[p] = s.nbf[p] + s.nbf[p] / s.ns[p];
        }
        acc = G;
    }
    s.np = s.N + extra;
    return acc == s.N;
}

// Cost of a schedule in rough per-thread instruction slots: every point of a pass is read and written once (address arithmetic included), every
// point of a pass behind the first takes a two-level table twiddle, a butterfly costs ~ R log2 R, a pass two barriers.
inline double cost_of(const Sched& s) {
    double c = 0;
    for (int p = 0; p < s.P; ++p) {
        const double pts = (double)s.M[p] * s.radix[p];
        c += pts * 8.0 + (p > 0 ? pts * 9.0 : 0.0) + pts * 2.5 * ilog2f(s.radix[p]) + (s.nbuf == 1 ? 120.0 : 60.0);
    }
    return c;
}

namespace detail {
inline void search(Sched& cur, int p, int rest, int lds_limit_elems, Sched& best) {
    if (rest == 1) {
        if (p < 2) return;
        Sched s = cur;
        s.P = p;
        if (!finish(s) || s.np > lds_limit_elems) return;
        s.cost = cost_of(s);
        if (best.P == 0 || s.cost < best.cost) best = s;
        return;
    }
    if (p == MDSP_GX_MAXP) return;
    for (int r = 2; r <= cur.emax && r <= rest; ++r) {
        if (rest % r != 0 || !radix_ok(r, cur.emax)) continue;
        cur.radix[p] = r;
        search(cur, p + 1, rest / r, lds_limit_elems, best);
    }
}
}  // namespace detail

// best schedule of N points for T threads; P == 0 in the result: none
inline Sched plan_t(int N, int T, int emax, int lds_limit_elems) {
    Sched cur, best;
    cur.N = N;
    cur.T = T;
    cur.emax = emax;
    if (N < 4 || !smooth7(N) || (int64_t)T * emax < N) return best;
    detail::search(cur, 0, N, lds_limit_elems, best);
    return best;
}

// Threads per workgroup: T in {256, 512, .. tmax}; LDS buffers: two (ping-pong) or one (in place).  A CU holds `cu_threads` threads of this kernel (its
// register budget) and `lds_bytes` of LDS; a schedule's throughput is (workgroups resident per CU) / (per-thread cost); of two equal ones the smaller
// workgroup wins (its barriers stall fewer waves and other workgroups fill them).  table_bytes: what a workgroup keeps in LDS next to its buffers,
// point_bytes: per point of the transform on top of them (the Welch sums).
inline Sched plan(int N, int emax, int tmax, int cu_threads, int lds_bytes, int elem_bytes, int table_bytes, int point_bytes = 0) {
    Sched best;
    double best_score = 0;
    const int fixed = table_bytes + point_bytes * N;
    for (int nbuf = 2; nbuf >= 1; --nbuf)
        for (int T = 256; T <= tmax; T *= 2) {
            if ((int64_t)T * emax < N || fixed >= lds_bytes) continue;
            Sched s = plan_t(N, T, emax, (lds_bytes - fixed) / (elem_bytes * nbuf));
            if (s.P == 0) continue;
            s.nbuf = nbuf;
            s.cost = cost_of(s);
            const int by_lds = lds_bytes / (s.np * elem_bytes * nbuf + fixed), by_regs = cu_threads / T;
            const int wgs = by_lds < by_regs ? by_lds : by_regs;
            if (wgs < 1) continue;
            const double score = s.cost / wgs * (wgs >= 4 ? 0.94 : wgs >= 2 ? 0.97 : 1.0);
            if (best.P == 0 || score < best_score) {
                best = s;
                best.wgs = wgs;
                best_score = score;
            }
        }
    return best;
}

}  // namespace gx
}  // namespace mdsp
