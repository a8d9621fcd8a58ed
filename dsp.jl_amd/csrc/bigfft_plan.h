// Host-side planning of the large transforms (bigfft_pass.h): the factorisation N = R_0 ... R_{P-1}, the radix schedule of every
// sub-transform, and the twiddle tables.  No HIP in here: tests/cpu_harness/bigfft_emul.cpp builds the same plans.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "bigfft_pass.h"
#include "hostfft.h"

namespace mdsp {
namespace big {

inline bool seven_smooth(int64_t n) {
    if (n < 1) return false;
    for (int p : {2, 3, 5, 7})
        while (n % p == 0) n /= p;
    return n == 1;
}

// Radix schedule of a sub-transform of length R: fewest passes, then the smallest largest radix; wide radices first (the first pass has no
// twiddles).  Radices stop at 16: the pass kernels are built for two workgroups per CU (256 registers), which a radix-20 / -25 butterfly plus
// its twiddles does not fit.
inline int sub_schedule(int R, bool dbl, int* radix, int* ns, unsigned* divm) {
    static const int cand32[] = {16, 15, 12, 10, 8, 7, 6, 5, 4, 3, 2};
    static const int cand64[] = {16, 15, 12, 10, 8, 7, 6, 5, 4, 3, 2};
    const int* cand = dbl ? cand64 : cand32;
    const int nc = dbl ? 11 : 11;
    if (R == 1) return 0;
    // depth-first over non-increasing radix sequences
    std::vector<int> best, cur;
    int best_max = 1 << 30;
    struct Rec {
        const int* cand; int nc; std::vector<int>&best, &cur; int& best_max;
        void go(int rest, int from) {
            if (rest == 1) {
                const int mx = cur.empty() ? 0 : cur[0];
                if (best.empty() || cur.size() < best.size() || (cur.size() == best.size() && mx < best_max)) {
                    best = cur;
                    best_max = mx;
                }
                return;
            }
            if (!best.empty() && cur.size() + 1 > best.size()) return;
            for (int i = from; i < nc; ++i)
                if (rest % cand[i] == 0) {
                    cur.push_back(cand[i]);
                    go(rest / cand[i], i);
                    cur.pop_back();
                }
        }
    } rec{cand, nc, best, cur, best_max};
    rec.go(R, 0);
    if (best.empty() || (int)best.size() > MAXSUB) return -1;
    int acc = 1;
    for (size_t i = 0; i < best.size(); ++i) {
        radix[i] = best[i];
        ns[i] = acc;
        divm[i] = (unsigned)(((1u << 24) + (unsigned)acc - 1u) / (unsigned)acc);
        acc *= best[i];
    }
    return (int)best.size();
}

// N = R_0 ... R_{P-1}: fewest passes (2..MAXP), every factor in [2, RMAX]; among those the split whose smallest factor is largest, then the one
// whose largest factor is smallest; factors in non-decreasing order (the widest sub-transform runs over contiguous memory in the last pass).
inline int factorise(int64_t N, int* R, int rmax = RMAX) {
    if ((N & (N - 1)) == 0 && rmax >= 64) {   // powers of two: factors the two-stage form takes (256 and 64 are its square geometries; 128 and 32 idle half of
                                              // a column's threads in the second stage; 512 runs the generic phases) -- fewest passes, then fewest non-square factors
        int k = 0;
        while (((int64_t)1 << k) < N) ++k;
        // (512 is never taken for a power of two: 2^17 as 32 x 64 x 64, three two-stage passes, measured 0.098 TB/s against 0.080 for 256 x 512 with the
        // generic phases in the second pass -- profiles/r05_bigfft_sessions.json)
        const int lgmax = rmax >= 256 ? 8 : rmax >= 128 ? 7 : 6;
        int bestP = 0, bestOdd = 0, bestE[MAXP] = {0, 0, 0, 0}, e[MAXP];
        for (int P = 2; P <= MAXP && !bestP; ++P) {
            // non-decreasing exponents e[0] <= ... <= e[P-1] in [5, lgmax] summing to k
            for (e[0] = 5; e[0] <= lgmax; ++e[0])
                for (e[1] = e[0]; e[1] <= lgmax; ++e[1])
                    for (e[2] = P > 2 ? e[1] : 0; e[2] <= (P > 2 ? lgmax : 0); ++e[2])
                        for (e[3] = P > 3 ? e[2] : 0; e[3] <= (P > 3 ? lgmax : 0); ++e[3]) {
                            int sum = 0, odd = 0;
                            for (int i = 0; i < P; ++i) sum += e[i], odd += (e[i] != 8 && e[i] != 6) ? (e[i] == 9 ? 2 : 1) : 0;
                            if (sum != k) continue;
                            if (!bestP || odd < bestOdd) {
                                bestP = P;
                                bestOdd = odd;
                                for (int i = 0; i < P; ++i) bestE[i] = e[i];
                            }
                        }
        }
        if (bestP) {
            for (int i = 0; i < bestP; ++i) R[i] = 1 << bestE[i];
            return bestP;
        }
    }
    std::vector<int> best, cur;
    struct Rec {
        std::vector<int>&best, &cur;
        int rmax;
        static bool better(const std::vector<int>& a, const std::vector<int>& b) {   // a, b sorted ascending, same length
            if (a.front() != b.front()) return a.front() > b.front();
            return a.back() < b.back();
        }
        void go(int64_t rest, int from, int left) {
            if (left == 0) {
                if (rest == 1 && (best.empty() || better(cur, best))) best = cur;
                return;
            }
            // rest must split into `left` factors, each >= from and <= RMAX
            for (int f = from; f <= rmax && (int64_t)f <= rest; ++f) {
                if (rest % f) continue;
                double need = std::pow((double)(rest / f), 1.0 / std::max(1, left - 1));
                if (left > 1 && need > rmax + 0.5) continue;
                if (left == 1 && rest != f) continue;
                cur.push_back(f);
                go(rest / f, f, left - 1);
                cur.pop_back();
            }
        }
    } rec{best, cur, rmax};
    for (int P = 2; P <= MAXP; ++P) {
        rec.go(N, 2, P);
        if (!best.empty()) {
            for (int p = 0; p < P; ++p) R[p] = best[p];
            return P;
        }
    }
    return 0;
}

template <typename R> struct HostPlan {
    int64_t N = 0;
    int P = 0;
    Pass pass[MAXP];
    std::vector<std::vector<cx<R>>> roots, T0, T1;   // per pass
};

// false: N cannot be planned (a prime factor above 7, or a factor that no schedule covers)
// first_only: the factors are (R_0, N / R_0) and only pass 0 is built -- the rows of N / R_0 points belong to somebody else (overlap-save: the
// single-workgroup transforms of ols.hip)
template <typename R> bool make_plan_factors(int64_t N, HostPlan<R>& hp, const int* Rf, int P, int fast, bool first_only = false);
template <typename R> bool make_plan(int64_t N, HostPlan<R>& hp, int rmax = RMAX, int fast = 1) {   // fast: 0 generic phases, 1 two-stage forms, 2 the same with 128 = 8 x 16 in Float32 too
    if (N < 4 || N >= ((int64_t)1 << 31) || !seven_smooth(N)) return false;
    int Rf[MAXP];
    const int P = factorise(N, Rf, rmax);
    if (P < 2) return false;
    return make_plan_factors<R>(N, hp, Rf, P, fast);
}
template <typename R> bool make_plan_factors(int64_t N, HostPlan<R>& hp, const int* Rf, int P, int fast, bool first_only) {
    constexpr bool dbl = sizeof(R) == 8;
    const int fast128 = fast == 2 ? 816 : 168;
    hp.N = N;
    hp.P = P;
    hp.roots.assign(P, {});
    hp.T0.assign(P, {});
    hp.T1.assign(P, {});
    int64_t M[MAXP];   // natural weights M_p = R_0 ... R_{p-1}
    M[0] = 1;
    for (int p = 1; p < P; ++p) M[p] = M[p - 1] * Rf[p - 1];
    for (int p = 0; p < (first_only ? 1 : P); ++p) {
        Pass& q = hp.pass[p];
        q = Pass{};
        q.N = N;
        q.Rp = Rf[p];
        q.R0 = Rf[0];
        q.last = p == P - 1;
        q.divR = (unsigned)(((1u << 24) + (unsigned)q.Rp - 1u) / (unsigned)q.Rp);
        int64_t S = 1;
        for (int k = p + 1; k < P; ++k) S *= Rf[k];
        q.Sp = S;
        q.Np = S * q.Rp;
        // the two-stage register form (bigfft_pass.h) where the sub-transform is 256 / 128 / 64 / 32 points; its tiles have 256 / TJ columns
        q.fRA = q.fTJ = 0;
        if (fast) {
            if (q.Rp == 256) q.fRA = 16, q.fTJ = 16;
            else if (q.Rp == 128 && !dbl && fast128 == 168) q.fRA = 16, q.fTJ = 8;   // Float32: every thread busy in both stages, 32-column tiles (bigfft_pass.h FastGeo)
            else if (q.Rp == 128) q.fRA = 8, q.fTJ = 16;
            else if (q.Rp == 64 && !dbl && fast == 3) q.fRA = 16, q.fTJ = 4;   // (MDSP_BIG_FAST=3, measured: 64-column tiles but two workgroups per CU instead of four -- 10 - 20 % slower than 8 x 8, profiles/r05_welch_rows.json)
            else if (q.Rp == 64) q.fRA = 8, q.fTJ = 8;
            else if (q.Rp == 32) q.fRA = 4, q.fTJ = 8;
        }
        const int B = q.fTJ ? TPB / q.fTJ : cols<R>();
        q.B = B;
        if (!q.last) {
            q.tpp = (S + B - 1) / B;
            q.ntiles = (N / q.Np) * q.tpp;
            q.Q = 0;
        } else {
            q.Q = N / ((int64_t)Rf[0] * q.Rp);
            q.tpp = (Rf[0] + B - 1) / B;
            q.ntiles = q.Q * q.tpp;
            q.nd = 0;
            for (int k = P - 2; k >= 1; --k) {   // rest = sum_{k=1}^{P-2} k_k (S_k / R_last), k_{P-2} fastest
                q.dR[q.nd] = Rf[k];
                q.dM[q.nd] = M[k];
                ++q.nd;
            }
        }
        q.nsub = sub_schedule(q.Rp, dbl, q.radix, q.ns, q.divm);
        if (q.nsub < 0) return false;
        hp.roots[p].resize((size_t)q.Rp);
        for (int k = 0; k < q.Rp; ++k) {
            const zd w = unit_root(k, q.Rp, -1);
            hp.roots[p][(size_t)k] = {(R)w.real(), (R)w.imag()};
        }
        if (!q.last) {
            int lg = 0;
            while (((int64_t)1 << lg) < q.Np) ++lg;
            q.logS = q.Np <= 8192 ? lg : (lg + 1) / 2;
            const int64_t Sz = (int64_t)1 << q.logS;
            q.nT1 = (int)((q.Np + Sz - 1) / Sz);
            hp.T0[p].resize((size_t)std::min<int64_t>(Sz, q.Np));
            for (size_t m = 0; m < hp.T0[p].size(); ++m) {
                const zd w = unit_root((int64_t)m, q.Np, -1);
                hp.T0[p][m] = {(R)w.real(), (R)w.imag()};
            }
            hp.T1[p].resize((size_t)q.nT1);
            for (int m = 0; m < q.nT1; ++m) {
                const zd w = unit_root((int64_t)m * Sz, q.Np, -1);
                hp.T1[p][(size_t)m] = {(R)w.real(), (R)w.imag()};
            }
        } else {
            q.logS = 0;
            q.nT1 = 1;
        }
    }
    if (first_only) hp.P = 1;
    return true;
}

}  // namespace big
}  // namespace mdsp
