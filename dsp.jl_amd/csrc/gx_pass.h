// Pass code of the run-time-schedule spectral kernel (spectral_gx.h): one thread's share of a Stockham pass on ONE LDS buffer, radix as a template
// constant, everything else (butterfly counts, strides, group padding) from a gx::Sched.  MDSP_HD: tests/cpu_harness/gx_emul.cpp runs these
// functions thread by thread on the host against a Float64 DFT.
//
// A pass is  read (every operand of every butterfly of the thread into v[])  |barrier|  twiddle, butterfly, write  |barrier|.  The first pass takes
// its operands from the caller (the windowed frame, straight from the signal), the last one hands its results -- natural order: butterfly j, output
// q is bin j + nbf q -- to a functor instead of writing them.
#pragma once

#include "fft_lds.h"
#include "gx_sched.h"

namespace mdsp {
namespace gx {

using fft::cx;

MDSP_HD unsigned mulhi_u32(unsigned a, unsigned b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (unsigned)(((unsigned long long)a * (unsigned long long)b) >> 32);
#endif
}

// Two-level root table: W^e = hi[e >> 7] (1 + lo1[e & 127]) with lo1[i] = W^i - 1 -- the small factor carries its own rounding, so the product is
// one rounding away from hi's (0.5 + 0.5 ulp), not the 1.5 ulp of hi[...] lo[...].
constexpr int TWS_LOG = 7, TWS = 1 << TWS_LOG;
MDSP_HD int tw_hi_entries(int64_t n) { return (int)(((n - 1) >> TWS_LOG) + 1); }
template <typename R> MDSP_HD cx<R> tw2(const cx<R>* lo1, const cx<R>* hi, unsigned e) {
    const cx<R> h = fft::ld2(hi + (e >> TWS_LOG)), d = fft::ld2(lo1 + (e & (unsigned)(TWS - 1)));
    return fft::cadd(h, fft::cmul(h, d));
}

// butterfly m of thread t in pass p: index (clamped to a valid one for the idle lanes of a partial last trip), whether it exists, j div Ns, j mod Ns
struct Bf {
    unsigned j, hi, k;
    bool on;
};
MDSP_HD Bf bf_of(const Sched& s, int p, int t, int m) {
    Bf b;
    const unsigned j0 = (unsigned)(t + s.T * m);
    b.on = j0 < (unsigned)s.nbf[p];
    b.j = b.on ? j0 : 0u;   // butterfly 0 exists in every pass
    b.hi = s.ns[p] > 1 ? mulhi_u32(b.j, s.divm[p]) : b.j;
    b.k = b.j - b.hi * (unsigned)s.ns[p];
    return b;
}

// operands of the thread's butterflies of pass p >= 1 into v[m RR + q]
template <int RR, int EMAX, typename R> MDSP_HD void pass_read(const Sched& s, int p, int t, const cx<R>* lds, cx<R> (&v)[EMAX]) {
    constexpr int MMAX = EMAX / RR;
    const int M = s.M[p];
    const unsigned rs = (unsigned)s.rstride[p];
#pragma unroll
    for (int m = 0; m < MMAX; ++m) {
        if (m < M) {
            const Bf b = bf_of(s, p, t, m);
            const unsigned jb = b.j + (s.gin[p] ? b.hi : 0u);
#pragma unroll
            for (int q = 0; q < RR; ++q) v[m * RR + q] = fft::ld2(lds + jb + rs * (unsigned)q);
        } else {   // defined on every path: an undefined slot becomes a value carried around the unit loop (and spilled across every radix case)
#pragma unroll
            for (int q = 0; q < RR; ++q) v[m * RR + q] = cx<R>{(R)0, (R)0};
        }
    }
#pragma unroll
    for (int i = MMAX * RR; i < EMAX; ++i) v[i] = cx<R>{(R)0, (R)0};
}

// twiddles (p >= 1) + butterflies of pass p on v[]; results stay in v[m RR + q] = output q of butterfly m
#if defined(__HIP_DEVICE_COMPILE__)
#define MDSP_GX_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define MDSP_GX_SCHED_FENCE() (void)0
#endif
template <int RR, int EMAX, typename R> MDSP_HD void pass_butterflies(const Sched& s, int p, int t, cx<R> (&v)[EMAX], const cx<R>* lo1, const cx<R>* hi) {
    constexpr int MMAX = EMAX / RR;
    const int M = s.M[p];
#pragma unroll
    for (int m = 0; m < MMAX; ++m) {
        if (m < M) {
            cx<R> u[RR];
#pragma unroll
            for (int q = 0; q < RR; ++q) u[q] = v[m * RR + q];
            if (p > 0) {
                const Bf b = bf_of(s, p, t, m);
                const unsigned e1 = b.k * (unsigned)s.stride[p];   // q e1 < N for q < RR
                // four table twiddles in flight at a time: hoisted all together, the 2 (RR - 1) LDS reads of a radix-16 butterfly are 60 registers
#pragma unroll
                for (int q0 = 1; q0 < RR; q0 += 4) {
#pragma unroll
                    for (int q = q0; q < q0 + 4 && q < RR; ++q) u[q] = fft::cmul(u[q], tw2(lo1, hi, (unsigned)q * e1));
                    MDSP_GX_SCHED_FENCE();
                }
            }
            fft::gen_bfly<RR>(u);
#pragma unroll
            for (int q = 0; q < RR; ++q) v[m * RR + q] = u[q];
        }
    }
}

// scatter of pass p < P - 1:  (j div Ns) gstride + k + Ns q
template <int RR, int EMAX, typename R> MDSP_HD void pass_write(const Sched& s, int p, int t, cx<R>* lds, const cx<R> (&v)[EMAX]) {
    constexpr int MMAX = EMAX / RR;
    const int M = s.M[p];
    const unsigned ns = (unsigned)s.ns[p];
#pragma unroll
    for (int m = 0; m < MMAX; ++m) {
        if (m < M) {
            const Bf b = bf_of(s, p, t, m);
            if (b.on) {
                cx<R>* o = lds + b.hi * (unsigned)s.gstride[p] + b.k;
#pragma unroll
                for (int q = 0; q < RR; ++q) fft::st2(o + ns * (unsigned)q, v[m * RR + q]);
            }
        }
    }
}

// natural-order write of the LAST pass (real-signal columns read the mirror bin of another thread): bin j + nbf q, no padding
template <int RR, int EMAX, typename R> MDSP_HD void pass_write_natural(const Sched& s, int t, cx<R>* lds, const cx<R> (&v)[EMAX]) {
    constexpr int MMAX = EMAX / RR;
    const int p = s.P - 1, M = s.M[p];
    const unsigned nbf = (unsigned)s.nbf[p];
#pragma unroll
    for (int m = 0; m < MMAX; ++m) {
        if (m < M) {
            const Bf b = bf_of(s, p, t, m);
            if (b.on) {
#pragma unroll
                for (int q = 0; q < RR; ++q) fft::st2(lds + b.j + nbf * (unsigned)q, v[m * RR + q]);
            }
        }
    }
}

// the first pass (Ns = 1: no twiddles) on operands the caller put into v[m RR + q] = x[j + nbf q]
template <int RR, int EMAX, typename R> MDSP_HD void pass0_butterflies(const Sched& s, cx<R> (&v)[EMAX]) {
    constexpr int MMAX = EMAX / RR;
    const int M = s.M[0];
#pragma unroll
    for (int m = 0; m < MMAX; ++m) {
        if (m < M) {
            cx<R> u[RR];
#pragma unroll
            for (int q = 0; q < RR; ++q) u[q] = v[m * RR + q];
            fft::gen_bfly<RR>(u);
#pragma unroll
            for (int q = 0; q < RR; ++q) v[m * RR + q] = u[q];
        }
    }
}

// results of the last pass: f(slot m RR + q, bin j + nbf q, value) for every butterfly the thread really owns
template <int RR, int EMAX, typename R, typename F> MDSP_HD void last_consume(const Sched& s, int t, const cx<R> (&v)[EMAX], F&& f) {
    constexpr int MMAX = EMAX / RR;
    const int p = s.P - 1, M = s.M[p];
    const unsigned nbf = (unsigned)s.nbf[p];
#pragma unroll
    for (int m = 0; m < MMAX; ++m) {
        if (m < M) {
            const unsigned j = (unsigned)(t + s.T * m);
            if (j < nbf) {
#pragma unroll
                for (int q = 0; q < RR; ++q) f(m * RR + q, j + nbf * (unsigned)q, v[m * RR + q]);
            }
        }
    }
}

// run `F(RR)` for the radix r (Float32: every radix of gx_sched.h radix_ok; Float64, EMAX = 16: up to 16)
#ifdef MDSP_GX_FEW
#define MDSP_GX_RADIX_CASES_16(F) case 4: F(4); break; case 16: F(16); break;
#else
#define MDSP_GX_RADIX_CASES_16(F) \
    case 2: F(2); break;   \
    case 3: F(3); break;   \
    case 4: F(4); break;   \
    case 5: F(5); break;   \
    case 6: F(6); break;   \
    case 7: F(7); break;   \
    case 8: F(8); break;   \
    case 9: F(9); break;   \
    case 10: F(10); break; \
    case 12: F(12); break; \
    case 14: F(14); break; \
    case 15: F(15); break; \
    case 16: F(16); break;
#endif
#define MDSP_GX_RADIX_CASES_32(F) \
    case 18: F(18); break; \
    case 20: F(20); break; \
    case 21: F(21); break; \
    case 24: F(24); break; \
    case 25: F(25); break; \
    case 27: F(27); break; \
    case 28: F(28); break; \
    case 30: F(30); break; \
    case 32: F(32); break;

}  // namespace gx
}  // namespace mdsp
