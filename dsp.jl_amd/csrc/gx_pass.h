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

#ifndef MDSP_GX_TW_CHUNK
#define MDSP_GX_TW_CHUNK 4
#endif
// what a pass needs of the schedule, fetched once per pass (the kernel reads the schedule from its argument block: a fetch per use is a scalar load and a
// wait that also drains the wave's LDS queue)
struct Pass {
    int T, M, nbf, ns, stride, gin, rstride, gstride;
    unsigned divm;
    bool first;   // pass 0: no twiddles
};
MDSP_HD Pass pass_of(const Sched& s, int p) {
    Pass q;
    q.T = s.T; q.M = s.M[p]; q.nbf = s.nbf[p]; q.ns = s.ns[p]; q.stride = s.stride[p]; q.gin = s.gin[p]; q.rstride = s.rstride[p]; q.gstride = s.gstride[p];
    q.divm = s.divm[p];
    q.first = p == 0;
    return q;
}

// butterfly m of thread t in a pass: index (a valid one for the idle lanes of a partial last trip), whether it exists, j div Ns, j mod Ns
struct Bf {
    unsigned j, hi, k;
    bool on;
};
MDSP_HD Bf bf_of(const Pass& ps, int t, int m) {
    Bf b;
    const unsigned j0 = (unsigned)(t + ps.T * m);
    b.on = j0 < (unsigned)ps.nbf;
    b.j = b.on ? j0 : 0u;   // butterfly 0 exists in every pass
    b.hi = ps.ns > 1 ? mulhi_u32(b.j, ps.divm) : b.j;
    b.k = b.j - b.hi * (unsigned)ps.ns;
    return b;
}

// operands of the thread's butterflies into v[m RR + q]
template <int RR, int EMAX, typename R> MDSP_HD void pass_read(const Pass& ps, int t, const cx<R>* lds, cx<R> (&v)[EMAX]) {
    constexpr int MMAX = EMAX / RR;
    const unsigned rs = (unsigned)ps.rstride;
#pragma unroll
    for (int m = 0; m < MMAX; ++m) {
        if (m < ps.M) {
            const Bf b = bf_of(ps, t, m);
            const cx<R>* o = lds + b.j + (ps.gin ? b.hi : 0u);   // (one address, stepped: a product per operand is a quarter-rate multiply-add each)
#pragma unroll
            for (int q = 0; q < RR; ++q, o += rs) v[m * RR + q] = fft::ld2(o);
        } else {   // defined on every path: an undefined slot becomes a value carried around the unit loop (and spilled across every radix case)
#pragma unroll
            for (int q = 0; q < RR; ++q) v[m * RR + q] = cx<R>{(R)0, (R)0};
        }
    }
#pragma unroll
    for (int i = MMAX * RR; i < EMAX; ++i) v[i] = cx<R>{(R)0, (R)0};
}

// twiddles (behind the first pass) + butterflies on v[]; results stay in v[m RR + q] = output q of butterfly m
template <int RR, int EMAX, typename R> MDSP_HD void pass_butterflies(const Pass& ps, int t, cx<R> (&v)[EMAX], const cx<R>* lo1, const cx<R>* hi) {
    constexpr int MMAX = EMAX / RR;
#pragma unroll
    for (int m = 0; m < MMAX; ++m) {
        if (m < ps.M) {
            cx<R>(&u)[RR] = *reinterpret_cast<cx<R>(*)[RR]>(&v[m * RR]);   // the butterfly's slice of v[], in place
            if (!ps.first) {
                const Bf b = bf_of(ps, t, m);
                const unsigned e1 = b.k * (unsigned)ps.stride;   // q e1 < N for q < RR
                // the table reads of up to CH twiddles are issued together, then consumed (one read pair at a time, each behind its own wait, was
                // what the compiler made of the plain loop: the full LDS latency per twiddle)
                constexpr int CH = sizeof(R) == 8 ? 8 : MDSP_GX_TW_CHUNK;
                unsigned e = 0;
#pragma unroll
                for (int q0 = 1; q0 < RR; q0 += CH) {
                    cx<R> h[CH], d[CH];
#pragma unroll
                    for (int q = q0; q < q0 + CH && q < RR; ++q) {
                        e += e1;   // q e1
                        h[q - q0] = fft::ld2(hi + (e >> TWS_LOG));
                        d[q - q0] = fft::ld2(lo1 + (e & (unsigned)(TWS - 1)));
                    }
#pragma unroll
                    for (int q = q0; q < q0 + CH && q < RR; ++q) u[q] = fft::cmul(u[q], fft::cadd(h[q - q0], fft::cmul(h[q - q0], d[q - q0])));
                }
            }
            fft::gen_bfly<RR>(u);
        }
    }
}

// scatter of a pass that is not the last:  (j div Ns) gstride + k + Ns q
template <int RR, int EMAX, typename R> MDSP_HD void pass_write(const Pass& ps, int t, cx<R>* lds, const cx<R> (&v)[EMAX]) {
    constexpr int MMAX = EMAX / RR;
    const unsigned ns = (unsigned)ps.ns;
#pragma unroll
    for (int m = 0; m < MMAX; ++m) {
        if (m < ps.M) {
            const Bf b = bf_of(ps, t, m);
            if (b.on) {
                cx<R>* o = lds + b.hi * (unsigned)ps.gstride + b.k;
#pragma unroll
                for (int q = 0; q < RR; ++q, o += ns) fft::st2(o, v[m * RR + q]);
            }
        }
    }
}

// natural-order write of the LAST pass (real-signal columns read the mirror bin of another thread): bin j + nbf q, no padding
template <int RR, int EMAX, typename R> MDSP_HD void pass_write_natural(const Pass& ps, int t, cx<R>* lds, const cx<R> (&v)[EMAX]) {
    constexpr int MMAX = EMAX / RR;
    const unsigned nbf = (unsigned)ps.nbf;
#pragma unroll
    for (int m = 0; m < MMAX; ++m) {
        if (m < ps.M) {
            const Bf b = bf_of(ps, t, m);
            if (b.on) {
                cx<R>* o = lds + b.j;
#pragma unroll
                for (int q = 0; q < RR; ++q, o += nbf) fft::st2(o, v[m * RR + q]);
            }
        }
    }
}

// results of the last pass: f(slot m RR + q, bin j + nbf q, value) for every butterfly the thread really owns
template <int RR, int EMAX, typename R, typename F> MDSP_HD void last_consume(const Pass& ps, int t, const cx<R> (&v)[EMAX], F&& f) {
    constexpr int MMAX = EMAX / RR;
    const unsigned nbf = (unsigned)ps.nbf;
#pragma unroll
    for (int m = 0; m < MMAX; ++m) {
        if (m < ps.M) {
            const unsigned j = (unsigned)(t + ps.T * m);
            if (j < nbf) {
                unsigned bin = j;
#pragma unroll
                for (int q = 0; q < RR; ++q, bin += nbf) f(m * RR + q, bin, v[m * RR + q]);
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
