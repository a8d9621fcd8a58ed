// One wavefront per 4096-point transform: 4096 = 64 x 64, 64 points per lane, ONE exchange (round 4).
//
// The workgroup transforms of fft_lds.h share a transform between T = N/E threads and pay P-1 exchanges through LDS, each with its
// workgroup barriers.  Here lane t of ONE wavefront holds the 64 points x[t + 64 e]:
//     X[ke + 64 kt] = sum_t W64^{t kt} [ W4096^{t ke} ( sum_e x[t + 64 e] W64^{e ke} ) ]
//   pass A   lane t: 64-point transform over e, in registers (bfly64: 8 x 8, the W64 roots between the two radix-8 layers are compile-time constants)
//   exchange lane ke receives Y_t[ke] from every lane t: a 64 x 64 transposition inside the wavefront -- one v_permlane32_swap per register
//            pair moves the two off-diagonal 32 x 32 blocks, then each half-wave transposes 32 x 32 blocks through a 16.5 KiB LDS buffer in two
//            rounds.  No barrier anywhere: a wave's LDS operations execute in order.
//   twiddle  W4096^{lane * register} -- symmetric in (t, ke), so ONE per-lane table serves whichever side applies it (the reader here)
//   pass B   lane ke: 64-point transform over t; lane ke ends with X[ke + 64 kt], kt = 0..63  (the "thread t holds X[t + T e]" invariant of fft_lds.h)
// LDS traffic per transform: 32 KiB written + 32 KiB read, half of the three-pass radix-16 form.
//
// Plain C++ (MDSP_HD) like fft_lds.h: tests/cpu_harness/fft_emul.cpp runs the same code lane by lane on the host against a long-double DFT.
#pragma once

#include "fft_lds.h"

namespace mdsp {
namespace fft {

// cos(pi m / 32), m = 0..16
template <typename R> MDSP_HD constexpr R cos64(int m) {
    constexpr long double c[17] = {1.0L,
                                   0.995184726672196886244836953109479922L,
                                   0.980785280403230449126182236134239037L,
                                   0.956940335732208864935797886980269969L,
                                   0.923879532511286756128183189396788287L,
                                   0.88192126434835502971275686366038835L,
                                   0.831469612302545237078788377617905757L,
                                   0.773010453362736960810906609758469801L,
                                   0.707106781186547524400844362104849039L,
                                   0.634393284163645498215171613225493371L,
                                   0.555570233019602224742830813948532874L,
                                   0.471396736825997648556387625905254378L,
                                   0.382683432365089771728459984030398867L,
                                   0.290284677254462367636192375817395275L,
                                   0.195090322016128267848284868477022241L,
                                   0.0980171403295606019941955638886418459L,
                                   0.0L};
    return (R)c[m];
}
// forward root W64^m = exp(-2 pi i m / 64), 0 <= m < 64, from the first-quadrant table
template <typename R> MDSP_HD constexpr cx<R> w64(int m) {
    const int q = m >> 4, r = m & 15;                  // m = 16 q + r: W64^m = (-i)^q W64^r
    const R c = cos64<R>(r), s = cos64<R>(16 - r);     // W64^r = (c, -s)
    return q == 0 ? cx<R>{c, -s} : q == 1 ? cx<R>{-s, -c} : q == 2 ? cx<R>{-c, s} : cx<R>{s, c};
}

// radix-8 butterfly from its first layer: S[j] = v[j] + v[j+4], D[j] = v[j] - v[j+4], j = 0..3; natural order out (same arithmetic as bfly8)
template <int DIR, typename R> MDSP_HD void bfly8_sd(const cx<R> (&S)[4], const cx<R> (&D)[4], cx<R> (&o)[8]) {
    constexpr R h = (R)0.70710678118654752440084436210485L;
    const cx<R> e0 = cadd(S[0], S[2]), e2 = csub(S[0], S[2]), e1 = add_mi<DIR>(D[0], D[2]), e3 = sub_mi<DIR>(D[0], D[2]);
    const cx<R> o0 = cadd(S[1], S[3]), o2 = csub(S[1], S[3]), o1 = add_mi<DIR>(D[1], D[3]), o3 = sub_mi<DIR>(D[1], D[3]);
    const cx<R> u1 = w8_1_unscaled<DIR>(o1), u3 = w8_3_unscaled_neg<DIR>(o3);
    o[0] = cadd(e0, o0);
    o[4] = csub(e0, o0);
    o[1] = caxpy_k(h, u1, e1);
    o[5] = caxmy_k(h, u1, e1);
    o[2] = add_mi<DIR>(e2, o2);
    o[6] = sub_mi<DIR>(e2, o2);
    o[3] = caxmy_k(h, u3, e3);
    o[7] = caxpy_k(h, u3, e3);
}

// multiply by the compile-time root W64^M (forward; conjugate for the inverse)
template <int DIR, int M, typename R> MDSP_HD cx<R> mul_w64(cx<R> a) {
    constexpr R h = (R)0.70710678118654752440084436210485L;
    if constexpr (M == 0) return a;
    else if constexpr (M == 16) return mul_mi<DIR>(a);
    else if constexpr (M == 8) return cscale_k(h, w8_1_unscaled<DIR>(a));
    else if constexpr (M == 24) return cscale_k(-h, w8_3_unscaled_neg<DIR>(a));
    else return twmul_k<DIR>(a, w64<R>(M));
}

// where bfly64 leaves output k = k1 + 8 k2:  slot k2 + 8 k1  (an involution: slot64(slot64(k)) == k)
MDSP_HD constexpr int slot64(int k) { return (k >> 3) + 8 * (k & 7); }

// second half of the 64-point butterfly: v[n1 + 8 k1] = A[n1][k1] (the first radix-8 layer's results) -> v[slot64(k)] = X[k]
template <int DIR, int K1 = 0, typename R> MDSP_HD void bfly64_tail(cx<R> (&v)[64]) {
    if constexpr (K1 < 8) {
        cx<R> u[8];
        u[0] = v[8 * K1];
        u[1] = mul_w64<DIR, (1 * K1) & 63>(v[1 + 8 * K1]);
        u[2] = mul_w64<DIR, (2 * K1) & 63>(v[2 + 8 * K1]);
        u[3] = mul_w64<DIR, (3 * K1) & 63>(v[3 + 8 * K1]);
        u[4] = mul_w64<DIR, (4 * K1) & 63>(v[4 + 8 * K1]);
        u[5] = mul_w64<DIR, (5 * K1) & 63>(v[5 + 8 * K1]);
        u[6] = mul_w64<DIR, (6 * K1) & 63>(v[6 + 8 * K1]);
        u[7] = mul_w64<DIR, (7 * K1) & 63>(v[7 + 8 * K1]);
        bfly8<DIR>(u);
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) v[k2 + 8 * K1] = u[k2];
        bfly64_tail<DIR, K1 + 1>(v);
    }
}

// 64-point transform in registers: natural order in, X[k] in v[slot64(k)] out
template <int DIR, typename R> MDSP_HD void bfly64(cx<R> (&v)[64]) {
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
        cx<R> u[8];
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) u[n2] = v[n1 + 8 * n2];
        bfly8<DIR>(u);
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) v[n1 + 8 * k1] = u[k1];
    }
    bfly64_tail<DIR>(v);
}

// ---- the Welch first layer: two real frames, window folded in ------------------------------------------------------------------------
// A unit transforms z[n] = w[n] (a[n] + i b[n]) with frame b starting half a frame after frame a: with the half-frames H0, H1, H2 of the signal,
//     z[p] = w[p] (H0[p] + i H1[p]),   z[p + N/2] = w[p + N/2] (H1[p] + i H2[p]),   p < N/2.
// The first radix-2 layer of the 64-point butterfly pairs exactly these two: S = z[p] + z[p + N/2], D = z[p] - z[p + N/2].  With the operands
// kept as  xp = (H0[p], H2[p]),  h1 = H1[p]  and  wp = (w[p], w[p + N/2])  that is three packed operations and no register holds H1 twice:
//     T = xp * wp (lane-wise);   S = (h1 w_hi + T.x,  h1 w_lo + T.y);   D = (T.x - h1 w_hi,  h1 w_lo - T.y)
template <typename R> MDSP_HD void win_sd(cx<R> xp, R h1, cx<R> wp, cx<R>& S, cx<R>& D) {
    const cx<R> T = {xp.x * wp.x, xp.y * wp.y};
    S = {h1 * wp.y + T.x, h1 * wp.x + T.y};
    D = {T.x - h1 * wp.y, h1 * wp.x - T.y};
}
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MDSP_NO_PACKED_F32)
// h1 rides in one half (HALF) of the pair hh; op_sel broadcasts it to both result halves and crosses the window pair
template <int HALF> __device__ __forceinline__ void win_sd_pk(cx<float> xp, cx<float> hh, cx<float> wp, cx<float>& S, cx<float>& D) {
    const cx<float> T = pk_mul(xp, wp);
    f2v s, d;
    if constexpr (HALF == 0) {
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,0,1]" : "=v"(s) : "v"(f2v{hh.x, hh.y}), "v"(f2v{wp.x, wp.y}), "v"(f2v{T.x, T.y}));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0] neg_hi:[0,0,1]"
            : "=v"(d) : "v"(f2v{hh.x, hh.y}), "v"(f2v{wp.x, wp.y}), "v"(f2v{T.x, T.y}));
    } else {
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(s) : "v"(f2v{hh.x, hh.y}), "v"(f2v{wp.x, wp.y}), "v"(f2v{T.x, T.y}));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[0,0,1]"
            : "=v"(d) : "v"(f2v{hh.x, hh.y}), "v"(f2v{wp.x, wp.y}), "v"(f2v{T.x, T.y}));
    }
    S = {s.x, s.y};
    D = {d.x, d.y};
}
#endif

// Pass A of a Welch unit, one first-layer group at a time (callers load group n1 + 1 while group n1 is being evaluated): with e_j = n1 + 8 j,
//   xp[j] = (H0, H2)[t + 64 e_j],  hh[j / 2] = (H1[t + 64 e_j], H1[t + 64 e_{j+1}]) for j = 0, 2 (a two-address LDS read),  wp[j] = (w[t + 64 e_j], w[t + 64 e_j + N/2])
// -> v[n1 + 8 k1] = A[n1][k1], the operands of bfly64_tail.  FRAME_B = false (the channel's odd last frame): frame b does not exist, its
// component is forced to zero.
template <bool FRAME_B = true, typename R> MDSP_HD void passA_welch_group(int n1, const cx<R> (&xp)[4], const cx<R> (&hh)[2], const cx<R> (&wp)[4], cx<R> (&v)[64]) {
    cx<R> S[4], D[4], o[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MDSP_NO_PACKED_F32)
        if constexpr (sizeof(R) == 4 && FRAME_B) {
            if (j & 1) win_sd_pk<1>(xp[j], hh[j >> 1], wp[j], S[j], D[j]);
            else win_sd_pk<0>(xp[j], hh[j >> 1], wp[j], S[j], D[j]);
        } else
#endif
        {
            win_sd(xp[j], (j & 1) ? hh[j >> 1].y : hh[j >> 1].x, wp[j], S[j], D[j]);
            if constexpr (!FRAME_B) {
                S[j].y = (R)0;
                D[j].y = (R)0;
            }
        }
    }
    bfly8_sd<-1>(S, D, o);
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) v[n1 + 8 * k1] = o[k1];
}
// the whole pass from full operand arrays (the host emulation; e' = 0..31 as in the header comment): -> v[slot64(ke)] = Y_t[ke]
template <bool FRAME_B = true, typename R> MDSP_HD void passA_welch(const cx<R> (&xp)[32], const R (&h1)[32], const cx<R> (&wp)[32], cx<R> (&v)[64]) {
    for (int n1 = 0; n1 < 8; ++n1) {
        const cx<R> x4[4] = {xp[n1], xp[n1 + 8], xp[n1 + 16], xp[n1 + 24]}, w4[4] = {wp[n1], wp[n1 + 8], wp[n1 + 16], wp[n1 + 24]};
        const cx<R> h2[2] = {{h1[n1], h1[n1 + 8]}, {h1[n1 + 16], h1[n1 + 24]}};
        passA_welch_group<FRAME_B>(n1, x4, h2, w4, v);
    }
    bfly64_tail<-1>(v);
}

// ---- the 64 x 64 transposition inside a wavefront -------------------------------------------------------------------------------------
// Logical register r of lane l holds M[l][r]; afterwards register r of lane l holds M[r][l].
//   step 1 (registers only): for r < 32, v_permlane32_swap(reg r, reg r + 32) exchanges lanes 32..63 of reg r with lanes 0..31 of reg r + 32.  Now each
//           32-lane half holds, in registers 0..31 and again in 32..63, a 32 x 32 block that must be transposed WITHIN that half.
//   step 2: two rounds (registers 0..31, then 32..63) through a buffer of 2 halves x 32 rows x 33 elements: lane l' of half h writes register r'
//           to row l', column r'; reads register T' from row T', column l'.  ds_write_b64: the 16 lanes of a group hit 16 distinct bank pairs (row
//           stride 66 dwords); ds_read_b64: 32 lanes read 64 consecutive dwords.  Both conflict-free.
constexpr int XP64_ROW = 33;                         // elements per row (one of padding)
constexpr int XP64_ELEMS = 2 * 32 * XP64_ROW;        // 2112 elements = 16.5 KiB of Float32 pairs
MDSP_HD int xp64_write_index(int lane, int r) { return ((lane >> 5) * 32 + (lane & 31)) * XP64_ROW + r; }   // r = register mod 32
MDSP_HD int xp64_read_index(int lane, int T) { return ((lane >> 5) * 32 + T) * XP64_ROW + (lane & 31); }    // T = register mod 32

}  // namespace fft
}  // namespace mdsp
