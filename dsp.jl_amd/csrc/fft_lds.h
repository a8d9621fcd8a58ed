// Workgroup-cooperative power-of-two complex FFT held in registers + LDS (the transform inside the fused
// MDSP_ENGINE_FUSED kernels).  Stockham autosort formulation, radix <= 16 register butterflies.
//
// Layout invariant (what makes the fused kernels cheap): a transform of N points is shared by T = N/E threads;
// thread t holds the E elements  X[t + T*e], e = 0..E-1  BEFORE the first pass AND AFTER the last pass, for the
// forward and the inverse transform alike.  So
//   * the first pass reads its operands straight from registers that were filled by coalesced global loads
//     (lane t -> address base + t + T*e),
//   * the last pass leaves natural-order results in the same slots (coalesced stores, or the input of the
//     inverse transform with no exchange in between),
//   * only the P-1 inter-pass exchanges go through LDS (write scattered with stride Ns, read contiguous).
//
// Everything here is plain C++ (no HIP builtins) and MDSP_HD-qualified so that tests/cpu_harness can run the
// exact same code thread-by-thread on the host against numpy.
#pragma once

#include <utility>

#ifdef __HIPCC__
#define MDSP_HD __host__ __device__ __forceinline__
#else
#define MDSP_HD inline
#endif
// Make a per-lane integer opaque to the optimiser (device only).  Used to stop LICM from hoisting the LDS twiddle
// reads of a persistent loop into ~40 extra VGPRs -- the point of the LDS table is to NOT hold them in registers.
#if defined(__HIP_DEVICE_COMPILE__)
#define MDSP_OPAQUE_INT(x) asm volatile("" : "+v"(x))
#else
#define MDSP_OPAQUE_INT(x) (void)(x)
#endif

namespace mdsp {
namespace fft {

template <typename R> struct cx {
    R x, y;
};

template <typename R> MDSP_HD cx<R> cadd(cx<R> a, cx<R> b) { return {a.x + b.x, a.y + b.y}; }
template <typename R> MDSP_HD cx<R> csub(cx<R> a, cx<R> b) { return {a.x - b.x, a.y - b.y}; }
template <typename R> MDSP_HD cx<R> cmul(cx<R> a, cx<R> b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
// a * conj(b)
template <typename R> MDSP_HD cx<R> cmulc(cx<R> a, cx<R> b) { return {a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y}; }
// multiply by the table root (forward: w = exp(-2 pi i k/N)) or its conjugate (inverse)
template <int DIR, typename R> MDSP_HD cx<R> twmul(cx<R> a, cx<R> w) { return DIR < 0 ? cmul(a, w) : cmulc(a, w); }
// multiply by -i (forward) / +i (inverse)
template <int DIR, typename R> MDSP_HD cx<R> mul_mi(cx<R> a) { return DIR < 0 ? cx<R>{a.y, -a.x} : cx<R>{-a.y, a.x}; }

// a + mul_mi<DIR>(b)  /  a - mul_mi<DIR>(b)  /  e + h u  /  e - h u   (h real): the shapes the butterflies are made of
template <int DIR, typename R> MDSP_HD cx<R> add_mi(cx<R> a, cx<R> b) { return cadd(a, mul_mi<DIR>(b)); }
template <int DIR, typename R> MDSP_HD cx<R> sub_mi(cx<R> a, cx<R> b) { return csub(a, mul_mi<DIR>(b)); }
template <typename R> MDSP_HD cx<R> caxpy(R h, cx<R> u, cx<R> e) { return {e.x + h * u.x, e.y + h * u.y}; }
template <typename R> MDSP_HD cx<R> caxmy(R h, cx<R> u, cx<R> e) { return {e.x - h * u.x, e.y - h * u.y}; }
template <typename R> MDSP_HD cx<R> cscale(R h, cx<R> u) { return {h * u.x, h * u.y}; }
// lane-wise a b + c on the (re, im) pair (NOT a complex product): the |z|^2 accumulation of the spectral kernels
template <typename R> MDSP_HD cx<R> lanefma(cx<R> a, cx<R> b, cx<R> c) { return {a.x * b.x + c.x, a.y * b.y + c.y}; }
// both halves of a times ONE half of the pair w (a real window pair {w_lo, w_hi}), optionally +- c: the window folded into the first butterfly stage
template <typename R> MDSP_HD cx<R> wmul_hi(cx<R> a, cx<R> w) { return {a.x * w.y, a.y * w.y}; }
template <typename R> MDSP_HD cx<R> wfma_lo(cx<R> a, cx<R> w, cx<R> c) { return {a.x * w.x + c.x, a.y * w.x + c.y}; }
template <typename R> MDSP_HD cx<R> wfms_lo(cx<R> a, cx<R> w, cx<R> c) { return {a.x * w.x - c.x, a.y * w.x - c.y}; }
// products with COMPILE-TIME constants (the W16 / W8 roots inside the butterflies); the packed Float32 forms keep the constant in an SGPR pair
template <int DIR, typename R> MDSP_HD cx<R> twmul_k(cx<R> a, cx<R> w) { return twmul<DIR>(a, w); }
template <typename R> MDSP_HD cx<R> cscale_k(R h, cx<R> u) { return cscale(h, u); }
template <typename R> MDSP_HD cx<R> caxpy_k(R h, cx<R> u, cx<R> e) { return caxpy(h, u, e); }
template <typename R> MDSP_HD cx<R> caxmy_k(R h, cx<R> u, cx<R> e) { return caxmy(h, u, e); }

#if defined(__HIP_DEVICE_COMPILE__) && !defined(MDSP_NO_PACKED_F32)
// ---------------------------------------------------------------------------------- packed-FP32 complex arithmetic
// gfx950 issues v_pk_{add,mul,fma}_f32 at the rate of their scalar forms, and a complex number IS a (lo, hi) register
// pair, so every complex add / rotate-by-i / twiddle product below is one or two VOP3P instructions whose op_sel /
// neg modifiers do the swaps and sign flips for free.  hipcc's SLP vectoriser finds only some of these and pays for
// the rest with v_mov shuffles (a third of the VALU stream of the fused kernels), hence the explicit forms.
// Non-volatile asm: the optimiser may still schedule, CSE and delete them.
typedef float f2v __attribute__((ext_vector_type(2)));
#define MDSP_PK2(NAME, INSN, MODS)                                                                  \
    __device__ __forceinline__ cx<float> NAME(cx<float> a, cx<float> b) {                           \
        f2v d;                                                                                      \
        asm(INSN " %0, %1, %2 " MODS : "=v"(d) : "v"(f2v{a.x, a.y}), "v"(f2v{b.x, b.y}));           \
        return {d.x, d.y};                                                                          \
    }
#define MDSP_PK3(NAME, INSN, MODS)                                                                  \
    __device__ __forceinline__ cx<float> NAME(cx<float> a, cx<float> b, cx<float> c) {              \
        f2v d;                                                                                      \
        asm(INSN " %0, %1, %2, %3 " MODS : "=v"(d) : "v"(f2v{a.x, a.y}), "v"(f2v{b.x, b.y}), "v"(f2v{c.x, c.y})); \
        return {d.x, d.y};                                                                          \
    }
// MDSP_PK_NATIVE: the forms the compiler can express itself (plain add / subtract / multiply / FMA, broadcasts and swaps of a multiplicand --
// it folds those into op_sel, and a whole-operand negation into neg_lo + neg_hi) are written on float2 vectors instead of asm.  What that buys:
// hipcc pads every asm statement whose result the NEXT instruction reads with an `s_nop 0` (the gfx940+ dst_sel-forwarding hazard: the asm
// might have written half a register), 240 of the 1994 instructions of welch_w64_kernel's unit loop; its own v_pk_* need no pad, and it knows
// their latency when it schedules.  The forms with a per-half negation (a +- i b, the second step of a complex product) stay asm: the
// compiler spends a v_xor + v_mov on those.
#ifndef MDSP_PK_NATIVE
#define MDSP_PK_NATIVE 0
#endif
#if MDSP_PK_NATIVE
#define MDSP_V(a) (f2v{(a).x, (a).y})
__device__ __forceinline__ cx<float> mdsp_c(f2v d) { return {d.x, d.y}; }
__device__ __forceinline__ cx<float> cadd(cx<float> a, cx<float> b) { return mdsp_c(MDSP_V(a) + MDSP_V(b)); }
__device__ __forceinline__ cx<float> csub(cx<float> a, cx<float> b) { return mdsp_c(MDSP_V(a) - MDSP_V(b)); }
#else
MDSP_PK2(cadd, "v_pk_add_f32", "")
MDSP_PK2(csub, "v_pk_add_f32", "neg_lo:[0,1] neg_hi:[0,1]")
#endif
MDSP_PK2(pk_add_ib, "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")   // a + i b = (ax - by, ay + bx)
MDSP_PK2(pk_sub_ib, "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")   // a - i b = (ax + by, ay - bx)
#if MDSP_PK_NATIVE
__device__ __forceinline__ cx<float> pk_mul(cx<float> a, cx<float> b) { return mdsp_c(MDSP_V(a) * MDSP_V(b)); }
__device__ __forceinline__ cx<float> pk_mul_swap(cx<float> a, cx<float> b) { return mdsp_c(f2v{a.y, a.x} * MDSP_V(b)); }
__device__ __forceinline__ cx<float> pk_mul_blo(cx<float> a, cx<float> b) { return mdsp_c(MDSP_V(a) * f2v{b.x, b.x}); }
__device__ __forceinline__ cx<float> pk_mul_bhi(cx<float> a, cx<float> b) { return mdsp_c(MDSP_V(a) * f2v{b.y, b.y}); }
__device__ __forceinline__ cx<float> pk_mul_yy(cx<float> a, cx<float> b) { return mdsp_c(f2v{a.y, a.y} * f2v{b.y, b.x}); }
__device__ __forceinline__ cx<float> pk_fma(cx<float> a, cx<float> b, cx<float> c) { return mdsp_c(__builtin_elementwise_fma(MDSP_V(a), MDSP_V(b), MDSP_V(c))); }
__device__ __forceinline__ cx<float> pk_fnma(cx<float> a, cx<float> b, cx<float> c) { return mdsp_c(__builtin_elementwise_fma(-MDSP_V(a), MDSP_V(b), MDSP_V(c))); }
#else
MDSP_PK2(pk_mul, "v_pk_mul_f32", "")
MDSP_PK2(pk_mul_swap, "v_pk_mul_f32", "op_sel:[1,0] op_sel_hi:[0,1]")              // (a.y b.x, a.x b.y)
MDSP_PK2(pk_mul_blo, "v_pk_mul_f32", "op_sel_hi:[1,0]")                          // (a.x b.x, a.y b.x): both lanes times b's low half
MDSP_PK2(pk_mul_bhi, "v_pk_mul_f32", "op_sel:[0,1] op_sel_hi:[1,1]")              // (a.x b.y, a.y b.y): both lanes times b's high half
MDSP_PK2(pk_mul_yy, "v_pk_mul_f32", "op_sel:[1,1] op_sel_hi:[1,0]")                // (a.y b.y, a.y b.x)
MDSP_PK3(pk_fma, "v_pk_fma_f32", "")                                               // a b + c, lane-wise
MDSP_PK3(pk_fnma, "v_pk_fma_f32", "neg_lo:[1,0,0] neg_hi:[1,0,0]")                 // c - a b
#endif
MDSP_PK3(pk_cmul_fin, "v_pk_fma_f32", "op_sel_hi:[0,1,1] neg_lo:[0,0,1]")          // (ax bx - c.x, ax by + c.y)
MDSP_PK3(pk_cmulc_fin, "v_pk_fma_f32", "op_sel_hi:[0,1,1] neg_hi:[1,0,0]")         // (ax bx + c.x, -ax by + c.y)
#undef MDSP_PK2
#undef MDSP_PK3
// MDSP_PK_FUSED: dependent packed instructions in ONE asm statement.  hipcc pads every (producer, consumer) pair of which at least one is an
// asm statement with an `s_nop 0` unless another non-asm instruction sits between them (the gfx940+ dst_sel-forwarding hazard rule cannot see
// inside an asm, and asm statements count as zero wait states): with one instruction per statement that was 171 s_nop in the 2333
// instructions of welch_half3_kernel, 11 % of overlap-save's unit loop.  The two halves of a complex product and the eight instructions of a
// radix-4 butterfly as single statements leave at most one pad per statement; the arithmetic (operations, operands, rounding) is unchanged.
#ifndef MDSP_PK_FUSED
#define MDSP_PK_FUSED 1
#endif
#if MDSP_PK_FUSED
__device__ __forceinline__ cx<float> cmul(cx<float> a, cx<float> w) {
    f2v d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1] neg_lo:[0,0,1]"
        : "=&v"(d) : "v"(f2v{a.x, a.y}), "v"(f2v{w.x, w.y}));
    return {d.x, d.y};
}
__device__ __forceinline__ cx<float> cmulc(cx<float> a, cx<float> w) {
    f2v d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
        : "=&v"(d) : "v"(f2v{a.x, a.y}), "v"(f2v{w.x, w.y}));
    return {d.x, d.y};
}
// two products at once, interleaved (the twiddle loops of the passes): no dependent pair is adjacent
__device__ __forceinline__ void cmul2(cx<float>& a0, cx<float> w0, cx<float>& a1, cx<float> w1, bool conj) {
    f2v d0, d1;
    if (!conj)
        asm("v_pk_mul_f32 %0, %2, %3 op_sel:[1,1] op_sel_hi:[1,0]\n\tv_pk_mul_f32 %1, %4, %5 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
            "v_pk_fma_f32 %0, %2, %3, %0 op_sel_hi:[0,1,1] neg_lo:[0,0,1]\n\tv_pk_fma_f32 %1, %4, %5, %1 op_sel_hi:[0,1,1] neg_lo:[0,0,1]"
            : "=&v"(d0), "=&v"(d1) : "v"(f2v{a0.x, a0.y}), "v"(f2v{w0.x, w0.y}), "v"(f2v{a1.x, a1.y}), "v"(f2v{w1.x, w1.y}));
    else
        asm("v_pk_mul_f32 %0, %2, %3 op_sel:[1,1] op_sel_hi:[1,0]\n\tv_pk_mul_f32 %1, %4, %5 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
            "v_pk_fma_f32 %0, %2, %3, %0 op_sel_hi:[0,1,1] neg_hi:[1,0,0]\n\tv_pk_fma_f32 %1, %4, %5, %1 op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
            : "=&v"(d0), "=&v"(d1) : "v"(f2v{a0.x, a0.y}), "v"(f2v{w0.x, w0.y}), "v"(f2v{a1.x, a1.y}), "v"(f2v{w1.x, w1.y}));
    a0 = {d0.x, d0.y};
    a1 = {d1.x, d1.y};
}
#else
__device__ __forceinline__ cx<float> cmul(cx<float> a, cx<float> w) { return pk_cmul_fin(a, w, pk_mul_yy(a, w)); }
__device__ __forceinline__ cx<float> cmulc(cx<float> a, cx<float> w) { return pk_cmulc_fin(a, w, pk_mul_yy(a, w)); }
#endif
template <int DIR> __device__ __forceinline__ cx<float> add_mi(cx<float> a, cx<float> b) { return DIR < 0 ? pk_sub_ib(a, b) : pk_add_ib(a, b); }
template <int DIR> __device__ __forceinline__ cx<float> sub_mi(cx<float> a, cx<float> b) { return DIR < 0 ? pk_add_ib(a, b) : pk_sub_ib(a, b); }
template <int DIR> __device__ __forceinline__ cx<float> mul_mi(cx<float> a) {
    return pk_mul_swap(a, DIR < 0 ? cx<float>{1.f, -1.f} : cx<float>{-1.f, 1.f});
}
__device__ __forceinline__ cx<float> caxpy(float h, cx<float> u, cx<float> e) { return pk_fma(u, cx<float>{h, h}, e); }
__device__ __forceinline__ cx<float> caxmy(float h, cx<float> u, cx<float> e) { return pk_fnma(u, cx<float>{h, h}, e); }
__device__ __forceinline__ cx<float> cscale(float h, cx<float> u) { return pk_mul(u, cx<float>{h, h}); }
__device__ __forceinline__ cx<float> lanefma(cx<float> a, cx<float> b, cx<float> c) { return pk_fma(a, b, c); }
__device__ __forceinline__ cx<float> wmul_hi(cx<float> a, cx<float> w) { return pk_mul_bhi(a, w); }
__device__ __forceinline__ cx<float> wfma_lo(cx<float> a, cx<float> w, cx<float> c) {
    f2v d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(f2v{a.x, a.y}), "v"(f2v{w.x, w.y}), "v"(f2v{c.x, c.y}));
    return {d.x, d.y};
}
__device__ __forceinline__ cx<float> wfms_lo(cx<float> a, cx<float> w, cx<float> c) {
    f2v d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(d) : "v"(f2v{a.x, a.y}), "v"(f2v{w.x, w.y}), "v"(f2v{c.x, c.y}));
    return {d.x, d.y};
}
// Constant operands from SGPR pairs (VOP3P takes one scalar source): the butterflies' roots cost neither VGPRs nor the v_mov_b64 that
// re-materialised them in front of every use (9 per radix-16 butterfly in the round-2 ISA).  -DMDSP_FFT_SGPR_CONST=0 restores the VGPR forms.
#ifndef MDSP_FFT_SGPR_CONST
#define MDSP_FFT_SGPR_CONST 1
#endif
#if MDSP_FFT_SGPR_CONST
#define MDSP_PK2S(NAME, INSN, MODS)                                                                 \
    __device__ __forceinline__ cx<float> NAME(cx<float> a, cx<float> k) {                           \
        f2v d;                                                                                      \
        asm(INSN " %0, %1, %2 " MODS : "=v"(d) : "v"(f2v{a.x, a.y}), "s"(f2v{k.x, k.y}));           \
        return {d.x, d.y};                                                                          \
    }
#define MDSP_PK3S(NAME, INSN, MODS)                                                                 \
    __device__ __forceinline__ cx<float> NAME(cx<float> a, cx<float> k, cx<float> c) {              \
        f2v d;                                                                                      \
        asm(INSN " %0, %1, %2, %3 " MODS : "=v"(d) : "v"(f2v{a.x, a.y}), "s"(f2v{k.x, k.y}), "v"(f2v{c.x, c.y})); \
        return {d.x, d.y};                                                                          \
    }
#if MDSP_PK_NATIVE >= 2   // compile-time constants: the compiler chooses an SGPR pair (or an inline constant) itself
__device__ __forceinline__ cx<float> pks_mul(cx<float> a, cx<float> k) { return pk_mul(a, k); }
__device__ __forceinline__ cx<float> pks_mul_yy(cx<float> a, cx<float> k) { return pk_mul_yy(a, k); }
__device__ __forceinline__ cx<float> pks_fma(cx<float> a, cx<float> k, cx<float> c) { return pk_fma(a, k, c); }
__device__ __forceinline__ cx<float> pks_fnma(cx<float> a, cx<float> k, cx<float> c) { return pk_fnma(a, k, c); }
#else
MDSP_PK2S(pks_mul, "v_pk_mul_f32", "")
MDSP_PK2S(pks_mul_yy, "v_pk_mul_f32", "op_sel:[1,1] op_sel_hi:[1,0]")
MDSP_PK3S(pks_fma, "v_pk_fma_f32", "")
MDSP_PK3S(pks_fnma, "v_pk_fma_f32", "neg_lo:[1,0,0] neg_hi:[1,0,0]")
#endif
MDSP_PK3S(pks_cmul_fin, "v_pk_fma_f32", "op_sel_hi:[0,1,1] neg_lo:[0,0,1]")
MDSP_PK3S(pks_cmulc_fin, "v_pk_fma_f32", "op_sel_hi:[0,1,1] neg_hi:[1,0,0]")
#undef MDSP_PK2S
#undef MDSP_PK3S
#if MDSP_PK_FUSED
template <int DIR> __device__ __forceinline__ cx<float> twmul_k(cx<float> a, cx<float> w) {
    f2v d;
    if constexpr (DIR < 0)
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1] neg_lo:[0,0,1]"
            : "=&v"(d) : "v"(f2v{a.x, a.y}), "s"(f2v{w.x, w.y}));
    else
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
            : "=&v"(d) : "v"(f2v{a.x, a.y}), "s"(f2v{w.x, w.y}));
    return {d.x, d.y};
}
#else
template <int DIR> __device__ __forceinline__ cx<float> twmul_k(cx<float> a, cx<float> w) {
    return DIR < 0 ? pks_cmul_fin(a, w, pks_mul_yy(a, w)) : pks_cmulc_fin(a, w, pks_mul_yy(a, w));
}
#endif
__device__ __forceinline__ cx<float> cscale_k(float h, cx<float> u) { return pks_mul(u, cx<float>{h, h}); }
__device__ __forceinline__ cx<float> caxpy_k(float h, cx<float> u, cx<float> e) { return pks_fma(u, cx<float>{h, h}, e); }
__device__ __forceinline__ cx<float> caxmy_k(float h, cx<float> u, cx<float> e) { return pks_fnma(u, cx<float>{h, h}, e); }
#endif
#endif

constexpr int ilog2(int n) { return n <= 1 ? 0 : 1 + ilog2(n >> 1); }

// ------------------------------------------------------------------------------------------------ butterflies
template <int DIR, typename R> MDSP_HD void bfly2(cx<R>& a, cx<R>& b) {
    const cx<R> t = csub(a, b);
    a = cadd(a, b);
    b = t;
}

#if defined(__HIP_DEVICE_COMPILE__) && !defined(MDSP_NO_PACKED_F32) && MDSP_PK_FUSED
// the radix-4 butterfly as ONE statement of eight packed instructions, in place with one temporary (see MDSP_PK_FUSED above):
//   T0 = a0 + a2, T1 = a0 - a2, T2 = a1 + a3, D = a1 - a3;  X0 = T0 + T2, X2 = T0 - T2, X1 = T1 -+ i D, X3 = T1 +- i D  (forward: X1 = T1 - i D)
// no instruction reads the result of its predecessor.  X0 lands in a0's registers, X2 in a1's, X1 in the temporary, X3 in a3's: the renaming
// behind the statement is free.
template <int DIR> __device__ __forceinline__ void bfly4(cx<float>& a0, cx<float>& a1, cx<float>& a2, cx<float>& a3) {
    f2v r0 = {a0.x, a0.y}, r1 = {a1.x, a1.y}, r2 = {a2.x, a2.y}, r3 = {a3.x, a3.y}, t;
    if constexpr (DIR < 0)
        asm("v_pk_add_f32 %4, %0, %2\n\t"                                               // t  = T0
            "v_pk_add_f32 %2, %0, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"                     // r2 = T1
            "v_pk_add_f32 %0, %1, %3\n\t"                                               // r0 = T2
            "v_pk_add_f32 %3, %1, %3 neg_lo:[0,1] neg_hi:[0,1]\n\t"                     // r3 = D
            "v_pk_add_f32 %1, %4, %0 neg_lo:[0,1] neg_hi:[0,1]\n\t"                     // r1 = X2 = T0 - T2
            "v_pk_add_f32 %0, %4, %0\n\t"                                               // r0 = X0 = T0 + T2
            "v_pk_add_f32 %4, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"     // t  = X1 = T1 - i D
            "v_pk_add_f32 %3, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"           // r3 = X3 = T1 + i D
            : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "=&v"(t));
    else
        asm("v_pk_add_f32 %4, %0, %2\n\t"
            "v_pk_add_f32 %2, %0, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %0, %1, %3\n\t"
            "v_pk_add_f32 %3, %1, %3 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %1, %4, %0 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %0, %4, %0\n\t"
            "v_pk_add_f32 %4, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"     // t  = X1 = T1 + i D
            "v_pk_add_f32 %3, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]"           // r3 = X3 = T1 - i D
            : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "=&v"(t));
    a0 = {r0.x, r0.y};
    a1 = {t.x, t.y};
    a2 = {r1.x, r1.y};
    a3 = {r3.x, r3.y};
}
#endif

// natural-order in, natural-order out
template <int DIR, typename R> MDSP_HD void bfly4(cx<R>& a0, cx<R>& a1, cx<R>& a2, cx<R>& a3) {
    const cx<R> t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), d = csub(a1, a3);
    a0 = cadd(t0, t2);
    a2 = csub(t0, t2);
    a1 = add_mi<DIR>(t1, d);
    a3 = sub_mi<DIR>(t1, d);
}

// o W8 / h  and  o W8^3 / (-h)  with W8 = exp(-+ 2 pi i/8) = h (1 -+ i):  o + mul_mi(o)  and  o - mul_mi(o)
template <int DIR, typename R> MDSP_HD cx<R> w8_1_unscaled(cx<R> o) { return add_mi<DIR>(o, o); }
template <int DIR, typename R> MDSP_HD cx<R> w8_3_unscaled_neg(cx<R> o) { return sub_mi<DIR>(o, o); }

template <int DIR, typename R> MDSP_HD void bfly8(cx<R> (&v)[8]) {
    constexpr R h = (R)0.70710678118654752440084436210485L;
    // even / odd radix-4 sub-transforms: E[k] lands in v[2k], O[k] in v[2k+1]
    bfly4<DIR>(v[0], v[2], v[4], v[6]);
    bfly4<DIR>(v[1], v[3], v[5], v[7]);
    const cx<R> e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    const cx<R> o0 = v[1], o2 = v[5];
    const cx<R> u1 = w8_1_unscaled<DIR>(v[3]);       // O[1] W8   =  h u1
    const cx<R> u3 = w8_3_unscaled_neg<DIR>(v[7]);   // O[3] W8^3 = -h u3
    v[0] = cadd(e0, o0);
    v[4] = csub(e0, o0);
    v[1] = caxpy_k(h, u1, e1);
    v[5] = caxmy_k(h, u1, e1);
    v[2] = add_mi<DIR>(e2, o2);                      // O[2] W8^2 = mul_mi(O[2])
    v[6] = sub_mi<DIR>(e2, o2);
    v[3] = caxmy_k(h, u3, e3);
    v[7] = caxpy_k(h, u3, e3);
}

// everything of the radix-16 butterfly after the four inner DFT4s: v[m + 4q] = y[m][q] in, natural order out
template <int DIR, typename R> MDSP_HD void bfly16_tail(cx<R> (&v)[16]) {
    constexpr R c1 = (R)0.92387953251128675612818318939679L;  // cos(pi/8)
    constexpr R s1 = (R)0.38268343236508977172845998403040L;  // sin(pi/8)
    constexpr R h = (R)0.70710678118654752440084436210485L;
    // twiddles W16^{mq}, forward root w = exp(-2 pi i/16) = (c1, -s1)
    const cx<R> w1 = {c1, -s1}, w3 = {s1, -c1}, w9 = {-c1, s1};
    v[1 + 4 * 1] = twmul_k<DIR>(v[1 + 4 * 1], w1);                          // m=1,q=1 : W^1
    v[1 + 4 * 2] = cscale_k(h, w8_1_unscaled<DIR>(v[1 + 4 * 2]));           // m=1,q=2 : W^2 = W8
    v[1 + 4 * 3] = twmul_k<DIR>(v[1 + 4 * 3], w3);                          // m=1,q=3 : W^3
    v[2 + 4 * 1] = cscale_k(h, w8_1_unscaled<DIR>(v[2 + 4 * 1]));           // m=2,q=1 : W^2
    v[2 + 4 * 2] = mul_mi<DIR>(v[2 + 4 * 2]);                               // m=2,q=2 : W^4 = -+i
    v[2 + 4 * 3] = cscale_k(-h, w8_3_unscaled_neg<DIR>(v[2 + 4 * 3]));      // m=2,q=3 : W^6 = W8^3
    v[3 + 4 * 1] = twmul_k<DIR>(v[3 + 4 * 1], w3);                          // m=3,q=1 : W^3
    v[3 + 4 * 2] = cscale_k(-h, w8_3_unscaled_neg<DIR>(v[3 + 4 * 2]));      // m=3,q=2 : W^6
    v[3 + 4 * 3] = twmul_k<DIR>(v[3 + 4 * 3], w9);                          // m=3,q=3 : W^9 = -W^1
    // outer DFT4 over m for each q; result p lands in slot m=p of the same group: X[4p+q] in v[p + 4q]
#pragma unroll
    for (int q = 0; q < 4; ++q) bfly4<DIR>(v[4 * q + 0], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    // reorder v[p + 4q] -> natural index 4p + q  (a 4x4 transpose of register names; free after unrolling)
    cx<R> t;
#define MDSP_SWAP(a, b) \
    t = v[a];           \
    v[a] = v[b];        \
    v[b] = t;
    MDSP_SWAP(1, 4) MDSP_SWAP(2, 8) MDSP_SWAP(3, 12) MDSP_SWAP(6, 9) MDSP_SWAP(7, 13) MDSP_SWAP(11, 14)
#undef MDSP_SWAP
}

template <int DIR, typename R> MDSP_HD void bfly16(cx<R> (&v)[16]) {
    // n = m + 4 s, k = 4 p + q :  X[4p+q] = sum_m W4^{mp} W16^{mq} ( sum_s v[m+4s] W4^{sq} )
#pragma unroll
    for (int m = 0; m < 4; ++m) bfly4<DIR>(v[m], v[m + 4], v[m + 8], v[m + 12]);  // y[m][q] in v[m + 4q]
    bfly16_tail<DIR>(v);
}

// The same butterfly on WINDOWED input with the window folded into the first add/subtract stage:
//     x[e] = q[e] w[e],  x[e + 8] = f[e] w[e + 8],  e = 0..7,   wp[e] = {w[e], w[e + 8]}  (real window values as a pair)
// a + b with a = q w_lo, b = f w_hi is one product and one FMA, a - b one more FMA: 3 operations per pair instead of 2 products + 2 additions
// (8 packed operations less per butterfly).  Rounding differs from "window, then transform" by one fused rounding per term.
template <int DIR, typename R> MDSP_HD void bfly16_win(const cx<R> (&q)[8], const cx<R> (&f)[8], const cx<R> (&wp)[8], cx<R> (&v)[16]) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const cx<R> p0 = wmul_hi(f[m], wp[m]), p1 = wmul_hi(f[m + 4], wp[m + 4]);
        const cx<R> t0 = wfma_lo(q[m], wp[m], p0), t1 = wfms_lo(q[m], wp[m], p0);              // x[m] +- x[m+8]
        const cx<R> t2 = wfma_lo(q[m + 4], wp[m + 4], p1), d = wfms_lo(q[m + 4], wp[m + 4], p1);  // x[m+4] +- x[m+12]
        v[m] = cadd(t0, t2);
        v[m + 8] = csub(t0, t2);
        v[m + 4] = add_mi<DIR>(t1, d);
        v[m + 12] = sub_mi<DIR>(t1, d);
    }
    bfly16_tail<DIR>(v);
}

template <int RDX, int DIR, typename R> MDSP_HD void bfly(cx<R> (&v)[RDX]) {
    if constexpr (RDX == 2) bfly2<DIR>(v[0], v[1]);
    else if constexpr (RDX == 4) bfly4<DIR>(v[0], v[1], v[2], v[3]);
    else if constexpr (RDX == 8) bfly8<DIR>(v);
    else bfly16<DIR>(v);
}

// v[r] *= w[r] (conjugated for the inverse), r = 1..RDX-1: the twiddle products in front of a butterfly (pairs of products per statement in the
// fused packed form)
template <int DIR, int RDX, typename R> MDSP_HD void twmul_all(cx<R> (&v)[RDX], const cx<R> (&w)[RDX]) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MDSP_NO_PACKED_F32) && MDSP_PK_FUSED
    if constexpr (sizeof(R) == 4) {
#pragma unroll
        for (int r = 1; r + 1 < RDX; r += 2) cmul2(v[r], w[r], v[r + 1], w[r + 1], DIR > 0);
        if constexpr (((RDX - 1) & 1) != 0) v[RDX - 1] = twmul<DIR>(v[RDX - 1], w[RDX - 1]);
        return;
    }
#endif
#pragma unroll
    for (int r = 1; r < RDX; ++r) v[r] = twmul<DIR>(v[r], w[r]);
}

// ------------------------------------------------------------------------------------------------ configuration
// N points, E elements per thread (power of two, 4..16), T = N/E threads per transform.
template <int N_, int E_> struct Cfg {
    static constexpr int N = N_, E = E_, T = N_ / E_;
    static constexpr int LOGN = ilog2(N_), LOGE = ilog2(E_);
    static_assert((1 << LOGN) == N_ && (1 << LOGE) == E_, "power of two sizes only");
    static_assert(E_ >= 2 && E_ <= 64 && N_ >= E_, "unsupported elements-per-thread");
    static constexpr int LOGR = LOGE < 4 ? LOGE : 4;    // register butterflies stop at radix 16; E > 16 runs E/16 of them per pass
    static constexpr int P = (LOGN + LOGR - 1) / LOGR;  // passes
    // spread the log2 radices as evenly as possible, larger radices first
    static constexpr int logradix(int p) { return LOGN / P + (p < LOGN % P ? 1 : 0); }
    static constexpr int radix(int p) { return 1 << logradix(p); }
    static constexpr int ns(int p) { return p == 0 ? 1 : ns(p - 1) * radix(p - 1); }
    // number of per-thread twiddles of pass p (pass 0 has none)
    static constexpr int ntw(int p) { return p == 0 ? 0 : E - E / radix(p); }
    static constexpr int twoff(int p) { return p == 0 ? 0 : twoff(p - 1) + ntw(p - 1); }
    static constexpr int NTW = twoff(P - 1) + ntw(P - 1);
    // LDS-resident twiddle table (TWMODE 2): pass p >= 1 stores W^{r k stride_p} at  ldsoff(p) + (r-1)*ns(p) + k,
    // r = 1..radix(p)-1, k = 0..ns(p)-1  (k contiguous: conflict-free reads, r and b become immediate offsets)
    static constexpr int ldstw(int p) { return p == 0 ? 0 : (radix(p) - 1) * ns(p); }
    static constexpr int ldsoff(int p) { return p <= 1 ? 0 : ldsoff(p - 1) + ldstw(p - 1); }
    static constexpr int NTWLDS = P <= 1 ? 1 : ldsoff(P - 1) + ldstw(P - 1);
};

// ------------------------------------------------------------------------------------------------ lane permutations
// Which butterfly a thread computes in a pass is free as long as reads and writes agree: thread t runs butterflies
// j = perm_p(t) + T*b of pass p, where perm_p permutes the low six bits of t (inside the wavefront, so global
// accesses stay coalesced).  tools/lds_perm_search.py picks, per geometry, the bit permutations (and padding) that
// minimise LDS bank conflicts under the MI355X banking rules (ds_write_b64: 16-lane groups / 32 banks; ds_read_b64:
// 32-lane groups / 64 banks): N = 4096, E = 16 becomes conflict-free with one element of padding per 32
// (192 array cycles per transform and wave instead of 256), N = 2048, E = 8 drops from 224 to 160.
// Measured on MI355X (profiles/r01e_tune_lanes.json): the permuted schedules lose to identity lanes + the same padding,
// because the permuted pass-0 ownership also permutes the lanes of the global loads / stores (same cache lines per
// wave, but no longer ascending by lane) -- that costs more than the saved LDS cycles.  They stay available as tuning
// variants (PERMUTE template flags); the defaults use identity lanes with pad shift 5.
// The first and the last pass share their permutation so a transform's input and output ownership coincide:
// thread t holds X[io_lane(t) + T*e] before the first and after the last pass.
template <int N, int E> struct LanePerm {
    static constexpr bool any = false;
    static constexpr int src(int, int bit) { return bit; }
    static constexpr int padshift = 4;
};
template <> struct LanePerm<2048, 8> {   // radices 8 8 8 4
    static constexpr bool any = true;
    static constexpr int src(int pass, int bit) {
        constexpr int A[6] = {0, 4, 1, 2, 3, 5}, B[6] = {0, 1, 2, 4, 5, 3};
        return (pass == 0 || pass == 3) ? A[bit] : (pass == 1 ? B[bit] : bit);
    }
    static constexpr int padshift = 5;
};
template <> struct LanePerm<4096, 16> {  // radices 16 16 16
    static constexpr bool any = true;
    static constexpr int src(int pass, int bit) {
        constexpr int A[6] = {4, 0, 1, 2, 3, 5};
        return (pass == 0 || pass == 2) ? A[bit] : bit;
    }
    static constexpr int padshift = 5;
};
// PERMUTE = 0 reproduces the identity mapping (kernels that were not re-tuned, and the host emulation's baseline), 1 the
// bank-conflict permutations above.
//
// PERMUTE = 2, "wave-private last exchange" (N = 4096, E = 16, T = 256, radices 16 16 16): passes 1 and 2 run butterfly
// j = nibble_swap(t) = (t mod 16) * 16 + t div 16.  With that ownership the 16 operands of a pass-2 butterfly are produced by the 16
// lanes of ONE 16-lane row of one wavefront (writer lane l, output r  ->  reader lane (l div 16) * 16 + r, operand l mod 16: a 16 x 16
// transpose inside each row), so the second exchange needs no s_barrier at all -- a wave's DS operations execute in order -- and only the
// first exchange (all-to-all between the four waves) keeps its barrier.  Input ownership stays t + T*e (coalesced loads); the OUTPUT
// ownership becomes nibble_swap(t) + T*e, which is why this mode is used where outputs are reduced, not stored (Welch).
template <typename C> constexpr bool wave_private_ok() { return C::N == 4096 && C::E == 16 && C::P == 3; }
template <typename C, int PASS, int PERMUTE> MDSP_HD int lane_perm(int t) {
    using LP = LanePerm<C::N, C::E>;
    if constexpr (PERMUTE == 2) {
        static_assert(wave_private_ok<C>(), "wave-private exchange is wired for N = 4096, E = 16");
        if constexpr (PASS == 0) return t;
        else return ((t & 15) << 4) | ((t >> 4) & 15);
    } else if constexpr (!PERMUTE || !LP::any || C::T < 64) return t;
    else {
        int out = t & ~63;
#pragma unroll
        for (int d = 0; d < 6; ++d) out |= ((t >> LP::src(PASS, d)) & 1) << d;
        return out;
    }
}
template <typename C, int PERMUTE> MDSP_HD int io_lane(int t) {
    static_assert(PERMUTE != 1 || !LanePerm<C::N, C::E>::any || LanePerm<C::N, C::E>::src(0, 0) == LanePerm<C::N, C::E>::src(C::P - 1, 0),
                  "first and last pass must share their lane permutation");
    return lane_perm<C, 0, PERMUTE>(t);
}
// ownership AFTER the last pass: thread t holds X[out_lane(t) + T*e]  (== io_lane(t) except in the wave-private mode)
template <typename C, int PERMUTE> MDSP_HD int out_lane(int t) { return lane_perm<C, C::P - 1, PERMUTE>(t); }
// wave-private exchange: element (row-local lane a, slot b) of wavefront w, 16-lane row q lives at  w*4*272 + q*272 + a*17 + b
// (17 = one element of padding per 16: ds_write_b64 from 16 contiguous lanes and ds_read_b64 from 32 are both conflict-free)
MDSP_HD int wave_private_base(int t_raw) { return (t_raw >> 6) * (4 * 272) + ((t_raw >> 4) & 3) * 272; }

// twiddle sources
enum { TW_GLOBAL = 0, TW_REG = 1, TW_LDS = 2, TW_HYB = 3 };
// TW_HYB: passes whose twiddles depend on few distinct k (Ns <= HYB_NS_MAX: a (radix-1) x Ns table of <= 2 KiB) read
// them from LDS -- consecutive lanes read consecutive entries, lanes l and l+Ns broadcast -- and only the wide
// passes keep theirs in VGPRs.  For N = 4096, E = 16 this frees 30 of the 60 twiddle registers.
constexpr int HYB_NS_MAX = 16;
template <typename C, int TWMODE, int PASS> constexpr bool tw_pass_in_lds() {
    return TWMODE == TW_LDS || (TWMODE == TW_HYB && PASS >= 1 && C::ns(PASS) <= HYB_NS_MAX);
}
template <typename C, int TWMODE, int PASS> constexpr bool tw_pass_in_regs() {
    return TWMODE == TW_REG || (TWMODE == TW_HYB && !(PASS >= 1 && C::ns(PASS) <= HYB_NS_MAX));
}
// LDS twiddle entries a mode needs (TW_HYB: the leading small-Ns passes only; their ldsoff() are the TW_LDS ones)
template <typename C, int TWMODE, int PASS = 1> constexpr int tw_lds_entries() {
    if constexpr (PASS >= C::P) return 1;
    else if constexpr (tw_pass_in_lds<C, TWMODE, PASS>()) {
        constexpr int here = C::ldsoff(PASS) + C::ldstw(PASS), rest = tw_lds_entries<C, TWMODE, PASS + 1>();
        return here > rest ? here : rest;
    } else return tw_lds_entries<C, TWMODE, PASS + 1>();
}

// Fill the LDS twiddle table cooperatively (T threads of one transform; callers sync afterwards).
template <typename C, typename R, int PASS = 1, int TWMODE = TW_LDS> MDSP_HD void fill_lds_twiddles(cx<R>* twl, int t, const cx<R>* table) {
    if constexpr (PASS < C::P) {
        if constexpr (tw_pass_in_lds<C, TWMODE, PASS>()) {
            constexpr int Ns = C::ns(PASS), Rdx = C::radix(PASS);
            for (int idx = t; idx < (Rdx - 1) * Ns; idx += C::T) {
                const int r = idx / Ns + 1, k = idx - (r - 1) * Ns;
                twl[C::ldsoff(PASS) + idx] = table[(r * k * (C::N / (Ns * Rdx))) & (C::N - 1)];
            }
        }
        fill_lds_twiddles<C, R, PASS + 1, TWMODE>(twl, t, table);
    }
}

// LDS index padding: one extra element every 2^PADSHIFT elements (PADSHIFT >= 31 disables it)
template <int PADSHIFT> MDSP_HD int lds_pad(int i) {
    if constexpr (PADSHIFT >= 31) return i;
    else return i + (i >> PADSHIFT);
}
template <int N, int PADSHIFT> constexpr int lds_elems() { return PADSHIFT >= 31 ? N : N + (N >> PADSHIFT); }
// Padding of a compile-time offset.  lds_pad(base + c) == lds_pad(base) + lds_padc(c) whenever the low PADSHIFT
// bits of base and c cannot carry into each other -- true for every (base, c) pair used below (c is a multiple
// of a power-of-two stride Ns or T, and base's low bits stay below that stride); spelling it this way lets the
// compiler fold c into the DS instruction's immediate offset instead of keeping one address VGPR per element.
template <int PADSHIFT> constexpr int lds_padc(int c) { return PADSHIFT >= 31 ? c : c + (c >> PADSHIFT); }

// Index into the N-entry root table (w[k] = exp(-2 pi i k/N)) of twiddle (butterfly b, element r) of pass p.
template <typename C, int PASS> MDSP_HD int tw_index(int t, int b, int r) {
    constexpr int Ns = C::ns(PASS), Rdx = C::radix(PASS);
    const int j = t + C::T * b;
    const int k = j & (Ns - 1);
    return (r * k * (C::N / (Ns * Rdx))) & (C::N - 1);
}

// Fill the per-thread twiddle registers (loop-invariant for a persistent workgroup).
template <typename C, typename R, int PASS = 1, int TWMODE = TW_REG, int PERMUTE = false>
MDSP_HD void load_twiddles(cx<R> (&tw)[C::NTW > 0 ? C::NTW : 1], int t, const cx<R>* table) {
    if constexpr (PASS < C::P) {
        if constexpr (tw_pass_in_regs<C, TWMODE, PASS>()) {
            constexpr int Rdx = C::radix(PASS), NB = C::E / Rdx;
            const int tp = lane_perm<C, PASS, PERMUTE>(t);
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int r = 1; r < Rdx; ++r) tw[C::twoff(PASS) + b * (Rdx - 1) + (r - 1)] = table[tw_index<C, PASS>(tp, b, r)];
        }
        load_twiddles<C, R, PASS + 1, TWMODE, PERMUTE>(tw, t, table);
    }
}

// One Stockham pass on the thread's registers.  Non-final passes scatter their results to `lds`
// (this transform's region); the final pass leaves X[t + T*e] in x[e].
template <typename C, int DIR, int PASS, int TWMODE, int PADSHIFT, int PERMUTE = false, typename R>
MDSP_HD void pass_compute(cx<R> (&x)[C::E], int t_raw, const cx<R> (&tw)[C::NTW > 0 ? C::NTW : 1], const cx<R>* table, cx<R>* lds) {
    constexpr int Rdx = C::radix(PASS), NB = C::E / Rdx, Ns = C::ns(PASS);
    constexpr bool LAST = PASS == C::P - 1;
    const int t = lane_perm<C, PASS, PERMUTE>(t_raw);   // the butterflies this thread owns in this pass
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        cx<R> v[Rdx];
#pragma unroll
        for (int r = 0; r < Rdx; ++r) v[r] = x[b + r * NB];
        if constexpr (PASS > 0) {
            [[maybe_unused]] int kt = (Ns > C::T) ? t : (t & (Ns - 1));   // TW_LDS: k = kt + (T*b mod Ns), see below
            constexpr bool IN_LDS = tw_pass_in_lds<C, TWMODE, PASS>(), IN_REGS = tw_pass_in_regs<C, TWMODE, PASS>();
            if constexpr (IN_LDS) MDSP_OPAQUE_INT(kt);
            cx<R> wv[Rdx];
#pragma unroll
            for (int r = 1; r < Rdx; ++r) {
                if constexpr (IN_REGS) wv[r] = tw[C::twoff(PASS) + b * (Rdx - 1) + (r - 1)];
                else if constexpr (IN_LDS) {
                    // `table` points at the LDS twiddle table.  k = (t + T*b) mod Ns = kt + KB with kt = t mod Ns (or t when
                    // Ns > T) and KB = (T*b) mod Ns a compile-time constant (no carry: KB is a multiple of T, kt < T).
                    const int KB = (C::T * b) & (Ns - 1);   // constant after unrolling
                    wv[r] = table[kt + (C::ldsoff(PASS) + (r - 1) * Ns + KB)];
                } else wv[r] = table[tw_index<C, PASS>(t, b, r)];
            }
            twmul_all<DIR, Rdx>(v, wv);
        }
        bfly<Rdx, DIR>(v);
        if constexpr (LAST) {
#pragma unroll
            for (int r = 0; r < Rdx; ++r) x[b + r * NB] = v[r];
        } else if constexpr (PERMUTE == 2 && PASS == C::P - 2) {   // wave-private transpose: writer (lane a of its row, output r) -> slot a*17 + r
            const int base = wave_private_base(t_raw) + (t_raw & 15) * 17;
#pragma unroll
            for (int r = 0; r < Rdx; ++r) lds[base + r] = v[r];
        } else {
            const int j = t + C::T * b;
            const int base = lds_pad<PADSHIFT>((j / Ns) * (Ns * Rdx) + (j & (Ns - 1)));
#pragma unroll
            for (int r = 0; r < Rdx; ++r) lds[base + lds_padc<PADSHIFT>(r * Ns)] = v[r];
        }
    }
}

// Pass 0 of a forward E = 16 transform whose input is q[e] w[e] (e < 8), f[e] w[e + 8]: the window rides in the butterfly's first stage
// (bfly16_win); results are scattered exactly as pass_compute<.., PASS = 0> does.  Identity lanes only.
template <typename C, int PADSHIFT, typename R> MDSP_HD void pass0_scatter(const cx<R> (&v)[16], int t, cx<R>* lds) {
    const int base = lds_pad<PADSHIFT>(t * 16);
#pragma unroll
    for (int r = 0; r < 16; ++r) lds[base + lds_padc<PADSHIFT>(r)] = v[r];
}
template <typename C, int PADSHIFT, typename R>
MDSP_HD void pass0_windowed(const cx<R> (&q)[8], const cx<R> (&f)[8], const cx<R> (&wp)[8], int t, cx<R>* lds) {
    static_assert(C::E == 16 && C::radix(0) == 16 && C::P > 1, "E = 16 geometries with a radix-16 first pass");
    cx<R> v[16];
    bfly16_win<-1>(q, f, wp, v);
    pass0_scatter<C, PADSHIFT>(v, t, lds);
}

// After the barrier that follows a non-final pass: fetch the operands of the next pass.
template <typename C, int PADSHIFT, int NEXT = 1, int PERMUTE = false, typename R> MDSP_HD void pass_reload(cx<R> (&x)[C::E], int t_raw, const cx<R>* lds) {
    const int t = lane_perm<C, NEXT, PERMUTE>(t_raw);   // operands of the pass that follows
    if constexpr (PERMUTE == 2 && NEXT == C::P - 1) {   // wave-private transpose: reader lane c of its row takes operand e from slot e*17 + c
        const int base = wave_private_base(t_raw) + (t_raw & 15);
#pragma unroll
        for (int e = 0; e < C::E; ++e) x[e] = lds[base + 17 * e];
    } else if constexpr (PADSHIFT >= 31 || C::T % (1 << (PADSHIFT >= 31 ? 0 : PADSHIFT)) == 0) {
        const int base = lds_pad<PADSHIFT>(t);
#pragma unroll
        for (int e = 0; e < C::E; ++e) x[e] = lds[base + lds_padc<PADSHIFT>(C::T * e)];
    } else {  // tiny transforms (T below the pad period): only reached by the host emulation
#pragma unroll
        for (int e = 0; e < C::E; ++e) x[e] = lds[lds_pad<PADSHIFT>(t + C::T * e)];
    }
}


// ================================================================================================ mixed-radix transforms (runtime N)
// nextfastfft (util.jl:134) returns 7-smooth sizes -- 1000, 1536, 3000 ... -- and periodogram / welch_pgram / stft use them by default
// (nfft = nextfastfft(n), periodograms.jl:393, :560, :872).  The register-resident Stockham above needs E | N with one radix set per
// geometry; for everything else the transform runs entirely through LDS: every pass reads its operands from one buffer and scatters its
// results into the other (Stockham autosort, natural order in, natural order out), radices 16 / 8 / 4 / 2 / 3 / 5 / 7 chosen per pass at run
// time, any thread count.  Slower per point than the register form, but fused all the same: the frame is windowed into LDS straight
// from the signal and the spectrum is consumed from LDS, so HBM sees the signal once and the output once.
// (round 4: the constants ride as SGPR-pair operands of the packed instructions -- cscale_k / caxpy_k / caxmy_k -- instead of VGPR pairs that hipcc
// re-materialised with a v_mov_b64 in front of every butterfly once the kernel was near its register cap; the -i products are folded into the last add / subtract)
template <typename R> MDSP_HD void bfly3f(cx<R>& a0, cx<R>& a1, cx<R>& a2) {   // forward: w = exp(-2 pi i / 3)
    constexpr R s = (R)0.86602540378443864676372317075294L;
    const cx<R> t1 = cadd(a1, a2), d = csub(a1, a2);
    const cx<R> t2 = caxmy_k((R)0.5, t1, a0);   // a0 - t1 / 2
    const cx<R> t3 = cscale_k(s, d);            // (sqrt(3)/2) (a1 - a2), times -i below
    a0 = cadd(a0, t1);
    a1 = add_mi<-1>(t2, t3);
    a2 = sub_mi<-1>(t2, t3);
}
template <typename R> MDSP_HD void bfly5f(cx<R> (&v)[5]) {
    constexpr R c1 = (R)0.30901699437494742410229341718282L, c2 = (R)-0.80901699437494742410229341718282L;
    constexpr R s1 = (R)0.95105651629515357211643933337938L, s2 = (R)0.58778525229247312916870595463907L;
    const cx<R> t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]), t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
    const cx<R> m1 = caxpy_k(c2, t2, caxpy_k(c1, t1, v[0]));
    const cx<R> m2 = caxpy_k(c1, t2, caxpy_k(c2, t1, v[0]));
    const cx<R> n1 = caxpy_k(s2, t4, cscale_k(s1, t3));   // times -i below
    const cx<R> n2 = caxmy_k(s1, t4, cscale_k(s2, t3));
    v[0] = cadd(v[0], cadd(t1, t2));
    v[1] = add_mi<-1>(m1, n1);
    v[4] = sub_mi<-1>(m1, n1);
    v[2] = add_mi<-1>(m2, n2);
    v[3] = sub_mi<-1>(m2, n2);
}
template <typename R> MDSP_HD void bfly7f(cx<R> (&v)[7]) {
    constexpr R c1 = (R)0.62348980185873353052500488400424L, c2 = (R)-0.22252093395631440428890256449679L, c3 = (R)-0.90096886790241912623610231950745L;
    constexpr R s1 = (R)0.78183148246802980870844452667406L, s2 = (R)0.97492791218182360701813168299393L, s3 = (R)0.43388373911755812047576833284836L;
    const cx<R> t1 = cadd(v[1], v[6]), t2 = cadd(v[2], v[5]), t3 = cadd(v[3], v[4]);
    const cx<R> u1 = csub(v[1], v[6]), u2 = csub(v[2], v[5]), u3 = csub(v[3], v[4]);
    // (round 6: chains of packed multiply-adds with the constant as the scalar operand, as the radix-5 butterfly -- the component-wise form compiled to
    // twice as many unpacked instructions)
    const cx<R> m1 = caxpy_k(c3, t3, caxpy_k(c2, t2, caxpy_k(c1, t1, v[0])));
    const cx<R> m2 = caxpy_k(c1, t3, caxpy_k(c3, t2, caxpy_k(c2, t1, v[0])));
    const cx<R> m3 = caxpy_k(c2, t3, caxpy_k(c1, t2, caxpy_k(c3, t1, v[0])));
    const cx<R> n1 = caxpy_k(s3, u3, caxpy_k(s2, u2, cscale_k(s1, u1)));    // times -i below
    const cx<R> n2 = caxmy_k(s1, u3, caxmy_k(s3, u2, cscale_k(s2, u1)));
    const cx<R> n3 = caxpy_k(s2, u3, caxmy_k(s1, u2, cscale_k(s3, u1)));
    v[0] = cadd(v[0], cadd(t1, cadd(t2, t3)));
    v[1] = add_mi<-1>(m1, n1);
    v[6] = sub_mi<-1>(m1, n1);
    v[2] = add_mi<-1>(m2, n2);
    v[5] = sub_mi<-1>(m2, n2);
    v[3] = add_mi<-1>(m3, n3);
    v[4] = sub_mi<-1>(m3, n3);
}
// ---- composite radices (round 4): 6, 10, 12, 15, 20, 24, 25 as two levels of the small butterflies INSIDE the registers of one thread -------------
// A pass of radix R1 R2 replaces two passes (one LDS round trip and one barrier less): 3000 = 5 x 24 x 25 runs three passes instead of the five
// of 3 x 5 x 5 x 5 x 8.  Cooley-Tukey inside the butterfly: R1 transforms of length R2 over the points n1 + R1 n2, the constant twiddles
// exp(-2 pi i n1 k2 / (R1 R2)), R2 transforms of length R1; result k2 + R2 k1, natural order.  cos / sin (2 pi m / N) to 21 digits (mpmath);
// multiples of a quarter turn are exact and cost a swap, not a product.
template <int N> struct CompRoots;
template <> struct CompRoots<6> {
    static constexpr double c[6] = {1.0, 0.5, -0.5, -1.0, -0.5, 0.5};
    static constexpr double s[6] = {0.0, 0.866025403784438646764, 0.866025403784438646764, 0.0, -0.866025403784438646764, -0.866025403784438646764};
};
template <> struct CompRoots<10> {
    static constexpr double c[10] = {1.0, 0.809016994374947424102, 0.309016994374947424102, -0.309016994374947424102, -0.809016994374947424102, -1.0, -0.809016994374947424102, -0.309016994374947424102, 0.309016994374947424102, 0.809016994374947424102};
    static constexpr double s[10] = {0.0, 0.587785252292473129169, 0.951056516295153572116, 0.951056516295153572116, 0.587785252292473129169, 0.0, -0.587785252292473129169, -0.951056516295153572116, -0.951056516295153572116, -0.587785252292473129169};
};
template <> struct CompRoots<12> {
    static constexpr double c[12] = {1.0, 0.866025403784438646764, 0.5, 0.0, -0.5, -0.866025403784438646764, -1.0, -0.866025403784438646764, -0.5, 0.0, 0.5, 0.866025403784438646764};
    static constexpr double s[12] = {0.0, 0.5, 0.866025403784438646764, 1.0, 0.866025403784438646764, 0.5, 0.0, -0.5, -0.866025403784438646764, -1.0, -0.866025403784438646764, -0.5};
};
template <> struct CompRoots<15> {
    static constexpr double c[15] = {1.0, 0.913545457642600895502, 0.669130606358858213826, 0.309016994374947424102, -0.1045284632676534714, -0.5, -0.809016994374947424102, -0.978147600733805637929, -0.978147600733805637929, -0.809016994374947424102, -0.5, -0.1045284632676534714, 0.309016994374947424102, 0.669130606358858213826, 0.913545457642600895502};
    static constexpr double s[15] = {0.0, 0.406736643075800207754, 0.743144825477394235015, 0.951056516295153572116, 0.994521895368273336923, 0.866025403784438646764, 0.587785252292473129169, 0.207911690817759337102, -0.207911690817759337102, -0.587785252292473129169, -0.866025403784438646764, -0.994521895368273336923, -0.951056516295153572116, -0.743144825477394235015, -0.406736643075800207754};
};
template <> struct CompRoots<20> {
    static constexpr double c[20] = {1.0, 0.951056516295153572116, 0.809016994374947424102, 0.587785252292473129169, 0.309016994374947424102, 0.0, -0.309016994374947424102, -0.587785252292473129169, -0.809016994374947424102, -0.951056516295153572116, -1.0, -0.951056516295153572116, -0.809016994374947424102, -0.587785252292473129169, -0.309016994374947424102, 0.0, 0.309016994374947424102, 0.587785252292473129169, 0.809016994374947424102, 0.951056516295153572116};
    static constexpr double s[20] = {0.0, 0.309016994374947424102, 0.587785252292473129169, 0.809016994374947424102, 0.951056516295153572116, 1.0, 0.951056516295153572116, 0.809016994374947424102, 0.587785252292473129169, 0.309016994374947424102, 0.0, -0.309016994374947424102, -0.587785252292473129169, -0.809016994374947424102, -0.951056516295153572116, -1.0, -0.951056516295153572116, -0.809016994374947424102, -0.587785252292473129169, -0.309016994374947424102};
};
template <> struct CompRoots<24> {
    static constexpr double c[24] = {1.0, 0.96592582628906828675, 0.866025403784438646764, 0.707106781186547524401, 0.5, 0.258819045102520762349, 0.0, -0.258819045102520762349, -0.5, -0.707106781186547524401, -0.866025403784438646764, -0.96592582628906828675, -1.0, -0.96592582628906828675, -0.866025403784438646764, -0.707106781186547524401, -0.5, -0.258819045102520762349, 0.0, 0.258819045102520762349, 0.5, 0.707106781186547524401, 0.866025403784438646764, 0.96592582628906828675};
    static constexpr double s[24] = {0.0, 0.258819045102520762349, 0.5, 0.707106781186547524401, 0.866025403784438646764, 0.96592582628906828675, 1.0, 0.96592582628906828675, 0.866025403784438646764, 0.707106781186547524401, 0.5, 0.258819045102520762349, 0.0, -0.258819045102520762349, -0.5, -0.707106781186547524401, -0.866025403784438646764, -0.96592582628906828675, -1.0, -0.96592582628906828675, -0.866025403784438646764, -0.707106781186547524401, -0.5, -0.258819045102520762349};
};
template <> struct CompRoots<25> {
    static constexpr double c[25] = {1.0, 0.96858316112863111949, 0.876306680043863587308, 0.728968627421411523147, 0.535826794978996618271, 0.309016994374947424102, 0.0627905195293133760762, -0.187381314585724630543, -0.425779291565072648863, -0.637423989748689710177, -0.809016994374947424102, -0.929776485888251403661, -0.99211470131447783105, -0.99211470131447783105, -0.929776485888251403661, -0.809016994374947424102, -0.637423989748689710177, -0.425779291565072648863, -0.187381314585724630543, 0.0627905195293133760762, 0.309016994374947424102, 0.535826794978996618271, 0.728968627421411523147, 0.876306680043863587308, 0.96858316112863111949};
    static constexpr double s[25] = {0.0, 0.248689887164854788242, 0.481753674101715274987, 0.684547105928688673732, 0.844327925502015078549, 0.951056516295153572116, 0.998026728428271561952, 0.982287250728688681086, 0.904827052466019527714, 0.770513242775789230803, 0.587785252292473129169, 0.368124552684677959157, 0.125333233564304245373, -0.125333233564304245373, -0.368124552684677959157, -0.587785252292473129169, -0.770513242775789230803, -0.904827052466019527714, -0.982287250728688681086, -0.998026728428271561952, -0.951056516295153572116, -0.844327925502015078549, -0.684547105928688673732, -0.481753674101715274987, -0.248689887164854788242};
};
#include "comp_roots_ext.h"   // 9, 14, 18, 21, 27, 28, 30, 32 (round 6: the run-time-schedule kernel of spectral_gx.h; tools/gen_comp_roots.py)
template <int RDX, typename R> MDSP_HD void gen_bfly(cx<R> (&v)[RDX]);
template <int N, int M, typename R> MDSP_HD cx<R> comp_twiddle(cx<R> a) {   // a exp(-2 pi i M / N), 0 <= M < N
    if constexpr (M == 0) return a;
    else if constexpr (4 * M == N) return mul_mi<-1>(a);      // -i a
    else if constexpr (2 * M == N) return cx<R>{-a.x, -a.y};
    else if constexpr (4 * M == 3 * N) return mul_mi<1>(a);   // +i a
    else return twmul_k<-1>(a, cx<R>{(R)CompRoots<N>::c[M], (R)-CompRoots<N>::s[M]});
}
template <int R1, int R2, int K2, typename R, int... N1>
MDSP_HD void comp_column(const cx<R> (&y)[R1][R2], cx<R> (&v)[R1 * R2], std::integer_sequence<int, N1...>) {
    cx<R> u[R1] = {comp_twiddle<R1 * R2, N1 * K2>(y[N1][K2])...};
    gen_bfly<R1>(u);
    ((v[K2 + R2 * N1] = u[N1]), ...);
}
template <int R1, int R2, typename R, int... K2> MDSP_HD void comp_columns(const cx<R> (&y)[R1][R2], cx<R> (&v)[R1 * R2], std::integer_sequence<int, K2...>) {
    (comp_column<R1, R2, K2>(y, v, std::make_integer_sequence<int, R1>{}), ...);
}
template <int R1, int R2, typename R> MDSP_HD void bfly_comp(cx<R> (&v)[R1 * R2]) {
    cx<R> y[R1][R2];
#pragma unroll
    for (int n1 = 0; n1 < R1; ++n1) {
        cx<R> u[R2];
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) u[n2] = v[n1 + R1 * n2];
        gen_bfly<R2>(u);
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) y[n1][k2] = u[k2];
    }
    comp_columns<R1, R2>(y, v, std::make_integer_sequence<int, R2>{});
}
template <int RDX, typename R> MDSP_HD void gen_bfly(cx<R> (&v)[RDX]) {
    if constexpr (RDX == 3) bfly3f(v[0], v[1], v[2]);
    else if constexpr (RDX == 5) bfly5f(v);
    else if constexpr (RDX == 7) bfly7f(v);
    else if constexpr (RDX == 6) bfly_comp<3, 2>(v);
    else if constexpr (RDX == 10) bfly_comp<5, 2>(v);
    else if constexpr (RDX == 12) bfly_comp<3, 4>(v);
    else if constexpr (RDX == 15) bfly_comp<3, 5>(v);
    else if constexpr (RDX == 20) bfly_comp<5, 4>(v);
    else if constexpr (RDX == 24) bfly_comp<3, 8>(v);
    else if constexpr (RDX == 25) bfly_comp<5, 5>(v);
    else if constexpr (RDX == 9) bfly_comp<3, 3>(v);
    else if constexpr (RDX == 14) bfly_comp<7, 2>(v);
    else if constexpr (RDX == 18) bfly_comp<3, 6>(v);
    else if constexpr (RDX == 21) bfly_comp<3, 7>(v);
    else if constexpr (RDX == 27) bfly_comp<3, 9>(v);
    else if constexpr (RDX == 28) bfly_comp<7, 4>(v);
    else if constexpr (RDX == 30) bfly_comp<5, 6>(v);
    else if constexpr (RDX == 32) bfly_comp<4, 8>(v);
    else bfly<RDX, -1>(v);
}

// LDS accesses of the runtime-indexed buffers: cx<R> is only R-aligned, so a plain access compiles to two 4-byte DS operations (32-bank
// rules, 2-way conflicts on every consecutive-element access: SQ_LDS_BANK_CONFLICT was 50 % of the LDS cycles).  The buffers ARE 2R-aligned:
// say so, and every access is one ds_read_b64 / ds_write_b64 (b128 for Float64).
#if defined(__HIP_DEVICE_COMPILE__)
template <typename R> struct alignas(2 * sizeof(R)) cxa { R x, y; };
template <typename R> __device__ __forceinline__ cx<R> ld2(const cx<R>* p) {
    const cxa<R> v = *reinterpret_cast<const cxa<R>*>(p);
    return {v.x, v.y};
}
template <typename R> __device__ __forceinline__ void st2(cx<R>* p, cx<R> v) { *reinterpret_cast<cxa<R>*>(p) = cxa<R>{v.x, v.y}; }
#else
template <typename R> MDSP_HD cx<R> ld2(const cx<R>* p) { return *p; }
template <typename R> MDSP_HD void st2(cx<R>* p, cx<R> v) { *p = v; }
#endif

// LDS index padding of the mixed-radix buffers: one element per 16 (first-pass scatters have stride = radix elements)
MDSP_HD int gen_pad(int i) { return i + (i >> 4); }
MDSP_HD constexpr int gen_lds_elems(int n) { return n + (n >> 4) + 1; }

// Radix schedule of a 7-smooth N: 16s and 8s first, then 4 / 2, then 7, 5, 3 (radix[] and ns[] have room for MDSP_GEN_MAXP passes).
// Returns the number of passes, 0 if N has another prime factor or needs more passes than that.
#define MDSP_GEN_MAXP 8
MDSP_HD int gen_schedule(int n, int* radix, int* ns) {
    int p = 0, rest = n, acc = 1;
    const int order[7] = {16, 8, 4, 2, 7, 5, 3};
    for (int k = 0; k < 7; ++k) {
        const int r = order[k];
        while (rest % r == 0 && rest > 1) {
            if (r == 16 && rest == 32) break;   // 32 = 8 * 4, not 16 * 2
            if (p == MDSP_GEN_MAXP) return 0;
            radix[p] = r;
            ns[p] = acc;
            acc *= r;
            rest /= r;
            ++p;
        }
    }
    return rest == 1 ? p : 0;
}

// One forward pass of radix RDX over the whole transform by T threads (thread t takes butterflies t, t + T, ...):
//   v[q] = in[j + (N/RDX) q] * w^(q k stride),  k = j mod Ns, stride = N / (Ns RDX);   out[(j div Ns) Ns RDX + k + Ns q'] = DFT_RDX(v)[q']
// roots[k] = exp(-2 pi i k / N), k < N.  divm = ceil(2^24 / Ns): j div Ns == (j * divm) >> 24 (64-bit product) for every j < N / RDX -- exact
// because j * Ns < 2^24 for N <= 8192 (checked exhaustively by the host harness for every size it runs).
template <int RDX, typename R>
MDSP_HD void gen_pass(const cx<R>* in, cx<R>* out, const cx<R>* roots, int N, int Ns, unsigned divm, int t, int T) {
    const int nbf = N / RDX, stride = N / (Ns * RDX);
    // U butterflies per trip (predicated): their LDS reads and twiddle fetches are all in flight before the first butterfly is evaluated
    constexpr int U = RDX >= 16 ? 1 : 2;
    for (int j0 = t; j0 < nbf; j0 += U * T) {
        cx<R> v[U][RDX];
        int hi[U], k[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + u * T;
            const bool on = j < nbf;
            const int jj = on ? j : t;   // a valid index for the idle half: nothing is stored for it
#pragma unroll
            for (int q = 0; q < RDX; ++q) v[u][q] = ld2(in + gen_pad(jj + nbf * q));
            hi[u] = Ns == 1 ? jj : (int)(((unsigned long long)(unsigned)jj * divm) >> 24);
            k[u] = jj - hi[u] * Ns;
        }
        if (Ns > 1) {
            cx<R> w[U][RDX];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = k[u] * stride;   // q * idx < N for q < RDX: no reduction needed
#pragma unroll
                for (int q = 1; q < RDX; ++q) w[u][q] = ld2(roots + q * idx);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int q = 1; q < RDX; ++q) v[u][q] = cmul(v[u][q], w[u][q]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            gen_bfly<RDX>(v[u]);
            if (j0 + u * T < nbf) {
                const int base = hi[u] * Ns * RDX + k[u];
#pragma unroll
                for (int q = 0; q < RDX; ++q) st2(out + gen_pad(base + Ns * q), v[u][q]);
            }
        }
    }
}
template <typename R> MDSP_HD void gen_pass_dispatch(int radix, const cx<R>* in, cx<R>* out, const cx<R>* roots, int N, int Ns, unsigned divm, int t, int T) {
    switch (radix) {
        case 16: gen_pass<16>(in, out, roots, N, Ns, divm, t, T); break;
        case 8: gen_pass<8>(in, out, roots, N, Ns, divm, t, T); break;
        case 4: gen_pass<4>(in, out, roots, N, Ns, divm, t, T); break;
        case 2: gen_pass<2>(in, out, roots, N, Ns, divm, t, T); break;
        case 7: gen_pass<7>(in, out, roots, N, Ns, divm, t, T); break;
        case 5: gen_pass<5>(in, out, roots, N, Ns, divm, t, T); break;
        default: gen_pass<3>(in, out, roots, N, Ns, divm, t, T); break;
    }
}

}  // namespace fft
}  // namespace mdsp
