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

constexpr int ilog2(int n) { return n <= 1 ? 0 : 1 + ilog2(n >> 1); }

// ------------------------------------------------------------------------------------------------ butterflies
template <int DIR, typename R> MDSP_HD void bfly2(cx<R>& a, cx<R>& b) {
    const cx<R> t = csub(a, b);
    a = cadd(a, b);
    b = t;
}

// natural-order in, natural-order out
template <int DIR, typename R> MDSP_HD void bfly4(cx<R>& a0, cx<R>& a1, cx<R>& a2, cx<R>& a3) {
    const cx<R> t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = mul_mi<DIR>(csub(a1, a3));
    a0 = cadd(t0, t2);
    a2 = csub(t0, t2);
    a1 = cadd(t1, t3);
    a3 = csub(t1, t3);
}

template <int DIR, typename R> MDSP_HD void bfly8(cx<R> (&v)[8]) {
    constexpr R h = (R)0.70710678118654752440084436210485L;
    // even / odd radix-4 sub-transforms
    bfly4<DIR>(v[0], v[2], v[4], v[6]);
    bfly4<DIR>(v[1], v[3], v[5], v[7]);
    // odd outputs times W8^k, k = 1,2,3  (W8 = exp(-+ 2 pi i / 8))
    {
        const cx<R> o = v[3];  // k = 1:  (1 -+ i)/sqrt2
        v[3] = DIR < 0 ? cx<R>{(o.x + o.y) * h, (o.y - o.x) * h} : cx<R>{(o.x - o.y) * h, (o.y + o.x) * h};
    }
    v[5] = mul_mi<DIR>(v[5]);  // k = 2
    {
        const cx<R> o = v[7];  // k = 3:  (-1 -+ i)/sqrt2
        v[7] = DIR < 0 ? cx<R>{(o.y - o.x) * h, -(o.x + o.y) * h} : cx<R>{-(o.x + o.y) * h, (o.x - o.y) * h};
    }
    // after the two bfly4 calls: E[k] sits in v[2k], O[k]*W in v[2k+1]
    const cx<R> e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    const cx<R> o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    v[0] = cadd(e0, o0);
    v[1] = cadd(e1, o1);
    v[2] = cadd(e2, o2);
    v[3] = cadd(e3, o3);
    v[4] = csub(e0, o0);
    v[5] = csub(e1, o1);
    v[6] = csub(e2, o2);
    v[7] = csub(e3, o3);
}

template <int DIR, typename R> MDSP_HD void bfly16(cx<R> (&v)[16]) {
    // n = m + 4 s, k = 4 p + q :  X[4p+q] = sum_m W4^{mp} W16^{mq} ( sum_s v[m+4s] W4^{sq} )
    constexpr R c1 = (R)0.92387953251128675612818318939679L;  // cos(pi/8)
    constexpr R s1 = (R)0.38268343236508977172845998403040L;  // sin(pi/8)
    constexpr R h = (R)0.70710678118654752440084436210485L;
#pragma unroll
    for (int m = 0; m < 4; ++m) bfly4<DIR>(v[m], v[m + 4], v[m + 8], v[m + 12]);  // y[m][q] in v[m + 4q]
    // twiddles W16^{mq}, forward root w = exp(-2 pi i/16) = (c1, -s1)
    const cx<R> w1 = {c1, -s1}, w2 = {h, -h}, w3 = {s1, -c1};
    v[1 + 4 * 1] = twmul<DIR>(v[1 + 4 * 1], w1);           // m=1,q=1 : W^1
    v[1 + 4 * 2] = twmul<DIR>(v[1 + 4 * 2], w2);           // m=1,q=2 : W^2
    v[1 + 4 * 3] = twmul<DIR>(v[1 + 4 * 3], w3);           // m=1,q=3 : W^3
    v[2 + 4 * 1] = twmul<DIR>(v[2 + 4 * 1], w2);           // m=2,q=1 : W^2
    v[2 + 4 * 2] = mul_mi<DIR>(v[2 + 4 * 2]);              // m=2,q=2 : W^4 = -+i
    v[2 + 4 * 3] = mul_mi<DIR>(twmul<DIR>(v[2 + 4 * 3], w2));  // m=2,q=3 : W^6 = W^4 W^2
    v[3 + 4 * 1] = twmul<DIR>(v[3 + 4 * 1], w3);           // m=3,q=1 : W^3
    v[3 + 4 * 2] = mul_mi<DIR>(twmul<DIR>(v[3 + 4 * 2], w2));  // m=3,q=2 : W^6
    {
        const cx<R> t = twmul<DIR>(v[3 + 4 * 3], w1);      // m=3,q=3 : W^9 = -W^1
        v[3 + 4 * 3] = {-t.x, -t.y};
    }
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

template <int RDX, int DIR, typename R> MDSP_HD void bfly(cx<R> (&v)[RDX]) {
    if constexpr (RDX == 2) bfly2<DIR>(v[0], v[1]);
    else if constexpr (RDX == 4) bfly4<DIR>(v[0], v[1], v[2], v[3]);
    else if constexpr (RDX == 8) bfly8<DIR>(v);
    else bfly16<DIR>(v);
}

// ------------------------------------------------------------------------------------------------ configuration
// N points, E elements per thread (power of two, 4..16), T = N/E threads per transform.
template <int N_, int E_> struct Cfg {
    static constexpr int N = N_, E = E_, T = N_ / E_;
    static constexpr int LOGN = ilog2(N_), LOGE = ilog2(E_);
    static_assert((1 << LOGN) == N_ && (1 << LOGE) == E_, "power of two sizes only");
    static_assert(E_ >= 2 && E_ <= 16 && N_ >= E_, "unsupported elements-per-thread");
    static constexpr int P = (LOGN + LOGE - 1) / LOGE;  // passes
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

// twiddle sources
enum { TW_GLOBAL = 0, TW_REG = 1, TW_LDS = 2 };

// Fill the LDS twiddle table cooperatively (T threads of one transform; callers sync afterwards).
template <typename C, typename R, int PASS = 1> MDSP_HD void fill_lds_twiddles(cx<R>* twl, int t, const cx<R>* table) {
    if constexpr (PASS < C::P) {
        constexpr int Ns = C::ns(PASS), Rdx = C::radix(PASS);
        for (int idx = t; idx < (Rdx - 1) * Ns; idx += C::T) {
            const int r = idx / Ns + 1, k = idx - (r - 1) * Ns;
            twl[C::ldsoff(PASS) + idx] = table[(r * k * (C::N / (Ns * Rdx))) & (C::N - 1)];
        }
        fill_lds_twiddles<C, R, PASS + 1>(twl, t, table);
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
template <typename C, typename R, int PASS = 1> MDSP_HD void load_twiddles(cx<R> (&tw)[C::NTW > 0 ? C::NTW : 1], int t, const cx<R>* table) {
    if constexpr (PASS < C::P) {
        constexpr int Rdx = C::radix(PASS), NB = C::E / Rdx;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 1; r < Rdx; ++r) tw[C::twoff(PASS) + b * (Rdx - 1) + (r - 1)] = table[tw_index<C, PASS>(t, b, r)];
        load_twiddles<C, R, PASS + 1>(tw, t, table);
    }
}

// One Stockham pass on the thread's registers.  Non-final passes scatter their results to `lds`
// (this transform's region); the final pass leaves X[t + T*e] in x[e].
template <typename C, int DIR, int PASS, int TWMODE, int PADSHIFT, typename R>
MDSP_HD void pass_compute(cx<R> (&x)[C::E], int t, const cx<R> (&tw)[C::NTW > 0 ? C::NTW : 1], const cx<R>* table, cx<R>* lds) {
    constexpr int Rdx = C::radix(PASS), NB = C::E / Rdx, Ns = C::ns(PASS);
    constexpr bool LAST = PASS == C::P - 1;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        cx<R> v[Rdx];
#pragma unroll
        for (int r = 0; r < Rdx; ++r) v[r] = x[b + r * NB];
        if constexpr (PASS > 0) {
            [[maybe_unused]] int kt = (Ns > C::T) ? t : (t & (Ns - 1));   // TW_LDS: k = kt + (T*b mod Ns), see below
            if constexpr (TWMODE == TW_LDS) MDSP_OPAQUE_INT(kt);
#pragma unroll
            for (int r = 1; r < Rdx; ++r) {
                cx<R> w;
                if constexpr (TWMODE == TW_REG) w = tw[C::twoff(PASS) + b * (Rdx - 1) + (r - 1)];
                else if constexpr (TWMODE == TW_LDS) {
                    // `table` points at the LDS twiddle table.  k = (t + T*b) mod Ns = kt + KB with kt = t mod Ns (or t when
                    // Ns > T) and KB = (T*b) mod Ns a compile-time constant (no carry: KB is a multiple of T, kt < T).
                    const int KB = (C::T * b) & (Ns - 1);   // constant after unrolling
                    w = table[kt + (C::ldsoff(PASS) + (r - 1) * Ns + KB)];
                } else w = table[tw_index<C, PASS>(t, b, r)];
                v[r] = twmul<DIR>(v[r], w);
            }
        }
        bfly<Rdx, DIR>(v);
        if constexpr (LAST) {
#pragma unroll
            for (int r = 0; r < Rdx; ++r) x[b + r * NB] = v[r];
        } else {
            const int j = t + C::T * b;
            const int base = lds_pad<PADSHIFT>((j / Ns) * (Ns * Rdx) + (j & (Ns - 1)));
#pragma unroll
            for (int r = 0; r < Rdx; ++r) lds[base + lds_padc<PADSHIFT>(r * Ns)] = v[r];
        }
    }
}

// After the barrier that follows a non-final pass: fetch the operands of the next pass.
template <typename C, int PADSHIFT, typename R> MDSP_HD void pass_reload(cx<R> (&x)[C::E], int t, const cx<R>* lds) {
    if constexpr (PADSHIFT >= 31 || C::T % (1 << (PADSHIFT >= 31 ? 0 : PADSHIFT)) == 0) {
        const int base = lds_pad<PADSHIFT>(t);
#pragma unroll
        for (int e = 0; e < C::E; ++e) x[e] = lds[base + lds_padc<PADSHIFT>(C::T * e)];
    } else {  // tiny transforms (T below the pad period): only reached by the host emulation
#pragma unroll
        for (int e = 0; e < C::E; ++e) x[e] = lds[lds_pad<PADSHIFT>(t + C::T * e)];
    }
}

}  // namespace fft
}  // namespace mdsp
