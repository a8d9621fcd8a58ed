// The row kernel of nfft = R0 x S Welch sums and its launch (included by spectral_ctcols.hip and spectral_ctcols_big.hip inside their anonymous namespaces,
// behind spectral_gen.h).
#pragma once

struct ColsArgs {
    GenArgs g;             // s, out (partials), roots (of S), lds_, K, hop, nch, units_per_ch, per_slot, n, N = S
    const void* winf;      // R window[nfft] (ones without a window, zero tail)
    const void* rootsN;    // nfft forward roots, cx<R>
    int nfft, R0;
};

// Rows above 8192 points (round 6, the single-workgroup schedules of spectral_ctbig.hip as rows: 32768 = 2 x 16384, 25000 = 2 x 12500): one workgroup per CU;
// CtSched flag 8192 keeps the sums in the working precision (flushed to the Float64 partials every MDSP_GEN_LEAN_FLUSH units), 4096 loads the column
// twiddles W_nfft^{i k1} beside the samples (from the nfft-root table, which stays in L2) instead of keeping them in 2 N / T registers.
template <typename R, bool CPLX, typename S>
__global__ __launch_bounds__(S::T, (sizeof(cx<R>) * S::NP > 80 * 1024 && S::T > 512) ? (S::T / 64 + 3) / 4 : 2) void gen_ct_cols_kernel(ColsArgs ca) {   // (waves per SIMD: one workgroup of the big rows, two of the others)
    const GenArgs& a = ca.g;
    using TT = std::conditional_t<CPLX, cx<R>, R>;
    constexpr int N = S::N, T = S::T;
    constexpr int PL = S::P - 1, RL = S::radix(PL), ML = S::M(PL), NBL = S::nbf(PL);
    constexpr int R0r = S::radix(0), M0 = S::M(0), NB0 = S::nbf(0), W0 = M0 * R0r;
    constexpr bool INPL = S::INPLACE;
    constexpr int SZ = (int)sizeof(TT), WZ = (int)sizeof(R);
    __shared__ __attribute__((aligned(16))) cx<R> buf[INPL ? S::NP : 2 * S::NP];
    cx<R>*bufA = buf, *bufB = INPL ? buf : buf + S::NP;
    const int t = threadIdx.x;
    const int64_t ch = blockIdx.y;
    const TT* sc = static_cast<const TT*>(a.s) + ch * a.lds_;
    // group of workgroups (one frame sequence) and row k1; the R0 rows of a group on one XCD (workgroups go to the XCDs round-robin)
    const unsigned b = blockIdx.x, xcd = b & 7u, wq = b >> 3;
    const int k1 = (int)(wq % (unsigned)ca.R0);
    const int64_t gslot = (int64_t)(wq / (unsigned)ca.R0) * 8 + xcd;
    const int64_t u0 = gslot * a.per_slot;
    cx<R> tw[S::NTW];
    ct_load_twiddles<S, 0>(tw, static_cast<const cx<R>*>(a.roots), t);
    __shared__ __attribute__((aligned(16))) cx<R> twlo[S::TW2L ? S::TWS : 1], twhi[S::TW2L ? S::NTWHI : 1];
    const CtTw<R> t2{twlo, twhi};
    if constexpr (S::TW2L) {
        const cx<R>* g = static_cast<const cx<R>*>(a.roots);
        for (int i = t; i < S::TWS; i += T) fft::st2(twlo + i, g[i]);
        for (int i = t; i < S::NTWHI; i += T) fft::st2(twhi + i, g[(unsigned)i * S::TWS]);
    }
    // W_nfft^{i k1} at the points of this thread's first-pass butterflies: loop invariants
    const cx<R>* rootsN = static_cast<const cx<R>*>(ca.rootsN);
    constexpr bool LEANW = S::LEANW, LEANA = S::LEANA;
    cx<R> twc[LEANW ? 1 : W0];
    auto col_twiddle = [&](int tt, int m, int q) __attribute__((always_inline)) {
        const unsigned i = (unsigned)(tt + T * m + NB0 * q);
        return rootsN[(unsigned)(((unsigned long long)(i < (unsigned)N ? i : 0u) * (unsigned)k1) % (unsigned)ca.nfft)];
    };
    if constexpr (!LEANW) {
#pragma unroll
        for (int m = 0; m < M0; ++m)
#pragma unroll
            for (int q = 0; q < R0r; ++q) twc[m * R0r + q] = col_twiddle(t, m, q);
    }
    std::conditional_t<LEANA, R, double> acc[ML * RL];
#pragma unroll
    for (int i = 0; i < ML * RL; ++i) acc[i] = 0;
    constexpr int NTOUCH = S::TOUCH ? (N * SZ * (CPLX ? 2 : 3) / 2 / 128 + T - 1) / T : 1;   // 128-byte lines of a unit's span per thread and workgroup of the group
    [[maybe_unused]] float touched[NTOUCH] = {};
    bool flushed = false;
    auto flush = [&]() __attribute__((always_inline)) {
        double* part = static_cast<double*>(a.out) + (gslot * a.nch + ch) * (int64_t)ca.nfft + k1;
#pragma unroll
        for (int m = 0; m < ML; ++m) {
            const int j = t + T * m;
            if ((m + 1) * T <= NBL || j < NBL) {
#pragma unroll
                for (int q = 0; q < RL; ++q) {
                    double* o = part + (int64_t)(j + NBL * q) * ca.R0;
                    *o = flushed ? *o + (double)acc[m * RL + q] : (double)acc[m * RL + q];
                    if constexpr (LEANA) acc[m * RL + q] = 0;
                }
            }
        }
        flushed = true;
    };
    const __amdgpu_buffer_rsrc_t dw = io::make_rsrc(ca.winf, (long long)ca.nfft * WZ);
    for (int64_t it = 0; it < a.per_slot; ++it) {
        int tl = t;   // (lean rows: a per-unit copy the compiler cannot see through, so that addresses are recomputed instead of living in spilled registers)
        if constexpr (LEANW || LEANA) asm volatile("" : "+v"(tl));
        const int64_t u = u0 + it;
        const bool live = u < a.units_per_ch;
        const int64_t f0 = live ? (CPLX ? u : 2 * u) : 0;
        const bool haveB = !CPLX && live && (f0 + 1) < a.K;
        const TT* fa = sc + f0 * a.hop;
        const __amdgpu_buffer_rsrc_t da = io::make_rsrc(fa, live ? (long long)a.n * SZ : 0);
        const __amdgpu_buffer_rsrc_t db = io::make_rsrc(fa + (CPLX ? 0 : a.hop), haveB ? (long long)a.n * SZ : 0);
        // ---- first pass: the k1-th combination of the R0 segments of the windowed frame (pair), formed while loading
        cx<R> v0[W0];
#pragma unroll
        for (int i = 0; i < W0; ++i) v0[i] = cx<R>{(R)0, (R)0};
        int off = tl * SZ, offw = tl * WZ;
        asm volatile("" : "+v"(off), "+v"(offw));
        unsigned cidx = 0;   // n1 k1 mod R0
        for (int n1 = 0; n1 < ca.R0; ++n1) {
            const cx<R> c = rootsN[(unsigned)cidx * (unsigned)N];   // W_R0^{n1 k1} = W_nfft^{S (n1 k1 mod R0)}: wave-uniform
            cidx += (unsigned)k1;
            if (cidx >= (unsigned)ca.R0) cidx -= (unsigned)ca.R0;
            // (lean rows: the 3 W0 loads of a segment in chunks of 16 points -- all of them in flight next to the 2 W0 sums do not fit the registers)
            constexpr int CHK = (LEANW || LEANA) && W0 > 16 ? 16 : W0;
#pragma unroll
            for (int i0 = 0; i0 < W0; i0 += CHK) {
                TT ra[CHK], rb[CPLX ? 1 : CHK];
                R w[CHK];
#pragma unroll
                for (int i = i0; i < i0 + CHK && i < W0; ++i) {
                    const int e = T * (i / R0r) + NB0 * (i % R0r);
                    ra[i - i0] = io::Ld<TT>::load(da, off + e * SZ);
                    if constexpr (!CPLX) rb[i - i0] = io::Ld<TT>::load(db, off + e * SZ);
                    w[i - i0] = io::Ld<R>::load(dw, offw + e * WZ);
                }
#pragma unroll
                for (int i = i0; i < i0 + CHK && i < W0; ++i) {
                    cx<R> z;
                    if constexpr (CPLX) z = {ra[i - i0].x * w[i - i0], ra[i - i0].y * w[i - i0]};
                    else z = {ra[i - i0] * w[i - i0], rb[i - i0] * w[i - i0]};
                    v0[i] = fft::cadd(v0[i], fft::cmul(z, c));
                }
                if constexpr (CHK < W0) asm volatile("" ::: "memory");
            }
            off += N * SZ;
            offw += N * WZ;
        }
#pragma unroll
        for (int m = 0; m < M0; ++m) {
            const int j = tl + T * m;
            if ((m + 1) * T <= NB0 || j < NB0) {
                cx<R> v[R0r];
                if constexpr (LEANW) {
                    if (k1 == 0) {
#pragma unroll
                        for (int q = 0; q < R0r; ++q) v[q] = v0[m * R0r + q];
                    } else {
                        cx<R> c[R0r];
#pragma unroll
                        for (int q = 0; q < R0r; ++q) c[q] = col_twiddle(tl, m, q);
#pragma unroll
                        for (int q = 0; q < R0r; ++q) v[q] = fft::cmul(v0[m * R0r + q], c[q]);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < R0r; ++q) v[q] = k1 == 0 ? v0[m * R0r + q] : fft::cmul(v0[m * R0r + q], twc[m * R0r + q]);
                }
                fft::gen_bfly<R0r>(v);
                cx<R>* o = bufA + (unsigned)j * (unsigned)(R0r + (S::padded(0) ? 1 : 0));
#pragma unroll
                for (int q = 0; q < R0r; ++q) fft::st2(o + q, v[q]);
            }
        }
        __syncthreads();
        if (S::TOUCH && ca.R0 >= 4) {   // (measured r06s40: +3 .. 6 % from R0 = 4, -3 .. 5 % at R0 = 2, 3) this workgroup's share (every R0-th 128-byte line) of the NEXT unit's span, into the L2 while the passes run (gen_ct_kernel does the same)
            const int64_t un = u + 1;
            const bool nlive = it + 1 < a.per_slot && un < a.units_per_ch;
            const int64_t fn = CPLX ? un : 2 * un;
            const long long nspan = nlive ? ((long long)a.n + ((!CPLX && fn + 1 < a.K) ? a.hop : 0)) * (long long)SZ : 0;
            const __amdgpu_buffer_rsrc_t dn = io::make_rsrc(sc + fn * a.hop, nspan);
#pragma unroll
            for (int i = 0; i < NTOUCH; ++i) touched[i] = io::Ld<float>::load(dn, (k1 + ca.R0 * (tl + T * i)) * 128);
        }
        // ---- the other passes exactly as gen_ct_kernel's register-consumed modes
        const cx<R>* src = bufA;
        if constexpr (INPL) ct_passes_inplace<S, 1, S::P - 1>(bufA, tw, tl, t2);
        else src = ct_passes<S, 1, S::P - 1>(bufA, bufB, tw, tl, t2);
        ct_last_pass_regs<S>(src, tw, tl, t2, [&](int m, int q, int, cx<R> z) { acc[m * RL + q] += (std::conditional_t<LEANA, R, double>)(z.x * z.x + z.y * z.y); });   // (a unit that does not exist transformed zeros)
        __syncthreads();
        if (S::TOUCH && ca.R0 >= 4) {
#pragma unroll
            for (int i = 0; i < NTOUCH; ++i) asm volatile("" ::"v"(touched[i]));
        }
        if constexpr (LEANA) {
            if ((it & (MDSP_GEN_LEAN_FLUSH - 1)) == MDSP_GEN_LEAN_FLUSH - 1) flush();
        }
    }
    flush();
}

template <typename R> __global__ __launch_bounds__(256) void cols_window_kernel(const double* __restrict__ win, R* __restrict__ out, int n, int nfft) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nfft) out[i] = i < n ? (win ? (R)win[i] : (R)1) : (R)0;
}

template <typename R, bool CPLX, typename S> int cols_launch(ColsArgs& ca, int64_t nch, hipStream_t st, int64_t* ngroups, DevBuf* partial) {
    auto kern = gen_ct_cols_kernel<R, CPLX, S>;
    GenArgs& a = ca.g;
    hipFuncAttributes fa{};
    MDSP_HIP(hipFuncGetAttributes(&fa, (const void*)kern));
    const int regs = std::max(8, (fa.numRegs + 7) / 8 * 8), waves = S::T / 64;
    const size_t lds_bytes = sizeof(cx<R>) * ((S::INPLACE ? 1 : 2) * (size_t)S::NP + (S::TW2L ? S::TWS + S::NTWHI : 0));
    int per_cu = std::min<int>({32 / waves, (512 / regs) * 4 / waves, (int)((size_t)160 * 1024 / std::max<size_t>(lds_bytes, 1))});
    if (per_cu < 1) per_cu = 1;
    if (tunables().wg_per_cu > 0) per_cu = tunables().wg_per_cu;
    const int64_t resident = std::max<int64_t>(1, (int64_t)device_cu_count() * per_cu / std::max<int64_t>(1, nch));
    int64_t groups = std::max<int64_t>(1, std::min<int64_t>(a.units_per_ch, resident / ca.R0));
    groups = std::max<int64_t>(8, groups / 8 * 8);   // the XCD mapping walks groups in eights; rounded DOWN: 85 -> 88 groups of three workgroups are 264 on 256 CUs, a second round for 8
    a.per_slot = cdiv(a.units_per_ch, groups);
    *ngroups = groups;
    MDSP_TRY(partial->reserve(sizeof(double) * (size_t)groups * (size_t)nch * (size_t)ca.nfft));
    a.out = partial->p;
    hipLaunchKernelGGL(kern, dim3((unsigned)(groups * ca.R0), (unsigned)nch), dim3(S::T), 0, st, ca);
    MDSP_LAUNCH_CHECK();
    return MDSP_OK;
}

template <typename R> int upload_roots_n(DevBuf& buf, int64_t n) {
    std::vector<cx<R>> w((size_t)n);
    for (int64_t k = 0; k < n; ++k) {
        const zd r = unit_root(k, n, -1);
        w[(size_t)k] = {(R)r.real(), (R)r.imag()};
    }
    MDSP_TRY(buf.reserve(sizeof(cx<R>) * (size_t)n));
    MDSP_HIP(hipMemcpy(buf.p, w.data(), sizeof(cx<R>) * (size_t)n, hipMemcpyHostToDevice));
    return MDSP_OK;
}

