// Welch, nfft = n = 4096, hop = n/2, real Float32 signals: ONE wavefront per transform, one exchange (round 4; fft_w64.h has the algebra).
// Included by spectral.hip (SpecArgs, set_schedule and the partial-sum protocol live there).
//
// Why: welch_half3_kernel shares a transform between four waves -- three radix-16 passes, two exchanges, four workgroup barriers per unit -- and its
// phase profile (profiles/r03c_welch_phases.txt) shows a wave issuing ~1650 clocks of vector work in a 4700-5300 clock unit: the rest is waiting
// (barriers, LDS round trips, the load burst).  A 64 x 64 factorisation needs one exchange and, with the whole transform inside one wave, no
// barrier at all; what it costs is registers: 64 points + 64 power accumulators per lane fill the 256 architectural VGPRs, so one wave per
// SIMD, and everything that is only READ once per unit lives in the accumulation registers (AGPRs: 512-entry unified file, the other half is
// free when a SIMD holds one wave) behind explicit v_accvgpr_read/write:
//   * the 63 per-lane twiddles W4096^{lane T} (126 AGPRs), * the half-frame a unit hands to its successor (32 AGPRs).
// A single wave per SIMD issues one instruction every ~4 clocks whatever its type, so the budget is counted in instructions per unit:
//   ~1190 packed arithmetic + 128 |Z|^2 FMAs + 64 sample reads + 32 window reads + 64 (carry) + 126 (twiddles) AGPR moves + 64 lane swaps +
//   128 exchange DS operations + 16 DMA  ~= 1850 -> ~7400 clocks per TRANSFORM and SIMD, against ~10600 for welch_half3_kernel (2 x 5300 for
//   half a transform per SIMD).
// Samples: the two new half-frames of a unit arrive in LDS by DMA (buffer_load_dwordx4 ... lds, 16 instructions per unit, no VGPRs), issued as soon
// as the previous unit's samples have been consumed -- a whole unit of cover; the wave waits with s_waitcnt vmcnt(0) at the top of a unit.
// LDS per workgroup of four independent waves: window pairs 16 KiB (shared) + 4 x (16 KiB staging + 16.5 KiB exchange) = 146 KiB: one workgroup per CU.
#pragma once
// (spectral.hip includes fft_w64.h at file scope)

namespace w64 {

__device__ __forceinline__ float agpr_read(float a, int token) {   // `token` changes per unit: the read cannot be hoisted out of the unit loop
    float v;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a), "s"(token));
    return v;
}
__device__ __forceinline__ float agpr_write(float v) {
    float a;
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v));
    return a;
}

__device__ __forceinline__ float fma_sq(float x, float c) {   // x x + c
    float d;
    asm("v_fma_f32 %0, %1, %1, %2" : "=v"(d) : "v"(x), "v"(c));
    return d;
}

typedef __attribute__((address_space(3))) float lds_cf;
typedef volatile __attribute__((address_space(3))) float lds_cvf;
constexpr int N = 4096, HALF = N / 2;
constexpr int STAGE_BYTES = 2 * HALF * 4;                         // H1 | H2
constexpr int XBUF_BYTES = fft::XP64_ELEMS * 8;
constexpr int WAVE_BYTES = STAGE_BYTES + XBUF_BYTES;
constexpr int WIN_BYTES = HALF * 8;                               // (w[p], w[p + N/2]) pairs
constexpr int LDS_BYTES = WIN_BYTES + 4 * WAVE_BYTES;
static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");

#ifdef MDSP_DEBUG_KNOBS   // the two HIP forms of this transform (variants 40 / 41) lost to the hand-allocated one: debug builds only (HISTORY.md section 4.3)
// the 64 x 64 transposition of fft_w64.h on the wave's registers: v[slot64(ke)] = M[lane][ke] in, v[T] = M[T][lane] out
__device__ __forceinline__ void transpose64(cx<float> (&v)[64], cx<float>* xb, int lane) {
    cx<float> m[64];   // logical registers
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        const cx<float> lo = v[fft::slot64(r)], hi = v[fft::slot64(r + 32)];
        const auto sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo.x), __float_as_uint(hi.x), false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo.y), __float_as_uint(hi.y), false, false);
        m[r] = {__uint_as_float(sx[0]), __uint_as_float(sy[0])};
        m[r + 32] = {__uint_as_float(sx[1]), __uint_as_float(sy[1])};
    }
    cx<float>* wr = xb + fft::xp64_write_index(lane, 0);
    const cx<float>* rd = xb + fft::xp64_read_index(lane, 0);
#pragma unroll
    for (int round = 0; round < 2; ++round) {
#pragma unroll
        for (int r = 0; r < 32; ++r) fft::st2(wr + r, m[32 * round + r]);
#pragma unroll
        for (int T = 0; T < 32; ++T) v[32 * round + T] = fft::ld2(rd + T * fft::XP64_ROW);
    }
}

template <int DUMMY = 0>
__global__ __launch_bounds__(256, 1) void welch_w64_kernel(SpecArgs a) {
    using R = float;
    extern __shared__ __attribute__((aligned(16))) unsigned char w64_smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    cx<R>* winl = reinterpret_cast<cx<R>*>(w64_smem);
    unsigned char* mine = w64_smem + WIN_BYTES + wave * WAVE_BYTES;
    R* stage = reinterpret_cast<R*>(mine);
    cx<R>* xb = reinterpret_cast<cx<R>*>(mine + STAGE_BYTES);
    const unsigned stage_lds = io::lds_byte_address(w64_smem) + (unsigned)(WIN_BYTES + wave * WAVE_BYTES);
    const cx<R>* table = static_cast<const cx<R>*>(a.table);
    const int64_t ch = blockIdx.y;

    // window pairs, shared by the four waves (the only barrier of the kernel)
    for (int p = tid; p < HALF; p += 256) {
        const double lo = p < a.n ? (a.win ? a.win[p] : 1.0) : 0.0, hi = (p + HALF) < a.n ? (a.win ? a.win[p + HALF] : 1.0) : 0.0;
        winl[p] = {(R)lo, (R)hi};
    }
    __syncthreads();

    // per-lane twiddles W^{lane T}, T = 1..63, in AGPRs
    float twx[64], twy[64];
#pragma unroll
    for (int T = 1; T < 64; ++T) {
        const cx<R> w = table[(lane * T) & (N - 1)];
        twx[T] = agpr_write(w.x);
        twy[T] = agpr_write(w.y);
    }

    float acc[64];
#pragma unroll
    for (int s = 0; s < 64; ++s) acc[s] = 0.f;
    const int64_t slot = (int64_t)blockIdx.x * 4 + wave;
    double* part = static_cast<double*>(a.out) + (slot * a.nch + ch) * N;
    const __amdgpu_buffer_rsrc_t prs = io::make_rsrc(part, (int64_t)N * 8);
    bool first = true;
    auto flush = [&]() __attribute__((always_inline)) {
        int off = lane * 8;
        asm volatile("" : "+v"(off));
#pragma unroll
        for (int kt = 0; kt < 64; ++kt) {   // bin lane + 64 kt sits in slot64(kt)
            double s = (double)acc[fft::slot64(kt)];
            if (!first) s += io::Ld<double>::load(prs, off + 64 * kt * 8);
            io::Ld<double>::store(s, prs, off + 64 * kt * 8);
            acc[fft::slot64(kt)] = 0.f;
        }
        first = false;
    };

    const R* sc = static_cast<const R*>(a.s) + ch * a.lds_;
    const int64_t u0 = slot * a.run_len;
    const int64_t uend = std::min<int64_t>(u0 + a.run_len, a.units_per_ch);   // one past this wave's last unit
    const int64_t klast = uend > u0 ? 2 * (uend - 1) + (((2 * (uend - 1) + 1) < a.K) ? 2 : 1) : -1;   // last half-frame any of its frames touches
    // half-frame k of the channel -> staging region `reg` (0: H1, 1: H2); always eight instructions (an unwanted half-frame moves nothing)
    auto dma_half = [&](int64_t k, int reg) __attribute__((always_inline)) {
        const bool want = k >= 2 * u0 && k <= klast;
        const io::dma_i4 r = io::dma_rsrc(sc + k * HALF, want ? (long long)HALF * 4 : 0);
#pragma unroll
        for (int g = 0; g < HALF / 256; ++g) io::dma256(r, stage_lds + (unsigned)reg * (unsigned)(HALF * 4) + (unsigned)g * 1024u, g * 1024 + lane * 16);
    };

    if (uend > u0) {
        // the first unit's H0 passes through the H2 region into the carry registers
        float carry[32];
        dma_half(2 * u0, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int e = 0; e < 32; ++e) carry[e] = agpr_write(stage[HALF + lane + 64 * e]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        dma_half(2 * u0 + 1, 0);
        dma_half(2 * u0 + 2, 1);
        int since = 0;
        auto unit = [&](auto frame_b, int64_t u, int token) __attribute__((always_inline)) {
            constexpr bool FRAME_B = decltype(frame_b)::value;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this unit's half-frames have landed
            cx<R> v[64];
            // first layer + first radix-8 layer, one group of four sample pairs at a time; group n1 + 1 is fetched from LDS while group n1 is evaluated
            // (all 32 groups' operands at once would need 160 registers next to the 64 accumulators and the growing result)
            struct Grp { cx<R> xp[4], hh[2], wp[4]; };
            auto fetch = [&](Grp& g, int n1) __attribute__((always_inline)) {
                const lds_cf* st = (const lds_cf*)stage_lds + lane;
                const lds_cvf* sv = (const lds_cvf*)stage_lds + HALF + lane;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = n1 + 8 * j;
                    const R h2 = sv[64 * e];                                   // volatile: never merged into a two-address read -- it lands in the pair's upper half
                    g.xp[j] = {agpr_read(carry[e], token), h2};
                    carry[e] = agpr_write(h2);
                    g.wp[j] = fft::ld2(winl + lane + 64 * e);
                }
                g.hh[0] = {st[64 * n1], st[64 * (n1 + 8)]};
                g.hh[1] = {st[64 * (n1 + 16)], st[64 * (n1 + 24)]};
            };
            Grp ga, gb;
            fetch(ga, 0);
#pragma unroll
            for (int n1 = 0; n1 < 8; n1 += 2) {
                fetch(gb, n1 + 1);
                fft::passA_welch_group<FRAME_B>(n1, ga.xp, ga.hh, ga.wp, v);
                __builtin_amdgcn_sched_barrier(0);
                if (n1 + 2 < 8) fetch(ga, n1 + 2);
                fft::passA_welch_group<FRAME_B>(n1 + 1, gb.xp, gb.hh, gb.wp, v);
                __builtin_amdgcn_sched_barrier(0);
            }
            fft::bfly64_tail<-1>(v);
            // the staging buffer is free: the next unit's half-frames (a whole unit of cover)
            dma_half(2 * u + 3, 0);
            dma_half(2 * u + 4, 1);
            transpose64(v, xb, lane);
#pragma unroll
            for (int T = 1; T < 64; ++T) v[T] = fft::cmul(v[T], cx<R>{agpr_read(twx[T], token), agpr_read(twy[T], token)});
            fft::bfly64<-1>(v);
#pragma unroll
            for (int s = 0; s < 64; ++s) {   // asm: hipcc's SLP pass would pair bins into v_pk_fma and pay three v_mov per pair to build the operands
                acc[s] = fma_sq(v[s].y, fma_sq(v[s].x, acc[s]));
            }
        };
        for (int64_t u = u0; u < uend; ++u) {
            const int token = (int)(u - u0);
            if ((2 * u + 1) < a.K) unit(std::true_type{}, u, token);
            else unit(std::false_type{}, u, token);
            if (++since == 128 || u + 1 == uend) {   // Float32 sums of at most 128 units, folded into the Float64 partial row (as welch_half3_kernel)
                flush();
                since = 0;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing of this wave may still be writing LDS when the workgroup exits
    } else {   // a slot without units still owns a partial row: zeros
#pragma unroll 1
        for (int kt = 0; kt < 64; ++kt) io::Ld<double>::store(0.0, prs, lane * 8 + 64 * kt * 8);
    }
}

// ---- the same transform with TWO waves per SIMD (variant 41) -----------------------------------------------------------------------------
// What the first GPU session said about welch_w64_kernel (gpurun_out/s1, 2^30 samples): parity green at the first attempt, 1.65 ms against
// 1.36 ms for welch_half3_kernel.  Its unit loop is 2177 instructions and a unit takes ~14 200 clocks: ~6.5 clocks per instruction, which is what
// ONE wave per SIMD issues (tools/ubench/valu_rate.hip, profiles/r02t_valu_rate.txt: v_pk_* at 5.69 clocks per instruction with one wave per SIMD,
// 4.75 with two, 4.40 with four).  The registers that forced one wave per SIMD were the AGPR-resident twiddles, carry and the DMA staging; here:
//   * twiddles W^{lane T}, T = t1 + 8 t2, as a product of two per-lane tables of seven roots each (28 VGPRs): W^{8 lane t2} on the operands of
//     pass B's first radix-8 layer, W^{lane t1} on its results (112 complex products instead of 63: +98 packed instructions per unit),
//   * no staging and no carry: a unit's three half-frames are loaded straight into the operand registers of the first layer (96 loads of 256
//     bytes per wave; the middle half-frame comes from L2 the second time) once the previous unit's spectrum has been accumulated -- the partner
//     wave of the SIMD computes while they are in flight,
// so a wave needs <= 256 registers, eight waves (one 512-thread workgroup, no barriers after the window table) share a CU, and LDS holds the
// window pairs + 8 x 16.5 KiB of exchange buffers = 148 KiB.
constexpr int LDS_BYTES_B = WIN_BYTES + 8 * XBUF_BYTES;
static_assert(LDS_BYTES_B <= 160 * 1024, "one workgroup per CU");

template <int DUMMY = 0>
__global__ __launch_bounds__(512, 2) void welch_w64b_kernel(SpecArgs a) {
    using R = float;
    extern __shared__ __attribute__((aligned(16))) unsigned char w64_smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    cx<R>* winl = reinterpret_cast<cx<R>*>(w64_smem);
    cx<R>* xb = reinterpret_cast<cx<R>*>(w64_smem + WIN_BYTES + wave * XBUF_BYTES);
    const cx<R>* table = static_cast<const cx<R>*>(a.table);
    const int64_t ch = blockIdx.y;
    for (int p = tid; p < HALF; p += 512) {
        const double lo = p < a.n ? (a.win ? a.win[p] : 1.0) : 0.0, hi = (p + HALF) < a.n ? (a.win ? a.win[p + HALF] : 1.0) : 0.0;
        winl[p] = {(R)lo, (R)hi};
    }
    __syncthreads();
    cx<R> twa[8], twb[8];   // [0] unused
#pragma unroll
    for (int j = 1; j < 8; ++j) {
        twa[j] = table[(8 * lane * j) & (N - 1)];
        twb[j] = table[(lane * j) & (N - 1)];
    }
    float acc[64];
#pragma unroll
    for (int s = 0; s < 64; ++s) acc[s] = 0.f;
    const int64_t slot = (int64_t)blockIdx.x * 8 + wave;
    double* part = static_cast<double*>(a.out) + (slot * a.nch + ch) * N;
    const __amdgpu_buffer_rsrc_t prs = io::make_rsrc(part, (int64_t)N * 8);
    bool first = true;
    auto flush = [&]() __attribute__((always_inline)) {
        int off = lane * 8;
        asm volatile("" : "+v"(off));
#pragma unroll
        for (int kt = 0; kt < 64; ++kt) {
            double s = (double)acc[fft::slot64(kt)];
            if (!first) s += io::Ld<double>::load(prs, off + 64 * kt * 8);
            io::Ld<double>::store(s, prs, off + 64 * kt * 8);
            acc[fft::slot64(kt)] = 0.f;
        }
        first = false;
    };
    const R* sc = static_cast<const R*>(a.s) + ch * a.lds_;
    const int64_t u0 = slot * a.run_len;
    const int64_t uend = std::min<int64_t>(u0 + a.run_len, a.units_per_ch);
    if (uend > u0) {
        struct Grp { cx<R> xp[4], hh[2]; };
        Grp g[8];   // the operands of the first layer: g[n1].xp[j] = (H0, H2)[lane + 64 (n1 + 8 j)], g[n1].hh = H1 at j = (0, 1), (2, 3)
        auto load_unit = [&](int64_t u) __attribute__((always_inline)) {
            const bool live = u < uend, haveB = live && (2 * u + 1) < a.K;
            const int64_t pos = u * N;
            const __amdgpu_buffer_rsrc_t r0 = io::make_rsrc(sc + pos, live ? (int64_t)HALF * 4 : 0);
            const __amdgpu_buffer_rsrc_t r1 = io::make_rsrc(sc + pos + HALF, live ? (int64_t)HALF * 4 : 0);
            const __amdgpu_buffer_rsrc_t r2 = io::make_rsrc(sc + pos + 2 * HALF, haveB ? (int64_t)HALF * 4 : 0);
            int off = lane * 4;
            asm volatile("" : "+v"(off));
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = n1 + 8 * j;
                    g[n1].xp[j] = {io::Ld<R>::load(r0, off + 256 * e), io::Ld<R>::load(r2, off + 256 * e)};
                }
                g[n1].hh[0] = {io::Ld<R>::load(r1, off + 256 * n1), io::Ld<R>::load(r1, off + 256 * (n1 + 8))};
                g[n1].hh[1] = {io::Ld<R>::load(r1, off + 256 * (n1 + 16)), io::Ld<R>::load(r1, off + 256 * (n1 + 24))};
            }
        };
        load_unit(u0);
        int since = 0;
        auto unit = [&](auto frame_b, int64_t u) __attribute__((always_inline)) {
            constexpr bool FRAME_B = decltype(frame_b)::value;
            cx<R> v[64];
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) {
                cx<R> wp[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) wp[j] = fft::ld2(winl + lane + 64 * (n1 + 8 * j));
                fft::passA_welch_group<FRAME_B>(n1, g[n1].xp, g[n1].hh, wp, v);
            }
            fft::bfly64_tail<-1>(v);
            transpose64(v, xb, lane);
            // pass B with the two-level twiddles: W^{8 lane t2} before the first radix-8 layer, W^{lane t1} behind it
#pragma unroll
            for (int t1 = 0; t1 < 8; ++t1) {
                cx<R> q[8];
                q[0] = v[t1];
#pragma unroll
                for (int t2 = 1; t2 < 8; ++t2) q[t2] = fft::cmul(v[t1 + 8 * t2], twa[t2]);
                fft::bfly8<-1>(q);
#pragma unroll
                for (int k1 = 0; k1 < 8; ++k1) v[t1 + 8 * k1] = t1 == 0 ? q[k1] : fft::cmul(q[k1], twb[t1]);
            }
            fft::bfly64_tail<-1>(v);
#pragma unroll
            for (int s = 0; s < 64; ++s) acc[s] = fma_sq(v[s].y, fma_sq(v[s].x, acc[s]));
            load_unit(u + 1);   // into the registers the spectrum has just left
        };
        for (int64_t u = u0; u < uend; ++u) {
            if ((2 * u + 1) < a.K) unit(std::true_type{}, u);
            else unit(std::false_type{}, u);
            if (++since == 128 || u + 1 == uend) {
                flush();
                since = 0;
            }
        }
    } else {
#pragma unroll 1
        for (int kt = 0; kt < 64; ++kt) io::Ld<double>::store(0.0, prs, lane * 8 + 64 * kt * 8);
    }
}

inline int welch_run_w64b(mdsp_welch_plan_s* pl, SpecArgs& a, hipStream_t st, int* nslices) {
    auto kern = welch_w64b_kernel<0>;
    static std::atomic<unsigned long long> lds_opt_in{0};
    int dev = 0;
    MDSP_HIP(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(lds_opt_in.load(std::memory_order_acquire) & bit)) {
        MDSP_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        lds_opt_in.fetch_or(bit, std::memory_order_release);
    }
    const int64_t per_ch = std::max<int64_t>(1, (int64_t)device_cu_count() / std::max<int64_t>(1, a.nch));
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(a.units_per_ch, 8), per_ch));
    const int64_t nslots = (int64_t)grid * 8;
    MDSP_TRY(pl->partial.reserve(sizeof(double) * (size_t)nslots * (size_t)a.nch * N));
    a.out = pl->partial.p;
    set_schedule(a, a.units_per_ch, nslots);
    a.run_len = cdiv(a.units_per_ch, nslots);
    hipLaunchKernelGGL(kern, dim3(grid, (unsigned)a.nch), dim3(512), LDS_BYTES_B, st, a);
    MDSP_LAUNCH_CHECK();
    *nslices = (int)nslots;
    return MDSP_OK;
}

inline int welch_run_w64(mdsp_welch_plan_s* pl, SpecArgs& a, hipStream_t st, int* nslices) {
    auto kern = welch_w64_kernel<0>;
    static std::atomic<unsigned long long> lds_opt_in{0};
    int dev = 0;
    MDSP_HIP(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(lds_opt_in.load(std::memory_order_acquire) & bit)) {
        MDSP_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        lds_opt_in.fetch_or(bit, std::memory_order_release);
    }
    const int64_t per_ch = std::max<int64_t>(1, (int64_t)device_cu_count() / std::max<int64_t>(1, a.nch));
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(a.units_per_ch, 4), per_ch));
    const int64_t nslots = (int64_t)grid * 4;
    MDSP_TRY(pl->partial.reserve(sizeof(double) * (size_t)nslots * (size_t)a.nch * N));
    a.out = pl->partial.p;
    set_schedule(a, a.units_per_ch, nslots);
    a.run_len = cdiv(a.units_per_ch, nslots);   // one contiguous run per wave
    hipLaunchKernelGGL(kern, dim3(grid, (unsigned)a.nch), dim3(256), LDS_BYTES, st, a);
    MDSP_LAUNCH_CHECK();
    *nslices = (int)nslots;
    return MDSP_OK;
}

#endif   // MDSP_DEBUG_KNOBS

// ---- the hand-allocated forms: csrc/welch_w64c_asm.s (variant 43, the default: tools/gen_welch_asm_c.py) and csrc/welch_w64_asm.s (variant 42:
// tools/gen_welch_asm.py) -----------------------------------------------------------------------------------------------------------------------------
// Same transform as welch_w64b_kernel (two waves per SIMD, two-level twiddles, direct loads) with every register assigned by the generator: all 256
// VGPRs, no spill, 1770-1790 instructions per unit.  Variant 43 keeps the half-frame two consecutive units share in registers (twiddles in LDS, compact
// operand banks, a two-unit loop body: 64 loads per unit, HBM traffic 1.01 x algorithmic); variant 42 loads it again (96 loads, 1.32 x).  The code
// objects are assembled by build.py and embedded as byte arrays (*_co.h in the object directory), loaded once per device with hipModuleLoadData.
// Host side of their contract:
//   * a prepared per-plan block: window pairs (w[p], w[p + N/2]) as Float32 (16 KiB) + the per-lane twiddles W^{8 lane j}, W^{lane j}, j = 1..7;
//   * Float32 partial rows part[((slot nch + ch) nflush + f) N + bin]: a wave with U units writes exactly ceil(U / 128) of its nflush rows, the row
//     reduction skips the others (no zeroing);
//   * the units they run all have both frames; the odd last frame of a channel goes through welch_half3_kernel and is added to the same sums.
#include "welch_w64c_asm_co.h"   // static const unsigned char welch_w64c_asm_co[]; generated by build.py from welch_w64c_asm.s (tools/gen_welch_asm_c.py)
// (variant 42, csrc/welch_w64_asm.s -- the first hand-allocated form, which re-read the shared half-frame: 1.32 x the algorithmic bytes -- was removed in round 5;
// its generator tools/gen_welch_asm.py stays: gen_welch_asm_c.py builds on its scheduler, allocator and emulator)

struct W64AsmArgs {
    const float* s;
    float* part;
    const float* winpairs;
    const float* tw;
    int64_t lds_, units, run_len, nch;
    int nflush, pad;
};
static_assert(sizeof(W64AsmArgs) == 72, "kernarg layout of mdsp_welch_w64_asm");

__global__ __launch_bounds__(256) void w64asm_prepare_kernel(const double* __restrict__ win, int n, const cx<float>* __restrict__ table, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < HALF) {
        out[2 * i] = (float)(i < n ? (win ? win[i] : 1.0) : 0.0);
        out[2 * i + 1] = (float)((i + HALF) < n ? (win ? win[i + HALF] : 1.0) : 0.0);
    }
    if (i < 64) {   // lane i: 28 floats behind the window pairs
        float* tw = out + 2 * HALF + 28 * i;
        for (int j = 1; j < 8; ++j) {
            const cx<float> a = table[(8 * i * j) & (N - 1)], b = table[(i * j) & (N - 1)];
            tw[2 * (j - 1)] = a.x;
            tw[2 * (j - 1) + 1] = a.y;
            tw[14 + 2 * (j - 1)] = b.x;
            tw[14 + 2 * (j - 1) + 1] = b.y;
        }
    }
}

// reduced[ch][k] (+)= sum over this channel's rows of part (Float32 rows, Float64 sum, fixed order), in two steps: 2048 slots x 16 KiB of rows
// are 32 MiB per launch, and one thread per bin walking all of them (the first form: 128 workgroups, 128-byte reads) took 116 us of a 1.37 ms stage.
// Step 1: a workgroup owns 1024 bins x one group of rows (16-byte reads, 4 KiB per wave and row), tmp[ch][g][k]; step 2 adds the groups in order.
constexpr int W64_RED_GROUPS = 128;
// (rows a wave never wrote -- it flushes ceil(units / 128) rows, mdsp_welch_w64_asm .Lflush -- are skipped: the row buffer needs no zeroing)
__global__ __launch_bounds__(256) void w64asm_reduce1_kernel(const float* __restrict__ part, double* __restrict__ tmp, int nslots, int nflush, int64_t nch, int rows_per_group,
                                                             int64_t units, int64_t run_len) {
    const int k4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int g = blockIdx.y;
    const int64_t ch = blockIdx.z;
    const int rows = nslots * nflush;
    const int r0 = g * rows_per_group, r1 = min(rows, r0 + rows_per_group);
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll 4
    for (int r = r0; r < r1; ++r) {
        const int sl = r / nflush, f = r - sl * nflush;
        const int64_t mine = min(run_len, units - (int64_t)sl * run_len);   // units of slot sl (<= 0: none)
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((int64_t)f * 128 < mine) v = *reinterpret_cast<const float4*>(part + (((int64_t)sl * nch + ch) * nflush + f) * N + k4);
        a0 += (double)v.x;
        a1 += (double)v.y;
        a2 += (double)v.z;
        a3 += (double)v.w;
    }
    double* o = tmp + ((int64_t)ch * W64_RED_GROUPS + g) * N + k4;
    o[0] = a0;
    o[1] = a1;
    o[2] = a2;
    o[3] = a3;
}
__global__ __launch_bounds__(256) void w64asm_reduce2_kernel(const double* __restrict__ tmp, double* __restrict__ reduced, int ngroups, int accumulate) {
    __shared__ double sm[8][32];
    const int b = threadIdx.x & 31, gl = threadIdx.x >> 5;   // 32 bins x 8 lanes of groups per workgroup; the eight lanes are added in order
    const int k = blockIdx.x * 32 + b;
    const int64_t ch = blockIdx.y;
    double t = 0;
    for (int g = gl; g < ngroups; g += 8) t += tmp[((int64_t)ch * W64_RED_GROUPS + g) * N + k];
    sm[gl][b] = t;
    __syncthreads();
    if (gl == 0) {
        t = sm[0][b];
#pragma unroll
        for (int i = 1; i < 8; ++i) t += sm[i][b];
        reduced[ch * N + k] = accumulate ? reduced[ch * N + k] + t : t;
    }
}

struct W64AsmModule {
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
};
// carry: mdsp_welch_w64c_asm (the shared half-frame of consecutive units stays in registers: 64 loads per unit, 1.0 x the algorithmic bytes) instead
// of mdsp_welch_w64_asm (96 loads, 1.32 x)
inline int w64asm_function(hipFunction_t* fn, bool carry) {
    static std::mutex mu;
    static W64AsmModule mods[2][64];
    int dev = 0;
    MDSP_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    W64AsmModule& m = mods[carry ? 1 : 0][dev & 63];
    if (!m.fn) {
        (void)carry;
        MDSP_HIP(hipModuleLoadData(&m.mod, welch_w64c_asm_co));
        MDSP_HIP(hipModuleGetFunction(&m.fn, m.mod, "mdsp_welch_w64c_asm"));
    }
    *fn = m.fn;
    return MDSP_OK;
}

// returns MDSP_OK with *handled = true when the sums have been added to pl->reduced (the caller skips its own slice reduction)
template <int DUMMY = 0> int welch_run_w64asm(mdsp_welch_plan_s* pl, SpecArgs& a, hipStream_t st, bool* handled, bool carry = true) {
    *handled = false;
    hipFunction_t fn = nullptr;
    MDSP_TRY(w64asm_function(&fn, carry));
    // the prepared block (per plan: the window and the root table never change)
    constexpr size_t PREP_FLOATS = 2 * HALF + 28 * 64;
    if (pl->w64prep.bytes == 0) {
        MDSP_TRY(pl->w64prep.reserve(PREP_FLOATS * sizeof(float)));
        hipLaunchKernelGGL(w64asm_prepare_kernel, dim3(HALF / 256), dim3(256), 0, st, a.win, a.n, static_cast<const cx<float>*>(a.table), pl->w64prep.as<float>());
        MDSP_LAUNCH_CHECK();
    }
    const int64_t units = a.K / 2;                                  // units with both frames
    MDSP_TRY(pl->reduced.reserve(sizeof(double) * (size_t)a.nch * N));
    bool fresh = pl->acc_fresh;
    if (units > 0) {
        const int64_t per_ch = std::max<int64_t>(1, (int64_t)device_cu_count() / std::max<int64_t>(1, a.nch));
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(units, 8), per_ch));
        const int64_t nslots = (int64_t)grid * 8;
        const int64_t run_len = cdiv(units, nslots);
        const int nflush = (int)cdiv(run_len, 128);
        const size_t part_bytes = sizeof(float) * (size_t)nslots * (size_t)a.nch * (size_t)nflush * N;
        const size_t tmp_bytes = sizeof(double) * (size_t)a.nch * W64_RED_GROUPS * N;   // the reduction's intermediate sums, behind the rows
        MDSP_TRY(pl->partial.reserve(part_bytes + tmp_bytes));
        W64AsmArgs ka;
        ka.s = static_cast<const float*>(a.s);
        ka.part = pl->partial.as<float>();
        ka.winpairs = pl->w64prep.as<float>();
        ka.tw = pl->w64prep.as<float>() + 2 * HALF;
        ka.lds_ = a.lds_;
        ka.units = units;
        ka.run_len = run_len;
        ka.nch = a.nch;
        ka.nflush = nflush;
        ka.pad = 0;
        size_t ksz = sizeof(ka);
        void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &ka, HIP_LAUNCH_PARAM_BUFFER_SIZE, &ksz, HIP_LAUNCH_PARAM_END};
        MDSP_HIP(hipModuleLaunchKernel(fn, (unsigned)grid, (unsigned)a.nch, 1, 512, 1, 1, 0, st, nullptr, cfg));
        {
            const int rows = (int)nslots * nflush;
            const int rpg = (int)cdiv(rows, W64_RED_GROUPS), ngroups = (int)cdiv(rows, rpg);
            double* tmp = reinterpret_cast<double*>(static_cast<char*>(pl->partial.p) + part_bytes);
            hipLaunchKernelGGL(w64asm_reduce1_kernel, dim3(N / 1024, (unsigned)ngroups, (unsigned)a.nch), dim3(256), 0, st, pl->partial.as<float>(), tmp, (int)nslots, nflush,
                               a.nch, rpg, units, run_len);
            MDSP_LAUNCH_CHECK();
            hipLaunchKernelGGL(w64asm_reduce2_kernel, dim3(N / 32, (unsigned)a.nch), dim3(256), 0, st, tmp, pl->reduced.as<double>(), ngroups, fresh ? 0 : 1);
            MDSP_LAUNCH_CHECK();
        }
        fresh = false;
    }
    if (a.K & 1) {   // the channel's odd last frame: one unit of welch_half3_kernel on the last n samples, added to the same sums
        SpecArgs t = a;
        t.s = static_cast<const float*>(a.s) + (a.K - 1) * a.hop;
        t.len = a.n;
        t.K = 1;
        t.units_per_ch = 1;
        int ns2 = 0;
        MDSP_TRY((welch_run_half3<N, 5, 1>(pl, t, st, &ns2)));
        hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)cdiv(N, 32), (unsigned)a.nch), dim3(256), 0, st, pl->partial.as<double>(), pl->reduced.as<double>(), ns2,
                           a.nch, N, fresh ? 0 : 1);
        MDSP_LAUNCH_CHECK();
        fresh = false;
    }
    *handled = true;
    return MDSP_OK;
}

}  // namespace w64
