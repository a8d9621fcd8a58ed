// Register-tap polyphase kernel, every signal type (fir_reg.h).  The reference loop it replaces: Filters/stream_filt.jl:496-509 (FIRRational filt!) with
// unsafe_dot util.jl:225-283 -- one dot product of tapsPerPhi taps per output, oldest sample first.
//
// Outputs m = q L + s (round q, residue s) of a rational resampler all use phase phi_s = (phi0 - 1 + s M) mod L and the window that starts at
// q M + c_s, c_s = d0 - 1 + (phi0 - 1 + s M) div L -- both independent of q (the closed form of fir.hip).  A thread owns P consecutive residues for the
// whole launch: their taps live in registers (pre-shifted by delta_k = c_{s+k} - c_s, which is < P when M <= L), and per round it reads ONE window of
// TPC + P - 1 staged samples from LDS for its P outputs -- one LDS read per multiply-add (P = 1) or half of one (P = 2) where the generic kernel issues an
// L2 load and an LDS read, and the multiply-add chain keeps the reference's order (zero taps in front of / behind a phase add exact zeros: results are bit
// for bit the generic kernel's for finite samples; a non-finite sample inside the P - 1 extra window positions widens the reference's hole, which is
// what mdsp_fir_set_exact is for).
#include "fir_reg.h"

#include "devio.h"
#include "fft_lds.h"

using namespace mdsp;
using mdsp::fft::cx;

namespace {

template <typename R> __device__ __forceinline__ R to_acc(float v, R*) { return (R)v; }
template <typename R> __device__ __forceinline__ R to_acc(double v, R*) { return (R)v; }
template <typename R> __device__ __forceinline__ cx<R> to_acc(cx<float> v, cx<R>*) { return {(R)v.x, (R)v.y}; }
template <typename R> __device__ __forceinline__ cx<R> to_acc(cx<double> v, cx<R>*) { return {(R)v.x, (R)v.y}; }
template <typename R> __device__ __forceinline__ void fma_acc(R& acc, R h, R x) { acc = fma(x, h, acc); }
template <typename R> __device__ __forceinline__ void fma_acc(cx<R>& acc, R h, cx<R> x) {
    acc.x = fma(x.x, h, acc.x);
    acc.y = fma(x.y, h, acc.y);
}
template <typename A> __device__ __forceinline__ A zero_acc() { return A{}; }

// XS: storage type of x; A: accumulate / output type (R or cx<R>); R: tap type
template <typename XS, typename A, typename R, int TPC, int P, int LB>
__global__ __launch_bounds__(LB) void polyphase_reg_kernel(FirRegArgs a) {
    constexpr int W = TPC + P - 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char reg_smem[];
    A* zs = reinterpret_cast<A*>(reg_smem);
    const int64_t ch = blockIdx.y;
    const int g = threadIdx.x % a.NP, r = threadIdx.x / a.NP;
    const XS* xc = static_cast<const XS*>(a.x) + ch * a.ldx;
    const XS* hc = static_cast<const XS*>(a.hist) + ch * (int64_t)a.hl;
    A* yc = static_cast<A*>(a.y) + ch * a.ldy;
    const R* pfb = static_cast<const R*>(a.pfbT);
    // per-thread constants: the window offset of residue 0 of the group, the taps of its P residues shifted to that window
    const int s0 = g * P;
    bool valid[P];
    R h[P][W];
    const int64_t cbase = a.d0 - 1;   // c_s = cbase + (phi0m1 + s M) div L
    int c0rel = 0;
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const int s = s0 + k;
        valid[k] = s < a.L;
        const int64_t p = a.phi0m1 + (int64_t)(valid[k] ? s : 0) * a.M;
        const int phi = (int)(p % a.L);
        const int crel = (int)(p / a.L);
        if (k == 0) c0rel = crel;
        const int delta = valid[k] ? crel - c0rel : 0;   // 0 .. P - 1 (M <= L whenever P > 1)
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const int i = j - delta;
            h[k][j] = (valid[k] && i >= 0 && i < a.tp) ? pfb[(int64_t)i * a.L + phi] : (R)0;
        }
    }
    const int64_t ntiles = (a.nrounds + a.Q - 1) / a.Q;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t q0 = tile * a.Q;
        const int nq = (int)std::min<int64_t>(a.Q, a.nrounds - q0);
        const int64_t z0 = q0 * a.M + cbase;          // first staged index into [history ; x]
        const int nz = nq * a.M + a.M + W;
        __syncthreads();   // the previous tile is consumed
        if (z0 >= a.hl) {  // steady state: the tile lies inside x (descriptor re-based at the tile start: zeros past the end of the signal)
            const XS* src = xc + (z0 - a.hl);
            const __amdgpu_buffer_rsrc_t rs = io::make_rsrc(src, (a.xlen - (z0 - a.hl)) * (int64_t)sizeof(XS));
            const int step = blockDim.x;
            for (int k2 = threadIdx.x; k2 < nz; k2 += 4 * step) {
                XS v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = io::Ld<XS>::load(rs, (k2 + u * step) * (int)sizeof(XS));
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (k2 + u * step < nz) zs[k2 + u * step] = to_acc(v[u], (A*)nullptr);
            }
        } else {           // the first tile(s) straddle the history
            for (int k2 = threadIdx.x; k2 < nz; k2 += blockDim.x) {
                const int64_t zi = z0 + k2;
                A v{};
                if (zi < a.hl) v = to_acc(hc[zi], (A*)nullptr);
                else if (zi - a.hl < a.xlen) v = to_acc(xc[zi - a.hl], (A*)nullptr);
                zs[k2] = v;
            }
        }
        __syncthreads();
        if (valid[0]) {
            for (int q = r; q < nq; q += a.RL) {
                const A* zp = zs + q * a.M + c0rel;
                A acc[P];
#pragma unroll
                for (int k = 0; k < P; ++k) acc[k] = zero_acc<A>();
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    const A xv = zp[j];
#pragma unroll
                    for (int k = 0; k < P; ++k) fma_acc(acc[k], h[k][j], xv);
                }
                const int64_t m = (q0 + q) * a.L + s0;
#pragma unroll
                for (int k = 0; k < P; ++k)
                    if (valid[k] && m + k < a.nout) yc[m + k] = acc[k];
            }
        }
    }
}

template <typename XS, typename A, typename R, int TPC, int P> int launch(FirRegArgs& a, int64_t nch, hipStream_t st) {
    a.NP = (int)cdiv(a.L, P);
    constexpr int W = TPC + P - 1;
    const bool small = a.NP <= 256;
    a.RL = small ? std::max(1, 256 / a.NP) : 1;
    // rounds per tile: ~48 KiB of staged samples, at least RL rounds
    const int64_t budget = (int64_t)48 * 1024 / (int64_t)sizeof(A);
    int Q = (int)std::max<int64_t>(a.RL, (budget - a.M - W) / std::max(1, a.M));
    Q = std::min(Q, 512);
    Q = (int)std::min<int64_t>(Q, std::max<int64_t>(a.nrounds, 1));
    a.Q = Q;
    a.span = Q * a.M + a.M + W;
    const size_t lds_bytes = (size_t)a.span * sizeof(A);
    if (lds_bytes > 150 * 1024) MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "decimation factor too large for the register-tap polyphase kernel (M=%d)", a.M);
    const int64_t ntiles = cdiv(a.nrounds, (int64_t)Q);
    // resident workgroups per CU: what the kernel's registers and the tile's LDS admit TOGETHER -- a launch of more than that runs a second, partial round
    // (ComplexF64 160//441: 160 threads = 3 waves at 226 registers, two workgroups per CU; the LDS alone says three: 5.0 ms against 3.8, r06s37)
    auto go = [&](auto kern) -> int {
        hipFuncAttributes fa{};
        MDSP_HIP(hipFuncGetAttributes(&fa, (const void*)kern));
        const int regs = std::max(8, (fa.numRegs + 7) / 8 * 8), waves = (a.NP * a.RL + 63) / 64;
        const int by_regs = std::max(1, std::min(8, 512 / regs) * 4 / waves), by_lds = (int)std::max<size_t>(1, (size_t)(150 * 1024) / lds_bytes);
        const int wgs = tunables().wg_per_cu > 0 ? tunables().wg_per_cu : std::min({4, by_regs, by_lds});
        const int64_t per = std::max<int64_t>(1, (int64_t)device_cu_count() * wgs / std::max<int64_t>(1, nch));
        const dim3 grid((unsigned)std::min<int64_t>(ntiles, per), (unsigned)nch);
        if (lds_bytes > 48 * 1024) MDSP_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        hipLaunchKernelGGL(kern, grid, dim3(a.NP * a.RL), lds_bytes, st, a);
        MDSP_LAUNCH_CHECK();
        return MDSP_OK;
    };
    if (small) return go(polyphase_reg_kernel<XS, A, R, TPC, P, 256>);
    if constexpr (P == 1 && TPC * (int)sizeof(R) / 4 <= 96) return go(polyphase_reg_kernel<XS, A, R, TPC, P, 1024>);   // (fits(): what 128 registers hold)
    else MDSP_FAIL(MDSP_ERR_ASSERTION, "register-tap polyphase kernel: %d phase groups of %d taps", a.NP, TPC);
}

// taps per phase rounded up to the next instantiated window
int tpc_of(int64_t tp) { return tp <= 64 ? (int)((tp + 7) / 8 * 8) : tp <= 80 ? 80 : tp <= 96 ? 96 : tp <= 112 ? 112 : 0; }
// registers of the taps: P (TPC + P - 1) values of R; what fits next to the accumulators, the window and the addresses
bool fits(int tpc, int P, int rbytes, int64_t NP) {
    const int regs = P * (tpc + P - 1) * (rbytes / 4);
    return regs <= (NP <= 256 ? 230 : 96);   // 256 threads: one wave per SIMD, 512 registers (the window in flight takes as many as the taps); up to 1024 threads (L > 256 with P = 1): 128
}

template <typename XS, typename A, typename R, int P> int dispatch_tpc(FirRegArgs& a, int64_t nch, hipStream_t st) {
    switch (tpc_of(a.tp)) {
        case 8: return launch<XS, A, R, 8, P>(a, nch, st);
        case 16: return launch<XS, A, R, 16, P>(a, nch, st);
        case 24: return launch<XS, A, R, 24, P>(a, nch, st);
        case 32: return launch<XS, A, R, 32, P>(a, nch, st);
        case 40: return launch<XS, A, R, 40, P>(a, nch, st);
        case 48: return launch<XS, A, R, 48, P>(a, nch, st);
        case 56:
            if constexpr (P == 1 || sizeof(R) == 4) return launch<XS, A, R, 56, P>(a, nch, st);
            break;
        case 64:
            if constexpr (P == 1 || sizeof(R) == 4) return launch<XS, A, R, 64, P>(a, nch, st);
            break;
        case 80:
            if constexpr (P == 1) return launch<XS, A, R, 80, P>(a, nch, st);
            break;
        case 96:
            if constexpr (P == 1) return launch<XS, A, R, 96, P>(a, nch, st);
            break;
        case 112:
            if constexpr (P == 1) return launch<XS, A, R, 112, P>(a, nch, st);
            break;
        default: break;
    }
    MDSP_FAIL(MDSP_ERR_ASSERTION, "no register-tap instantiation for %d taps per phase", a.tp);
}
int choose_p(int x_dtype, bool acc_double, int64_t tp, int64_t L, int64_t M) {
    const int tpc = tpc_of(tp), rbytes = acc_double ? 8 : 4;
    if (tpc == 0 || L > 1024) return 0;
    (void)x_dtype;
    if (M <= L && L >= 2 && fits(tpc, 2, rbytes, cdiv(L, 2)) && !(rbytes == 8 && tpc > 48) && tpc <= 64) return 2;
    if (fits(tpc, 1, rbytes, L)) return 1;
    return 0;
}
template <typename XS, typename A, typename R> int dispatch_p(int P, FirRegArgs& a, int64_t nch, hipStream_t st) {
    return P == 2 ? dispatch_tpc<XS, A, R, 2>(a, nch, st) : dispatch_tpc<XS, A, R, 1>(a, nch, st);
}

}  // namespace

namespace mdsp {
bool fir_reg_ok(int x_dtype, bool acc_double, int64_t tp, int64_t L, int64_t M) {
    if (x_dtype == MDSP_F32 && !acc_double) return false;   // the Float32 register-tap kernel of fir.hip (lane maps, packed FMAs) keeps that case
    if (acc_double != dtype_is_double(x_dtype)) return false;   // (mixed precision: generic kernel)
    return choose_p(x_dtype, acc_double, tp, L, M) != 0;
}
int fir_reg_run(int x_dtype, bool acc_double, FirRegArgs& a, int64_t nch, hipStream_t st) {
    const int P = choose_p(x_dtype, acc_double, a.tp, a.L, a.M);
    if (P == 0) MDSP_FAIL(MDSP_ERR_ASSERTION, "register-tap polyphase kernel: shape not instantiated");
    a.nrounds = cdiv(a.nout, (int64_t)a.L);
    switch (x_dtype) {
        case MDSP_F64: return dispatch_p<double, double, double>(P, a, nch, st);
        case MDSP_C32: return dispatch_p<cx<float>, cx<float>, float>(P, a, nch, st);
        case MDSP_C64: return dispatch_p<cx<double>, cx<double>, double>(P, a, nch, st);
        default: MDSP_FAIL(MDSP_ERR_ASSERTION, "register-tap polyphase kernel: dtype %d", x_dtype);
    }
}
}  // namespace mdsp
