// Stateful polyphase FIR filtering / rational resampling (mdsp_fir_*) and time-domain FIR (mdsp_tdfir_exec).
//
// Reference loop being replaced (src/Filters/stream_filt.jl:496-509, and its :409-428 / :452-463 / :541-552 twins):
//     while inputIdx <= xLen
//         y[bufIdx += 1] = unsafe_dot(pfb, phiIdx, (history,) x, inputIdx)      util.jl:225-283 (BLAS.dot per sample)
//         inputIdx += div(phiIdx + M - 1, L);  phiIdx = mod1(phiIdx + mod(M, L), L)
// The recurrence is serial; its closed form makes every output independent: for 0-based output m
//     p = (phi0 - 1) + m*M,   phi_m = p mod L (0-based column),   inputIdx_m = d0 + p div L (1-based, as the reference)
//     y[m] = sum_{i=0}^{tp-1} pfb[i, phi_m] * z[inputIdx_m - 1 + i],     z = [history (tp-1 samples) ; x]
// FIRStandard is L = M = 1, FIRDecimator L = 1, FIRInterpolator M = 1 (pfb = reversed taps for L = 1), so one
// kernel serves all four reference kernels.  phi0 / d0 / history are exactly the reference's state and are
// advanced on the host with the same integer arithmetic (bit-exact, see mdsp_fir_exec).
//
// Kernel: each workgroup produces a tile of consecutive outputs of one channel.  The input span of the tile
// (plus the tp-1 sample history halo) is staged once into LDS with coalesced loads, the transposed filter bank
// pfbT[i][phi] lives in LDS as well (phase index contiguous: lanes advance by M mod L phases per output, which
// spreads them over the banks), and every output is a tp-term fused multiply-add chain, oldest sample first.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <numeric>
#include <vector>

#include <atomic>

#include "common.h"
#include "hostpipe.h"
#include "devio.h"
#include "arb_scan.h"
#include "fir_reg.h"
#include "fft_lds.h"

using namespace mdsp;
using mdsp::fft::cx;

// The SI load / store optimizer per KERNEL (round 5 prep; round 4 switched it off for this whole file): merged ds_read2_b64 run at half rate and cost
// FIRArbitrary 8 %, 160//441 Float32 7 %, but the matrix-core kernel's fetched-tap and padded-run forms LOSE 6 - 13 % without the pass
// (profiles/r04_fir_lso_ab.json).  A function attribute decides it, and a body inlined into a kernel takes the kernel's setting.
#define MDSP_NO_LSO __attribute__((target("no-load-store-opt")))
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
template <typename R> __device__ __forceinline__ R mul_first(R h, R x) { return h * x; }
template <typename R> __device__ __forceinline__ cx<R> mul_first(R h, cx<R> x) { return {h * x.x, h * x.y}; }

// 2 a - b  (extrapolate_signal!)
template <typename R> __device__ __forceinline__ R sub2(R a, R b) { return (R)2 * a - b; }
template <typename R> __device__ __forceinline__ cx<R> sub2(cx<R> a, cx<R> b) { return {(R)2 * a.x - b.x, (R)2 * a.y - b.y}; }
// muladd(yUpper, alpha::Float64, yLower) evaluated in Float64 and rounded once to the buffer's element type
__device__ __forceinline__ float arb_combine(float up, double al, float lo) { return (float)fma((double)up, al, (double)lo); }
__device__ __forceinline__ double arb_combine(double up, double al, double lo) { return fma(up, al, lo); }
__device__ __forceinline__ cx<float> arb_combine(cx<float> up, double al, cx<float> lo) {
    return {(float)fma((double)up.x, al, (double)lo.x), (float)fma((double)up.y, al, (double)lo.y)};
}
__device__ __forceinline__ cx<double> arb_combine(cx<double> up, double al, cx<double> lo) { return {fma(up.x, al, lo.x), fma(up.y, al, lo.y)}; }

struct FirArgs {
    const void* x;       // (xlen, nch), ld ldx, storage type XS
    const void* hist;    // (hl, nch) storage type XS
    void* y;             // (ycap, nch), ld ldy, type A
    const void* pfbT;    // tp * L taps (compute real type R), pfbT[i*L + phi]
    int64_t xlen, ldx, ldy, nout;
    int64_t phi0m1;      // phi0 - 1
    int64_t d0;          // input deficit (1-based first input index)
    int L, M, tp, hl;
    int tile;            // outputs per workgroup
    int span;            // LDS samples per tile (upper bound)
    int pfb_in_lds;
};

// XS: storage type of x (float, double, cx<float>, cx<double>);  A: accumulate/output type (R or cx<R>)
template <typename XS, typename A, typename R>
MDSP_NO_LSO __global__ __launch_bounds__(256) void polyphase_fir_kernel(FirArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    A* zs = reinterpret_cast<A*>(smem);                                  // staged input span (converted to A)
    R* ps = reinterpret_cast<R*>(smem + (size_t)a.span * sizeof(A));      // pfbT (optional)
    const int64_t ch = blockIdx.y;
    const int64_t m0 = (int64_t)blockIdx.x * a.tile;
    if (m0 >= a.nout) return;
    const int cnt = (int)std::min<int64_t>(a.tile, a.nout - m0);
    const XS* xc = static_cast<const XS*>(a.x) + ch * a.ldx;
    const XS* hc = static_cast<const XS*>(a.hist) + ch * (int64_t)a.hl;
    // z-index range of the tile: first = inputIdx(m0) - 1, last = inputIdx(m0+cnt-1) - 1 + tp - 1
    const int64_t p_first = a.phi0m1 + m0 * a.M;
    const int64_t p_last = a.phi0m1 + (m0 + cnt - 1) * a.M;
    const int64_t z_first = a.d0 + p_first / a.L - 1;
    const int64_t z_last = a.d0 + p_last / a.L - 1 + a.tp - 1;
    const int nz = (int)(z_last - z_first + 1);
    for (int k = threadIdx.x; k < nz; k += blockDim.x) {
        const int64_t zi = z_first + k;  // index into [history ; x]
        A v{};
        if (zi < a.hl) v = to_acc(hc[zi], (A*)nullptr);
        else if (zi - a.hl < a.xlen) v = to_acc(xc[zi - a.hl], (A*)nullptr);
        zs[k] = v;
    }
    const R* pf = static_cast<const R*>(a.pfbT);
    if (a.pfb_in_lds) {
        const int np = a.tp * a.L;
        for (int k = threadIdx.x; k < np; k += blockDim.x) ps[k] = pf[k];
        pf = ps;
    }
    __syncthreads();
    A* yc = static_cast<A*>(a.y) + ch * a.ldy;
    for (int j = threadIdx.x; j < cnt; j += blockDim.x) {
        const int64_t p = p_first + (int64_t)j * a.M;
        const int phi = (int)(p % a.L);
        const int zoff = (int)(a.d0 + p / a.L - 1 - z_first);
        const A* zp = zs + zoff;
        const R* hp = pf + phi;
        A acc = mul_first(hp[0], zp[0]);
        for (int i = 1; i < a.tp; ++i) fma_acc(acc, hp[(int64_t)i * a.L], zp[i]);
        yc[m0 + j] = acc;
    }
}

// history update (shiftin!, util.jl:299-314): new = last hl samples of [old ; x]
template <typename XS>
MDSP_NO_LSO __global__ __launch_bounds__(256) void shiftin_kernel(const XS* __restrict__ x, const XS* __restrict__ old, XS* __restrict__ neu, int64_t xlen,
                                                      int64_t ldx, int hl) {
    const int64_t ch = blockIdx.y;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < hl; k += gridDim.x * blockDim.x) {
        const int64_t zi = (int64_t)hl + xlen - hl + k;  // index into [old ; x] of new[k]  = xlen + k
        XS v;
        if (zi < hl) v = old[ch * hl + zi];
        else v = x[ch * ldx + (zi - hl)];
        neu[ch * hl + k] = v;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Fast path (Float32 taps and real Float32 signal, tapsPerPhase <= 64): "a thread owns P consecutive output residues".
//
// Outputs m = q*L + s (round q, residue s) all use phase phi_s = (phi0-1 + s*M) mod L and the z-window that starts at
// q*M + c_s,  c_s = d0-1 + (phi0-1 + s*M) div L  -- both independent of q.  So a thread keeps the taps of its P residues
// in registers for the whole kernel, pre-shifted by delta_k = c_{s+k} - c_s (0 <= delta_k <= P-1 when M <= L), and per
// round reads ONE window of W = TPC+P-1 samples from LDS for its P outputs:  LDS reads per output drop from 2*tp
// (generic kernel: tap + sample) to W/P, and the FMA chain keeps the oldest-sample-first order.
// ------------------------------------------------------------------------------------------------------------
struct FirFastArgs {
    const float* x;
    const float* hist;
    float* y;
    const float* pfbT;   // tp * L
    int64_t xlen, ldx, ldy, nout, nrounds;
    int64_t phi0m1, d0;
    int L, M, tp, hl;
    int Q;               // rounds per tile
    int NP, RL;          // phase groups, round lanes (blockDim.x = NP * RL)
    int span;            // LDS floats per tile
    unsigned char gmap[256];   // thread -> phase group (a permutation of 0..NP-1 per round lane): see fir_lane_map
};

template <int TPC, int P>
MDSP_NO_LSO __global__ __launch_bounds__(256) void polyphase_fast_kernel(FirFastArgs a) {
    constexpr int W = TPC + P - 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* zs = reinterpret_cast<float*>(smem);
    const int64_t ch = blockIdx.y;
    const int g = a.gmap[threadIdx.x], r = threadIdx.x / a.NP;
    const float* xc = a.x + ch * a.ldx;
    const float* hc = a.hist + ch * (int64_t)a.hl;
    float* yc = a.y + ch * a.ldy;
    // per-thread constants: offsets c_k (relative to c of residue 0 of the tile), shifted taps
    const int s0 = g * P;
    int coff[P];
    bool valid[P];
    float h[P][W];
    const int64_t cbase = a.d0 - 1;   // c_s = cbase + (phi0m1 + s*M) div L
    int c0rel = 0;
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const int s = s0 + k;
        valid[k] = s < a.L;
        const int64_t p = a.phi0m1 + (int64_t)(valid[k] ? s : 0) * a.M;
        const int phi = (int)(p % a.L);
        const int crel = (int)(p / a.L);           // c_s - cbase
        if (k == 0) c0rel = crel;
        const int delta = valid[k] ? crel - c0rel : 0;   // 0..P-1 (M <= L)
        coff[k] = delta;
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const int i = j - delta;
            h[k][j] = (valid[k] && i >= 0 && i < a.tp) ? a.pfbT[(int64_t)i * a.L + phi] : 0.0f;
        }
    }
    (void)coff;
    const int64_t ntiles = (a.nrounds + a.Q - 1) / a.Q;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t q0 = tile * a.Q;
        const int nq = (int)std::min<int64_t>(a.Q, a.nrounds - q0);
        // z range of the tile: [q0*M + cbase, q0*M + cbase + nq*M + M + W)
        const int64_t z0 = q0 * a.M + cbase;
        const int nz = nq * a.M + a.M + W;
        __syncthreads();   // previous tile fully consumed
        if (z0 >= a.hl) {
            // steady state: the tile lies inside x.  Descriptor re-based at the tile start (hardware zero fill past the end
            // of the signal); 8 independent coalesced loads in flight per thread before the LDS writes.
            const float* src = xc + (z0 - a.hl);
            const __amdgpu_buffer_rsrc_t rs = io::make_rsrc(src, (a.xlen - (z0 - a.hl)) * 4);
            const int step = blockDim.x;
            for (int k2 = threadIdx.x; k2 < nz; k2 += 8 * step) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = io::Ld<float>::load(rs, (k2 + u * step) * 4);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (k2 + u * step < nz) zs[k2 + u * step] = v[u];
            }
        } else {  // the first tile(s) straddle the history
            for (int k2 = threadIdx.x; k2 < nz; k2 += blockDim.x) {
                const int64_t zi = z0 + k2;
                float v = 0.0f;
                if (zi < a.hl) v = hc[zi];
                else if (zi - a.hl < a.xlen) v = xc[zi - a.hl];
                zs[k2] = v;
            }
        }
        __syncthreads();
        if (valid[0]) {
            for (int q = r; q < nq; q += a.RL) {
                const float* zp = zs + q * a.M + c0rel;
                float acc[P];
                if constexpr (P == 2) {   // both residues in one v_pk_fma_f32 per tap (sample broadcast by op_sel)
                    typedef float f2v __attribute__((ext_vector_type(2)));
                    f2v a2 = {0.0f, 0.0f};
#pragma unroll
                    for (int j = 0; j < W; ++j) {
                        const float xv = zp[j];
                        a2 = __builtin_elementwise_fma(f2v{xv, xv}, f2v{h[0][j], h[1][j]}, a2);
                    }
                    acc[0] = a2.x;
                    acc[1] = a2.y;
                } else if constexpr (P == 4) {   // four residues: two packed FMAs per tap, W / 4 LDS reads per output instead of W / 2
                    typedef float f2v __attribute__((ext_vector_type(2)));
                    f2v a01 = {0.0f, 0.0f}, a23 = {0.0f, 0.0f};
#pragma unroll
                    for (int j = 0; j < W; ++j) {
                        const float xv = zp[j];
                        a01 = __builtin_elementwise_fma(f2v{xv, xv}, f2v{h[0][j], h[1][j]}, a01);
                        a23 = __builtin_elementwise_fma(f2v{xv, xv}, f2v{h[2][j], h[3][j]}, a23);
                    }
                    acc[0] = a01.x; acc[1] = a01.y; acc[2] = a23.x; acc[3] = a23.y;
                } else {
#pragma unroll
                    for (int k = 0; k < P; ++k) acc[k] = 0.0f;
#pragma unroll
                    for (int j = 0; j < W; ++j) {
                        const float xv = zp[j];
#pragma unroll
                        for (int k = 0; k < P; ++k) acc[k] = fmaf(xv, h[k][j], acc[k]);
                    }
                }
                const int64_t m = (q0 + q) * a.L + s0;
#pragma unroll
                for (int k = 0; k < P; ++k)
                    if (valid[k] && m + k < a.nout) yc[m + k] = acc[k];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Matrix-core path: the polyphase bank as the B operand of v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64 (Float32 taps on Float32 /
// ComplexF32 signals, Float64 arithmetic on Float64 / ComplexF64 signals; any ratio with L <= 1024 and filters of any length).
//
// The register-tap kernel above is bound by the LDS and the vector ALU: P = 2 residues share one window (16.5 ds_read_b32 and
// 16.5 v_pk_fma_f32 per output; 68 % LDS-array busy, the packed FMAs at 8 - 10 clocks each between their LDS reads against 4.4 in
// isolation), more residues per thread cost VGPRs and occupancy (P = 4: 228 VGPRs, slower) -- 2.5 ms on BASELINE config 5 against an
// HBM floor of 1.6 - 1.9.
// Outputs of the same residue s in different rounds q share their taps, and neighbouring residues read windows that start delta_j =
// c_{s0+j} - c_{s0} <= j samples apart: for 16 rounds x 16 residues,
//      Y[q][j] = sum_k  X[q][k] * H[k][j],    X[q][k] = z[q M + c_{s0} + k],    H[k][j] = pfb[phase_j][k - delta_j]  (0 outside the bank)
// is a (16 x K)(K x 16) product with K = tp + delta_15 -- what the f32 matrix instruction computes, four k per issue: 1024 multiply-adds
// per operand read from LDS, at the vector unit's peak FMA rate, and with the VALU left free.  Its arithmetic is bit for bit a k-ordered
// fmaf chain (one rounding per product, no wider accumulator: verified bit for bit against the register-tap kernel, tests/test_gpu_boundary.py), i.e. exactly the oldest-sample-first chain of the
// register-tap kernel: the zero taps add exact zeros and the two kernels agree bit for bit (tests/test_gpu_boundary.py).
// This is not a GEMM reshaping of the problem: no operand is materialised or reordered in memory, X is the staged signal tile
// itself (lane l reads z[(q0 + l%16) M + c + 4 t + l/16], one ds_read_b32 per MFMA), H lives in T VGPRs per wave for the whole
// kernel (lane l: tap k = 4 t + l/16 of residue j = l%16), and the roofline that bounds the kernel stays HBM.
//   tile   = 64 rounds = 64 M input samples + the window tail, staged in LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no VGPRs, no
//            ds_write pass); its 64 L outputs go through an LDS output buffer (row pitch 16 NB + 4) and leave as one contiguous
//            run of 16-byte stores
//   waves  = NB multiplying waves (one block of 16 residues each, the 64 rounds as four independent accumulators), then nd waves
//            that only issue the DMA of the next tile and ns waves that only store the previous tile's outputs: the memory
//            waves work through the whole tile period beside the MFMAs, one s_barrier per tile
//   LDS    = two sample buffers + two output buffers (config 5: 2 x 38 KiB + 2 x 41 KiB = 158 KiB, one workgroup of 16 waves per CU)
// Variations, all chosen by fir_mm_geo(): for L < 16 a row of the product is RB whole rounds (RB L <= 16 consecutive outputs, RB M
// samples) and several groups of rows share a tile; 2 or 1 chunks of 16 rows per wave when the tile would not fit; rows staged one by
// one (padded pitch) when their sample stride is bank-hostile; taps fetched from L2 per tile when they do not fit registers (long
// filters, or several column blocks per wave for L > 192).
// Measured on BASELINE config 5 (4 ch x 2^28, 160//147, 5120 taps): 1.86 ms = 4.8 TB/s of algorithmic traffic, the device-copy
// rate of this GPU, against 2.4 - 2.5 ms for the register-tap kernel (profiles/r02q_*); other ratios and types: DESIGN.md section 4.6.
// ------------------------------------------------------------------------------------------------------------
struct FirMArgs {
    const void* x;               // (xlen, nch) samples: R or (R, R) pairs
    const void* hist;
    void* y;
    const void* pfbT;            // tp * L taps of type R
    int64_t xlen, ldx, ldy, nout;
    int64_t nrows;               // rows of Lr outputs in this call: ceil(nout / Lr)
    int64_t d0;
    int L, M, hl, tp;            // the filter's ratio L // M
    int Lr, Mr;                  // a ROW = RB rounds = Lr = RB L consecutive outputs, Mr = RB M input samples (RB = 1 when L >= 16)
    int NB, NG;                  // blocks of 16 columns (outputs of a row); groups of 16 CH rows per tile
    int NBW;                     // multiplying waves per row group: wave w takes the column blocks w % NBW, w % NBW + NBW, .. (more than one only with T = 0)
    int Lp;                      // row pitch of the output buffer in LDS (R elements): Lr CS when NB = 1 (contiguous outputs), else 16 NB CS + 16 bytes
    int bufsz;                   // dwords per LDS sample buffer (two of them, then two output buffers)
    int steps;                   // k-steps of four taps when they do not fit registers (template T = 0): the taps are then fetched per tile
    int pitch;                   // 0: the samples of a tile are staged as one run (row r starts at r Mr); else every row is staged on its own, pitch dwords apart
    int rowpad;                  // > 0 (with pitch == 0): one run per tile, staged in the same 256-dword DMA granules as the plain run, with rowpad dwords of
                                 // padding behind every granule (dword d of the tile sits at d + (d / 256) rowpad): rows whose stride is bank-hostile spread
                                 // over the banks without the duplicated window tails of the row-staged form
    int memprio;                 // 1: the DMA and store waves run at raised priority (s_setprio)
    int vstore;                  // 1: output rows that are not whole vectors leave as gathered wide stores (0: element by element, round 2)
    int ablate;                  // MDSP_DEBUG_KNOBS builds (MDSP_ABLATE): 1 no tile DMA after the first, 2 no matrix products, 4 no output stores
    int nd, ns;                  // waves that issue the LDS-DMA / that store, after the multiplying waves
    unsigned lmagic, rmagic;     // ceil(2^32 / L), ceil(2^32 / (Lr CS)): quotients of small numbers by multiply-high
    int phi0m1;                  // phi0 - 1: output j of a row has phase (phi0-1 + j M) mod L and window start (phi0-1 + j M) div L
};

#ifndef MDSP_FIR_TAP_CARRY
#define MDSP_FIR_TAP_CARRY 1   // long-filter mode, ComplexF64: fetched taps carried raw into the next group of k-steps (0: masked next to the fetch)
#endif
typedef int mm_i4 __attribute__((ext_vector_type(4)));
// 64 consecutive dwords of a raw buffer straight into LDS (no VGPRs, no ds_write pass): lane l moves the dword at byte offset voff
// to lds_byte + 4 l.  Issued by hand: the compiler's wait-count bookkeeping makes every later ds_read wait for a DMA it knows of
// (it cannot tell the two tile buffers apart), which would serialise the next tile's loads with this tile's products.  M0 carries
// the LDS address and is the compiler's to use, so it is saved and restored inside the statement.
__device__ __forceinline__ void mm_dma64(mm_i4 rsrc, unsigned lds_byte, int voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_byte), "v"(voff), "s"(rsrc) : "memory");
}
// 256 consecutive dwords, 16 bytes per lane (lane l: byte offset voff -> lds_byte + 16 l).  A 16-byte access that crosses the end of
// the buffer is dropped whole, so the caller uses this form only for granules that lie inside the signal.
__device__ __forceinline__ void mm_dma256(mm_i4 rsrc, unsigned lds_byte, int voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_byte), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ mm_i4 mm_rsrc(const void* base, long long bytes) {   // raw buffer: zero fill past `bytes`
    const unsigned long long p = (unsigned long long)base;
    const long long nb = bytes < 0 ? 0 : (bytes > 0x7ffffff0ll ? 0x7ffffff0ll : bytes);
    return mm_i4{(int)__builtin_amdgcn_readfirstlane((unsigned)p), (int)(__builtin_amdgcn_readfirstlane((unsigned)(p >> 32)) & 0xffffu),
                 (int)__builtin_amdgcn_readfirstlane((unsigned)nb), 0x00020000};
}
// the matrix instruction per element type: D (16 x 16) += A (16 x 4) B (4 x 16); lane l supplies A[l % 16][l / 16] and B[l / 16][l % 16]
template <typename R> struct Mm;
template <> struct Mm<float> {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int lk, int r) { return 4 * lk + r; }   // D register r of lane l: row 4 (l / 16) + r, column l % 16
};
template <> struct Mm<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int lk, int r) { return lk + 4 * r; }   // D register r of lane l: row (l / 16) + 4 r, column l % 16
};

// R: Float32 / Float64 arithmetic (signal and taps of that type); CS: 1 real signal, 2 complex signal (interleaved pairs: the two parts
// are two products against the same taps); CH: 16-row chunks per multiplying wave (independent accumulators); T: k-steps of four taps
// RP: padded runs (a separate instantiation: the two forms of the product loop in one function cost the plain form 20 - 70 VGPRs)
// NBLK > 1: a multiplying wave owns NBLK column blocks (L > 192) with the taps of ALL of them in registers (T k-steps each)
template <typename R, int CS, int CH, int T, bool RP = false, int NBLK = 1>
__device__ __forceinline__ void polyphase_mfma_body(const FirMArgs& a) {
    typedef typename Mm<R>::acc_t acc_t;
    constexpr int DW = (int)(sizeof(R) / 4) * CS;   // dwords per sample
    constexpr int VW = 16 / (int)sizeof(R);         // R elements per 16-byte store
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    R* zs = reinterpret_cast<R*>(smem);            // two sample buffers of bufsz dwords, then two output buffers [Q][Lp]
    const int Q = 16 * CH * a.NG;                  // rows per tile
    const int64_t ch = blockIdx.y;
    const R* xc = static_cast<const R*>(a.x) + ch * a.ldx * CS;
    const R* hc = static_cast<const R*>(a.hist) + ch * (int64_t)a.hl * CS;
    R* yc = static_cast<R*>(a.y) + ch * a.ldy * CS;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int lj = lane & 15, lk = lane >> 4;
    // Wave roles: the first NB NG waves multiply (wave w: column block w % NB, rows 16 CH (w / NB) ..), the next nd waves issue the
    // LDS-DMA of the samples, the last ns waves store the outputs.  The memory waves work through the whole tile period beside the
    // MFMAs -- a wave that first moved data and then multiplied would hold up its SIMD's matrix pipe (measured: 2 000 - 6 000 clocks
    // of DMA issue / store issue per tile, against 4 600 of MFMA) -- and no wave waits on a vmcnt that mixes loads with younger stores.
    const int ncomp = a.NBW * a.NG;
    const bool is_comp = wave < ncomp, is_dma = wave >= ncomp && wave < ncomp + a.nd, is_store = wave >= ncomp + a.nd;
    const int wb0 = wave % a.NBW, wg = wave / a.NBW;
    if (!is_comp && a.memprio) __builtin_amdgcn_s_setprio(3);   // the memory waves' few instructions go first: a late DMA or store holds the whole tile up
    // H: this wave's taps, in registers for the whole kernel (T k-steps) -- or, for filters too long for that (T = 0), fetched from the
    // bank in L2 for every tile (a dword per lane and k-step, beside the four to eight MFMAs it feeds)
    constexpr int TR = T ? T : 1;
    const int steps = T ? T : a.steps;
    R hreg[NBLK][TR];
    int wb = wb0, c0 = 0, tap_phase = 0, tap_delta = 0;
    bool tap_valid = false;
    const R* pf = static_cast<const R*>(a.pfbT);
    const auto block_setup = [&](int b) {   // window start of column block b and this lane's column of it
        wb = b;
        const int s0 = 16 * b, sj = s0 + lj;
        const unsigned p0 = (unsigned)(a.phi0m1 + s0 * a.M), pj = (unsigned)(a.phi0m1 + sj * a.M);
        c0 = a.L == 1 ? (int)p0 : (int)__umulhi(p0, a.lmagic);   // (ceil(2^32 / 1) does not fit the magic)
        const int cj = a.L == 1 ? (int)pj : (int)__umulhi(pj, a.lmagic);
        tap_phase = (int)pj - cj * a.L;
        tap_delta = cj - c0;
        tap_valid = sj < a.Lr;
    };
    if (is_comp) block_setup(wb0);
    const auto tap = [&](int t) -> R {   // H[k = 4 t + lane / 16][j = lane % 16]
        const int i = 4 * t + lk - tap_delta;
        const bool ok = tap_valid && i >= 0 && i < a.tp;
        // branch-free: lanes without a tap read the bank's first entry and discard it -- a skipped load is an exec-masked branch per fetch,
        // and the long-filter mode's eight fetches in flight then come out one by one with a wait between them
        const R v = pf[ok ? (int64_t)i * a.L + tap_phase : (int64_t)0];
        return ok ? v : (R)0;
    };
#pragma unroll
    for (int kb = 0; kb < NBLK; ++kb) {
        if (NBLK > 1 && is_comp && wb0 + kb * a.NBW < a.NB) block_setup(wb0 + kb * a.NBW);
#pragma unroll
        for (int t = 0; t < TR; ++t) hreg[kb][t] = (T && is_comp && wb0 + kb * a.NBW < a.NB) ? tap(t) : (R)0;
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): the taps are in (their first use must not look like a pending load inside the tile loop)
    const int64_t cbase = a.d0 - 1;
    const int wtail = 4 * steps + 4;   // samples read past a window start
    const int64_t ntiles = (a.nrows + Q - 1) / Q;
    // Software pipeline over tiles with ONE barrier per tile.  LDS holds two sample buffers and two output buffers; in iteration t
    //   sample buffer t+1 receives the NEXT tile by LDS-DMA (no registers, no ds_write pass), issued right after the barrier;
    //   sample buffer t   feeds the products, and every multiplying wave writes its outputs into output buffer t as soon as its
    //                     own MFMAs are done (nobody else touches those rows and columns);
    //   output buffer t-1 leaves as coalesced stores.
    // The first tile(s), which straddle the history, are filled by ordinary loads (all waves).
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) R*)zs;
    const auto dma_ok = [&](int64_t tile) { return tile < ntiles && tile * Q * a.Mr + cbase >= a.hl; };
    // Row-staged tiles (pitch != 0): when the sample stride of a row, Mr, is a multiple of 8 the 16 rows of an A operand would sit on
    // 2 - 8 banks of a linear tile (147//160: all on one); every row is then DMA-ed on its own -- its Mr samples and the window tail
    // again -- to rows pitch = 256 g + 4 dwords apart (two-way conflicts at worst); the duplicates come from L2.  (Round 3: where a window is
    // shorter than a granule the padded run -- rowpad, below -- replaces this form.)
    const int rgran = a.pitch ? (a.pitch - 4) / 256 : 0;                    // 256-dword granules per row
    const int padE = RP ? a.rowpad / (int)(sizeof(R) / 4) : 0;                 // R elements of padding behind a granule (padded runs)
    const int zpitch = a.pitch ? a.pitch / (int)(sizeof(R) / 4) : a.Mr * CS;   // R elements between rows (padded runs: + padE from segment to segment)
    const auto dma = [&](int64_t tile, int buf) {
        if (!is_dma || !dma_ok(tile)) return;
        const int64_t q0 = tile * Q, z0 = q0 * a.Mr + cbase;
        const int nq = (int)std::min<int64_t>(Q, a.nrows - q0);
        const mm_i4 rs = mm_rsrc(xc + (z0 - a.hl) * CS, (a.xlen - (z0 - a.hl)) * 4 * DW);   // re-based at the tile start: zero fill past the signal
        const int64_t exist = (a.xlen - (z0 - a.hl)) * DW;                     // dwords of the signal from the tile start on
        if (a.pitch == 0) {
            const int nzd = (nq * a.Mr + a.Mr + wtail) * DW;   // dwords of the tile
            for (int i = wave - ncomp; 256 * i < nzd; i += a.nd) {   // granules of 256 dwords, round-robin over the DMA waves
                const unsigned dst = lds0 + (unsigned)(buf * a.bufsz + (256 + (RP ? a.rowpad : 0)) * i) * 4u;   // (padded runs: rowpad dwords between granules)
                if (256 * (int64_t)(i + 1) <= exist) mm_dma256(rs, dst, (256 * i + 4 * lane) * 4);
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) mm_dma64(rs, dst + 256u * j, (256 * i + 64 * j + lane) * 4);
                }
            }
        } else {
            for (int i = wave - ncomp; i < nq * rgran; i += a.nd) {   // (row, granule) pairs
                const int row = i / rgran, gr = i - row * rgran;
                const int src = row * a.Mr * DW + 256 * gr;                     // dwords from the tile start
                const unsigned dst = lds0 + (unsigned)(buf * a.bufsz + row * a.pitch + 256 * gr) * 4u;
                if (src + 256 <= exist) mm_dma256(rs, dst, (src + 4 * lane) * 4);
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) mm_dma64(rs, dst + 256u * j, (src + 64 * j + lane) * 4);
                }
            }
        }
    };
    // the outputs of a tile, m = q0 Lr .., are contiguous in y: coalesced 16-byte stores (element-aligned; global memory takes them
    // unaligned) by the storing waves, four LDS reads in flight per thread.  Indices count R elements.
    const int st0 = (int)threadIdx.x - 64 * (ncomp + a.nd), stn = 64 * a.ns;
    const int lrc = a.Lr * CS;
    const auto copy_out = [&](int64_t tile, const R* zo) {
        if (!is_store) return;
        const int64_t q0 = tile * Q, mbase = q0 * a.Lr;
        const int total = (int)std::min<int64_t>(std::min<int64_t>(Q, a.nrows - q0) * a.Lr, a.nout - mbase) * CS;
        R* yo = yc + mbase * CS;
        typedef R vec_t __attribute__((ext_vector_type(VW)));
        const bool flat = a.Lp == lrc;   // rows without padding: the buffer IS the run of outputs
        if (flat || lrc % VW == 0) {
            const int nv = total / VW;
            for (int iv = st0; iv < nv; iv += 4 * stn) {
                vec_t v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = VW * std::min(iv + u * stn, nv - 1), row = flat ? 0 : (int)__umulhi((unsigned)idx, a.rmagic), col = idx - row * lrc;
                    v[u] = *reinterpret_cast<const vec_t*>(zo + row * a.Lp + col);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (iv + u * stn < nv) __builtin_memcpy(yo + VW * (iv + u * stn), &v[u], sizeof(vec_t));
            }
            for (int idx = nv * VW + st0; idx < total; idx += stn) {
                const int row = flat ? 0 : (int)__umulhi((unsigned)idx, a.rmagic), col = idx - row * lrc;
                yo[idx] = zo[row * a.Lp + col];
            }
        } else if (a.vstore) {
            // rows of Lr elements that are not whole vectors (147//160: ten column blocks, 147 outputs per row): a vector's elements are
            // gathered one by one -- they may straddle a row end -- and leave as ONE wide store; the run of outputs itself is contiguous
            const int nv = total / VW;
            for (int iv = st0; iv < nv; iv += 2 * stn) {
                vec_t v[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int idx = VW * std::min(iv + u * stn, nv - 1);
                    int row = (int)__umulhi((unsigned)idx, a.rmagic), col = idx - row * lrc;
#pragma unroll
                    for (int j = 0; j < VW; ++j) {
                        v[u][j] = zo[row * a.Lp + col];
                        if (++col == lrc) { col = 0; ++row; }
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (iv + u * stn < nv) __builtin_memcpy(yo + VW * (iv + u * stn), &v[u], sizeof(vec_t));
            }
            for (int idx = nv * VW + st0; idx < total; idx += stn) {
                const int row = (int)__umulhi((unsigned)idx, a.rmagic), col = idx - row * lrc;
                yo[idx] = zo[row * a.Lp + col];
            }
        } else {
            for (int idx = st0; idx < total; idx += stn) {
                const int row = (int)__umulhi((unsigned)idx, a.rmagic), col = idx - row * lrc;
                yo[idx] = zo[row * a.Lp + col];
            }
        }
    };
    // Which row of the tile a row of the 16 x 16 product is: with odd Mr, the 16 EVEN (then the 16 odd) rows of a 32-row span put the
    // 2 x 16 A-operand reads of a lane group on 32 different banks (16 consecutive rows collide two ways).
    const bool eo = CH >= 2 && (a.Mr & 1) && a.pitch == 0;   // (a single 16-row chunk per wave keeps consecutive rows)
    const int ra = eo ? 2 : 1;
    const auto rbase = [&](int c) { return 16 * CH * wg + (eo ? 32 * (c >> 1) + (c & 1) : 16 * c); };
    R* zout = zs + 2 * (a.bufsz / (int)(sizeof(R) / 4));
    const int osz = Q * a.Lp;
    int cur = 0;
    int64_t prev_tile = -1;
    dma(blockIdx.x, 0);
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, cur ^= 1) {
        const int64_t q0 = tile * Q;
        const int nq = (int)std::min<int64_t>(Q, a.nrows - q0);
        const int64_t z0 = q0 * a.Mr + cbase;
        const int nz = nq * a.Mr + a.Mr + wtail;
        const R* zt = zs + cur * (a.bufsz / (int)(sizeof(R) / 4));   // this tile's samples
        R* zo = zout + cur * osz;                                      // this tile's outputs
        if (dma_ok(tile)) {
            if (is_dma) __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): this wave's share of the tile has landed
        } else {  // the first tile(s) straddle the history
            R* zw = zs + cur * (a.bufsz / (int)(sizeof(R) / 4));
            const int rowlen = a.pitch ? (a.Mr + wtail) * CS : nz * CS, nrow = a.pitch ? nq : 1;   // R elements per staged row; rows
            for (int k2 = threadIdx.x; k2 < rowlen * nrow; k2 += blockDim.x) {
                const int row = k2 / rowlen, e = k2 - row * rowlen;
                const int64_t zi = z0 + (int64_t)row * a.Mr + e / CS;
                const int part = e % CS;
                R v = (R)0;
                if (zi < a.hl) v = hc[zi * CS + part];
                else if (zi - a.hl < a.xlen) v = xc[(zi - a.hl) * CS + part];
                zw[row * zpitch + e + (RP ? ((e * (int)(sizeof(R) / 4)) >> 8) * padE : 0)] = v;
            }
            __builtin_amdgcn_s_waitcnt(0x0f70);
        }
        __syncthreads();   // tile t is in; the outputs of t-1 are complete; the other sample buffer and the other output buffer are free
        if (!MDSP_ABLATED(a, 1)) dma(tile + gridDim.x, cur ^ 1);
        if (prev_tile >= 0 && !MDSP_ABLATED(a, 4)) copy_out(prev_tile, zout + (cur ^ 1) * osz);
        if (is_comp && !MDSP_ABLATED(a, 2)) {
            constexpr int KBU = T == 0 ? 1 : NBLK;   // blocks of the unrolled inner loop (their taps are registers hreg[kb])
            for (int b0 = wb0; b0 < a.NB; b0 += KBU * a.NBW)   // (T = 0: the run-time walk over a wave's blocks; else one trip)
#pragma unroll
            for (int kb = 0; kb < KBU; ++kb) {   // (one block per wave unless the taps are fetched, T = 0, or NBLK > 1)
                const int b = b0 + kb * a.NBW;
                if (b >= a.NB) break;
                if (a.NBW < a.NB) block_setup(b);
                acc_t acc[CS][CH];
#pragma unroll
                for (int p = 0; p < CS; ++p)
#pragma unroll
                    for (int c = 0; c < CH; ++c) acc[p][c] = acc_t{(R)0, (R)0, (R)0, (R)0};
                const R* ap[CH];   // A operand: row = lane % 16, k = lane / 16
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const int row = ra * lj + rbase(c);
                    ap[c] = zt + row * zpitch + (RP ? (((row * a.Mr + c0 + lk) * DW) >> 8) * padE : 0) + (c0 + lk) * CS;
                }
                // padded runs (T != 0, a window shorter than a granule: it meets ONE pad at most): the positions behind the granule boundary sit
                // padE elements further on -- a per-lane choice between two base pointers, so that every read keeps its immediate offset (computed
                // addresses cost the read-ahead of the A operands: measured 1.08 against 0.78 ms at 147//160)
                if constexpr (T != 0 && RP) {
                    // dwords from this lane's first window position to the next granule boundary (16 rows are a whole number of granules,
                    // so the distance is the same for every chunk)
                    const int th = 256 - (((ra * lj + rbase(0)) * a.Mr + c0 + lk) * DW & 255);   // (rbase(0): single-chunk waves need not start on a granule, round 5 prep)
#pragma unroll
                    for (int t = 0; t < TR; ++t) {
                        const bool hi = 4 * t * DW >= th;
                        // (round 5 prep) windows longer than a granule -- 4 T DW dwords up to 512 -- meet a second pad: a third base pointer
                        const bool hi2 = 4 * (T - 1) * DW + DW > 256 && 4 * t * DW >= th + 256;
#pragma unroll
                        for (int c = 0; c < CH; ++c) {
                            const R* pp = hi2 ? ap[c] + 2 * padE : hi ? ap[c] + padE : ap[c];
#pragma unroll
                            for (int p = 0; p < CS; ++p) acc[p][c] = Mm<R>::mfma(pp[4 * t * CS + p], hreg[T == 0 ? 0 : kb][t], acc[p][c]);
                        }
                    }
                } else if constexpr (T != 0) {
#pragma unroll
                    for (int t = 0; t < T; ++t)
#pragma unroll
                        for (int c = 0; c < CH; ++c)
#pragma unroll
                            for (int p = 0; p < CS; ++p) acc[p][c] = Mm<R>::mfma(ap[c][4 * t * CS + p], hreg[T == 0 ? 0 : kb][t], acc[p][c]);
                } else {
                    // steps is a multiple of 8: the taps of eight k-steps are fetched while the previous eight are multiplied (a fetch is an L2 round
                    // trip; a wave that waits for it in front of every group leaves its SIMD's matrix pipe idle half of the time).  The fetched
                    // words are carried RAW into the next group and only then masked: a select next to the fetch would wait for it on the spot.
                    // Measured against masking next to the fetch, alternating processes on one box (tools/r03_session35.sh): ComplexF64 3//8 4.94 -> 4.37 ms,
                    // 1//4 2.99 -> 2.76, 1//8 5.16 -> 5.04; Float32 1//8 +3 %, but 1//16 -7 %, Float64 1//16 -4 %: the carried form for ComplexF64 only.
                    // (round 5 prep) padded runs with fetched taps: the window spans any number of granules, so the pads in front of a position are
                    // computed per k-step (one shift and one multiply-add beside a tap fetch and CH matrix instructions; the vector unit idles here)
                    const int off0 = RP ? (((ra * lj + rbase(0)) * a.Mr + c0 + lk) * DW & 255) : 0;
                    const auto padx = [&](int t) { return RP ? ((off0 + 4 * t * DW) >> 8) * padE : 0; };
                    if constexpr (MDSP_FIR_TAP_CARRY && CS == 2 && sizeof(R) == 8) {
                    R hraw[8];
                    unsigned okm = 0;
                    const auto fetch = [&](int tb) {
                        okm = 0;
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int i = 4 * (tb + u) + lk - tap_delta;
                            const bool ok = tap_valid && i >= 0 && i < a.tp;
                            hraw[u] = pf[ok ? (int64_t)i * a.L + tap_phase : (int64_t)0];
                            okm |= ok ? 1u << u : 0u;
                        }
                    };
                    fetch(0);
                    for (int t0 = 0; t0 < steps; t0 += 8) {
                        R h[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) h[u] = (okm >> u & 1u) ? hraw[u] : (R)0;
                        fetch(t0 + 8 < steps ? t0 + 8 : t0);   // (the last group re-reads its own: no branch around the loads)
#pragma unroll
                        for (int u = 0; u < 8; ++u)
#pragma unroll
                            for (int c = 0; c < CH; ++c)
#pragma unroll
                                for (int p = 0; p < CS; ++p) acc[p][c] = Mm<R>::mfma(ap[c][4 * (t0 + u) * CS + p + padx(t0 + u)], h[u], acc[p][c]);
                    }
                    } else {
                    R h[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) h[u] = tap(u);
                    for (int t0 = 0; t0 < steps; t0 += 8) {
                        R hn[8];
                        const int tn = t0 + 8 < steps ? t0 + 8 : t0;
#pragma unroll
                        for (int u = 0; u < 8; ++u) hn[u] = tap(tn + u);
#pragma unroll
                        for (int u = 0; u < 8; ++u)
#pragma unroll
                            for (int c = 0; c < CH; ++c)
#pragma unroll
                                for (int p = 0; p < CS; ++p) acc[p][c] = Mm<R>::mfma(ap[c][4 * (t0 + u) * CS + p + padx(t0 + u)], h[u], acc[p][c]);
#pragma unroll
                        for (int u = 0; u < 8; ++u) h[u] = hn[u];
                    }
                    }
                }
                if (16 * wb + lj < a.Lr) {
#pragma unroll
                    for (int c = 0; c < CH; ++c)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int p = 0; p < CS; ++p) zo[(ra * Mm<R>::row(lk, r) + rbase(c)) * a.Lp + (16 * wb + lj) * CS + p] = acc[p][c][r];
                }
            }
        }
        prev_tile = tile;
    }
    __syncthreads();
    if (prev_tile >= 0) copy_out(prev_tile, zout + (cur ^ 1) * osz);
}
// the two kernels around the body: taps in registers on a plain run -> no merged LDS reads; fetched taps (T = 0) and padded runs -> the pass stays on
template <typename R, int CS, int CH, int T, bool RP = false, int NBLK = 1>
MDSP_NO_LSO __global__ __launch_bounds__(1024) void polyphase_mfma_kernel(FirMArgs a) { polyphase_mfma_body<R, CS, CH, T, RP, NBLK>(a); }
template <typename R, int CS, int CH, int T, bool RP = false, int NBLK = 1>
__global__ __launch_bounds__(1024) void polyphase_mfma_kernel_lso(FirMArgs a) { polyphase_mfma_body<R, CS, CH, T, RP, NBLK>(a); }

// ------------------------------------------------------------------------------------------------------------
// Decimator kernel (round 5): L = 1, any M <= 64, any filter length.  FIRDecimator (stream_filt.jl:43-56, :522-558) is the commonest multirate
// call and was the matrix-core kernel's worst shape: its product has 15 M structural-zero rows, the taps do not fit registers and the row stride M
// is bank-hostile (1//16 Float32 at 0.08 of the HBM roof, 87 % of the LDS cycles bank conflicts: profiles/r04_fir_decim16_pmc.json).
//
//     y[m] = sum_i h'[i] z[c + m M + i]     (h' = pfbT, the taps oldest sample first; z = [history ; x]; c = d0 - 1 + phi0 - 1)
//          = sum_{r < M}  sum_q h'[M q + r] z[c + M (m + q) + r]
//
// A lane owns ONE PHASE r of the M-decimated input: its inner sum is a short correlation (ceil(tp / M) taps) over every M-th sample, and the M
// lanes of a group read M CONSECUTIVE samples per step -- conflict-free whatever M is.  Per block of P = 16 outputs and chunk of QC = 8 taps a
// lane reads P + QC - 1 samples and QC taps from LDS for P QC multiply-adds; the M partial sums of an output meet through LDS (one write per
// lane and output, M reads by the lane that stores it).  Multiply-adds are packed pairs: a complex sample is one (real taps), and a real signal
// pairs the two halves of its tile (the staging loop writes sample k of the first half next to sample k of the second).
// Every output reads exactly its own tp-sample window, as the reference does: the zero taps that pad the last chunk are masked, not multiplied.
// The summation order differs from the reference's oldest-first chain (phases first, then across phases): results agree to rounding, state
// (history, phase, deficit) stays bit-exact -- it never was arithmetic.
// ------------------------------------------------------------------------------------------------------------
template <typename R> struct DecV { R x, y; };
__device__ __forceinline__ DecV<float> dec_fma(DecV<float> w, float g, DecV<float> a) {
    typedef float f2v __attribute__((ext_vector_type(2)));
    const f2v d = __builtin_elementwise_fma(f2v{w.x, w.y}, f2v{g, g}, f2v{a.x, a.y});
    return {d.x, d.y};
}
__device__ __forceinline__ DecV<double> dec_fma(DecV<double> w, double g, DecV<double> a) { return {fma(w.x, g, a.x), fma(w.y, g, a.y)}; }

struct DecArgs {
    const void* x;
    const void* hist;
    void* y;
    const void* pfbT;      // tp taps, oldest sample first
    int64_t xlen, ldx, ldy, nout, zb;   // zb: index into [history ; x] of the first sample of output 0's window
    int M, Mp, logMp, tp, hl;
    int nq;                // ceil(tp / M) tap steps; the LDS tap table is padded to a multiple of QC steps
    int nbh;               // blocks of P outputs per half tile (real signals: a tile is two halves) / per tile (complex)
    int nz;                // staged samples per (half) tile
    unsigned blkmagic;     // ceil(2^32 / (P M)): k div (P M) == umulhi(k, blkmagic) for the staged indices
    int ablate;            // MDSP_FIR_DEC_ABLATE (profiling; garbage results): 1 no multiply-adds, 2 no staging, 4 no reduction / stores
};

// LDS position of staged sample k of a tile: M elements of padding behind every block of P M samples, so that the groups of a wavefront -- whose
// blocks start P M samples apart, a multiple of the 256-byte bank row for every power-of-two M -- read bank rows that interleave instead of coincide
// (first run without it: 2-way conflicts on every read at M = 16, 16-way at M = 2; profiles/r05_fir_dec_ab.json "first").
__device__ __forceinline__ int dec_pos(int k, int M, unsigned blkmagic) { return k + (int)__umulhi((unsigned)k, blkmagic) * M; }

// MC: M as a compile-time constant (0: run-time M) -- the window reads are then one address register plus immediates
// NC: the filter's chunks of QC taps per phase as a compile-time constant (0: run-time loop).  The chunk loop is then unrolled, so a lane's taps live in registers
// for the whole tile and its sample window SLIDES -- QC new samples per chunk instead of P + QC - 1: 8 LDS reads per 128 packed multiply-adds instead of 31.
// NC = 5 is every decimator resample_filter designs (36.4 taps per phase and M, stream_filt.jl / design.jl:700-720): the LDS was as busy as the vector unit
// with the run-time loop (per CU: 4 x 640 clocks of LDS reads against 2800 of multiply-adds per block).
template <typename R, bool CPLX, int P, int MC, int NC = 0>
MDSP_NO_LSO __global__ __launch_bounds__(256, (sizeof(R) == 4 ? 3 : 2)) void decimator_kernel(DecArgs a) {
    using V = DecV<R>;
    using XS = std::conditional_t<CPLX, cx<R>, R>;
    constexpr int QC = 8, PH = 4;   // taps per chunk; outputs per reduction round
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int M = MC ? MC : a.M, Mp = a.Mp;
    const int nqp = (a.nq + QC - 1) / QC * QC;
    const int nzp = dec_pos(a.nz - 1, M, a.blkmagic) + 1;
    V* zs = reinterpret_cast<V*>(smem);
    R* tl = reinterpret_cast<R*>(zs + nzp);
    V* red = reinterpret_cast<V*>(tl + (((size_t)nqp * M + 3) & ~(size_t)3));
    const int64_t ch = blockIdx.y;
    const int TO = (CPLX ? 1 : 2) * a.nbh * P;                  // outputs per tile
    const int64_t ntile = (a.nout + TO - 1) / TO;
    const XS* xc = static_cast<const XS*>(a.x) + ch * a.ldx;
    const XS* hc = static_cast<const XS*>(a.hist) + ch * (int64_t)a.hl;
    const int H = a.nbh * P * M;                                // samples between the two halves of a real tile
    // Float64 / ComplexF64 (PRE): a workgroup walks tiles blockIdx.x, + gridDim.x, ... with the NEXT tile's samples on their way from HBM (in registers: at
    // most 40 KiB a tile, ten 16-byte loads a thread) while this tile is multiplied; they are written to the LDS behind the barrier that ends it.  One
    // tile per workgroup -- the first form -- left staging and arithmetic to overlap across co-resident workgroups only, and they added up
    // (profiles/r05_fir_dec_ab.json "ablation").  Float64 1//4 ... 1//16 0.92 - 1.00 -> 0.82 - 0.98 ms, ComplexF64 1.93 - 2.23 -> 1.54 - 2.02.
    // Float32 / ComplexF32 keep one tile per workgroup: their 170 registers leave no room for the tile in flight at three workgroups per CU, and every
    // form that made room measured slower -- two workgroups per CU 0.51 ms at 1//8 against 0.44, eight outputs per block with three 0.64, with four
    // workgroups and 30 KiB tiles 0.88, 24 KiB tiles 0.64 (profiles/r05_fir_dec_ab.json "prefetch_forms").
    // Steady state: the tile (both halves of a real one) lies wholly inside x.  16-byte loads -- four Float32 samples, two Float64 / ComplexF32, one
    // ComplexF64; consecutive samples sit at consecutive LDS positions (a block's padding never falls inside a group: block lengths are multiples
    // of 16 samples, groups start at multiples of 4)
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    constexpr int NV = 16 / (int)sizeof(XS);
    constexpr bool PRE = sizeof(R) == 8;
    constexpr int NIT = PRE ? (CPLX ? 11 : 6) : 1;              // 256 threads x NIT x 16 (32: both halves) bytes: the 40 KiB of blocks (fir_dec_geo) plus the window tail
                                                                // (M = 16 with 40 tap steps: 43 KiB -- with ten / five loads those tiles fell out of the prefetching form)
    u4v pa[NIT], pb[CPLX ? 1 : NIT];
    const bool fits = a.nz <= 256 * NIT * NV;                   // (filters of several thousand taps per phase: the tile outgrows the registers -- staged in place)
    auto inside = [&](int64_t tile) {
        const int64_t zf = a.zb + tile * TO * M;
        return tile < ntile && !(a.ablate & 2) && zf >= a.hl && zf - a.hl + a.nz + (CPLX ? 0 : H) <= a.xlen;
    };
    auto steady = [&](int64_t tile) { return PRE && fits && inside(tile); };
    auto issue = [&](int64_t tile) __attribute__((always_inline)) {
        const int64_t zf = a.zb + tile * TO * M;
        const __amdgpu_buffer_rsrc_t rs = io::make_rsrc(xc + (zf - a.hl), (a.xlen - (zf - a.hl)) * (long long)sizeof(XS));
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int k = (threadIdx.x + u * 256) * NV;
            pa[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, k < a.nz ? k * (int)sizeof(XS) : io::OOB, 0, 0);
            if constexpr (!CPLX) pb[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, k < a.nz ? (k + H) * (int)sizeof(XS) : io::OOB, 0, 0);
        }
    };
    auto elem = [](const u4v& q, int i) -> XS {
        if constexpr (std::is_same_v<XS, float>) return __uint_as_float(q[i]);
        else if constexpr (std::is_same_v<XS, double>) return __hiloint2double((int)q[2 * i + 1], (int)q[2 * i]);
        else if constexpr (std::is_same_v<XS, cx<float>>) return XS{__uint_as_float(q[2 * i]), __uint_as_float(q[2 * i + 1])};
        else return XS{__hiloint2double((int)q[1], (int)q[0]), __hiloint2double((int)q[3], (int)q[2])};
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int k = (threadIdx.x + u * 256) * NV;
            if (k < a.nz) {
                V* dst = zs + dec_pos(k, M, a.blkmagic);
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    if constexpr (CPLX) {
                        const XS v = elem(pa[u], i);
                        dst[i] = {v.x, v.y};
                    } else dst[i] = {elem(pa[u], i), elem(pb[u], i)};
                }
            }
        }
    };
    auto stage_slow = [&](int64_t tile) {   // tiles that straddle the history or the end of x
        const int64_t zf = a.zb + tile * TO * M;
        if (inside(tile)) {                 // Float32: the whole tile in ONE round of 16-byte loads (ten a thread: the registers of the multiply-adds are not live
                                            // yet; four at a time were three HBM latencies per tile); Float64 tiles beyond the registers: four in flight
            constexpr int U = PRE ? 4 : (CPLX ? 10 : 5);
            const __amdgpu_buffer_rsrc_t rs = io::make_rsrc(xc + (zf - a.hl), (a.xlen - (zf - a.hl)) * (long long)sizeof(XS));
            for (int k0 = threadIdx.x * NV; k0 < a.nz; k0 += U * 256 * NV) {
                u4v va[U], vb[CPLX ? 1 : U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int k = k0 + u * 256 * NV;
                    va[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, k < a.nz ? k * (int)sizeof(XS) : io::OOB, 0, 0);
                    if constexpr (!CPLX) vb[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, k < a.nz ? (k + H) * (int)sizeof(XS) : io::OOB, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int k = k0 + u * 256 * NV;
                    if (k < a.nz) {
                        V* dst = zs + dec_pos(k, M, a.blkmagic);
#pragma unroll
                        for (int i = 0; i < NV; ++i) {
                            if constexpr (CPLX) {
                                const XS v = elem(va[u], i);
                                dst[i] = {v.x, v.y};
                            } else dst[i] = {elem(va[u], i), elem(vb[u], i)};
                        }
                    }
                }
            }
            return;
        }
        auto sample = [&](int64_t zi) -> XS {
            if (zi < a.hl) return hc[zi];
            if (zi - a.hl < a.xlen) return xc[zi - a.hl];
            return XS{};
        };
        for (int k = threadIdx.x; k < a.nz; k += 256) {
            if constexpr (CPLX) {
                const XS v = sample(zf + k);
                zs[dec_pos(k, M, a.blkmagic)] = {v.x, v.y};
            } else zs[dec_pos(k, M, a.blkmagic)] = {sample(zf + k), sample(zf + k + H)};
        }
    };
    int64_t tile = blockIdx.x;
    bool pre = steady(tile);                                    // workgroup-uniform
    if (pre) issue(tile);
    {
        const R* pf = static_cast<const R*>(a.pfbT);
        for (int k = threadIdx.x; k < nqp * M; k += 256) tl[k] = k < a.tp ? pf[k] : (R)0;
    }
    const int r = threadIdx.x & (Mp - 1), gq = threadIdx.x >> a.logMp, ng = 256 >> a.logMp;
    const bool lane_on = r < M;
    const int rr = lane_on ? r : 0;
    V* myred = red + (size_t)gq * PH * (Mp + 1);
    using Y = std::conditional_t<CPLX, cx<R>, R>;
    Y* yc = static_cast<Y*>(a.y) + ch * a.ldy;
    for (; tile < ntile; tile += gridDim.x) {
        const int64_t m0 = tile * TO;
        if (PRE && pre) commit();
        else if (!(a.ablate & 2)) stage_slow(tile);
        __syncthreads();
        if constexpr (PRE) {
            pre = steady(tile + gridDim.x);
            if (pre) issue(tile + gridDim.x);
        }
        for (int bl = gq; bl < a.nbh; bl += ng) {
            V acc[P];
#pragma unroll
            for (int p = 0; p < P; ++p) acc[p] = {(R)0, (R)0};
            // sample (bl P + t) M + r of the tile sits at  bl (P + 1) M + (t + t div P) M + r : t = q0 + j walks the window
            const V* zb = zs + (size_t)bl * (P + 1) * M + rr;
            if constexpr (NC > 0) {
                if (!(a.ablate & 1)) {
                    constexpr bool GREG = sizeof(R) == 4;   // Float64: the taps of a chunk from LDS as it starts (80 registers of taps next to the tile in flight spilt)
                    R g[GREG ? NC * QC : QC];
                    if constexpr (GREG) {
#pragma unroll
                        for (int u = 0; u < NC * QC; ++u) g[u] = tl[u * M + rr];
                    }
                    V w[P + QC - 1];
#pragma unroll
                    for (int j = 0; j < P + QC - 1; ++j) w[j] = zb[(j + j / P) * M];
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const int q0 = c * QC, g0 = GREG ? q0 : 0;
                        if constexpr (!GREG) {
#pragma unroll
                            for (int u = 0; u < QC; ++u) g[u] = tl[(q0 + u) * M + rr];
                        }
                        if (c > 0) {
#pragma unroll
                            for (int j = 0; j < P - 1; ++j) w[j] = w[j + QC];
#pragma unroll
                            for (int j = P - 1; j < P + QC - 1; ++j) {
                                const int t = q0 + j;
                                w[j] = zb[(t + t / P) * M];
                            }
                        }
                        if (c < NC - 1 || (q0 + QC) * M <= a.tp) {
#pragma unroll
                            for (int u = 0; u < QC; ++u)
#pragma unroll
                                for (int p = 0; p < P; ++p) acc[p] = dec_fma(w[p + u], g[g0 + u], acc[p]);
                        } else {   // the last chunk, masked as below
#pragma unroll
                            for (int u = 0; u < QC; ++u) {
                                if ((q0 + u) * M + rr < a.tp) {
#pragma unroll
                                    for (int p = 0; p < P; ++p) acc[p] = dec_fma(w[p + u], g[g0 + u], acc[p]);
                                }
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                }
            } else
            for (int q0 = 0; q0 < ((a.ablate & 1) ? 0 : a.nq); q0 += QC) {
                R g[QC];
                V w[P + QC - 1];
#pragma unroll
                for (int u = 0; u < QC; ++u) g[u] = tl[(q0 + u) * M + rr];
#pragma unroll
                for (int j = 0; j < P + QC - 1; ++j) {
                    const int t = q0 + j;
                    w[j] = zb[(t + t / P) * M];
                }
                if ((q0 + QC) * M <= a.tp) {   // every tap of the chunk exists (uniform)
#pragma unroll
                    for (int u = 0; u < QC; ++u)
#pragma unroll
                        for (int p = 0; p < P; ++p) acc[p] = dec_fma(w[p + u], g[u], acc[p]);
                } else {                       // the last chunk: positions past the filter's end are not read (a NaN there must not reach the output).  A lane
                                               // whose tap does not exist sits the step out under the execution mask (first form: a select per multiply-add --
                                               // 256 v_cndmask next to 128 packed FMAs, the chunk cost as much as three others: profiles/r05_fir_dec_ab.json)
#pragma unroll
                    for (int u = 0; u < QC; ++u) {
                        if ((q0 + u) * M + rr < a.tp) {
#pragma unroll
                            for (int p = 0; p < P; ++p) acc[p] = dec_fma(w[p + u], g[u], acc[p]);
                        }
                        __builtin_amdgcn_sched_barrier(0);   // keeps the eight guarded groups apart (if-conversion would bring the selects back)
                    }
                }
            }
            // the M partial sums of every output meet in LDS (the lanes of a group sit in one wavefront: its DS operations execute in order)
#pragma unroll
            for (int p0 = 0; p0 < ((a.ablate & 4) ? 0 : P); p0 += PH) {
                __builtin_amdgcn_wave_barrier();
                if (lane_on) {
#pragma unroll
                    for (int p = 0; p < PH; ++p) myred[p * (Mp + 1) + r] = acc[p0 + p];
                }
                __builtin_amdgcn_wave_barrier();
                for (int pp = r; pp < PH; pp += Mp) {
                    V s = myred[pp * (Mp + 1)];
                    for (int k = 1; k < M; ++k) {
                        const V t = myred[pp * (Mp + 1) + k];
                        s.x += t.x;
                        s.y += t.y;
                    }
                    const int64_t m = m0 + (int64_t)bl * P + p0 + pp;
                    if constexpr (CPLX) {
                        if (m < a.nout) yc[m] = Y{s.x, s.y};
                    } else {
                        if (m < a.nout) yc[m] = s.x;
                        if (m + (int64_t)a.nbh * P < a.nout) yc[m + (int64_t)a.nbh * P] = s.y;
                    }
                }
            }
        }
        if constexpr (!PRE) break;   // (one tile per workgroup: straight-line code -- as a loop the Float32 form spilt 27 registers and ran 1.5x slower)
        __syncthreads();   // every group has read its last window before the next tile's samples overwrite them
    }
}

int64_t gcd64(int64_t a, int64_t b) { return std::gcd(a, b); }


// ------------------------------------------------------------------------------------------------------------
// FIRArbitrary (floating-point rate): polyphase bank + derivative bank with linear interpolation between
// neighbouring phases (stream_filt.jl:80-134, filt! :579-625).
//
// The reference advances a Float64 phase accumulator serially (update! :567-577): phiAcc += Delta; on overflow
// (dx, phiAcc) = divrem(phiAcc, Nphi), xIdx += dx; alpha = frac(phiAcc), phiIdx = 1 + trunc(phiAcc).  Every step
// rounds, so the index sequence depends on the accumulated roundings.  To stay bit-exact with it, anchors -- the exact
// (xIdx, phiAcc) of every ARB_BLK-th output -- are produced either by the same IEEE operations run serially on the host
// (arb_trajectory, short streams) or by the parallel integer evaluation of arb_scan.h on the device (long streams), cached
// per (state, length) and shared by all channels.  Wave 0 of a workgroup replays ARB_BLK updates per lane from those anchors
// (v_add_f64 / v_floor_f64 are IEEE-exact; ArbStep::fast) into an LDS record per output, then all threads evaluate
//   y = muladd(yUpper, alpha, yLower)   (:616) with yLower/yUpper the two tapsPerPhase-term
// chains (oldest sample first) over the input span staged in LDS.
// ------------------------------------------------------------------------------------------------------------
constexpr int ARB_BLK = 16;   // outputs per anchor (16 bytes each: 1 B of extra traffic per output); one lane of wave 0 replays one block

struct ArbStep {              // one update! (stream_filt.jl:567-577), shared by host and device
    double delta, nphi, inv_nphi;
    double c1, c2;            // qd nphi and (qd + 1) nphi, qd = floor(delta / nphi): the only two quotients an update can produce
    int64_t qd;
    static ArbStep make(double delta, double nphi) {
        ArbStep s{};
        s.delta = delta;
        s.nphi = nphi;
        s.inv_nphi = 1.0 / nphi;
        const double rd = std::fmod(delta, nphi);   // exact
        s.qd = (int64_t)((delta - rd) / nphi);       // exact: delta - rd is a multiple of nphi
        s.c1 = (double)s.qd * nphi;
        s.c2 = (double)(s.qd + 1) * nphi;
        return s;
    }
    // The reference's form, operation by operation.
    __host__ __device__ inline void operator()(double& acc, int64_t& xidx) const {
        acc = acc + delta;
        if (acc >= nphi) {
            if (acc < 2.0 * nphi) {
                // q = 1, the common overflow: nphi <= acc < 2 nphi makes acc - nphi exact (Sterbenz), which is what divrem
                // returns -- and keeps floor / multiply / compare-and-fix off the loop-carried chain
                acc -= nphi;
                xidx += 1;
                return;
            }
            // divrem(acc, nphi): q = floor(acc/nphi) up to one unit, r = acc - q nphi (exact), then fix q
            double q = floor(acc * inv_nphi);
            double r = acc - q * nphi;   // q nphi is an exact integer and |r| <= acc is a multiple of ulp(acc): exact with or without FMA
            if (r < 0.0) {
                q -= 1.0;
                r += nphi;
            } else if (r >= nphi) {
                q += 1.0;
                r -= nphi;
            }
            xidx += (int64_t)q;
            acc = r;
        }
    }
    // Branch-free equivalent for the device replay.  s = fl(acc + delta) lies in [delta, nphi + delta], so the quotient of
    // divrem(s, nphi) is qd or qd + 1 (none when qd = 0 and s < nphi); for a quotient q >= 1, q nphi <= s < (q + 1) nphi <= 2 q nphi
    // makes s - q nphi exact (Sterbenz) -- the exact remainder divrem returns.  Valid while qd nphi < 2^53.
    __host__ __device__ inline void fast(double& acc, int64_t& xidx) const {
        const double s = acc + delta;
        const double r1 = s - c1, r2 = s - c2;   // both candidates next to the compare: three dependent operations per update instead of four
        const bool big = s >= c2;
        acc = big ? r2 : r1;
        xidx += qd + (big ? 1 : 0);
    }
};

struct ArbArgs {
    const void* x;
    const void* hist;
    void* y;
    const void* taps2;    // tp * Nphi pairs (pfb, dpfb), taps2[i*Nphi + phi]
    const int64_t* tab_x; // 1-based xIdx of output ARB_BLK b
    const double* tab_acc;
    int64_t xlen, ldx, ldy, nout, nch;
    ArbStep step;
    int nphi, tp, hl;
    int tile;             // outputs per workgroup (multiple of ARB_BLK)
    int span;             // staged samples per tile (0: read [history ; x] straight from global/L2)
    int taps_in_lds;
    long long* prof;      // profiling only (MDSP_ARB_PROF): 8 clock64() stamps per workgroup
    int ablate;           // profiling only (MDSP_ABLATE): 1 no phase-A replay, 2 one tap instead of tp, 4 no staging loads, 8 no tap copy
    int prio;             // MDSP_ARB_PRIO: 1 = the prologue (replay, tap copy, staging) runs at raised wave priority
};

// Per-output trajectory records of a tile, written by phase A: the phase accumulator (phi = floor(acc), alpha = acc - phi: both exact, taken
// at the point of use) and xIdx relative to the tile's first output.  Two arrays (8 + 4 bytes per output instead of one 16-byte record):
// with 1185 taps x 32 phases and four interleaved channels the workgroup needs 38 KiB of LDS instead of 43 -- FOUR workgroups per CU instead
// of three, which is what this kernel's prologue / compute overlap lives on.
struct ArbRecs {
    double* acc;
    int* xrel;
};

// both arrays are written by phase A with a lane stride of ARB_BLK records: one pad record per ARB_BLK keeps those stores off a single
// bank group
__host__ __device__ constexpr int arb_rec_slot(int j) { return j + j / ARB_BLK; }
__host__ __device__ constexpr size_t arb_rec_bytes(int tile) { return ((size_t)arb_rec_slot(tile) * 12 + 15) & ~size_t(15); }

template <typename R> struct alignas(2 * sizeof(R)) Tap2 {   // (pfb, dpfb) of one (tap, phase): one 8 / 16-byte LDS read feeds both chains
    R p, d;
};

// One tile of outputs for one group of NCH channels.  `pf` is either the LDS copy of the tap pairs or the global
// table (the caller branches, so each copy of this body sees one address space and the LDS copy compiles to
// ds_read_b64); the samples of the NCH channels are interleaved in LDS, zs[k * NCH + c], so one tap-pair read and one
// NCH-wide sample read feed 2 NCH FMAs: 4 + 8 / NCH bytes of LDS traffic per tap and output instead of 12.
// ds_read_b128 is serviced in four groups of 16 NON-contiguous lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31}, and the same
// +32; MI355X_MICROARCH.md, LDS): lane l of a wave takes output ARB_B128_ORDER[l & 31] + (l & 32) of the wave's 64, so each
// group reads the windows of 16 CONSECUTIVE outputs -- consecutive 16-byte slots, no bank conflicts for rates >= 1
// (measured before: half of the kernel's LDS cycles were conflicts on this read).
__device__ const unsigned char ARB_B128_ORDER[32] = {0,  1,  2,  3,  16, 17, 18, 19, 20, 21, 22, 23, 4,  5,  6,  7,
                                                     24, 25, 26, 27, 8,  9,  10, 11, 12, 13, 14, 15, 28, 29, 30, 31};

// NPHI: the number of phases when it is known at compile time (32, the reference's default: the tap reads of an unrolled batch
// then differ by immediate offsets from one address register instead of one register and one v_add each), 0: run-time nphi
template <typename A, typename R, int NCH, int NPHI>
__device__ __forceinline__ void arb_tile_staged_n(const ArbRecs rec, const Tap2<R>* __restrict__ pf, const A* zs, A* const (&yc)[NCH], int nc, int cnt,
                                                  int tp, int nphi_rt) {
    const int nphi = NPHI ? NPHI : nphi_rt;
    struct alignas(sizeof(A) * NCH <= 16 ? sizeof(A) * NCH : 16) ZV {
        A v[NCH];
    };
    const ZV* zv = reinterpret_cast<const ZV*>(zs);
    int j0 = threadIdx.x;
    if constexpr (sizeof(ZV) == 16) j0 = (j0 & ~31) + ARB_B128_ORDER[j0 & 31];
    for (int j = j0; j < cnt; j += blockDim.x) {
        const double racc = rec.acc[arb_rec_slot(j)];
        const double rfl = floor(racc);
        const double ralpha = racc - rfl;
        const Tap2<R>* hq = pf + (int)rfl;
        const ZV* zp = zv + rec.xrel[arb_rec_slot(j)];
        A lo[NCH], up[NCH];
        {
            const Tap2<R> t = *hq;
            const ZV z = zp[0];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                lo[c] = mul_first(t.p, z.v[c]);
                up[c] = mul_first(t.d, z.v[c]);
            }
        }
        // taps 1 .. tp-1 in batches of 8 (sixteen LDS reads in flight, then their FMAs in tap order), the remainder as one batch of 4, 2 and 1
        // (wave-uniform branches) instead of single taps that each wait for their own two reads
        auto batch = [&](auto n, int i0) __attribute__((always_inline)) {
            constexpr int N = decltype(n)::value;
            Tap2<R> t[N];
            ZV z[N];
#pragma unroll
            for (int r = 0; r < N; ++r) {
                t[r] = hq[(i0 + r) * nphi];
                z[r] = zp[i0 + r];
            }
#pragma unroll
            for (int r = 0; r < N; ++r) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    fma_acc(lo[c], t[r].p, z[r].v[c]);
                    fma_acc(up[c], t[r].d, z[r].v[c]);
                }
            }
        };
        int i = 1;
        for (; i + 8 <= tp; i += 8) batch(std::integral_constant<int, 8>{}, i);
        if ((tp - i) & 4) {
            batch(std::integral_constant<int, 4>{}, i);
            i += 4;
        }
        if ((tp - i) & 2) {
            batch(std::integral_constant<int, 2>{}, i);
            i += 2;
        }
        if ((tp - i) & 1) batch(std::integral_constant<int, 1>{}, i);
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (c < nc) yc[c][j] = arb_combine(up[c], ralpha, lo[c]);
    }
}
template <typename A, typename R, int NCH>
__device__ __forceinline__ void arb_tile_staged(const ArbRecs rec, const Tap2<R>* __restrict__ pf, const A* zs, A* const (&yc)[NCH], int nc, int cnt,
                                                int tp, int nphi) {
    if (nphi == 32) arb_tile_staged_n<A, R, NCH, 32>(rec, pf, zs, yc, nc, cnt, tp, nphi);
    else arb_tile_staged_n<A, R, NCH, 0>(rec, pf, zs, yc, nc, cnt, tp, nphi);
}

// (the second launch bound: four workgroups per CU -- what the 38 KiB of LDS of the Float32 four-channel form admit -- need <= 128 VGPRs)
template <typename XS, typename A, typename R, int NCH>
MDSP_NO_LSO __global__ __launch_bounds__(256, (sizeof(A) * NCH <= 16 ? 4 : 1)) void arbitrary_fir_kernel(ArbArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const ArbRecs rec{reinterpret_cast<double*>(smem), reinterpret_cast<int*>(smem + (size_t)arb_rec_slot(a.tile) * sizeof(double))};
    A* zs = reinterpret_cast<A*>(smem + arb_rec_bytes(a.tile));
    Tap2<R>* ps = reinterpret_cast<Tap2<R>*>(smem + arb_rec_bytes(a.tile) + (size_t)a.span * NCH * sizeof(A));
    const int64_t m0 = (int64_t)blockIdx.x * a.tile;
    if (m0 >= a.nout) return;
    const int cnt = (int)std::min<int64_t>(a.tile, a.nout - m0);
    const int64_t b0 = m0 / ARB_BLK;
    const int tid = threadIdx.x;
    auto stamp = [&](int k) {
        if (a.prof && tid == 0) a.prof[(int64_t)blockIdx.x * 8 + k] = clock64();
    };
    if ((a.prio & 1) || ((a.prio & 2) && tid < 64)) __builtin_amdgcn_s_setprio(3);   // a workgroup in its latency-bound prologue (1), or only its replaying wave (2), goes first among the co-resident ones
    stamp(0);
    const int64_t x_first = a.tab_x[b0];
    const int64_t z_first = x_first - 1;                                    // z = [history ; x], output n reads z[n-1 .. n-1+tp)
    if (a.prof && tid == 0) a.prof[(int64_t)blockIdx.x * 8 + 1] = clock64() + (x_first & 0);
    auto zload = [&](int64_t ch, int64_t zi) -> A {   // branch-free: the loads of a staging batch all issue before the first wait
        const XS* xc = static_cast<const XS*>(a.x) + ch * a.ldx;
        const XS* hc = static_cast<const XS*>(a.hist) + ch * (int64_t)a.hl;
        const int64_t xi = zi - a.hl;
        const bool in_hist = zi < a.hl, ok = in_hist || xi < a.xlen;
        const XS* p = in_hist ? hc + zi : xc + (xi < a.xlen ? xi : 0);
        const A v = to_acc(*p, (A*)nullptr);
        return ok ? v : A{};
    };
    // stage z[z_first .. z_first + count) of the channel group at c0 into zs, interleaved; `lanes` threads starting at `first`.  Loads and
    // stores of a round are separate steps so that the prologue can put the tap copy's loads in front of the same wait.
    constexpr int RB = sizeof(A) * NCH <= 16 ? 6 : 4;   // RB * NCH independent loads in flight per thread (six rounds of 192 threads cover a 1024-output tile's span at rates >= 1)
    // a tile whose whole span lies inside x (every tile but the first and the last few of a call) loads through one uniform base pointer per
    // channel: the history / end-of-input selects of zload cost a dozen vector instructions per load, which is most of what the staging
    // waves issue -- and they issue in competition with three computing workgroups
    const bool interior = z_first >= a.hl && z_first + a.span <= (int64_t)a.hl + a.xlen;
    auto stage_loads = [&](A(&v)[RB][NCH], int64_t c0, int nc, int k0, int lanes, int count) __attribute__((always_inline)) {
        if (interior) {
            const XS* xb[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) xb[c] = static_cast<const XS*>(a.x) + (c0 + (c < nc ? c : 0)) * a.ldx + (z_first - a.hl);
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int k = min(k0 + r * lanes, count - 1);
#pragma unroll
                for (int c = 0; c < NCH; ++c) v[r][c] = to_acc(xb[c][k], (A*)nullptr);
            }
        } else {
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int k = k0 + r * lanes;
                const int64_t zi = z_first + (k < count ? k : count - 1);
#pragma unroll
                for (int c = 0; c < NCH; ++c) v[r][c] = zload(c0 + (c < nc ? c : 0), zi);
            }
        }
    };
    auto stage_stores = [&](const A(&v)[RB][NCH], int nc, int k0, int lanes, int count) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int k = k0 + r * lanes;
            if (k < count) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) zs[k * NCH + c] = c < nc ? v[r][c] : A{};
            }
        }
    };
    auto stage = [&](int64_t c0, int nc, int first, int lanes, int count) __attribute__((always_inline)) {
        for (int k0 = first; k0 < count; k0 += RB * lanes) {
            A v[RB][NCH];
            stage_loads(v, c0, nc, k0, lanes, count);
            stage_stores(v, nc, k0, lanes, count);
        }
    };
    const Tap2<R>* pg = static_cast<const Tap2<R>*>(a.taps2);
    const int64_t c_first = (int64_t)blockIdx.y * NCH;
    // Prologue, wave-specialised so that its two latency chains overlap: wave 0 replays the recurrence from the anchors (phase A,
    // once per tile, shared by every channel; one lane per ARB_BLK outputs), waves 1..3 meanwhile copy the tap pairs and stage the
    // first channel group's samples (the whole span: its exact extent is only known after phase A).
    if (tid < 64) {
        if (tid * ARB_BLK < cnt) {
            int64_t xi = a.tab_x[b0 + tid];
            double acc = a.tab_acc[b0 + tid];
            const int base = tid * ARB_BLK;
            const int n = min(ARB_BLK, cnt - base);
            for (int k = 0; k < n; ++k) {
                rec.acc[arb_rec_slot(base + k)] = acc;
                rec.xrel[arb_rec_slot(base + k)] = (int)(xi - x_first);
                if (!MDSP_ABLATED(a, 1)) a.step.fast(acc, xi);
            }
        }
        stamp(2);
    } else {
        const int lanes = (int)blockDim.x - 64, u = tid - 64;
        // the tap pairs travel as 16-byte units (two Float32 pairs, one Float64 pair); round 0 of the tap copy and round 0 of the staging
        // issue ALL their loads before the first wait -- one memory latency for the tap table and the whole span of a 1024-output tile
        constexpr int RT = 4;
        const bool do_taps = a.taps_in_lds && !MDSP_ABLATED(a, 8);
        const bool do_stage = a.span > 0 && c_first < a.nch && !MDSP_ABLATED(a, 4);
        const int np = a.tp * a.nphi;
        const int nu = do_taps ? (int)((size_t)np * sizeof(Tap2<R>) / 16) : 0;
        const int nc0 = (int)std::min<int64_t>(NCH, a.nch - c_first);
        const uint4* pg16 = reinterpret_cast<const uint4*>(pg);
        uint4* ps16 = reinterpret_cast<uint4*>(ps);
        uint4 tv[RT];
        A sv[RB][NCH];
#pragma unroll
        for (int r = 0; r < RT; ++r) tv[r] = pg16[min(u + r * lanes, max(nu, 1) - 1)];   // unconditional (the table always exists): no private copy of tv
        stage_loads(sv, do_stage ? c_first : 0, do_stage ? nc0 : 1, u, lanes, max(a.span, 1));
        if (do_taps) {
#pragma unroll
            for (int r = 0; r < RT; ++r)
                if (u + r * lanes < nu) ps16[u + r * lanes] = tv[r];
        }
        if (do_stage) stage_stores(sv, nc0, u, lanes, a.span);
        if (do_taps) {
            for (int k0 = u + RT * lanes; k0 < nu; k0 += RT * lanes) {
#pragma unroll
                for (int r = 0; r < RT; ++r) tv[r] = pg16[min(k0 + r * lanes, nu - 1)];
#pragma unroll
                for (int r = 0; r < RT; ++r)
                    if (k0 + r * lanes < nu) ps16[k0 + r * lanes] = tv[r];
            }
            if constexpr (sizeof(Tap2<R>) == 8) {
                if ((np & 1) && u == 0) ps[np - 1] = pg[np - 1];   // an odd number of 8-byte pairs: the last one is not part of a 16-byte unit
            }
        }
        if (do_stage) stage(c_first, nc0, u + RB * lanes, lanes, a.span);
    }
    __syncthreads();
    if (a.prio) __builtin_amdgcn_s_setprio(0);
    stamp(3);
    const int64_t nz = (int64_t)rec.xrel[arb_rec_slot(cnt - 1)] + a.tp;
    const bool staged = nz <= a.span;                                       // workgroup-uniform
    for (int64_t c0 = c_first; c0 < a.nch; c0 += (int64_t)gridDim.y * NCH) {
        const int nc = (int)std::min<int64_t>(NCH, a.nch - c0);
        if (staged) {
            if (c0 != c_first) {
                __syncthreads();   // the previous channel group's readers are done with zs
                stage(c0, nc, tid, (int)blockDim.x, (int)nz);
                __syncthreads();
            }
            A* yc[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) yc[c] = static_cast<A*>(a.y) + (c0 + (c < nc ? c : 0)) * a.ldy + m0;
            const int tpe = MDSP_ABLATED(a, 2) ? 1 : a.tp;
            if (a.taps_in_lds) arb_tile_staged<A, R, NCH>(rec, ps, zs, yc, nc, cnt, tpe, a.nphi);
            else arb_tile_staged<A, R, NCH>(rec, pg, zs, yc, nc, cnt, tpe, a.nphi);
            stamp(4);
        } else {
            const Tap2<R>* pf = a.taps_in_lds ? ps : pg;
            for (int c = 0; c < nc; ++c) {
                A* yc = static_cast<A*>(a.y) + (c0 + c) * a.ldy;
                for (int j = threadIdx.x; j < cnt; j += blockDim.x) {
                    const double racc = rec.acc[arb_rec_slot(j)];
                    const double rfl = floor(racc);                                 // alpha = modf(acc)[1] is exact
                    const Tap2<R>* hp = pf + (int)rfl;
                    const int64_t z0 = z_first + rec.xrel[arb_rec_slot(j)];
                    A z = zload(c0 + c, z0);
                    Tap2<R> t = hp[0];
                    A lo = mul_first(t.p, z);
                    A up = mul_first(t.d, z);
                    for (int i = 1; i < a.tp; ++i) {
                        z = zload(c0 + c, z0 + i);
                        t = hp[(int64_t)i * a.nphi];
                        fma_acc(lo, t.p, z);
                        fma_acc(up, t.d, z);
                    }
                    yc[m0 + j] = arb_combine(up, racc - rfl, lo);
                }
            }
        }
    }
    __syncthreads();
    stamp(5);
}

// ------------------------------------------------------------------------------------------------------------
// Stateful time-domain FIR: DF2TFilter{PolynomialRatio} with a = [1] (Filters/filt.jl:153-181) advanced by
// _filt_fir! (dspbase.jl:95-105).  The state si[0..nb-2] is the TDF-II register file, NOT an input history:
//     y[n]   = fma(x[n], b0, fma(x[n-1], b1, ... start))            start = si0[n]                (n <  nb-1)
//                                                                       = b[nb-1] * x[n-nb+1]      (n >= nb-1)
//     si'[j] = fma(x[N-1], b[j+1], fma(x[N-2], b[j+2], ... start))  start = si0[j+N]              (N <= nb-2-j)
//                                                                       = b[nb-1] * x[N-1-(nb-2-j)] otherwise
// -- the same innermost-first accumulation order the serial recursion produces, so outputs and state agree with
// the reference to the last fused-multiply-add.
// ------------------------------------------------------------------------------------------------------------
template <typename A, typename R>
MDSP_NO_LSO __global__ __launch_bounds__(256) void tdfir_state_out_kernel(const A* __restrict__ x, const A* __restrict__ si0, A* __restrict__ y, const R* __restrict__ b,
                                                              int64_t nx, int64_t ldx, int64_t ldy, int nb) {
    const int64_t col = blockIdx.y;
    const A* xc = x + col * ldx;
    const A* sc = si0 + col * (int64_t)(nb - 1);
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < nx; n += (int64_t)gridDim.x * blockDim.x) {
        A acc;
        int k;
        if (n < nb - 1) {
            acc = sc[n];
            k = (int)n;
        } else {
            acc = mul_first(b[nb - 1], xc[n - (nb - 1)]);
            k = nb - 2;
        }
        for (; k >= 0; --k) fma_acc(acc, b[k], xc[n - k]);
        y[col * ldy + n] = acc;
    }
}

// one workgroup per column; every thread first computes its new registers (reading the OLD state), then all write
template <typename A, typename R>
MDSP_NO_LSO __global__ __launch_bounds__(256) void tdfir_state_next_kernel(const A* __restrict__ x, A* __restrict__ si, const R* __restrict__ b, int64_t N, int64_t ldx,
                                                               int nb) {
    constexpr int MAXPER = 16;   // nb - 1 <= 256 * MAXPER
    const int64_t col = blockIdx.x;
    const A* xc = x + col * ldx;
    A* sc = si + col * (int64_t)(nb - 1);
    A res[MAXPER];
#pragma unroll
    for (int q = 0; q < MAXPER; ++q) {
        const int j = threadIdx.x + 256 * q;
        if (j >= nb - 1) break;
        const int span = nb - 2 - j;             // largest m with b[j+1+m] defined
        A acc;
        int64_t m;
        if (N <= span) {
            acc = sc[j + N];
            m = N - 1;
        } else {
            acc = mul_first(b[nb - 1], xc[N - 1 - span]);
            m = span - 1;
        }
        for (; m >= 0; --m) fma_acc(acc, b[j + 1 + m], xc[N - 1 - m]);
        res[q] = acc;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < MAXPER; ++q) {
        const int j = threadIdx.x + 256 * q;
        if (j >= nb - 1) break;
        sc[j] = res[q];
    }
}

// extrapolate_signal! (Filters/filt.jl:243-257): out = [2 x[1] .- x[pad+1:-1:2]; x; 2 x[end] .- x[end-1:-1:end-pad]]
template <typename A>
MDSP_NO_LSO __global__ __launch_bounds__(256) void extrapolate_kernel(const A* __restrict__ x, A* __restrict__ out, int64_t n, int64_t ldx, int64_t ldo, int64_t pad) {
    const int64_t col = blockIdx.y;
    const A* xc = x + col * ldx;
    A* oc = out + col * ldo;
    const int64_t total = n + 2 * pad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        A v;
        if (i < pad) v = sub2(xc[0], xc[pad - i]);                     // 1-based: out[i] = 2 sig[1] - sig[2 + pad - i]
        else if (i < pad + n) v = xc[i - pad];
        else v = sub2(xc[n - 1], xc[n - 1 - (i - pad - n + 1)]);     // out[n + pad + i'] = 2 sig[n] - sig[n - i']
        oc[i] = v;
    }
}

}  // namespace

struct mdsp_fir_s {
    int kind = 0;  // 0 standard, 1 interpolator, 2 decimator, 3 rational
    int64_t L = 1, M = 1, hlen = 0, tp = 0, hl = 0, nch = 1;
    int taps_dtype = MDSP_F32, x_dtype = MDSP_F32, out_dtype = MDSP_F32;
    bool acc_double = false;
    int64_t phi_idx = 1, input_deficit = 1;  // the reference's 1-based state
    DevBuf pfbT;
    DevBuf hist[2];
    int cur = 0;
    bool exact = false;   // mdsp_fir_set_exact: the generic kernel only -- every output reads exactly its own tapsPerPhi-sample window (stream_filt.jl:496-509)
};


struct mdsp_firarb_s {
    mdsp_fir_s base;          // history buffers, dtypes, tp / hl / nch (kind = 4)
    double rate = 1.0, delta = 0.0;
    int64_t nphi = 32;
    // the reference's state (stream_filt.jl:96-104); phi_idx and alpha are functions of phi_acc
    double phi_acc = 0.0;
    int64_t input_deficit = 1, x_idx = 1;
    DevBuf tab_x, tab_acc;
    DevBuf prof;   // MDSP_ARB_PROF: per-workgroup clock stamps of the filtering kernel
    DevBuf scan_t0, scan_wide, scan_E, scan_mind, scan_cb, scan_res;   // work space of the parallel trajectory scan (arb_scan.h)
    // trajectory cache: anchors of the last (phi_acc, input_deficit, xlen) evaluated
    bool cache_valid = false;
    double c_acc0 = 0.0;
    int64_t c_def0 = 0, c_xlen = -1, c_nout = 0, c_def_end = 0, c_xidx_end = 0;
    double c_acc_end = 0.0;
    int64_t n_scanned = 0, n_serial = 0;   // trajectories evaluated by the device scan / the serial host loop
};

namespace {

template <typename XS, typename A, typename R> int fir_launch(mdsp_fir_s* f, FirArgs& a, hipStream_t st) {
    // tile size: keep the staged span + filter bank within 64 KiB of LDS
    const int64_t pfb_bytes = (int64_t)f->tp * f->L * (int64_t)sizeof(R);
    a.pfb_in_lds = pfb_bytes <= 40 * 1024;
    int tile = 4096;
    int64_t span = 0;
    while (true) {
        span = ((int64_t)tile * f->M + f->L - 1) / f->L + f->tp + 2;
        const int64_t bytes = span * (int64_t)sizeof(A) + (a.pfb_in_lds ? pfb_bytes : 0);
        if (bytes <= 64 * 1024 || tile <= 64) break;
        tile /= 2;
    }
    const int64_t lds_bytes = span * (int64_t)sizeof(A) + (a.pfb_in_lds ? pfb_bytes : 0);
    if (lds_bytes > 150 * 1024) MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "filter too long for the polyphase kernel (tapsPerPhase=%lld)", (long long)f->tp);
    a.tile = tile;
    a.span = (int)span;
    auto kern = polyphase_fir_kernel<XS, A, R>;
    if (lds_bytes > 48 * 1024) MDSP_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    const dim3 grid((unsigned)cdiv(a.nout, tile), (unsigned)f->nch);
    hipLaunchKernelGGL(kern, grid, dim3(256), (size_t)lds_bytes, st, a);
    MDSP_LAUNCH_CHECK();
    return MDSP_OK;
}

// Which phase group each thread of the fast kernel owns.  A thread's LDS reads are at (round) M + c_g + j, c_g = (phi0-1 + g P M)
// div L: with groups in thread order, the 32 lanes of a ds_read_b32 group span ~32 P M / L addresses and collide two ways
// whenever that exceeds 32 (160//147: 1.88 LDS cycles per access instead of 1, 45 % of the kernel's LDS cycles measured as
// bank conflicts).  Any assignment of groups to lanes is legal (the taps live in registers, stores stay inside the same
// L-sample row), so the host picks, per 32-lane group, phase groups with distinct banks (greedy), keeps the identity when
// that is not better, and passes the table in the kernel arguments.
template <int P> void fir_lane_map(FirFastArgs& b) {
    const int NP = b.NP, nthreads = b.NP * b.RL;
    std::vector<int> c0(NP), ident(nthreads), greedy(nthreads);
    for (int g = 0; g < NP; ++g) c0[g] = (int)((b.phi0m1 + (int64_t)g * P * b.M) / b.L);
    const auto cycles = [&](const std::vector<int>& tab) {
        int total = 0;
        for (int w0 = 0; w0 < nthreads; w0 += 32) {
            int cnt[32] = {0}, addr[32][32];
            int worst = 1;
            for (int t = w0; t < std::min(w0 + 32, nthreads); ++t) {
                const int a = (t / NP) * b.M + c0[tab[t]], bank = a & 31;
                bool seen = false;
                for (int i = 0; i < cnt[bank]; ++i) seen = seen || addr[bank][i] == a;
                if (!seen) addr[bank][cnt[bank]++] = a;
                worst = std::max(worst, cnt[bank]);
            }
            total += worst;
        }
        return total;
    };
    std::vector<std::vector<char>> used(b.RL, std::vector<char>(NP, 0));
    for (int t = 0; t < nthreads; ++t) ident[t] = t % NP;
    for (int w0 = 0; w0 < nthreads; w0 += 32) {
        bool bank_used[32] = {false};
        for (int t = w0; t < std::min(w0 + 32, nthreads); ++t) {
            const int r = t / NP;
            int pick = -1, any = -1;
            for (int g = 0; g < NP && pick < 0; ++g) {
                if (used[r][g]) continue;
                if (any < 0) any = g;
                if (!bank_used[(r * b.M + c0[g]) & 31]) pick = g;
            }
            if (pick < 0) pick = any;
            used[r][pick] = 1;
            bank_used[(r * b.M + c0[pick]) & 31] = true;
            greedy[t] = pick;
        }
    }
    const std::vector<int>& best = cycles(greedy) < cycles(ident) && !MDSP_DBG(fir_identity_lanes) ? greedy : ident;
    for (int t = 0; t < 256; ++t) b.gmap[t] = (unsigned char)(t < nthreads ? best[t] : 0);
}

template <int TPC, int P> int fir_fast_launch(mdsp_fir_s* f, const FirArgs& a, hipStream_t st) {
    FirFastArgs b{};
    b.x = (const float*)a.x;
    b.hist = (const float*)a.hist;
    b.y = (float*)a.y;
    b.pfbT = (const float*)a.pfbT;
    b.xlen = a.xlen; b.ldx = a.ldx; b.ldy = a.ldy; b.nout = a.nout;
    b.phi0m1 = a.phi0m1; b.d0 = a.d0;
    b.L = a.L; b.M = a.M; b.tp = a.tp; b.hl = a.hl;
    b.nrounds = cdiv(a.nout, (int64_t)a.L);
    b.NP = (int)cdiv(a.L, P);
    b.RL = std::max(1, 256 / b.NP);
    fir_lane_map<P>(b);
    // rounds per tile: LDS budget ~48 KiB, at least RL rounds
    const int lds_kib = tunables().fir_lds_kib;
    int Q = (int)std::max<int64_t>(b.RL, ((int64_t)lds_kib * 1024 / 4 - a.M - (TPC + P)) / std::max(1, a.M));
    Q = std::min(Q, 512);
    Q = (int)std::min<int64_t>(Q, std::max<int64_t>(b.nrounds, 1));
    b.Q = Q;
    b.span = Q * a.M + a.M + TPC + P;
    const size_t lds_bytes = (size_t)b.span * sizeof(float);
    auto kern = polyphase_fast_kernel<TPC, P>;
    if (lds_bytes > 48 * 1024) MDSP_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    const int64_t ntiles = cdiv(b.nrounds, (int64_t)Q);
    int wgs = P >= 4 ? 2 : (P == 3 ? 3 : 5);   // resident workgroups per CU (P = 2: 94 VGPRs x 4 waves, ~20 KiB LDS; 8 / 5 / 4 measured 2.64 / 2.59 / 2.67 ms on config 5; P = 4: 228 VGPRs)
    if (tunables().wg_per_cu > 0) wgs = tunables().wg_per_cu;
    const int64_t per = std::max<int64_t>(1, (int64_t)device_cu_count() * wgs / std::max<int64_t>(1, f->nch));
    const dim3 grid((unsigned)std::min<int64_t>(ntiles, per), (unsigned)f->nch);
    hipLaunchKernelGGL(kern, grid, dim3(b.NP * b.RL), lds_bytes, st, b);
    MDSP_LAUNCH_CHECK();
    return MDSP_OK;
}

// Fast path applies to Float32 taps x real Float32 signal with <= 64 taps per phase and <= 1024 phase groups.
bool fir_fast_ok(const mdsp_fir_s* f, int P) {
    if (f->acc_double || f->x_dtype != MDSP_F32 || f->tp > 64) return false;
    if (MDSP_DBG(fir_generic) || tunables().fir_exact || f->exact) return false;
    if (P >= 2 && (f->M > f->L || f->L < P)) return false;
    return cdiv(f->L, P) <= 256;
}

template <int P> int fir_fast_dispatch(mdsp_fir_s* f, const FirArgs& a, hipStream_t st) {
    const int tpc = (int)((f->tp + 7) / 8 * 8);
    switch (tpc) {
        case 8: return fir_fast_launch<8, P>(f, a, st);
        case 16: return fir_fast_launch<16, P>(f, a, st);
        case 24: return fir_fast_launch<24, P>(f, a, st);
        case 32: return fir_fast_launch<32, P>(f, a, st);
        case 40: return fir_fast_launch<40, P>(f, a, st);
        case 48: return fir_fast_launch<48, P>(f, a, st);
        case 56: return fir_fast_launch<56, P>(f, a, st);
        default: return fir_fast_launch<64, P>(f, a, st);
    }
}

// ---- matrix-core kernel, host side ------------------------------------------------------------------------------
// Shapes it is built for: Float32 taps x Float32 / ComplexF32 signals, and Float64 arithmetic on Float64 / ComplexF64 signals; L <= 1024,
// a tile that fits the LDS.  Up to 12 column blocks (L <= 192) and 256 (Float64: 128) window positions per block the taps of a wave's block
// live in registers; beyond either, a wave fetches the taps of its block(s) from L2 for every tile.
// For L < 16 a row of the product is RB whole rounds (Lr = RB L <= 16 consecutive outputs, Mr = RB M samples): the columns of a
// row still repeat their phases from row to row, which is all the kernel needs; RB is chosen so that rows (lane stride Mr samples)
// spread over the LDS banks (odd Mr: conflict-free).
struct FirMGeo {
    bool ok = false;
    int esz = 4, CS = 1, CH = 4;   // bytes of R; parts per sample; 16-row chunks per multiplying wave
    int RB = 1, Lr = 0, Mr = 0, NB = 0, NBW = 0, NG = 1, T = 0, steps = 0, Lp = 0, nd = 1, ns = 1;   // T = 0: `steps` k-steps with the taps fetched per tile
    int NBLK = 1;                  // column blocks per multiplying wave whose taps ALL live in registers (L > 192 with T = 12 / 16; else 1)
    int64_t bufsz = 0;             // dwords per sample buffer
    int pitch = 0;                 // dwords between separately staged rows (0: one linear run per tile)
    int rowpad = 0;                // dwords of padding behind every 256-dword granule of a linear run (0: none)
    size_t lds_bytes = 0;
};
int fir_mm_tsel(int64_t steps) { return steps <= 4 ? 4 : steps <= 8 ? 8 : steps <= 12 ? 12 : steps <= 16 ? 16 : steps <= 20 ? 20 : steps <= 24 ? 24 : steps <= 32 ? 32 : steps <= 48 ? 48 : 64; }
// tight (round 4, only tried where nothing else fits the LDS -- shapes that would otherwise fall to the generic kernel at 0.03 - 0.07 of the roofline):
//   rb_cap > 0: rows of at most rb_cap rounds for L < 16 (ComplexF64 1//16: 14 rounds of 16 samples fit where the conflict-best 15 miss by 1 KiB);
//   Float32 windows of 33 - 40 k-steps take the 40-step register form with single-chunk waves instead of 48 (ComplexF32 160//441: the 32 samples
//   less of window tail per buffer are what the tile misses); ComplexF64 windows of 13 - 14 k-steps take a 14-step form (round 5).
FirMGeo fir_mm_geo_compute(const mdsp_fir_s* f, bool allow_regs, int rb_cap = 0) {   // allow_regs: round 3's register-tap rules (fewer chunks / shorter rows)
    FirMGeo g;
    const bool t64 = allow_regs && tunables().fir_mm_t64 != 0;
    // element type: signal and compute type must agree (Float32 taps x Float32 samples, or Float64 arithmetic on Float64 samples)
    if (f->acc_double != dtype_is_double(f->x_dtype)) return g;
    g.esz = f->acc_double ? 8 : 4;
    g.CS = dtype_is_complex(f->x_dtype) ? 2 : 1;
    int chmax = (g.esz == 4 && g.CS == 1) ? 4 : 2;   // 16-row chunks per multiplying wave: fewer when the tile would not fit the LDS
    if (tunables().fir_mm_ch > 0) chmax = std::min(chmax, tunables().fir_mm_ch);
    if (f->L > 1024 || f->M > 4096) return g;
    if (f->L < 16) {   // rounds per row: most outputs per LDS cycle of the A-operand reads (16 rows x 2 taps per lane group)
        double best = -1;
        for (int rb = 1; rb * f->L <= 16 && (rb_cap <= 0 || rb <= rb_cap); ++rb) {
            const int64_t mr = rb * f->M;
            int worst = 1;
            if (!(mr & 1)) {   // odd strides are conflict-free with the even / odd row order of the kernel
                int cnt[32] = {0};
                for (int i = 0; i < 16; ++i)
                    for (int k = 0; k < 2; ++k) worst = std::max(worst, ++cnt[(int)((i * mr + k) & 31)]);
            }
            double score = (double)(rb * f->L) / worst + 1e-3 * rb;
            // rows short enough for the taps to stay in registers (single-chunk waves) run up to twice as fast per k-step as rows whose taps are
            // fetched per tile (profiles/r03o_fir_register_taps.json): 1//8 with 293 taps takes 11 outputs per row (94 k-steps) instead of 15 (104)
            if (t64) {
                const int64_t st = cdiv(f->tp + ((f->L - 1) + (int64_t)(std::min<int64_t>(rb * f->L, 16) - 1) * f->M) / f->L, (int64_t)4);
                const int64_t treg = g.esz == 4 ? 96 : (g.CS == 1 ? 48 : 40);
                if (st <= treg) score *= 1.6;   // (2.0 takes 9 of 16 columns for ComplexF64 3//8: measured 14 % slower than fetching with 15)
            }
            if (score > best) { best = score; g.RB = rb; }
        }
    }
    g.Lr = g.RB * (int)f->L;
    g.Mr = g.RB * (int)f->M;
    g.NB = (int)cdiv((int64_t)g.Lr, (int64_t)16);
    const int64_t steps = cdiv(f->tp + ((f->L - 1) + (int64_t)(std::min(g.Lr, 16) - 1) * f->M) / f->L, (int64_t)4);   // tp + max delta within a block
    if (steps > 1024 || (int64_t)(f->L + (int64_t)(g.Lr + 16) * f->M) * f->L >= ((int64_t)1 << 32)) return g;   // (the multiply-high quotients stay exact)
    g.NBW = std::min(g.NB, 12);   // more than 12 column blocks (L > 192): a wave takes several, with their taps fetched per tile
    if (g.NBW < g.NB) g.NBW = (int)cdiv((int64_t)g.NB, cdiv((int64_t)g.NB, (int64_t)12));   // as even as it gets
    // Taps in registers whenever SOME chunk count admits it (round 3): a wave that carries fewer chunks of 16 rows has fewer accumulators and
    // room for more taps -- Float32: 48 k-steps with four chunks, 64 with two, 96 with one (ComplexF32 64 / 96); Float64 32 / 48 (ComplexF64
    // 24 / 40).  Fetching the taps per tile instead (T = 0) costs far more than the smaller tile: 1//4 (56 steps) 0.67 -> 0.46 ms with two
    // chunks, profiles/r03o_fir_register_taps.json.  MDSP_FIR_MM_T64=0: round 2's limits.
    const auto tmax_of = [&](int ch) {
        if (g.esz == 4) return g.CS == 1 ? (ch >= 4 ? 48 : ch == 2 ? 64 : 96) : (ch >= 2 ? 64 : 96);   // (112 steps spill 43 registers: slower than fetching)
        return g.CS == 1 ? (ch >= 2 ? 32 : 48) : (ch >= 2 ? 24 : 40);
    };
    bool regs = false;
    if (g.NBW == g.NB && t64 && steps > tmax_of(chmax)) {
        for (int ch = chmax / 2; ch >= 1 && !regs; ch /= 2)
            if (steps <= tmax_of(ch)) {
                chmax = ch;
                regs = true;
            }
    }
    // L > 192 (more than 12 column blocks): a wave walks two or three blocks.  With short windows (12 / 16 k-steps: the 32 taps per phase of the
    // default resampling filters) the taps of ALL its blocks fit registers -- single-chunk waves -- instead of being fetched per tile.
    const int nblk = (int)cdiv((int64_t)g.NB, (int64_t)g.NBW);
    if (g.NBW < g.NB && t64 && nblk <= 3 && steps <= 16 && tunables().fir_mm_nblk != 0) {
        chmax = g.esz == 4 ? std::min(chmax, 2) : 1;   // (Float32: two chunks of 16 rows per wave still leave room for three blocks' taps)
        g.NBLK = nblk;
        g.T = steps <= 12 ? 12 : 16;
        g.steps = g.T;
    } else
    if (regs) {
        g.T = (g.esz == 8 && steps > 32 && steps <= 40) ? 40 : steps <= 64 ? fir_mm_tsel(steps) : steps <= 80 ? 80 : 96;   // (Float64: 40 between 32 and 48 -- a shorter window tail)
        g.steps = g.T;
    } else
    if (steps > (g.esz == 8 ? (g.CS == 2 ? 24 : 32) : (g.CS == 2 ? 64 : 48)) || g.NBW < g.NB) {   // (beyond: too many registers -- measured: ComplexF64 at T = 32 and Float32 at T = 64 spill and lose 40 - 50 %)   // the taps do not fit registers (T, Float64 2 T, VGPRs): fetched per tile
        g.T = 0;
        g.steps = (int)cdiv(steps, (int64_t)8) * 8;
    } else
    if (rb_cap < 0 && g.esz == 4 && steps > 32 && steps <= 40) {
        g.T = 40;
        g.steps = 40;
        chmax = 1;
    } else
    if (rb_cap < 0 && g.esz == 8 && g.CS == 2 && steps > 12 && steps <= 14) {   // (round 5) ComplexF64 160//147 with resample_filter's 5921 taps (13 k-steps): the 16-step
        g.T = 14;                                                                  // form's window tail misses the LDS by 2 KiB and the cell fell to the generic kernel (0.125)
        g.steps = 14;
        chmax = 1;
    } else {
        g.T = fir_mm_tsel(steps);
        g.steps = g.T;
    }
    g.Lp = g.NB == 1 ? g.Lr * g.CS : 16 * g.NB * g.CS + 16 / g.esz;
    const int dw = g.esz / 4 * g.CS;
    // Rows (lane stride Mr samples) of a linear tile hit banks / gcd(Mr dw, banks) places.  From four-way conflicts on, staging the rows
    // one by one (pitch 256 g + 4 dwords: two-way conflicts in Float32, none in Float64) is the alternative -- unless the window tail it
    // duplicates per row (long filters) shrinks the tile too much.
    const int wtail = 4 * g.steps + 4;
    // conflict ways of the 16 rows of an A operand: 4-byte reads see 32 banks, 8-byte reads 64; rows s dwords apart land on banks / gcd(s, banks) places
    const int banks = g.esz == 8 ? 64 : 32;
    const auto ways_of = [&](int64_t s) { return (double)std::max<int64_t>(1, 16 / std::min<int64_t>(16, banks / std::gcd(s, (int64_t)banks))); };
    const int rpitch = (int)cdiv((int64_t)(g.Mr + wtail) * dw, (int64_t)256) * 256 + 4;
    const double ways_lin = ((g.Mr * dw) & 1) ? 1.0 : ways_of((int64_t)g.Mr * dw), ways_row = ways_of(rpitch);
    // cost per row of outputs ~ max(1, 0.2 ways) [the A-operand reads keep the LDS about 20 % busy when conflict-free] + 32 / rows [one
    // barrier and pipeline turn-around per tile, worth about 32 rows]: the staging mode and tile with the smallest cost
    // Two ways to pick the tile (16 CH rows per multiplying wave, NG groups of NBW waves):
    //   * ratios above one (and anything with several column blocks per wave): the largest tile that fits, scored by
    //     max(1, 0.2 ways) + 32 / rows as in round 2 -- with at most FOUR groups when there is one column block (2//1, 3//2, 4//1 ...):
    //     4 + 2 + 2 waves put one multiplying and one memory wave of a workgroup on every SIMD and three such workgroups share a CU
    //     (2//1 0.95 -> 0.91 ms, 4//1 2.03 -> 1.82, ComplexF32 2//1 1.84 -> 1.49; 3 and 6 groups leave the SIMDs unevenly loaded and lose
    //     10 - 20 %).  profiles/r03h_fir_groups.json
    //   * ratios up to one (decimators, single-rate FIR): one big workgroup per CU, so what a tile costs is the matrix work of its BUSIEST
    //     SIMD -- waves go round the four SIMDs, ceil(mw / 4) multiplying waves on the fullest -- plus the tap fetches of the long-filter
    //     mode (every multiplying wave fetches every k-step's taps: ~25 clocks each through the CU's one L1 pipe) plus ~3900 clocks of
    //     barrier and pipeline turn-around, per row:  cost = (ceil(mw/4) CH steps c_mfma max(1, 0.2 ways) + [T = 0] mw steps 25 + 3900) / rows.
    //     Fitted on 1//8 (0.98 : 0.87 : 1.22 measured for CH = 4, 2, 1 against 1 : 0.88 : 1.27) and checked on 1//2 ... 1//16, 3//8
    //     (profiles/r03h_fir_groups.json): 1//3 and 3//8 leave five groups for eight half-size ones (+25 %), 1//8 and 1//16 two / one
    //     groups of four chunks for twice as many of two (+12 %, +30 %).  Costs within 3 % of each other go to the larger tile.
    const bool by_model = f->L <= f->M && tunables().fir_mm_ng <= 0 && tunables().fir_mm_ch <= 0;
    double best_score = -1, best_cost = 0;
    int best_rows = 0, best_mw = 0;
    for (int pad = tunables().fir_mm_pad == 0 ? 0 : 1; pad >= 0 && !g.ok; --pad) {   // (a last resort: output rows without their 16 bytes of padding -- ComplexF64 at 160//147 then fits exactly)
    if (g.NB > 1) g.Lp = 16 * g.NB * g.CS + (pad ? 16 / g.esz : 0);
    else if (!pad) break;
    // round 3: a third staging form for the bank-hostile strides -- ONE run per tile, moved by the same full 256-dword DMA granules as mode 0,
    // with 4 dwords of padding behind every GRANULE (dword d of the tile at d + 4 (d / 256)).  Rows then spread over the banks (counted below),
    // no window tail is fetched twice (row-staged 147//160: 228 samples per 160 new ones in 256-dword granules, 1.6x the L2 -> LDS traffic; its
    // DMA alone took 0.53 of the kernel's 0.78 ms) and the tile is not shrunk by the duplicates.  A window shorter than a granule meets one pad
    // at most, at a lane-dependent step: the A-operand reads choose between two base pointers per lane and keep their immediate offsets.
    // What did NOT work on the way (profiles/r03k_fir_padded_runs.json): padding behind every row or every two rows -- the segments then are
    // 640 or 1280 bytes, i.e. 16-byte DMA with lanes switched off, or a 16-byte plus a 4-byte instruction: 0.87 - 1.0 ms of DMA alone against 0.36
    // for whole granules -- and computed (instead of immediate) read offsets.  147//160 0.76 -> 0.60 ms, Float64 1.41 -> 1.08, 49//48 1.64 -> 0.73.
    const int RPAD = tunables().fir_mm_rpad > 0 ? tunables().fir_mm_rpad : 4;
    double ways_pad = 1;
    {   // conflict ways of the 16 rows of an A operand, counted on the layout itself
        int cnt[64] = {0};
        for (int i = 0; i < 16; ++i) {
            const int64_t d = (int64_t)i * g.Mr * dw, ad = d + (d >> 8) * RPAD;
            ways_pad = std::max<double>(ways_pad, ++cnt[(int)((g.esz == 8 ? ad / 2 : ad) % 32)]);
        }
    }
    const int fm = tunables().fir_mm_rows;
    for (int mode = 0; mode < 3; ++mode) {   // 0: one linear run per tile, 1: row by row, 2: one run with padded rows
        if (mode >= 1 && ways_lin < 4 && fm < 0) continue;
        if (fm >= 0 && mode != fm) continue;
        // 16 rows are whole granules; taps in registers; a window (4 T dw dwords) meets one pad at most.  (round 5 prep, MDSP_FIR_MM_RPX=1: fetched taps
        // with computed pads, and register forms of single-chunk waves whose window meets two pads at most -- 4 T dw <= 512)
        const bool rpx = tunables().fir_mm_rpx != 0;
        const bool gran16 = !((g.Mr * dw) & 15);   // 16 rows are whole granules: every chunk of a wave crosses at the same step (single-chunk waves do not need it)
        const bool pad_ok = (gran16 && g.T != 0 && g.T <= 32 && g.T * dw <= 64) || (rpx && g.T == 0 && (gran16 || chmax == 1)) ||
                            (rpx && g.T != 0 && g.NBLK == 1 && g.T * dw <= 128 && chmax == 1);
        if (mode == 2 && !pad_ok) continue;
        if (mode == 1 && fm < 0 && pad_ok && ways_pad <= 2 * ways_row) continue;   // the padded run replaces the row-staged form wherever it applies and spreads the
                                                                                     // rows comparably (rows shorter than a granule share its pad: Mr dw = 16 stays 15-way)
        const double ways = mode == 1 ? ways_row : mode == 2 ? ways_pad : ways_lin;
        for (int ch = chmax; ch >= 1; ch /= 2) {   // the largest tile (16 CH NG rows) that leaves four memory waves and fits the LDS
            const int rows = 16 * ch;
            const int ngdef = (g.NBW == 1 && f->L > f->M) ? 4 : 8;
            for (int ng = std::min(tunables().fir_mm_ng > 0 ? tunables().fir_mm_ng : ngdef, 12 / g.NBW); ng >= 1; --ng) {
                const int64_t bufsz = mode == 1 ? cdiv((int64_t)rows * ng * rpitch, (int64_t)256) * 256
                                    : mode == 2 ? cdiv((cdiv(((int64_t)rows * ng * g.Mr + g.Mr + wtail) * dw, (int64_t)256) + 1) * (256 + RPAD), (int64_t)256) * 256
                                                : cdiv(((int64_t)rows * ng * g.Mr + g.Mr + wtail) * dw, (int64_t)256) * 256;
                const size_t bytes = (size_t)(2 * bufsz) * 4 + (size_t)(2 * rows * ng * g.Lp) * (size_t)g.esz;
                if (bytes > 160 * 1024) continue;
                const auto take = [&] { g.CH = ch; g.NG = ng; g.bufsz = bufsz; g.lds_bytes = bytes; g.ok = true; g.pitch = mode == 1 ? rpitch : 0; g.rowpad = mode == 2 ? RPAD : 0; };
                if (!by_model) {
                    const double score = 1.0 / (std::max(1.0, 0.2 * ways) + 32.0 / (rows * ng)) + 1e-6 * ch;
                    if (score > best_score) { best_score = score; take(); }
                    break;   // (only the largest tile of this chunk count)
                }
                const int mw = g.NBW * ng, trows = rows * ng;
                const double c_mfma = (g.esz == 8 ? 64.0 : 32.0) * g.CS;
                const double cost = (((mw + 3) / 4) * (double)ch * cdiv(g.NB, g.NBW) * g.steps * c_mfma * std::max(1.0, 0.2 * ways) + (g.T == 0 ? mw * g.steps * 25.0 : 0.0) + 3900.0) / trows;
                // (round 5 prep, MDSP_FIR_MM_TIEWAVES=1, unmeasured as a rule) equal tiles at equal cost: the one with more multiplying waves -- the k-steps of
                // a wave are a latency chain (profiles/r04_fir_decim16_pmc.json), and profiles/r04_fir_ch1_ab.json has both tie cases faster that way
                // (1//16 Float32 1.74 -> 1.46 ms, ComplexF32 1//4 1.01 -> 0.83)
                const bool tie_waves = tunables().fir_mm_tiewaves != 0 && g.ok && cost < 1.03 * best_cost && trows == best_rows && mw > best_mw;
                if (!g.ok || cost < 0.97 * best_cost || (cost < 1.03 * best_cost && trows > best_rows) || tie_waves) {
                    best_cost = g.ok ? std::min(best_cost, cost) : cost;
                    best_rows = trows;
                    best_mw = mw;
                    take();
                }
            }
        }
    }
    }
    if (!g.ok) return g;
    const int extra = 16 - g.NBW * g.NG;   // memory waves beside the multiplying ones (16 waves per workgroup at most)
    g.nd = extra >= 4 ? 2 : 1;
    g.ns = std::max(1, std::min(f->L >= f->M ? 2 : 4, extra - g.nd));   // measured: two store waves when the ratio is >= 1 (160//147 1.94 -> 1.87 ms, 2//1 1.01 -> 0.93), four below (1//2 0.48 -> 0.45)
    if (tunables().fir_mm_nd > 0 && tunables().fir_mm_ns > 0 && tunables().fir_mm_nd + tunables().fir_mm_ns <= extra) {   // tuning knobs
        g.nd = tunables().fir_mm_nd;
        g.ns = tunables().fir_mm_ns;
    }
    return g;
}
// every exec asks twice (use? / dispatch): one memo per calling thread, keyed by what the geometry depends on (the search above is a few
// hundred iterations -- microseconds, but a sample-at-a-time stream pays them per call)
FirMGeo fir_mm_geo(const mdsp_fir_s* f) {
    struct Memo {
        int64_t L = -1, M = 0, hlen = 0;
        int td = 0, xd = 0;
        uint64_t gen = 0;
        FirMGeo g;
    };
    static thread_local Memo m;
    const uint64_t gen = tunables_generation();
    if (m.L != f->L || m.M != f->M || m.hlen != f->hlen || m.td != f->taps_dtype || m.xd != f->x_dtype || m.gen != gen) {
        m.g = fir_mm_geo_compute(f, true);
        if (!m.g.ok) m.g = fir_mm_geo_compute(f, false);   // (longer register forms read a longer window tail: where that no longer fits the LDS, fetch the taps)
        if (!m.g.ok && tunables().fir_mm_tight != 0) {     // the tight forms: see fir_mm_geo_compute
            if (f->L < 16)
                for (int cap = (int)(16 / f->L) - 1; cap >= 1 && !m.g.ok; --cap) m.g = fir_mm_geo_compute(f, true, cap);
            else m.g = fir_mm_geo_compute(f, true, -1);
        }
        m.L = f->L; m.M = f->M; m.hlen = f->hlen; m.td = f->taps_dtype; m.xd = f->x_dtype; m.gen = gen;
    }
    return m.g;
}
bool fir_mm_shape_ok(const mdsp_fir_s* f) { return fir_mm_geo(f).ok; }

template <typename R, int CS, int CH, int T, bool RP = false, int NBLK = 1> int fir_mm_launch(mdsp_fir_s* f, const FirArgs& a, const FirMGeo& g, hipStream_t st) {
    FirMArgs b{};
    b.x = a.x;
    b.hist = a.hist;
    b.y = a.y;
    b.pfbT = a.pfbT;
    b.xlen = a.xlen; b.ldx = a.ldx; b.ldy = a.ldy; b.nout = a.nout;
    b.d0 = a.d0;
    b.L = a.L; b.M = a.M; b.hl = a.hl; b.tp = a.tp;
    b.Lr = g.Lr; b.Mr = g.Mr; b.NB = g.NB; b.NBW = g.NBW; b.NG = g.NG; b.Lp = g.Lp; b.nd = g.nd; b.ns = g.ns;
    b.nrows = cdiv(a.nout, (int64_t)g.Lr);
    b.lmagic = (unsigned)((((uint64_t)1 << 32) + (uint64_t)a.L - 1) / (uint64_t)a.L);
    b.rmagic = (unsigned)((((uint64_t)1 << 32) + (uint64_t)(g.Lr * CS) - 1) / (uint64_t)(g.Lr * CS));
    b.phi0m1 = (int)a.phi0m1;
    b.bufsz = (int)g.bufsz;
    b.pitch = g.pitch;
    b.rowpad = g.rowpad;
    b.steps = g.steps;
    b.vstore = tunables().fir_mm_vstore;
    // measured on 19 ratio x type combinations (profiles/r03l_fir_memprio.json): +2 ... +16 % (config 5 1.86 -> 1.76 ms) wherever a wave owns one column
    // block; -4 % where it walks several (441//160), which keep the default priority
    b.memprio = tunables().fir_mm_prio >= 0 ? tunables().fir_mm_prio : (g.NBW == g.NB ? 1 : 0);
    b.ablate = MDSP_DBG(ablate);
    const int nw = g.NBW * g.NG + g.nd + g.ns;
    void (*kern)(FirMArgs);
    if constexpr (T == 0 || RP) kern = polyphase_mfma_kernel_lso<R, CS, CH, T, RP, NBLK>;
    else kern = polyphase_mfma_kernel<R, CS, CH, T, RP, NBLK>;
    static std::atomic<unsigned long long> lds_opt_in{0};   // once per instantiation and device (later calls may sit inside a stream capture): the whole 160 KiB
    int dev = 0;
    MDSP_HIP(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(lds_opt_in.load(std::memory_order_acquire) & bit)) {
        MDSP_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        lds_opt_in.fetch_or(bit, std::memory_order_release);
    }
    const int64_t ntiles = cdiv(b.nrows, (int64_t)16 * CH * g.NG);
    int wgs = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)(160 * 1024) / (int64_t)g.lds_bytes, 32 / nw));
    if (tunables().wg_per_cu > 0) wgs = tunables().wg_per_cu;
    const int64_t per = std::max<int64_t>(1, (int64_t)device_cu_count() * wgs / std::max<int64_t>(1, f->nch));
    const dim3 grid((unsigned)std::min<int64_t>(ntiles, per), (unsigned)f->nch);
    hipLaunchKernelGGL(kern, grid, dim3(64 * nw), g.lds_bytes, st, b);
    MDSP_LAUNCH_CHECK();
    return MDSP_OK;
}

template <typename R, int CS, int CH> int fir_mm_dispatch_t(mdsp_fir_s* f, const FirArgs& a, const FirMGeo& g, hipStream_t st) {
    if constexpr (CH == 1 || (CH == 2 && sizeof(R) == 4)) {
        if (g.NBLK == 2) return g.T == 12 ? fir_mm_launch<R, CS, CH, 12, false, 2>(f, a, g, st) : fir_mm_launch<R, CS, CH, 16, false, 2>(f, a, g, st);
        if (g.NBLK == 3) return g.T == 12 ? fir_mm_launch<R, CS, CH, 12, false, 3>(f, a, g, st) : fir_mm_launch<R, CS, CH, 16, false, 3>(f, a, g, st);
    }
    if (g.NBLK > 1) MDSP_FAIL(MDSP_ERR_ASSERTION, "no matrix-core instantiation for %d blocks per wave with %d chunks", g.NBLK, CH);
    switch (g.T) {
        case 0: return g.rowpad > 0 ? fir_mm_launch<R, CS, CH, 0, true>(f, a, g, st) : fir_mm_launch<R, CS, CH, 0>(f, a, g, st);
        case 4: return g.rowpad > 0 ? fir_mm_launch<R, CS, CH, 4, true>(f, a, g, st) : fir_mm_launch<R, CS, CH, 4>(f, a, g, st);
        case 8: return g.rowpad > 0 ? fir_mm_launch<R, CS, CH, 8, true>(f, a, g, st) : fir_mm_launch<R, CS, CH, 8>(f, a, g, st);
        case 12: return g.rowpad > 0 ? fir_mm_launch<R, CS, CH, 12, true>(f, a, g, st) : fir_mm_launch<R, CS, CH, 12>(f, a, g, st);
        case 14:
            if constexpr (sizeof(R) == 8 && CS == 2 && CH == 1) return g.rowpad > 0 ? fir_mm_launch<R, CS, CH, 14, true>(f, a, g, st) : fir_mm_launch<R, CS, CH, 14>(f, a, g, st);
            MDSP_FAIL(MDSP_ERR_ASSERTION, "the 14-step form exists for ComplexF64 single-chunk waves only");
        case 16: return g.rowpad > 0 ? fir_mm_launch<R, CS, CH, 16, true>(f, a, g, st) : fir_mm_launch<R, CS, CH, 16>(f, a, g, st);
        case 20: return g.rowpad > 0 ? fir_mm_launch<R, CS, CH, 20, true>(f, a, g, st) : fir_mm_launch<R, CS, CH, 20>(f, a, g, st);
        case 24: return g.rowpad > 0 ? fir_mm_launch<R, CS, CH, 24, true>(f, a, g, st) : fir_mm_launch<R, CS, CH, 24>(f, a, g, st);
        case 32: return g.rowpad > 0 ? fir_mm_launch<R, CS, CH, 32, true>(f, a, g, st) : fir_mm_launch<R, CS, CH, 32>(f, a, g, st);
        default:
            if constexpr (CH == 1) {   // the long register forms exist for single-chunk waves only
                if constexpr (sizeof(R) == 4) {
                    if (g.rowpad > 0) {   // (round 5 prep) padded runs with two pads per window: Float32 up to 96 k-steps (ComplexF32: 64)
                        if (g.T == 40) return fir_mm_launch<R, CS, CH, 40, true>(f, a, g, st);
                        if (g.T == 48) return fir_mm_launch<R, CS, CH, 48, true>(f, a, g, st);
                        if (g.T == 64) return fir_mm_launch<R, CS, CH, 64, true>(f, a, g, st);
                        if constexpr (CS == 1) {
                            if (g.T == 80) return fir_mm_launch<R, CS, CH, 80, true>(f, a, g, st);
                            if (g.T == 96) return fir_mm_launch<R, CS, CH, 96, true>(f, a, g, st);
                        }
                        MDSP_FAIL(MDSP_ERR_ASSERTION, "no padded-run instantiation for %d k-steps", g.T);
                    }
                    if (g.T == 40) return fir_mm_launch<R, CS, CH, 40>(f, a, g, st);
                    if (g.T == 80) return fir_mm_launch<R, CS, CH, 80>(f, a, g, st);
                    if (g.T == 96) return fir_mm_launch<R, CS, CH, 96>(f, a, g, st);
                } else {
                    if (g.rowpad > 0) {   // Float64 up to 48 k-steps (ComplexF64: 32 -- one pad, above)
                        if constexpr (CS == 1) {
                            if (g.T == 40) return fir_mm_launch<R, CS, CH, 40, true>(f, a, g, st);
                            if (g.T == 48) return fir_mm_launch<R, CS, CH, 48, true>(f, a, g, st);
                        }
                        MDSP_FAIL(MDSP_ERR_ASSERTION, "no padded-run instantiation for %d k-steps", g.T);
                    }
                    if (g.T == 40) return fir_mm_launch<R, CS, CH, 40>(f, a, g, st);
                    if constexpr (CS == 1) { if (g.T == 48) return fir_mm_launch<R, CS, CH, 48>(f, a, g, st); }
                }
            }
            if (g.T > 64 || (sizeof(R) == 8 && g.T > 32)) MDSP_FAIL(MDSP_ERR_ASSERTION, "no matrix-core instantiation for %d k-steps", g.T);
            if constexpr (sizeof(R) == 4) return g.T == 48 ? fir_mm_launch<R, CS, CH, 48>(f, a, g, st) : fir_mm_launch<R, CS, CH, 64>(f, a, g, st);
            else return fir_mm_launch<R, CS, CH, 32>(f, a, g, st);
    }
}
int fir_mm_dispatch(mdsp_fir_s* f, const FirArgs& a, hipStream_t st) {
    const FirMGeo g = fir_mm_geo(f);
    if (g.esz == 4 && g.CS == 1) return g.CH == 4 ? fir_mm_dispatch_t<float, 1, 4>(f, a, g, st) : g.CH == 2 ? fir_mm_dispatch_t<float, 1, 2>(f, a, g, st) : fir_mm_dispatch_t<float, 1, 1>(f, a, g, st);
    if (g.esz == 4) return g.CH == 2 ? fir_mm_dispatch_t<float, 2, 2>(f, a, g, st) : fir_mm_dispatch_t<float, 2, 1>(f, a, g, st);
    if (g.CS == 1) return g.CH == 2 ? fir_mm_dispatch_t<double, 1, 2>(f, a, g, st) : fir_mm_dispatch_t<double, 1, 1>(f, a, g, st);
    return g.CH == 2 ? fir_mm_dispatch_t<double, 2, 2>(f, a, g, st) : fir_mm_dispatch_t<double, 2, 1>(f, a, g, st);
}

// where the matrix-core kernel is used: wherever the shape fits, whatever the chunk length -- one channel of 2^8 ... 2^16 samples takes
// 17 - 29 us against 32 - 61 us at 2//1, 18 - 23 against 21 - 31 at 160//147, 20 - 26 against 16 - 44 at 1//2 (profiles/r03h_fir_small_chunks.txt:
// no size gate pays), up to 2^28 (profiles/r02r_tune_fir); MDSP_FIR_MM=0 turns it off
bool fir_mm_use(const mdsp_fir_s* f, const FirArgs& a) {
    (void)a;
    if (tunables().fir_mm == 0 || tunables().fir_exact || f->exact) return false;
    const FirMGeo g = fir_mm_geo(f);
    if (!g.ok) return false;
    // L > 192 (several column blocks per wave, small tiles): measured slower than the register-tap kernel where that one applies
    // (Float32, 441//160: 1.1 against 2.3 TB/s) and 4 - 5x faster than the generic kernel everything else would take (Float64: 0.26 -> 1.2)
    // (round 3: with the taps of TWO blocks per wave in registers the matrix cores win -- 320//147 1.11 against 1.32 ms, 250//249 0.70 against 0.92;
    // with three the register-tap kernel still does: 441//160 1.67 against 2.39)
    if (g.NBW < g.NB && g.NBLK != 2 && tunables().fir_mm != 1 && fir_fast_ok(f, 2)) return false;
    return true;
}

// ---- decimator kernel: geometry and launch ---------------------------------------------------------------------------------------------------
struct DecGeo {
    bool ok = false;
    int Mp = 0, logMp = 0, nq = 0, nbh = 0, nz = 0, P = 0;
    size_t lds = 0;
};
DecGeo fir_dec_geo(const mdsp_fir_s* f) {
    DecGeo g;
    if (f->L != 1 || f->M < 2 || f->M > 64 || tunables().fir_dec == 0 || tunables().fir_exact || f->exact || MDSP_DBG(fir_generic)) return g;
    if (f->acc_double != dtype_is_double(f->x_dtype)) return g;   // Float32 samples under Float64 taps: the generic kernel converts as it stages
    // Where it wins (profiles/r05_fir_dec_ab.json, 4 channels x 2^26 samples, resample_filter taps): Float64 / ComplexF64 from M = 4 on (1.1 - 4.1x),
    // Float32 / ComplexF32 at M = 4 (1.17 - 1.19x) and from M = 8 on (1.1 - 3.9x); at M = 2, 3 and the Float32 M = 5, 6, 7 the matrix-core kernel's
    // short products stay ahead (0.31 - 0.54 of the roof against 0.24 - 0.38).  MDSP_FIR_DEC=3 takes it for every M <= 64 (tests).
    // (re-measured with the unrolled chunk loop, profiles/r05_fir_dec_ab.json "rule_recheck": unchanged but for real Float32 at M = 3 -- 0.52 against 0.61 ms)
    if (tunables().fir_dec == 1 && !(f->acc_double ? f->M >= 4 : (f->M == 4 || f->M >= 8 || (f->M == 3 && !dtype_is_complex(f->x_dtype))))) return g;
    const bool dbl = f->acc_double, cplx = dtype_is_complex(f->x_dtype);
    const int M = (int)f->M;
    g.P = dbl ? 8 : 16;
    g.Mp = 2;
    g.logMp = 1;
    while (g.Mp < M) g.Mp *= 2, ++g.logMp;
    if (f->tp > (int64_t)1 << 20) return g;
    g.nq = (int)cdiv(f->tp, M);
    const int nqp = (g.nq + 7) / 8 * 8, ng = 256 / g.Mp;
    const size_t vsz = dbl ? 16 : 8, rsz = dbl ? 8 : 4;
    // blocks per (half) tile: a multiple of the groups of a workgroup (every group the same number of blocks), as many as ~40 KiB of samples hold
    int kb = 1;
    while (((size_t)ng * (kb + 1) * g.P * M + (size_t)nqp * M) * vsz <= 40 * 1024) ++kb;
    g.nbh = ng * kb;
    g.nz = g.nbh * g.P * M + nqp * M;
    const size_t nzp = (size_t)g.nz + (size_t)((g.nz - 1) / (g.P * M)) * M;   // with M elements of padding behind every block (dec_pos)
    g.lds = nzp * vsz + (((size_t)nqp * M + 3) & ~(size_t)3) * rsz + (size_t)ng * 4 * (g.Mp + 1) * vsz;
    (void)cplx;
    g.ok = g.lds <= 150 * 1024;
    return g;
}
template <typename R, bool CPLX, int P> int fir_dec_launch(mdsp_fir_s* f, const FirArgs& a, const DecGeo& g, hipStream_t st) {
    DecArgs d{};
    d.x = a.x;
    d.hist = a.hist;
    d.y = a.y;
    d.pfbT = a.pfbT;
    d.xlen = a.xlen;
    d.ldx = a.ldx;
    d.ldy = a.ldy;
    d.nout = a.nout;
    d.zb = a.d0 - 1 + a.phi0m1;
    d.M = a.M;
    d.Mp = g.Mp;
    d.logMp = g.logMp;
    d.tp = a.tp;
    d.hl = a.hl;
    d.nq = g.nq;
    d.nbh = g.nbh;
    d.nz = g.nz;
    d.ablate = tunables().fir_dec_ablate;
    d.blkmagic = (unsigned)((((uint64_t)1 << 32) + (uint64_t)(P * a.M) - 1) / (uint64_t)(P * a.M));
    void (*kern)(DecArgs) = decimator_kernel<R, CPLX, P, 0>;
    // chunks of taps per phase: five is what resample_filter designs for every decimator; two .. six unrolled for M = 4, 8, 16 (the run-time loop otherwise)
    const int nc = tunables().fir_dec_nc != 0 ? (g.nq + 7) / 8 : 0;
#define MDSP_DEC_NC(MCV)                                                                      \
    switch (nc) {                                                                             \
        case 2: kern = decimator_kernel<R, CPLX, P, MCV, 2>; break;                           \
        case 3: kern = decimator_kernel<R, CPLX, P, MCV, 3>; break;                           \
        case 4: kern = decimator_kernel<R, CPLX, P, MCV, 4>; break;                           \
        case 5: kern = decimator_kernel<R, CPLX, P, MCV, 5>; break;                           \
        case 6: kern = decimator_kernel<R, CPLX, P, MCV, 6>; break;                           \
        default: kern = decimator_kernel<R, CPLX, P, MCV>; break;                             \
    }
    switch (a.M) {
        case 2: kern = decimator_kernel<R, CPLX, P, 2>; break;
        case 4: MDSP_DEC_NC(4) break;
        case 8: MDSP_DEC_NC(8) break;
        case 16: MDSP_DEC_NC(16) break;
        default:
            if (nc == 5) kern = decimator_kernel<R, CPLX, P, 0, 5>;
            break;
    }
#undef MDSP_DEC_NC
    if (tunables().fir_dec == 2) kern = decimator_kernel<R, CPLX, P, 0>;   // MDSP_FIR_DEC=2: the run-time M form for every M
    if (g.lds > 48 * 1024) MDSP_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds));
    const int64_t TO = (int64_t)(CPLX ? 1 : 2) * g.nbh * P;
    // Float64: the workgroups that stay resident (two a CU), each walking its share of the tiles with the next one's samples in flight; Float32: one tile each
    const int64_t ntile = cdiv(a.nout, TO);
    const int wgs = tunables().fir_dec_wgs > 0 ? tunables().fir_dec_wgs : 2;
    const int64_t per = sizeof(R) == 8 ? std::max<int64_t>(1, (int64_t)device_cu_count() * wgs / std::max<int64_t>(1, f->nch)) : ntile;
    hipLaunchKernelGGL(kern, dim3((unsigned)std::min<int64_t>(ntile, per), (unsigned)f->nch), dim3(256), g.lds, st, d);
    MDSP_LAUNCH_CHECK();
    return MDSP_OK;
}
int fir_dec_dispatch(mdsp_fir_s* f, const FirArgs& a, const DecGeo& g, hipStream_t st) {
    switch (f->x_dtype) {
        case MDSP_F32: return fir_dec_launch<float, false, 16>(f, a, g, st);
        case MDSP_F64: return fir_dec_launch<double, false, 8>(f, a, g, st);
        case MDSP_C32: return fir_dec_launch<float, true, 16>(f, a, g, st);
        default: return fir_dec_launch<double, true, 8>(f, a, g, st);
    }
}

// A filter whose shape has a line in the box's choice file (common.h FirChoice) is dispatched under that line's knob values: the calling thread's view of
// the tunables for the duration of the scope.
struct FirChoiceScope {
    Tunables t;
    const Tunables* prev = nullptr;
    bool on = false;
    explicit FirChoiceScope(const mdsp_fir_s* f) {
        const Tunables& base = tunables();
        if (base.fir_choices.empty()) return;   // (the usual case: no choice file named)
        for (const FirChoice& c : base.fir_choices) {
            if (c.L != f->L || c.M != f->M || c.hlen != f->hlen || c.taps_dtype != f->taps_dtype || c.x_dtype != f->x_dtype) continue;
            t = base;
            t.fir_choices.clear();   // the view carries the knob values only
            for (int i = 0; i < c.nset; ++i)
                if (int* field = fir_choice_field(t, c.field[i])) *field = c.value[i];
            prev = tunables_override(&t);
            on = true;
            break;
        }
    }
    ~FirChoiceScope() {
        if (on) tunables_override(prev);
    }
    FirChoiceScope(const FirChoiceScope&) = delete;
    FirChoiceScope& operator=(const FirChoiceScope&) = delete;
};

bool fir_reg_use(const mdsp_fir_s* f) {
    if (MDSP_DBG(fir_generic) || tunables().fir_exact || f->exact || f->kind == 4) return false;
    return fir_reg_ok(f->x_dtype, f->acc_double, f->tp, f->L, f->M);
}

int fir_dispatch(mdsp_fir_s* f, FirArgs& a, hipStream_t st) {
    const FirChoiceScope choice(f);
    {
        const DecGeo dg = fir_dec_geo(f);
        if (dg.ok) return fir_dec_dispatch(f, a, dg, st);
    }
    if (fir_mm_use(f, a)) return fir_mm_dispatch(f, a, st);
    if (tunables().fir_p == 4 && f->tp <= 32 && fir_fast_ok(f, 4)) return fir_fast_dispatch<4>(f, a, st);   // tuning: four residues per thread
    if (tunables().fir_p == 3 && f->tp <= 32 && fir_fast_ok(f, 3)) return fir_fast_dispatch<3>(f, a, st);   // tuning: three residues per thread
    if (fir_fast_ok(f, 2)) return fir_fast_dispatch<2>(f, a, st);
    if (fir_fast_ok(f, 1)) return fir_fast_dispatch<1>(f, a, st);
    if (fir_reg_use(f)) {   // round 6: register taps for every other signal type where the matrix-core tile missed the LDS (fir_reg.hip)
        FirRegArgs b{};
        b.x = a.x; b.hist = a.hist; b.y = a.y; b.pfbT = a.pfbT;
        b.xlen = a.xlen; b.ldx = a.ldx; b.ldy = a.ldy; b.nout = a.nout;
        b.phi0m1 = a.phi0m1; b.d0 = a.d0;
        b.L = a.L; b.M = a.M; b.tp = a.tp; b.hl = a.hl;
        return fir_reg_run(f->x_dtype, f->acc_double, b, f->nch, st);
    }
    const bool d = f->acc_double;
    switch (f->x_dtype) {
        case MDSP_F32: return d ? fir_launch<float, double, double>(f, a, st) : fir_launch<float, float, float>(f, a, st);
        case MDSP_F64: return fir_launch<double, double, double>(f, a, st);
        case MDSP_C32: return d ? fir_launch<cx<float>, cx<double>, double>(f, a, st) : fir_launch<cx<float>, cx<float>, float>(f, a, st);
        default: return fir_launch<cx<double>, cx<double>, double>(f, a, st);
    }
}

template <typename XS> int shiftin_launch(mdsp_fir_s* f, const void* x, int64_t xlen, int64_t ldx, hipStream_t st) {
    if (f->hl == 0) return MDSP_OK;
    const int nxt = f->cur ^ 1;
    const dim3 grid((unsigned)std::min<int64_t>(cdiv(f->hl, 256), 64), (unsigned)f->nch);
    hipLaunchKernelGGL(shiftin_kernel<XS>, grid, dim3(256), 0, st, (const XS*)x, f->hist[f->cur].as<XS>(), f->hist[nxt].as<XS>(), xlen, ldx, (int)f->hl);
    MDSP_LAUNCH_CHECK();
    f->cur = nxt;
    return MDSP_OK;
}

int shiftin_dispatch(mdsp_fir_s* f, const void* x, int64_t xlen, int64_t ldx, hipStream_t st) {
    if (xlen == 0) return MDSP_OK;
    switch (f->x_dtype) {
        case MDSP_F32: return shiftin_launch<float>(f, x, xlen, ldx, st);
        case MDSP_F64: return shiftin_launch<double>(f, x, xlen, ldx, st);
        case MDSP_C32: return shiftin_launch<cx<float>>(f, x, xlen, ldx, st);
        default: return shiftin_launch<cx<double>>(f, x, xlen, ldx, st);
    }
}

// Julia's round(Int, x): round half to even
int64_t round_half_even(double v) { return (int64_t)std::nearbyint(v); }

}  // namespace

extern "C" {

int mdsp_fir_create(mdsp_fir* fo, const void* taps_host, int64_t hlen, int64_t L, int64_t M, int taps_dtype, int x_dtype, int64_t nch) {
    if (!fo) MDSP_FAIL(MDSP_ERR_ARGUMENT, "handle pointer is NULL");
    *fo = nullptr;
    if (!taps_host || hlen < 1) MDSP_FAIL(MDSP_ERR_ARGUMENT, "filter taps must be non-empty");
    if (L < 1 || M < 1) MDSP_FAIL(MDSP_ERR_DOMAIN, "resampling ratio must be positive");
    if (taps_dtype != MDSP_F32 && taps_dtype != MDSP_F64) MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "only real Float32/Float64 taps are supported on the device");
    if (!dtype_valid(x_dtype)) MDSP_FAIL(MDSP_ERR_ARGUMENT, "invalid x dtype");
    if (nch < 1 || nch > 65535) MDSP_FAIL(MDSP_ERR_ARGUMENT, "nch must be in [1, 65535]");
    const int64_t g = gcd64(L, M);  // Rational normalisation (numerator / denominator of L//M)
    L /= g;
    M /= g;
    if (L > (int64_t(1) << 20) || M > (int64_t(1) << 30)) MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "ratio terms too large");
    auto f = new mdsp_fir_s();
    f->L = L;
    f->M = M;
    f->hlen = hlen;
    f->nch = nch;
    f->taps_dtype = taps_dtype;
    f->x_dtype = x_dtype;
    f->kind = (L == 1 && M == 1) ? 0 : (M == 1 ? 1 : (L == 1 ? 2 : 3));
    f->tp = cdiv(hlen, L);                       // taps2pfb: tapsPerPhase = ceil(hLen / Nphi)   (stream_filt.jl:296)
    f->hl = f->tp - 1;                           // historyLen (:163,:166,:169,:172)
    f->acc_double = (taps_dtype == MDSP_F64) || dtype_is_double(x_dtype);  // promote_type(Th, Tx)
    f->out_dtype = dtype_is_complex(x_dtype) ? (f->acc_double ? MDSP_C64 : MDSP_C32) : (f->acc_double ? MDSP_F64 : MDSP_F32);
    // pfbT[i*L + c] = pfb[i+1, c+1] with pfb[row, col] = h[(tp-row)*L + col] (1-based rows from the bottom, :300-304)
    const size_t np = (size_t)(f->tp * L);
    std::vector<double> pd(np, 0.0);
    for (int64_t row = 0; row < f->tp; ++row)       // row 0 = top of the matrix
        for (int64_t col = 0; col < L; ++col) {
            const int64_t hidx = (f->tp - 1 - row) * L + col;  // bottom row holds h[0..L)
            double v = 0;
            if (hidx < hlen) v = taps_dtype == MDSP_F32 ? (double)((const float*)taps_host)[hidx] : ((const double*)taps_host)[hidx];
            pd[(size_t)(row * L + col)] = v;
        }
    int st = MDSP_OK;
    if (f->acc_double) {
        st = f->pfbT.reserve(sizeof(double) * np);
        if (st == MDSP_OK && hipMemcpy(f->pfbT.p, pd.data(), sizeof(double) * np, hipMemcpyHostToDevice) != hipSuccess) st = set_error(MDSP_ERR_DEVICE, "tap upload failed");
    } else {
        std::vector<float> pf(np);
        for (size_t i = 0; i < np; ++i) pf[i] = (float)pd[i];
        st = f->pfbT.reserve(sizeof(float) * np);
        if (st == MDSP_OK && hipMemcpy(f->pfbT.p, pf.data(), sizeof(float) * np, hipMemcpyHostToDevice) != hipSuccess) st = set_error(MDSP_ERR_DEVICE, "tap upload failed");
    }
    const size_t hbytes = dtype_size(x_dtype) * (size_t)std::max<int64_t>(1, f->hl) * (size_t)nch;
    for (int b = 0; b < 2 && st == MDSP_OK; ++b) {
        st = f->hist[b].reserve(hbytes);
        if (st == MDSP_OK && hipMemset(f->hist[b].p, 0, hbytes) != hipSuccess) st = set_error(MDSP_ERR_DEVICE, "history init failed");
    }
    if (st != MDSP_OK) {
        delete f;
        return st;
    }
    *fo = f;
    return MDSP_OK;
}

int mdsp_fir_destroy(mdsp_fir f) {
    delete f;
    return MDSP_OK;
}

int mdsp_fir_set_exact(mdsp_fir f, int exact) {
    if (!f) MDSP_FAIL(MDSP_ERR_ARGUMENT, "handle is NULL");
    f->exact = exact != 0;
    return MDSP_OK;
}

static int fir_reset_on(mdsp_fir f, hipStream_t st, bool async) {
    if (!f) MDSP_FAIL(MDSP_ERR_ARGUMENT, "handle is NULL");
    const size_t hbytes = dtype_size(f->x_dtype) * (size_t)std::max<int64_t>(1, f->hl) * (size_t)f->nch;
    if (async) MDSP_HIP(hipMemsetAsync(f->hist[f->cur].p, 0, hbytes, st));   // stream-ordered: the caller's next exec runs on the same stream
    else MDSP_HIP(hipMemset(f->hist[f->cur].p, 0, hbytes));
    f->phi_idx = 1;
    f->input_deficit = 1;
    return MDSP_OK;
}

int mdsp_fir_reset(mdsp_fir f) { return fir_reset_on(f, nullptr, false); }

int mdsp_fir_setphase(mdsp_fir f, double phi) {
    if (!f) MDSP_FAIL(MDSP_ERR_ARGUMENT, "handle is NULL");
    if (!(phi >= 0)) MDSP_FAIL(MDSP_ERR_DOMAIN, "phi must be >= 0");
    if (f->kind == 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "setphase! is not defined for a single-rate FIRFilter");
    if (f->kind == 2) {  // :216-221
        f->input_deficit += round_half_even(phi);
    } else {  // :223-229
        const int64_t q = round_half_even(phi * (double)f->L);
        f->input_deficit += q / f->L;
        f->phi_idx = q % f->L + 1;
    }
    return MDSP_OK;
}

int mdsp_fir_timedelay(mdsp_fir f, double* tau) {
    if (!f || !tau) MDSP_FAIL(MDSP_ERR_ARGUMENT, "NULL argument");
    *tau = (f->kind == 1 || f->kind == 3) ? (double)(f->hlen - 1) / (double)(2 * f->L) : (double)(f->hlen - 1) / 2.0;
    return MDSP_OK;
}

int mdsp_fir_outputlength(mdsp_fir f, int64_t inputlength, int64_t* outlen) {
    if (!f || !outlen) MDSP_FAIL(MDSP_ERR_ARGUMENT, "NULL argument");
    if (f->kind == 0) *outlen = inputlength;
    else *outlen = mdsp_outputlength(inputlength - f->input_deficit + 1, f->L, f->M, f->kind == 2 ? 1 : f->phi_idx);
    return MDSP_OK;
}

int mdsp_fir_inputlength(mdsp_fir f, int64_t outputlength, int round_up, int64_t* inlen) {
    if (!f || !inlen) MDSP_FAIL(MDSP_ERR_ARGUMENT, "NULL argument");
    if (f->kind == 0) *inlen = outputlength;
    else *inlen = mdsp_inputlength(outputlength, f->L, f->M, f->kind == 2 ? 1 : f->phi_idx, round_up) + f->input_deficit - 1;
    return MDSP_OK;
}

int mdsp_fir_info(mdsp_fir f, int* kind, int64_t* L, int64_t* M, int64_t* taps_per_phase, int64_t* history_len, int* out_dtype) {
    if (!f) MDSP_FAIL(MDSP_ERR_ARGUMENT, "handle is NULL");
    if (kind) *kind = f->kind;
    if (L) *L = f->L;
    if (M) *M = f->M;
    if (taps_per_phase) *taps_per_phase = f->tp;
    if (history_len) *history_len = f->hl;
    if (out_dtype) *out_dtype = f->out_dtype;
    return MDSP_OK;
}

int mdsp_fir_mm_geometry(int64_t L, int64_t M, int64_t hlen, int taps_dtype, int x_dtype, int64_t* out12) {
    if (!out12) MDSP_FAIL(MDSP_ERR_ARGUMENT, "out is NULL");
    if (L < 1 || M < 1 || hlen < 1) MDSP_FAIL(MDSP_ERR_ARGUMENT, "L, M, hlen must be positive");
    if ((taps_dtype != MDSP_F32 && taps_dtype != MDSP_F64) || !dtype_valid(x_dtype)) MDSP_FAIL(MDSP_ERR_ARGUMENT, "invalid dtype");
    const int64_t g0 = gcd64(L, M);
    mdsp_fir_s f;   // no device objects are touched: pure geometry
    f.L = L / g0;
    f.M = M / g0;
    f.hlen = hlen;
    f.tp = cdiv(hlen, f.L);
    f.hl = f.tp - 1;
    f.taps_dtype = taps_dtype;
    f.x_dtype = x_dtype;
    f.acc_double = (taps_dtype == MDSP_F64) || dtype_is_double(x_dtype);
    const FirChoiceScope choice(&f);
    const FirMGeo g = fir_mm_geo(&f);
    const int64_t v[12] = {g.ok ? 1 : 0, g.RB, g.Lr, g.Mr, g.NB, g.NG, g.steps, g.CH, g.CS, g.nd, g.ns, (int64_t)g.lds_bytes};
    for (int i = 0; i < 12; ++i) out12[i] = v[i];
    return MDSP_OK;
}

int mdsp_fir_kernel_path(mdsp_fir f, int64_t xlen, int* path) {
    if (!f || !path) MDSP_FAIL(MDSP_ERR_ARGUMENT, "NULL argument");
    if (xlen < 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "negative size");
    const int64_t phi0 = f->kind == 2 ? 1 : f->phi_idx;
    const int64_t d0 = f->kind == 0 ? 1 : f->input_deficit;
    FirArgs a{};
    a.nout = f->kind == 0 ? xlen : (xlen < d0 ? 0 : mdsp_outputlength(xlen - d0 + 1, f->L, f->M, phi0));
    a.L = (int)f->L;
    a.M = (int)f->M;
    const FirChoiceScope choice(f);
    *path = fir_dec_geo(f).ok ? 3 : fir_mm_use(f, a) ? 2 : (fir_fast_ok(f, 2) || fir_fast_ok(f, 1) || fir_reg_use(f)) ? 1 : 0;   // 3: decimator kernel, 2: matrix cores, 1: register taps, 0: generic
    return MDSP_OK;
}

int mdsp_fir_get_state(mdsp_fir f, int64_t* phi_idx, int64_t* input_deficit, void* history_host) {
    if (!f) MDSP_FAIL(MDSP_ERR_ARGUMENT, "handle is NULL");
    if (phi_idx) *phi_idx = f->phi_idx;
    if (input_deficit) *input_deficit = f->input_deficit;
    if (history_host && f->hl > 0) {
        MDSP_HIP(hipDeviceSynchronize());
        MDSP_HIP(hipMemcpy(history_host, f->hist[f->cur].p, dtype_size(f->x_dtype) * (size_t)f->hl * (size_t)f->nch, hipMemcpyDeviceToHost));
    }
    return MDSP_OK;
}

int mdsp_fir_set_state(mdsp_fir f, int64_t phi_idx, int64_t input_deficit, const void* history_host) {
    if (!f) MDSP_FAIL(MDSP_ERR_ARGUMENT, "handle is NULL");
    if (phi_idx < 1 || phi_idx > f->L || input_deficit < 1) MDSP_FAIL(MDSP_ERR_ARGUMENT, "state out of range");
    f->phi_idx = phi_idx;
    f->input_deficit = input_deficit;
    if (history_host && f->hl > 0) {
        MDSP_HIP(hipDeviceSynchronize());
        MDSP_HIP(hipMemcpy(f->hist[f->cur].p, history_host, dtype_size(f->x_dtype) * (size_t)f->hl * (size_t)f->nch, hipMemcpyHostToDevice));
    }
    return MDSP_OK;
}

int mdsp_fir_exec(mdsp_fir f, const void* x_dev, int64_t xlen, int64_t ldx, void* y_dev, int64_t ycap, int64_t ldy, int64_t* nwritten,
                  void* stream) {
    if (!f) MDSP_FAIL(MDSP_ERR_ARGUMENT, "handle is NULL");
    if (xlen < 0 || ycap < 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "negative size");
    if (f->nch > 1 && ldx < xlen) MDSP_FAIL(MDSP_ERR_ARGUMENT, "ldx smaller than xlen");
    hipStream_t st = as_stream(stream);
    if (nwritten) *nwritten = 0;
    if (f->kind != 0 && xlen < f->input_deficit) {  // stream_filt.jl:483-487 (and :443-447, :529-533)
        MDSP_TRY(shiftin_dispatch(f, x_dev, xlen, ldx, st));
        f->input_deficit -= xlen;
        return MDSP_OK;
    }
    const int64_t phi0 = f->kind == 2 ? 1 : f->phi_idx;
    const int64_t d0 = f->kind == 0 ? 1 : f->input_deficit;
    const int64_t nout = f->kind == 0 ? xlen : mdsp_outputlength(xlen - d0 + 1, f->L, f->M, phi0);
    if (ycap < nout) MDSP_FAIL(MDSP_ERR_ARGUMENT, "buffer is too small: need %lld, have %lld", (long long)nout, (long long)ycap);
    if (f->nch > 1 && ldy < nout) MDSP_FAIL(MDSP_ERR_ARGUMENT, "ldy smaller than the output length");
    if (nout > 0) {
        if (!x_dev || !y_dev) MDSP_FAIL(MDSP_ERR_ARGUMENT, "NULL buffer");
        FirArgs a{};
        a.x = x_dev;
        a.hist = f->hist[f->cur].p;
        a.y = y_dev;
        a.pfbT = f->pfbT.p;
        a.xlen = xlen;
        a.ldx = ldx;
        a.ldy = ldy;
        a.nout = nout;
        a.phi0m1 = phi0 - 1;
        a.d0 = d0;
        a.L = (int)f->L;
        a.M = (int)f->M;
        a.tp = (int)f->tp;
        a.hl = (int)f->hl;
        MDSP_TRY(fir_dispatch(f, a, st));
    }
    // state advance: closed form of the loop's final (inputIdx, phiIdx)   (:506-511)
    const __int128 p_end = (__int128)(phi0 - 1) + (__int128)nout * f->M;
    const int64_t input_idx_end = d0 + (int64_t)(p_end / f->L);
    if (f->kind == 1 || f->kind == 3) f->phi_idx = (int64_t)(p_end % f->L) + 1;
    if (f->kind == 1) f->input_deficit = 1;                              // :465
    else if (f->kind != 0) f->input_deficit = input_idx_end - xlen;      // :511, :554
    MDSP_TRY(shiftin_dispatch(f, x_dev, xlen, ldx, st));                 // :512
    if (nwritten) *nwritten = nout;
    return MDSP_OK;
}

// filt(::FIRFilter, x) / resample(x, ratio, h) of host arrays (stream_filt.jl:627-637, :688-775): the stream goes through the filter in time
// chunks of all channels -- exactly the reference's streaming use of a FIRFilter, whose state (phase index, input deficit, history) the object
// carries from chunk to chunk, so the chunked result is the one-shot result bit for bit and the filter ends in the same state.
int mdsp_fir_exec_host(mdsp_fir f, const void* x_host, int64_t xlen, int64_t ldx, void* y_host, int64_t ycap, int64_t ldy, int64_t* nwritten,
                       int flags) {
    if (!f) MDSP_FAIL(MDSP_ERR_ARGUMENT, "handle is NULL");
    if (xlen < 0 || ycap < 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "negative size");
    if (f->nch > 1 && ldx < xlen) MDSP_FAIL(MDSP_ERR_ARGUMENT, "ldx smaller than xlen");
    if (nwritten) *nwritten = 0;
    int64_t total = 0;
    MDSP_TRY(mdsp_fir_outputlength(f, xlen, &total));
    if (ycap < total) MDSP_FAIL(MDSP_ERR_ARGUMENT, "buffer is too small: need %lld, have %lld", (long long)total, (long long)ycap);
    if (f->nch > 1 && ldy < total) MDSP_FAIL(MDSP_ERR_ARGUMENT, "ldy smaller than the output length");
    if (xlen == 0) return MDSP_OK;
    if (!x_host || (total > 0 && !y_host)) MDSP_FAIL(MDSP_ERR_ARGUMENT, "NULL buffer");
    const bool pinned = (flags & MDSP_HOST_PINNED) != 0;
    const size_t esz = dtype_size(f->x_dtype), osz = dtype_size(f->out_dtype);
    const int64_t nch = f->nch;
    // samples per channel and chunk: ~host_chunk_mib of input and output together
    const double per_sample = (double)esz + (double)osz * (double)f->L / (double)f->M;
    const int64_t cl_max = std::max<int64_t>(4096, (int64_t)((double)((int64_t)tunables().host_chunk_mib << 20) / per_sample / (double)nch));
    const int64_t ocap = (int64_t)(((__int128)cl_max * f->L) / f->M) + 2;      // outputs of a chunk never exceed ceil(cl L / M) + 1
    hostpipe::Session ss((size_t)cl_max * (size_t)nch * esz, (size_t)ocap * (size_t)nch * osz, pinned);
    int rc = ss.status();
    int64_t done = 0;
    for (int64_t x0 = 0; x0 < xlen && rc == MDSP_OK; x0 += cl_max) {
        hostpipe::Lane* ln = nullptr;
        if ((rc = ss.acquire(&ln)) != MDSP_OK) break;
        const int64_t cl = std::min(cl_max, xlen - x0);
        if ((rc = ss.upload(ln, static_cast<const char*>(x_host) + (size_t)x0 * esz, (size_t)ldx * esz, (size_t)cl * esz, (size_t)nch)) != MDSP_OK) break;
        int64_t nw = 0;
        if ((rc = mdsp_fir_exec(f, ln->din.p, cl, cl, ln->dout.p, ocap, ocap, &nw, ss.kstream())) != MDSP_OK) break;
        rc = ss.download(ln, static_cast<char*>(y_host) + (size_t)done * osz, (size_t)ldy * osz, (size_t)nw * osz, (size_t)nch, 0, (size_t)ocap * osz);
        done += nw;
    }
    rc = ss.finish(rc);
    if (rc == MDSP_OK && done != total) rc = set_error(MDSP_ERR_ASSERTION, "chunked filtering wrote %lld outputs, the closed form says %lld", (long long)done, (long long)total);
    if (nwritten) *nwritten = done;
    return rc;
}

// ---- FIRArbitrary host side ---------------------------------------------------------------------------------
}  // extern "C"

namespace {

// The loop of filt!(buffer, ::FIRFilter{FIRArbitrary}, x) (stream_filt.jl:593-622) without the dot products.
// Anchors (xIdx, phiAcc) of outputs 0, blk, 2 blk, ... go to ax / aa when given.  Same IEEE operations as the
// reference, so the trajectory -- and with it the output count and the final state -- is bit-exact.
// kstop: stop after that many outputs (a multiple of blk) with the state of output kstop -- the pilot of the parallel scan.
void arb_trajectory(double acc, int64_t deficit, const ArbStep& st, int64_t xlen, int64_t blk, std::vector<int64_t>* ax, std::vector<double>* aa,
                    int64_t* nout, double* acc_end, int64_t* xidx_end, int64_t kstop = INT64_MAX) {
    int64_t n = 0, xi = deficit;
    const double nphi = st.nphi, nphi2 = 2.0 * st.nphi, nphi3 = 3.0 * st.nphi, nphi4 = 4.0 * st.nphi, delta = st.delta;
    while (xi <= xlen && n < kstop) {
        if (ax) {
            ax->push_back(xi);
            aa->push_back(acc);
        }
        // one anchor block: the loop-carried chain is add -> compare -> (subtract); q = 1 is the common overflow and
        // acc - nphi is exact there (Sterbenz), which is what divrem returns
        const int64_t stop = n + blk;
        while (n < stop && xi <= xlen) {
            ++n;
            const double prev = acc;
            acc = prev + delta;
            if (acc >= nphi) {
                if (acc < nphi2) {
                    acc -= nphi;
                    xi += 1;
                } else if (acc < nphi3) {   // q = 2, 3: same argument (q nphi <= acc < 2 q nphi)
                    acc -= nphi2;
                    xi += 2;
                } else if (acc < nphi4) {
                    acc -= nphi3;
                    xi += 3;
                } else {               // many input samples skipped at once (rate < 1/4): the general divrem
                    acc = prev;
                    st(acc, xi);
                }
            }
        }
    }
    *nout = n;
    *acc_end = acc;
    *xidx_end = xi;
}

// ---- parallel trajectory scan (arb_scan.h): kernels and drivers ------------------------------------------------
static_assert(arbscan::ANCH == ARB_BLK && arbscan::BLK == 2 * ARB_BLK, "two anchors per scan block");

template <int RC> __global__ __launch_bounds__(256) void arb_scan_tables_kernel(arbscan::ScanArgs a) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < a.nb) arbscan::scan_tables_body<RC>(a, b);
}
template <int RC, typename TIn> __global__ __launch_bounds__(256) void arb_scan_compose_kernel(arbscan::Grid G, const TIn* in, int64_t n, int64_t* out, int64_t ng) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < ng) arbscan::scan_compose_body<RC>(G, in, n, out, g);
}
template <typename TIn> void launch_compose(const arbscan::Grid& G, const TIn* in, int64_t n, int64_t* out, int64_t ng, hipStream_t st) {
    const dim3 grid((unsigned)cdiv(ng, (int64_t)256));
    switch (G.R) {
        case 2: hipLaunchKernelGGL((arb_scan_compose_kernel<2, TIn>), grid, dim3(256), 0, st, G, in, n, out, ng); break;
        case 4: hipLaunchKernelGGL((arb_scan_compose_kernel<4, TIn>), grid, dim3(256), 0, st, G, in, n, out, ng); break;
        case 8: hipLaunchKernelGGL((arb_scan_compose_kernel<8, TIn>), grid, dim3(256), 0, st, G, in, n, out, ng); break;
        default: hipLaunchKernelGGL((arb_scan_compose_kernel<16, TIn>), grid, dim3(256), 0, st, G, in, n, out, ng); break;
    }
}
__global__ void arb_scan_top_kernel(arbscan::Grid G, const int64_t* in, int64_t n, int64_t* Eout) {
    if (blockIdx.x == 0 && threadIdx.x == 0) arbscan::scan_top_body(G, in, n, Eout);
}
template <typename TIn>
__global__ __launch_bounds__(256) void arb_scan_expand_kernel(arbscan::Grid G, const TIn* in, int64_t n, const int64_t* Ecoarse, int64_t* Efine, int64_t ng) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < ng) arbscan::scan_expand_body(G, in, n, Ecoarse, Efine, g);
}
__global__ __launch_bounds__(256) void arb_scan_finalize_kernel(arbscan::ScanArgs a) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < a.nb) arbscan::scan_finalize_body(a, b);
}

struct ScanLayout {
    std::vector<int64_t> n;      // elements per level: n[0] blocks, n[l+1] = ceil(n[l] / FAN); the last level is scanned serially
    std::vector<int64_t> woff;   // wide tables of level l >= 1 start at woff[l] (int64 elements)
    std::vector<int64_t> eoff;   // E of level l starts at eoff[l]
    int64_t wide_total = 0, e_total = 0;
};
ScanLayout scan_layout(int64_t nb, int R) {
    ScanLayout L;
    L.n.push_back(nb);
    do L.n.push_back(cdiv(L.n.back(), (int64_t)arbscan::FAN));
    while (L.n.back() > 1024);
    L.woff.assign(L.n.size(), 0);
    L.eoff.assign(L.n.size(), 0);
    for (size_t l = 0; l < L.n.size(); ++l) {
        L.eoff[l] = L.e_total;
        L.e_total += L.n[l];
        if (l >= 1) {
            L.woff[l] = L.wide_total;
            L.wide_total += L.n[l] * R;
        }
    }
    return L;
}

// One pass of the scan: tables -> up-sweep -> serial top -> down-sweep -> finalize.  device = false runs the same bodies in
// host loops (CPU tests of the arithmetic; the product path always passes true).
int scan_pass(const arbscan::ScanArgs& a, const ScanLayout& L, int64_t* wide, int64_t* Eall, bool device, hipStream_t st) {
    using namespace arbscan;
    const int nl = (int)L.n.size() - 1;   // top level
    const auto blocks = [](int64_t n) { return dim3((unsigned)cdiv(n, (int64_t)256)); };
    if (device) {
        switch (a.G.R) {   // table size 2^(J+1) as a template argument: the candidates stay in registers
            case 2: hipLaunchKernelGGL(arb_scan_tables_kernel<2>, blocks(a.nb), dim3(256), 0, st, a); break;
            case 4: hipLaunchKernelGGL(arb_scan_tables_kernel<4>, blocks(a.nb), dim3(256), 0, st, a); break;
            case 8: hipLaunchKernelGGL(arb_scan_tables_kernel<8>, blocks(a.nb), dim3(256), 0, st, a); break;
            default: hipLaunchKernelGGL(arb_scan_tables_kernel<16>, blocks(a.nb), dim3(256), 0, st, a); break;
        }
        launch_compose<int32_t>(a.G, (const int32_t*)a.t0, L.n[0], wide + L.woff[1], L.n[1], st);
        for (int l = 2; l <= nl; ++l)
            launch_compose<int64_t>(a.G, (const int64_t*)(wide + L.woff[l - 1]), L.n[l - 1], wide + L.woff[l], L.n[l], st);
        hipLaunchKernelGGL(arb_scan_top_kernel, dim3(1), dim3(64), 0, st, a.G, (const int64_t*)(wide + L.woff[nl]), L.n[nl], Eall + L.eoff[nl]);
        for (int l = nl; l >= 2; --l)
            hipLaunchKernelGGL(arb_scan_expand_kernel<int64_t>, blocks(L.n[l]), dim3(256), 0, st, a.G, (const int64_t*)(wide + L.woff[l - 1]), L.n[l - 1],
                               (const int64_t*)(Eall + L.eoff[l]), Eall + L.eoff[l - 1], L.n[l]);
        hipLaunchKernelGGL(arb_scan_expand_kernel<int32_t>, blocks(L.n[1]), dim3(256), 0, st, a.G, (const int32_t*)a.t0, L.n[0], (const int64_t*)(Eall + L.eoff[1]),
                           Eall + L.eoff[0], L.n[1]);
        hipLaunchKernelGGL(arb_scan_finalize_kernel, blocks(a.nb), dim3(256), 0, st, a);
        MDSP_LAUNCH_CHECK();
        return MDSP_OK;
    }
    for (int64_t b = 0; b < a.nb; ++b) scan_tables_body<0>(a, b);
    for (int64_t g = 0; g < L.n[1]; ++g) scan_compose_body<0>(a.G, (const int32_t*)a.t0, L.n[0], wide + L.woff[1], g);
    for (int l = 2; l <= nl; ++l)
        for (int64_t g = 0; g < L.n[l]; ++g) scan_compose_body<0>(a.G, (const int64_t*)(wide + L.woff[l - 1]), L.n[l - 1], wide + L.woff[l], g);
    scan_top_body(a.G, wide + L.woff[nl], L.n[nl], Eall + L.eoff[nl]);
    for (int l = nl; l >= 2; --l)
        for (int64_t g = 0; g < L.n[l]; ++g) scan_expand_body(a.G, (const int64_t*)(wide + L.woff[l - 1]), L.n[l - 1], Eall + L.eoff[l], Eall + L.eoff[l - 1], g);
    for (int64_t g = 0; g < L.n[1]; ++g) scan_expand_body(a.G, (const int32_t*)a.t0, L.n[0], Eall + L.eoff[1], Eall + L.eoff[0], g);
    for (int64_t b = 0; b < a.nb; ++b) scan_finalize_body(a, b);
    return MDSP_OK;
}

// What the scan needs from the serial pilot: anchors of the first `pilot` outputs, the exact state of output `pilot`
// on the grid, the slope of the rounding drift, and how many blocks can still follow.
struct ScanSetup {
    arbscan::Grid G;
    std::vector<int64_t> ax;
    std::vector<double> aa;
    uint64_t As = 0;
    int64_t xs = 0, k0 = 0, nb = 0;
    double sigma = 0.0;
};
// false: not applicable (short stream, recurrence outside the integer model) -- the caller runs the serial loop
bool scan_setup(double acc, int64_t deficit, const ArbStep& st, int64_t nphi, int64_t xlen, int64_t pilot, ScanSetup& S) {
    using namespace arbscan;
    if (pilot < 2 * BLK || pilot % BLK) return false;
    if (!make_grid(st.delta, nphi, S.G)) return false;
    int64_t n = 0, xe = 0;
    double ae = 0.0;
    arb_trajectory(acc, deficit, st, xlen, ANCH, &S.ax, &S.aa, &n, &ae, &xe, pilot);
    if (n < pilot || xe > xlen) return false;   // the stream ends inside the pilot
    if (!to_grid(S.G, ae, S.As)) return false;
    S.xs = xe;
    S.k0 = pilot;
    // drift slope over outputs BLK .. pilot (both on the grid):  E = U(pilot) - U(BLK) - (pilot - BLK) D
    uint64_t A1 = 0;
    if (!to_grid(S.G, S.aa[BLK / ANCH], A1)) return false;
    const __int128 U1 = (__int128)A1 + (__int128)S.G.N * S.ax[BLK / ANCH], U0 = (__int128)S.As + (__int128)S.G.N * xe;
    const __int128 E = U0 - U1 - (__int128)(pilot - BLK) * (__int128)S.G.D;
    S.sigma = (double)E / (double)(pilot - BLK);
    // outputs that can follow output k0: xIdx first exceeds xlen after about ((xlen + 1 - xs) N - As) / D updates; two spare blocks
    const __int128 room = (__int128)(xlen + 1 - xe) * (__int128)S.G.N;
    const __int128 kmax = room / (__int128)S.G.D + 2;
    if (kmax > ((__int128)1 << 33)) return false;   // scan work space is ~40 B per 32 outputs: keep it within a few GiB, longer streams go serial
    S.nb = (int64_t)kmax / BLK + 2;
    return true;
}

// Host emulation of the whole scan (same integer bodies as the kernels), for the CPU tests.  Returns false when the caller
// has to fall back to the serial loop.
bool arb_scan_host(double acc, int64_t deficit, const ArbStep& st, int64_t nphi, int64_t xlen, int64_t pilot, std::vector<int64_t>& ax,
                   std::vector<double>& aa, int64_t* nout, double* acc_end, int64_t* xidx_end, int* passes) {
    using namespace arbscan;
    ScanSetup S;
    if (!scan_setup(acc, deficit, st, nphi, xlen, pilot, S)) return false;
    const ScanLayout L = scan_layout(S.nb, S.G.R);
    const int64_t nanch = S.k0 / ANCH + 2 * S.nb;
    ax = S.ax;
    aa = S.aa;
    ax.resize((size_t)nanch);
    aa.resize((size_t)nanch);
    std::vector<int32_t> t0((size_t)S.nb * S.G.R);
    std::vector<uint32_t> mind((size_t)S.nb);
    std::vector<int64_t> cb((size_t)S.nb), wide((size_t)L.wide_total), Eall((size_t)L.e_total), res(4, 0);
    ScanArgs a{};
    a.G = S.G;
    a.As = S.As;
    a.xs = S.xs;
    a.k0 = S.k0;
    a.xlen = xlen;
    a.nb = S.nb;
    a.sigma = S.sigma;
    a.t0 = t0.data();
    a.mind = mind.data();
    a.cb = cb.data();
    a.E = Eall.data() + L.eoff[0];
    a.baseA = reinterpret_cast<uint64_t*>(aa.data() + S.k0 / ANCH);
    a.baseW = ax.data() + S.k0 / ANCH;
    a.tab_x = ax.data();
    a.tab_acc = aa.data();
    a.result = res.data();
    for (int pass = 0; pass < 2; ++pass) {
        a.pass = pass;
        std::fill(res.begin(), res.end(), 0);
        scan_pass(a, L, wide.data(), Eall.data(), false, nullptr);
        if (passes) *passes = pass + 1;
        if (!(res[0] & 1)) break;
    }
    if ((res[0] & 1) || !(res[0] & 2)) return false;
    *nout = res[1];
    *xidx_end = res[2];
    std::memcpy(acc_end, &res[3], 8);
    ax.resize((size_t)cdiv(*nout, (int64_t)ANCH));
    aa.resize(ax.size());
    return true;
}

// Device scan for one call: pilot on the host, everything else on the stream; the anchors are written straight into the
// filter's device tables.  false: fall back to the serial loop (nothing the caller relies on has been modified).
int arb_scan_device(mdsp_firarb_s* f, const ArbStep& step, int64_t xlen, int64_t pilot, hipStream_t st, bool* used, int64_t* nout, double* acc_end,
                    int64_t* xidx_end) {
    using namespace arbscan;
    *used = false;
    ScanSetup S;
    if (!scan_setup(f->phi_acc, f->input_deficit, step, f->nphi, xlen, pilot, S)) return MDSP_OK;
    const ScanLayout L = scan_layout(S.nb, S.G.R);
    const int64_t nanch = S.k0 / ANCH + 2 * S.nb;
    MDSP_TRY(f->tab_x.reserve(sizeof(int64_t) * (size_t)nanch));
    MDSP_TRY(f->tab_acc.reserve(sizeof(double) * (size_t)nanch));
    MDSP_TRY(f->scan_t0.reserve(sizeof(int32_t) * (size_t)S.nb * S.G.R));
    MDSP_TRY(f->scan_mind.reserve(sizeof(uint32_t) * (size_t)S.nb));
    MDSP_TRY(f->scan_cb.reserve(sizeof(int64_t) * (size_t)S.nb));
    MDSP_TRY(f->scan_wide.reserve(sizeof(int64_t) * (size_t)L.wide_total));
    MDSP_TRY(f->scan_E.reserve(sizeof(int64_t) * (size_t)L.e_total));
    MDSP_TRY(f->scan_res.reserve(sizeof(int64_t) * 4));
    MDSP_HIP(hipStreamSynchronize(st));   // a previous launch on this stream may still read the anchor tables
    MDSP_HIP(hipMemcpy(f->tab_x.p, S.ax.data(), sizeof(int64_t) * S.ax.size(), hipMemcpyHostToDevice));
    MDSP_HIP(hipMemcpy(f->tab_acc.p, S.aa.data(), sizeof(double) * S.aa.size(), hipMemcpyHostToDevice));
    ScanArgs a{};
    a.G = S.G;
    a.As = S.As;
    a.xs = S.xs;
    a.k0 = S.k0;
    a.xlen = xlen;
    a.nb = S.nb;
    a.sigma = S.sigma;
    a.t0 = f->scan_t0.as<int32_t>();
    a.mind = f->scan_mind.as<uint32_t>();
    a.cb = f->scan_cb.as<int64_t>();
    a.E = f->scan_E.as<int64_t>() + L.eoff[0];
    a.baseA = reinterpret_cast<uint64_t*>(f->tab_acc.as<double>() + S.k0 / ANCH);
    a.baseW = f->tab_x.as<int64_t>() + S.k0 / ANCH;
    a.tab_x = f->tab_x.as<int64_t>();
    a.tab_acc = f->tab_acc.as<double>();
    a.result = f->scan_res.as<int64_t>();
    int64_t res[4] = {0, 0, 0, 0};
    for (int pass = 0; pass < 2; ++pass) {
        a.pass = pass;
        MDSP_HIP(hipMemsetAsync(f->scan_res.p, 0, sizeof(res), st));
        MDSP_TRY(scan_pass(a, L, f->scan_wide.as<int64_t>(), f->scan_E.as<int64_t>(), true, st));
        MDSP_HIP(hipMemcpyAsync(res, f->scan_res.p, sizeof(res), hipMemcpyDeviceToHost, st));
        MDSP_HIP(hipStreamSynchronize(st));
        if (!(res[0] & 1)) break;
    }
    if ((res[0] & 1) || !(res[0] & 2)) return MDSP_OK;   // ambiguous twice, or no end found: serial loop
    *nout = res[1];
    *xidx_end = res[2];
    std::memcpy(acc_end, &res[3], 8);
    *used = true;
    return MDSP_OK;
}

template <typename XS, typename A, typename R, int NCH> int arb_launch_n(mdsp_firarb_s* f, ArbArgs& a, int tile, int64_t span, hipStream_t st) {
    const int64_t taps_bytes = 2 * (int64_t)f->base.tp * f->nphi * (int64_t)sizeof(R);
    a.tile = tile;
    a.span = (int)span;
    const size_t lds_bytes = arb_rec_bytes(tile) + (size_t)span * NCH * sizeof(A) + (a.taps_in_lds ? (size_t)taps_bytes : 0);
    auto kern = arbitrary_fir_kernel<XS, A, R, NCH>;
    if (lds_bytes > 48 * 1024) MDSP_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    // channels share a tile's replayed trajectory (and, NCH at a time, its tap reads): loop over the channel groups inside the
    // workgroup unless there are too few tiles to fill the GPU
    const int64_t tiles = cdiv(a.nout, tile);
    const int64_t groups = cdiv(f->base.nch, NCH);
    const unsigned gy = (unsigned)std::min<int64_t>(groups, std::max<int64_t>(1, cdiv((int64_t)device_cu_count() * 8, tiles)));
    const dim3 grid((unsigned)tiles, gy);
    const bool prof = MDSP_DBG(arb_prof) && gy == 1;
    if (prof) {
        MDSP_TRY(f->prof.reserve(sizeof(long long) * 8 * (size_t)tiles));
        MDSP_HIP(hipMemsetAsync(f->prof.p, 0, sizeof(long long) * 8 * (size_t)tiles, st));
        a.prof = f->prof.as<long long>();
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, st, a);
    MDSP_LAUNCH_CHECK();
    if (prof) {   // phase durations (shader clocks) averaged over the workgroups of the middle half of the launch
        std::vector<long long> h((size_t)tiles * 8);
        MDSP_HIP(hipStreamSynchronize(st));
        MDSP_HIP(hipMemcpy(h.data(), f->prof.p, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        double sum[5] = {0, 0, 0, 0, 0};
        int64_t n = 0;
        long long tmin = LLONG_MAX, tmax = 0;
        for (int64_t t = 0; t < tiles; ++t) {
            tmin = std::min(tmin, h[t * 8]);
            tmax = std::max(tmax, h[t * 8 + 5]);
        }
        for (int64_t t = tiles / 4; t < tiles * 3 / 4; ++t, ++n)
            for (int k = 0; k < 5; ++k) sum[k] += (double)(h[t * 8 + k + 1] - h[t * 8 + (k == 2 ? 1 : k)]);
        fprintf(stderr, "[MDSP_ARB_PROF] tiles %lld, kernel span %lld clk; per workgroup (clk): x_first load %.0f, replay(wave0) %.0f, prologue->barrier %.0f, compute %.0f, drain %.0f\n",
                (long long)tiles, tmax - tmin, sum[0] / n, sum[1] / n, sum[2] / n, sum[3] / n, sum[4] / n);
    }
    return MDSP_OK;
}

template <typename XS, typename A, typename R> int arb_launch(mdsp_firarb_s* f, ArbArgs& a, hipStream_t st) {
    const int64_t taps_bytes = 2 * (int64_t)f->base.tp * f->nphi * (int64_t)sizeof(R);
    a.taps_in_lds = taps_bytes <= 32 * 1024;
    a.ablate = MDSP_DBG(ablate);
    a.prio = tunables().arb_prio;
    const int nch_max = tunables().arb_nch;   // tuning knob
    // channels per group: the most whose workgroup (trajectory records + interleaved input span of tile * Delta / Nphi + tp
    // samples per channel + tap pairs) stays within 44 KiB of LDS, i.e. leaves >= 3 workgroups per CU; measured on MI355X,
    // larger footprints lose more to occupancy than the shared tap reads save
    int nchg = f->base.nch >= 3 ? 4 : (int)std::max<int64_t>(1, f->base.nch);
    nchg = std::min(nchg, nch_max >= 4 ? 4 : nch_max >= 2 ? 2 : 1);
    int tile = 1024;
    if (tunables().arb_tile > 0) tile = std::min(64 * ARB_BLK, std::max(ARB_BLK, tunables().arb_tile / ARB_BLK * ARB_BLK));   // tuning knob (one wave replays the tile)
    const int tile0 = tile;
    int64_t span = 0;
    const auto span_of = [&](int t) { return ((int64_t)std::ceil((double)t * f->delta / (double)f->nphi) + f->base.tp + 4 + 3) & ~int64_t(3); };   // multiple of 4: the tap pairs that follow stay 16-byte aligned
    const int64_t fixed = (int64_t)arb_rec_bytes(tile0) + (a.taps_in_lds ? taps_bytes : 0);
    span = span_of(tile);
    while (nchg > 1 && fixed + span * nchg * (int64_t)sizeof(A) > 44 * 1024) nchg /= 2;
    if (nchg == 1) {   // one channel at a time: shrink the tile until its span fits 48 KiB
        while ((span = span_of(tile)) * (int64_t)sizeof(A) > 48 * 1024 && tile > ARB_BLK) tile /= 2;
        if (span * (int64_t)sizeof(A) > 48 * 1024) span = 0;   // very low rates: outputs are far apart, read through L2 instead
    }
    switch (nchg) {
        case 4: return arb_launch_n<XS, A, R, 4>(f, a, tile, span, st);
        case 2: return arb_launch_n<XS, A, R, 2>(f, a, tile, span, st);
        default: return arb_launch_n<XS, A, R, 1>(f, a, tile, span, st);
    }
}

int arb_dispatch(mdsp_firarb_s* f, ArbArgs& a, hipStream_t st) {
    const bool d = f->base.acc_double;
    switch (f->base.x_dtype) {
        case MDSP_F32: return d ? arb_launch<float, double, double>(f, a, st) : arb_launch<float, float, float>(f, a, st);
        case MDSP_F64: return arb_launch<double, double, double>(f, a, st);
        case MDSP_C32: return d ? arb_launch<cx<float>, cx<double>, double>(f, a, st) : arb_launch<cx<float>, cx<float>, float>(f, a, st);
        default: return arb_launch<cx<double>, cx<double>, double>(f, a, st);
    }
}

int upload_bank(DevBuf& buf, const std::vector<double>& pd, bool as_double) {
    if (as_double) {
        MDSP_TRY(buf.reserve(sizeof(double) * pd.size()));
        MDSP_HIP(hipMemcpy(buf.p, pd.data(), sizeof(double) * pd.size(), hipMemcpyHostToDevice));
    } else {
        std::vector<float> pf(pd.size());
        for (size_t i = 0; i < pd.size(); ++i) pf[i] = (float)pd[i];
        MDSP_TRY(buf.reserve(sizeof(float) * pf.size()));
        MDSP_HIP(hipMemcpy(buf.p, pf.data(), sizeof(float) * pf.size(), hipMemcpyHostToDevice));
    }
    return MDSP_OK;
}

}  // namespace

extern "C" {

int mdsp_arb_trajectory(double phi_acc, int64_t input_deficit, double rate, int64_t nphi, int64_t xlen, int64_t block, int64_t* anchors_x,
                        double* anchors_acc, int64_t anchors_cap, int64_t* nout, double* phi_acc_end, int64_t* input_deficit_end) {
    if (!(rate > 0.0) || nphi < 1 || xlen < 0 || input_deficit < 1 || block < 1) MDSP_FAIL(MDSP_ERR_ARGUMENT, "invalid trajectory arguments");
    if (!nout || !phi_acc_end || !input_deficit_end) MDSP_FAIL(MDSP_ERR_ARGUMENT, "NULL result pointer");
    if (xlen < input_deficit) {   // stream_filt.jl:586-590
        *nout = 0;
        *phi_acc_end = phi_acc;
        *input_deficit_end = input_deficit - xlen;
        return MDSP_OK;
    }
    const double delta = (double)nphi / rate;
    const ArbStep st = ArbStep::make(delta, (double)nphi);
    std::vector<int64_t> ax;
    std::vector<double> aa;
    int64_t xe = 0;
    arb_trajectory(phi_acc, input_deficit, st, xlen, block, anchors_x ? &ax : nullptr, anchors_x ? &aa : nullptr, nout, phi_acc_end, &xe);
    *input_deficit_end = xe - xlen;
    if (anchors_x) {
        if ((int64_t)ax.size() > anchors_cap || !anchors_acc) MDSP_FAIL(MDSP_ERR_ARGUMENT, "anchor buffers too small: need %lld", (long long)ax.size());
        std::copy(ax.begin(), ax.end(), anchors_x);
        std::copy(aa.begin(), aa.end(), anchors_acc);
    }
    return MDSP_OK;
}

// Host emulation of the device's parallel trajectory scan (same integer code, arb_scan.h): for the CPU tests.  *used = 0
// when the scan does not apply (short stream, recurrence outside the integer model, ambiguous twice) -- the product then
// runs the serial loop, and so should the caller.
int mdsp_arb_trajectory_scan(double phi_acc, int64_t input_deficit, double rate, int64_t nphi, int64_t xlen, int64_t pilot, int64_t* anchors_x,
                             double* anchors_acc, int64_t anchors_cap, int64_t* nout, double* phi_acc_end, int64_t* input_deficit_end, int* used,
                             int* passes) {
    if (!(rate > 0.0) || nphi < 1 || xlen < 0 || input_deficit < 1) MDSP_FAIL(MDSP_ERR_ARGUMENT, "invalid trajectory arguments");
    if (!nout || !phi_acc_end || !input_deficit_end || !used) MDSP_FAIL(MDSP_ERR_ARGUMENT, "NULL result pointer");
    *used = 0;
    if (xlen < input_deficit) return MDSP_OK;
    const double delta = (double)nphi / rate;
    const ArbStep st = ArbStep::make(delta, (double)nphi);
    std::vector<int64_t> ax;
    std::vector<double> aa;
    int64_t xe = 0;
    if (!arb_scan_host(phi_acc, input_deficit, st, nphi, xlen, pilot, ax, aa, nout, phi_acc_end, &xe, passes)) return MDSP_OK;
    *input_deficit_end = xe - xlen;
    if (anchors_x) {
        if ((int64_t)ax.size() > anchors_cap || !anchors_acc) MDSP_FAIL(MDSP_ERR_ARGUMENT, "anchor buffers too small: need %lld", (long long)ax.size());
        std::copy(ax.begin(), ax.end(), anchors_x);
        std::copy(aa.begin(), aa.end(), anchors_acc);
    }
    *used = 1;
    return MDSP_OK;
}

// Test helper: number of updates (out of nsteps from phi_acc) at which the branch-free update of the device replay
// (ArbStep::fast) differs from the reference-form update (ArbStep::operator()) in phase or in index increment.
int mdsp_arb_replay_check(double phi_acc, double rate, int64_t nphi, int64_t nsteps, int64_t* mismatches) {
    if (!(rate > 0.0) || nphi < 1 || nsteps < 0 || !mismatches) MDSP_FAIL(MDSP_ERR_ARGUMENT, "invalid arguments");
    const ArbStep st = ArbStep::make((double)nphi / rate, (double)nphi);
    double a1 = phi_acc, a2 = phi_acc;
    int64_t x1 = 0, x2 = 0, bad = 0;
    for (int64_t k = 0; k < nsteps; ++k) {
        st(a1, x1);
        st.fast(a2, x2);
        if (a1 != a2 || x1 != x2) {
            ++bad;
            a2 = a1;
            x2 = x1;
        }
    }
    *mismatches = bad;
    return MDSP_OK;
}

int mdsp_firarb_scan_stats(mdsp_firarb f, int64_t* scanned, int64_t* serial) {
    if (!f) MDSP_FAIL(MDSP_ERR_ARGUMENT, "handle is NULL");
    if (scanned) *scanned = f->n_scanned;
    if (serial) *serial = f->n_serial;
    return MDSP_OK;
}

int mdsp_firarb_create(mdsp_firarb* fo, const void* taps_host, int64_t hlen, double rate, int64_t nphi, int taps_dtype, int x_dtype, int64_t nch) {
    if (!fo) MDSP_FAIL(MDSP_ERR_ARGUMENT, "handle pointer is NULL");
    *fo = nullptr;
    if (!(rate > 0.0)) MDSP_FAIL(MDSP_ERR_DOMAIN, "rate must be greater than 0");          // stream_filt.jl:151
    if (!taps_host || hlen < 1) MDSP_FAIL(MDSP_ERR_ARGUMENT, "filter taps must be non-empty");
    if (nphi < 1 || nphi > (int64_t(1) << 20)) MDSP_FAIL(MDSP_ERR_ARGUMENT, "Nphi must be in [1, 2^20]");
    if (taps_dtype != MDSP_F32 && taps_dtype != MDSP_F64) MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "only real Float32/Float64 taps are supported on the device");
    if (!dtype_valid(x_dtype)) MDSP_FAIL(MDSP_ERR_ARGUMENT, "invalid x dtype");
    if (nch < 1 || nch > 65535) MDSP_FAIL(MDSP_ERR_ARGUMENT, "nch must be in [1, 65535]");
    std::unique_ptr<mdsp_firarb_s> f(new mdsp_firarb_s());
    mdsp_fir_s& b = f->base;
    b.kind = 4;
    b.L = nphi;
    b.M = 1;
    b.hlen = hlen;
    b.nch = nch;
    b.taps_dtype = taps_dtype;
    b.x_dtype = x_dtype;
    b.tp = cdiv(hlen, nphi);
    b.hl = b.tp - 1;
    b.acc_double = (taps_dtype == MDSP_F64) || dtype_is_double(x_dtype);
    b.out_dtype = dtype_is_complex(x_dtype) ? (b.acc_double ? MDSP_C64 : MDSP_C32) : (b.acc_double ? MDSP_F64 : MDSP_F32);
    f->rate = rate;
    f->nphi = nphi;
    f->delta = (double)nphi / rate;                                                         // :114
    // dh = [diff(h); 0] in the taps' own precision (:107), both banks through taps2pfb (:108-109, :294-307)
    const size_t np = (size_t)(b.tp * nphi);
    std::vector<double> pd(np, 0.0), dd(np, 0.0);
    auto tap = [&](int64_t i) -> double { return taps_dtype == MDSP_F32 ? (double)((const float*)taps_host)[i] : ((const double*)taps_host)[i]; };
    auto dtap = [&](int64_t i) -> double {
        if (i + 1 >= hlen) return 0.0;
        if (taps_dtype == MDSP_F32) return (double)(((const float*)taps_host)[i + 1] - ((const float*)taps_host)[i]);
        return ((const double*)taps_host)[i + 1] - ((const double*)taps_host)[i];
    };
    for (int64_t row = 0; row < b.tp; ++row)
        for (int64_t col = 0; col < nphi; ++col) {
            const int64_t hidx = (b.tp - 1 - row) * nphi + col;
            if (hidx < hlen) {
                pd[(size_t)(row * nphi + col)] = tap(hidx);
                dd[(size_t)(row * nphi + col)] = dtap(hidx);
            }
        }
    std::vector<double> both(2 * np);
    for (size_t i = 0; i < np; ++i) {
        both[2 * i] = pd[i];
        both[2 * i + 1] = dd[i];
    }
    MDSP_TRY(upload_bank(b.pfbT, both, b.acc_double));   // interleaved (pfb, dpfb) pairs
    const size_t hbytes = dtype_size(x_dtype) * (size_t)std::max<int64_t>(1, b.hl) * (size_t)nch;
    for (int k = 0; k < 2; ++k) {
        MDSP_TRY(b.hist[k].reserve(hbytes));
        MDSP_HIP(hipMemset(b.hist[k].p, 0, hbytes));
    }
    *fo = f.release();
    return MDSP_OK;
}

int mdsp_firarb_destroy(mdsp_firarb f) {
    delete f;
    return MDSP_OK;
}

int mdsp_firarb_reset(mdsp_firarb f) {   // reset! stream_filt.jl:260-276
    if (!f) MDSP_FAIL(MDSP_ERR_ARGUMENT, "handle is NULL");
    const mdsp_fir_s& b = f->base;
    MDSP_HIP(hipMemset(b.hist[b.cur].p, 0, dtype_size(b.x_dtype) * (size_t)std::max<int64_t>(1, b.hl) * (size_t)b.nch));
    f->phi_acc = 0.0;
    f->input_deficit = 1;
    f->x_idx = 1;
    return MDSP_OK;
}

int mdsp_firarb_setphase(mdsp_firarb f, double phi) {   // setphase! :231-239
    if (!f) MDSP_FAIL(MDSP_ERR_ARGUMENT, "handle is NULL");
    if (!(phi >= 0)) MDSP_FAIL(MDSP_ERR_DOMAIN, "phi must be >= 0");
    double whole = 0.0;
    const double frac = std::modf(phi, &whole);
    f->input_deficit += round_half_even(whole);
    f->phi_acc = frac * (double)f->nphi;
    return MDSP_OK;
}

int mdsp_firarb_timedelay(mdsp_firarb f, double* tau) {   // :400-401
    if (!f || !tau) MDSP_FAIL(MDSP_ERR_ARGUMENT, "NULL argument");
    *tau = (double)(f->base.hlen - 1) / (double)(2 * f->nphi);
    return MDSP_OK;
}

int mdsp_firarb_outputlength(mdsp_firarb f, int64_t inputlength, int64_t* outlen) {   // :340-342
    if (!f || !outlen) MDSP_FAIL(MDSP_ERR_ARGUMENT, "NULL argument");
    const double a = (double)(inputlength - f->input_deficit + 1) * f->rate;   // separate statements: no contraction
    const double b = f->phi_acc / f->delta;
    const double c = a - b;
    *outlen = (int64_t)std::ceil(c);
    return MDSP_OK;
}

int mdsp_firarb_inputlength(mdsp_firarb f, int64_t outputlength, int round_up, int64_t* inlen) {   // :385-389
    if (!f || !inlen) MDSP_FAIL(MDSP_ERR_ARGUMENT, "NULL argument");
    const int64_t d = round_up ? 1 : 0;
    const double b = f->phi_acc / f->delta;
    const double s = (double)(outputlength - d) + b;
    const double q = s / f->rate;
    *inlen = (int64_t)std::floor(q) + d + f->input_deficit - 1;
    return MDSP_OK;
}

int mdsp_firarb_info(mdsp_firarb f, int64_t* nphi, int64_t* taps_per_phase, int64_t* history_len, int* out_dtype, double* delta) {
    if (!f) MDSP_FAIL(MDSP_ERR_ARGUMENT, "handle is NULL");
    if (nphi) *nphi = f->nphi;
    if (taps_per_phase) *taps_per_phase = f->base.tp;
    if (history_len) *history_len = f->base.hl;
    if (out_dtype) *out_dtype = f->base.out_dtype;
    if (delta) *delta = f->delta;
    return MDSP_OK;
}

int mdsp_firarb_get_state(mdsp_firarb f, double* phi_acc, double* alpha, int64_t* phi_idx, int64_t* input_deficit, int64_t* x_idx,
                          void* history_host) {
    if (!f) MDSP_FAIL(MDSP_ERR_ARGUMENT, "handle is NULL");
    const double fl = std::floor(f->phi_acc);
    if (phi_acc) *phi_acc = f->phi_acc;
    if (alpha) *alpha = f->phi_acc - fl;               // modf(phiAcc)[1]  (:576, :238)
    if (phi_idx) *phi_idx = 1 + (int64_t)fl;           // 1 + Int(foffset) (:577, :237)
    if (input_deficit) *input_deficit = f->input_deficit;
    if (x_idx) *x_idx = f->x_idx;
    const mdsp_fir_s& b = f->base;
    if (history_host && b.hl > 0) {
        MDSP_HIP(hipDeviceSynchronize());
        MDSP_HIP(hipMemcpy(history_host, b.hist[b.cur].p, dtype_size(b.x_dtype) * (size_t)b.hl * (size_t)b.nch, hipMemcpyDeviceToHost));
    }
    return MDSP_OK;
}

int mdsp_firarb_set_state(mdsp_firarb f, double phi_acc, int64_t input_deficit, const void* history_host) {
    if (!f) MDSP_FAIL(MDSP_ERR_ARGUMENT, "handle is NULL");
    if (!(phi_acc >= 0.0) || !(phi_acc < (double)f->nphi) || input_deficit < 1) MDSP_FAIL(MDSP_ERR_ARGUMENT, "state out of range");
    f->phi_acc = phi_acc;
    f->input_deficit = input_deficit;
    const mdsp_fir_s& b = f->base;
    if (history_host && b.hl > 0) {
        MDSP_HIP(hipDeviceSynchronize());
        MDSP_HIP(hipMemcpy(b.hist[b.cur].p, history_host, dtype_size(b.x_dtype) * (size_t)b.hl * (size_t)b.nch, hipMemcpyHostToDevice));
    }
    return MDSP_OK;
}

int mdsp_firarb_exec(mdsp_firarb f, const void* x_dev, int64_t xlen, int64_t ldx, void* y_dev, int64_t ycap, int64_t ldy, int64_t* nwritten,
                     void* stream) {
    if (!f) MDSP_FAIL(MDSP_ERR_ARGUMENT, "handle is NULL");
    if (xlen < 0 || ycap < 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "negative size");
    mdsp_fir_s& b = f->base;
    if (b.nch > 1 && ldx < xlen) MDSP_FAIL(MDSP_ERR_ARGUMENT, "ldx smaller than xlen");
    hipStream_t st = as_stream(stream);
    if (nwritten) *nwritten = 0;
    if (xlen < f->input_deficit) {   // stream_filt.jl:586-590
        MDSP_TRY(shiftin_dispatch(&b, x_dev, xlen, ldx, st));
        f->input_deficit -= xlen;
        return MDSP_OK;
    }
    const ArbStep step = ArbStep::make(f->delta, (double)f->nphi);
    const bool hit = f->cache_valid && f->c_acc0 == f->phi_acc && f->c_def0 == f->input_deficit && f->c_xlen == xlen;
    if (!hit) {
        int64_t nout = 0, xe = 0;
        double ae = 0.0;
        bool scanned = false;
        f->cache_valid = false;
        // long streams: the recurrence is evaluated in parallel on the device, bit for bit (arb_scan.h); MDSP_ARB_SCAN=0 and
        // MDSP_ARB_SCAN_MIN (outputs) are test / tuning knobs
        const int64_t scan_min = tunables().arb_scan_min;
        const int64_t pilot = std::min<int64_t>(65536, std::max<int64_t>(2 * arbscan::BLK, (scan_min / 4) & ~int64_t(arbscan::BLK - 1)));
        if (tunables().arb_scan != 0 && (double)xlen * f->rate >= (double)scan_min)
            MDSP_TRY(arb_scan_device(f, step, xlen, pilot, st, &scanned, &nout, &ae, &xe));
        if (scanned) {
            ++f->n_scanned;
        } else {
            std::vector<int64_t> ax;
            std::vector<double> aa;
            ax.reserve((size_t)((double)xlen * f->rate / ARB_BLK) + 16);
            aa.reserve(ax.capacity());
            arb_trajectory(f->phi_acc, f->input_deficit, step, xlen, ARB_BLK, &ax, &aa, &nout, &ae, &xe);
            MDSP_TRY(f->tab_x.reserve(sizeof(int64_t) * std::max<size_t>(1, ax.size())));
            MDSP_TRY(f->tab_acc.reserve(sizeof(double) * std::max<size_t>(1, aa.size())));
            MDSP_HIP(hipStreamSynchronize(st));   // a previous launch on this stream may still read the anchor tables
            if (!ax.empty()) {
                MDSP_HIP(hipMemcpy(f->tab_x.p, ax.data(), sizeof(int64_t) * ax.size(), hipMemcpyHostToDevice));
                MDSP_HIP(hipMemcpy(f->tab_acc.p, aa.data(), sizeof(double) * aa.size(), hipMemcpyHostToDevice));
            }
            ++f->n_serial;
        }
        f->c_acc0 = f->phi_acc;
        f->c_def0 = f->input_deficit;
        f->c_xlen = xlen;
        f->c_nout = nout;
        f->c_acc_end = ae;
        f->c_xidx_end = xe;
        f->c_def_end = xe - xlen;
        f->cache_valid = true;
    }
    const int64_t nout = f->c_nout;
    // the reference indexes buffer[bufIdx] unchecked beyond allocate_output's outputlength + 1 (:639-655): a short
    // buffer is a BoundsError there, an ArgumentError here -- raised before anything is written
    if (ycap < nout) MDSP_FAIL(MDSP_ERR_ARGUMENT, "buffer is too small: need %lld, have %lld", (long long)nout, (long long)ycap);
    if (b.nch > 1 && ldy < nout) MDSP_FAIL(MDSP_ERR_ARGUMENT, "ldy smaller than the output length");
    if (nout > 0) {
        if (!x_dev || !y_dev) MDSP_FAIL(MDSP_ERR_ARGUMENT, "NULL buffer");
        ArbArgs a{};
        a.x = x_dev;
        a.hist = b.hist[b.cur].p;
        a.y = y_dev;
        a.taps2 = b.pfbT.p;
        a.nch = b.nch;
        a.tab_x = f->tab_x.as<int64_t>();
        a.tab_acc = f->tab_acc.as<double>();
        a.xlen = xlen;
        a.ldx = ldx;
        a.ldy = ldy;
        a.nout = nout;
        a.step = step;
        a.nphi = (int)f->nphi;
        a.tp = (int)b.tp;
        a.hl = (int)b.hl;
        MDSP_TRY(arb_dispatch(f, a, st));
    }
    f->phi_acc = f->c_acc_end;
    f->x_idx = f->c_xidx_end;
    f->input_deficit = f->c_def_end;                                       // :619
    MDSP_TRY(shiftin_dispatch(&b, x_dev, xlen, ldx, st));                  // :620
    if (nwritten) *nwritten = nout;
    return MDSP_OK;
}

// ---- stateful time-domain FIR (DF2TFilter with FIR coefficients) and filtfilt's odd extension --------------------
}  // extern "C"

namespace {

// Device copies of the taps of the recently used stateful filters: a streaming caller passes the same coefficients chunk after chunk
// (often one sample at a time), so the upload -- and any allocation -- happens once per filter, not once per call.  Entries live in the
// library's LRU (plancache.hip), keyed by (device, thread, stream, tap bytes): filters with different taps that alternate each keep
// their buffer, nothing synchronises the device, and no lock is held across a launch.
template <typename A, typename R>
int tdfir_state_run(const void* taps_host, int64_t nb, const void* x, int64_t nx, int64_t ncols, int64_t ldx, void* y, int64_t ldy, void* si, hipStream_t st) {
    const size_t bytes = sizeof(R) * (size_t)nb;
    std::string key = plan_cache_key('t', st);
    key.append(static_cast<const char*>(taps_host), bytes);
    void* h = nullptr;
    MDSP_TRY(plan_cache_get(key, &h,
                            [&](void** out) -> int {
                                auto buf = new DevBuf();
                                int rc = buf->reserve(bytes);
                                if (rc == MDSP_OK && hipMemcpyAsync(buf->p, taps_host, bytes, hipMemcpyHostToDevice, st) != hipSuccess)
                                    rc = set_error(MDSP_ERR_DEVICE, "tap upload failed");
                                if (rc == MDSP_OK && hipStreamSynchronize(st) != hipSuccess) rc = set_error(MDSP_ERR_DEVICE, "tap upload failed");   // taps_host may be freed on return
                                if (rc != MDSP_OK) { delete buf; return rc; }
                                *out = buf;
                                return MDSP_OK;
                            },
                            [](void* p) { delete static_cast<DevBuf*>(p); }));
    const R* taps = static_cast<DevBuf*>(h)->as<R>();
    if (nx > 0) {
        const dim3 g((unsigned)std::min<int64_t>(cdiv(nx, 256), 4096), (unsigned)ncols);
        hipLaunchKernelGGL((tdfir_state_out_kernel<A, R>), g, dim3(256), 0, st, (const A*)x, (const A*)si, (A*)y, taps, nx, ldx, ldy, (int)nb);
        MDSP_LAUNCH_CHECK();
        hipLaunchKernelGGL((tdfir_state_next_kernel<A, R>), dim3((unsigned)ncols), dim3(256), 0, st, (const A*)x, (A*)si, taps, nx, ldx, (int)nb);
        MDSP_LAUNCH_CHECK();
    }
    return MDSP_OK;
}

template <typename A> int extrapolate_run(const void* x, int64_t n, int64_t ncols, int64_t ldx, void* out, int64_t ldo, int64_t pad, hipStream_t st) {
    const dim3 g((unsigned)std::min<int64_t>(cdiv(n + 2 * pad, 256), 65535), (unsigned)ncols);
    hipLaunchKernelGGL(extrapolate_kernel<A>, g, dim3(256), 0, st, (const A*)x, (A*)out, n, ldx, ldo, pad);
    MDSP_LAUNCH_CHECK();
    return MDSP_OK;
}

}  // namespace

extern "C" {

int mdsp_tdfir_state_exec(const void* taps_host, int64_t nb, int dtype, const void* x_dev, int64_t nx, int64_t ncols, int64_t ldx, void* y_dev,
                          int64_t ldy, void* si_dev, void* stream) {
    if (!taps_host || nb < 1) MDSP_FAIL(MDSP_ERR_ARGUMENT, "filter vector b must be non-empty");
    if (!dtype_valid(dtype)) MDSP_FAIL(MDSP_ERR_ARGUMENT, "invalid dtype");
    if (nx < 0 || ncols < 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "negative size");
    if (nb < 2) MDSP_FAIL(MDSP_ERR_ARGUMENT, "a one-tap filter has no state; scale the signal instead");   // mul!(out, x, b[1]), filt.jl:163
    if (nb - 1 > 256 * 16) MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "stateful time-domain FIR supports at most 4097 taps");
    if (nx == 0 || ncols == 0) return MDSP_OK;
    if (ncols > 65535) MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "more than 65535 columns per call");
    if (!x_dev || !y_dev || !si_dev) MDSP_FAIL(MDSP_ERR_ARGUMENT, "NULL buffer");
    hipStream_t st = as_stream(stream);
    switch (dtype) {
        case MDSP_F32: return tdfir_state_run<float, float>(taps_host, nb, x_dev, nx, ncols, ldx, y_dev, ldy, si_dev, st);
        case MDSP_F64: return tdfir_state_run<double, double>(taps_host, nb, x_dev, nx, ncols, ldx, y_dev, ldy, si_dev, st);
        case MDSP_C32: return tdfir_state_run<cx<float>, float>(taps_host, nb, x_dev, nx, ncols, ldx, y_dev, ldy, si_dev, st);
        default: return tdfir_state_run<cx<double>, double>(taps_host, nb, x_dev, nx, ncols, ldx, y_dev, ldy, si_dev, st);
    }
}

int mdsp_extrapolate(const void* x_dev, int64_t n, int64_t ncols, int64_t ldx, int dtype, int64_t pad, void* out_dev, int64_t ldo, void* stream) {
    if (!dtype_valid(dtype)) MDSP_FAIL(MDSP_ERR_ARGUMENT, "invalid dtype");
    if (n < 1 || pad < 0 || pad > n - 1) MDSP_FAIL(MDSP_ERR_ARGUMENT, "pad_length must be in [0, length(x) - 1]");     // sig[2 + pad - i] must exist
    if (ldo < n + 2 * pad) MDSP_FAIL(MDSP_ERR_ARGUMENT, "output is incorrectly sized");                               // filt.jl:245
    if (ncols <= 0) return MDSP_OK;
    if (ncols > 65535) MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "more than 65535 columns per call");
    if (!x_dev || !out_dev) MDSP_FAIL(MDSP_ERR_ARGUMENT, "NULL buffer");
    hipStream_t st = as_stream(stream);
    switch (dtype) {
        case MDSP_F32: return extrapolate_run<float>(x_dev, n, ncols, ldx, out_dev, ldo, pad, st);
        case MDSP_F64: return extrapolate_run<double>(x_dev, n, ncols, ldx, out_dev, ldo, pad, st);
        case MDSP_C32: return extrapolate_run<cx<float>>(x_dev, n, ncols, ldx, out_dev, ldo, pad, st);
        default: return extrapolate_run<cx<double>>(x_dev, n, ncols, ldx, out_dev, ldo, pad, st);
    }
}

// ---- time-domain FIR: filt(b, a::Number, x) / tdfilt (dspbase.jl:95-105) -------------------------------------
// taps_host: nb REAL taps in the precision of `dtype`; x / y of `dtype`.  Same accumulation order as the TDF-II
// recursion: oldest sample first, one multiply then a chain of fused multiply-adds.
int mdsp_tdfir_exec(const void* taps_host, int64_t nb, int dtype, const void* x_dev, int64_t nx, int64_t ncols, int64_t ldx, void* y_dev,
                    int64_t ldy, void* stream) {
    if (!taps_host || nb < 1) MDSP_FAIL(MDSP_ERR_ARGUMENT, "filter vector b must be non-empty");
    if (!dtype_valid(dtype)) MDSP_FAIL(MDSP_ERR_ARGUMENT, "invalid dtype");
    if (nx < 0 || ncols < 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "negative size");
    if (nx == 0 || ncols == 0) return MDSP_OK;
    if (ncols > 65535) MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "more than 65535 columns per call");
    // The filter object (device taps, history) comes from the library's LRU, keyed by (device, thread, stream, taps, dtype, columns):
    // repeated filt(b, 1, x) / tdfilt / direct-conv calls allocate and upload nothing, and nothing synchronises the caller's stream.
    std::string key = plan_cache_key('f', stream);
    key.append(reinterpret_cast<const char*>(&dtype), sizeof(dtype));
    key.append(reinterpret_cast<const char*>(&ncols), sizeof(ncols));
    key.append(static_cast<const char*>(taps_host), (size_t)nb * dtype_size(dtype_real_of(dtype)));
    void* h = nullptr;
    MDSP_TRY(plan_cache_get(key, &h, [&](void** out) { return mdsp_fir_create(reinterpret_cast<mdsp_fir*>(out), taps_host, nb, 1, 1, dtype_real_of(dtype), dtype, ncols); },
                            [](void* p) { (void)mdsp_fir_destroy(static_cast<mdsp_fir>(p)); }));
    mdsp_fir f = static_cast<mdsp_fir>(h);
    MDSP_TRY(fir_reset_on(f, as_stream(stream), true));   // zero initial state: every call is a fresh filt(b, a, x)
    int64_t nw = 0;
    return mdsp_fir_exec(f, x_dev, nx, ldx, y_dev, nx, ldy, &nw, stream);
}

}  // extern "C"
