// Welch sums at nfft = R0 x S in TWO kernels (round 6): what the fused column step of spectral_ctcols*.hip costs from R0 = 5 on -- every one of the R0 workgroups
// of a group reads the whole frame, R0^2 segment reads per frame -- is replaced by one trip through a work buffer (the multi-pass engine of bigfft.hip takes three):
//   columns : one thread per column i forms, from the R0 windowed segments x[S n1 + i] w[S n1 + i] (two real frames per transform), the R0-point DFT over n1,
//             multiplies bin k1 by W_nfft^{i k1} and writes row k1 -- every sample read once, 8 bytes per point written;
//   rows    : the single-workgroup Welch kernel of spectral_ctbig.hip over the rows -- complex frames of S points, (channel, k1) as its channels -- summing
//             |Z|^2 into Float64 partial rows that persist over the chunks of a call;
//   the sums leave in natural order: X[k1 + R0 k2] = FFT_S(row k1)[k2].
// The work buffer is cut into chunks of MDSP_BIG_CHUNK_MIB (1 GiB; chunks of 128 MiB, to read the rows back from the Infinity Cache, measured 5 - 15 % SLOWER: more
// launches, no cache effect -- r06s47).  Taken from R0 = 6 (spectral.hip ctrows_r0: the fused step wins to R0 = 4, a tie at 5).  (Also measured and dropped, r06s51: the row kernel of chunk c on a second stream beside the column kernel of chunk c + 1, six chunks on
// two buffers -- 0.27 - 0.38 TB/s against 0.35 - 0.53 in order: at these call lengths the extra launches and event waits cost more than the overlap returns.)
// Float32 / ComplexF32 with rows of 8193 .. 16384 points, Float64 / ComplexF64 with rows of 4097 .. 9600;
// S any size of ctbig_sizes.h, R0 any radix fft_lds.h has a butterfly for (2 .. 32): 125000 = 8 x 15625, 200000 = 16 x 12500, 2^19 = 32 x 16384.
// Reference loops: periodograms.jl:746-759 (welch_pgram_helper!), :57-69 (ArraySplit), :142-172 (fft2pow!).
#include <algorithm>

#include "common.h"
#include "devio.h"
#include "fft_lds.h"
#include "hostfft.h"
#include "spectral_ctrows.h"

using namespace mdsp;
using mdsp::fft::cx;

namespace {

struct RowsColArgs {
    const void* s;        // signal, channel stride lds_
    void* work;           // cx<R> [ch][k1][unit][S]
    const void* win;      // R: nfft values (ones without a window, zero tail)
    const void* rootsN;   // cx<R>: the nfft forward roots
    int64_t lds_, K, hop, u0, cnt;   // frames of the call, first unit and units of this chunk
    int n, nfft, S;
};

template <int R0, bool CPLX, typename R> __global__ __launch_bounds__(256) void rows_col_kernel(RowsColArgs a) {
    using TT = std::conditional_t<CPLX, cx<R>, R>;
    constexpr int SZ = (int)sizeof(TT);
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    const int64_t ul = blockIdx.y, ch = blockIdx.z, u = a.u0 + ul;
    const int64_t f0 = CPLX ? u : 2 * u;
    const TT* fa = static_cast<const TT*>(a.s) + ch * a.lds_ + f0 * a.hop;
    const bool haveB = !CPLX && (f0 + 1) < a.K;
    const __amdgpu_buffer_rsrc_t da = io::make_rsrc(fa, (long long)a.n * SZ);
    const __amdgpu_buffer_rsrc_t db = io::make_rsrc(fa + (CPLX ? 0 : a.hop), haveB ? (long long)a.n * SZ : 0);
    const __amdgpu_buffer_rsrc_t dw = io::make_rsrc(a.win, (long long)a.nfft * (int)sizeof(R));
    const cx<R>* rootsN = static_cast<const cx<R>*>(a.rootsN);
    if (i >= a.S) return;
    cx<R> z[R0];
    TT ra[R0], rb[CPLX ? 1 : R0];
    R w[R0];
#pragma unroll
    for (int n1 = 0; n1 < R0; ++n1) {
        const int idx = a.S * n1 + i;
        ra[n1] = io::Ld<TT>::load(da, idx * SZ);
        if constexpr (!CPLX) rb[n1] = io::Ld<TT>::load(db, idx * SZ);
        w[n1] = io::Ld<R>::load(dw, idx * (int)sizeof(R));
    }
#pragma unroll
    for (int n1 = 0; n1 < R0; ++n1) {
        if constexpr (CPLX) z[n1] = {ra[n1].x * w[n1], ra[n1].y * w[n1]};
        else z[n1] = {ra[n1] * w[n1], rb[n1] * w[n1]};
    }
    fft::gen_bfly<R0>(z);
    cx<R>* o = static_cast<cx<R>*>(a.work) + ((ch * R0) * a.cnt + ul) * (int64_t)a.S + i;
    fft::st2(o, z[0]);
    // W_nfft^{i k1}: a lane's k1-th twiddle sits k1 table entries from its neighbour's -- fetched directly, the R0 - 1 twiddles of a wave touch ~R0^2 / 2 cache lines
    // (r06s54: this kernel at 0.35 ms of the 0.55 a 2^26-sample call takes).  Only the powers of two are fetched (W^{i 2^j}: R0 lines in all); the others are products
    // of at most log2(R0) of them -- two or three roundings more on a twiddle.
    constexpr int NB = R0 <= 2 ? 1 : R0 <= 4 ? 2 : R0 <= 8 ? 3 : R0 <= 16 ? 4 : 5;
    cx<R> wp[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) wp[j] = rootsN[(unsigned)(((unsigned long long)(unsigned)i << j) % (unsigned)a.nfft)];
#pragma unroll
    for (int k1 = 1; k1 < R0; ++k1) {
        cx<R> w{(R)1, (R)0};
        bool first = true;
#pragma unroll
        for (int j = 0; j < NB; ++j)
            if (k1 & (1 << j)) {
                w = first ? wp[j] : fft::cmul(w, wp[j]);
                first = false;
            }
        fft::st2(o + (int64_t)k1 * a.cnt * a.S, fft::cmul(z[k1], w));
    }
}

// acc[ch][k1 + R0 k2] (+)= sum_slot partial[slot][ch R0 + k1][k2], fixed order; 32 x 32 tiles through LDS (runs along k2 in, along k1 out)
__global__ __launch_bounds__(256) void rows_reduce_kernel(const double* __restrict__ part, double* __restrict__ acc, int R0, int S, int nslots, int64_t nrows, int add) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int k2_0 = blockIdx.x * 32, k1_0 = blockIdx.y * 32;
    const int64_t ch = blockIdx.z;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k1 = k1_0 + ty + 8 * j, k2 = k2_0 + tx;
        double v = 0.0;
        if (k1 < R0 && k2 < S)
            for (int sl = 0; sl < nslots; ++sl) v += part[((int64_t)sl * nrows + ch * R0 + k1) * S + k2];
        tile[ty + 8 * j][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k1 = k1_0 + tx, k2 = k2_0 + ty + 8 * j;
        if (k1 < R0 && k2 < S) {
            double* o = acc + ch * (int64_t)R0 * S + (int64_t)k1 + (int64_t)R0 * k2;
            *o = add ? *o + tile[tx][ty + 8 * j] : tile[tx][ty + 8 * j];
        }
    }
}

template <typename R> __global__ __launch_bounds__(256) void rows_window_kernel(const double* __restrict__ win, R* __restrict__ out, int n, int nfft) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i < nfft) out[i] = i < n ? (win ? (R)win[i] : (R)1) : (R)0;
}

template <bool CPLX, typename R> int launch_cols(int R0, const RowsColArgs& a, int64_t nch, hipStream_t st) {
    const dim3 grid((unsigned)cdiv(a.S, 256), (unsigned)a.cnt, (unsigned)nch);
    switch (R0) {
#define MDSP_X(R_)                                                                  \
    case R_:                                                                        \
        hipLaunchKernelGGL((rows_col_kernel<R_, CPLX, R>), grid, dim3(256), 0, st, a); \
        break;
        MDSP_X(2) MDSP_X(3) MDSP_X(4) MDSP_X(5) MDSP_X(6) MDSP_X(7) MDSP_X(8) MDSP_X(9) MDSP_X(10) MDSP_X(12) MDSP_X(14) MDSP_X(15) MDSP_X(16) MDSP_X(18) MDSP_X(20) MDSP_X(21)
        MDSP_X(24) MDSP_X(25) MDSP_X(27) MDSP_X(28) MDSP_X(30) MDSP_X(32)
#undef MDSP_X
        default: MDSP_FAIL(MDSP_ERR_ASSERTION, "no column kernel of %d rows", R0);
    }
    MDSP_LAUNCH_CHECK();
    return MDSP_OK;
}

bool radix_ok(int r) {
    switch (r) {
        case 2: case 3: case 4: case 5: case 6: case 7: case 8: case 9: case 10: case 12: case 14: case 15: case 16: case 18: case 20: case 21: case 24: case 25: case 27:
        case 28: case 30: case 32: return true;
        default: return false;
    }
}

}  // namespace

namespace mdsp {
// rows: Float32 -- a single-workgroup schedule above 8192 points; Float64 (round 6, later) -- one above 4096 points whose buffer of 16-byte elements fits (to 9600)
static bool row_ok(bool dbl, int64_t S) { return dbl ? (S > 4096 && ctbig_ok(MDSP_C64, S)) : (S > 8192 && ctbig_ok(MDSP_C32, S)); }

int ctrows_split(int dtype, int64_t nfft, int r0_min) {
    const bool dbl = dtype_is_double(dtype);
    for (int R0 = std::max(2, r0_min); R0 <= 32; ++R0)
        if (nfft % R0 == 0 && radix_ok(R0) && row_ok(dbl, nfft / R0)) return R0;
    return 0;
}

template <typename R>
static int ctrows_run(CtRowsPlan& rp, bool cplx, int R0, const void* s, int64_t lds_, int64_t K, int64_t hop, int64_t nch, int n, int64_t nfft, const double* win_dev, double* acc,
                      bool fresh, hipStream_t st) {
    const int64_t S = nfft / R0;
    const int64_t units = cplx ? K : cdiv(K, 2);
    if (units == 0) return MDSP_OK;
    if (!rp.ready) {
        std::vector<cx<R>> w((size_t)nfft);
        for (int64_t k = 0; k < nfft; ++k) {
            const zd r = unit_root(k, nfft, -1);
            w[(size_t)k] = {(R)r.real(), (R)r.imag()};
        }
        MDSP_TRY(rp.rootsN.reserve(sizeof(cx<R>) * (size_t)nfft));
        MDSP_HIP(hipMemcpy(rp.rootsN.p, w.data(), sizeof(cx<R>) * (size_t)nfft, hipMemcpyHostToDevice));
        MDSP_TRY(rp.win.reserve(sizeof(R) * (size_t)nfft));
        hipLaunchKernelGGL(rows_window_kernel<R>, dim3((unsigned)cdiv(nfft, 256)), dim3(256), 0, st, win_dev, rp.win.as<R>(), n, (int)nfft);
        MDSP_LAUNCH_CHECK();
        rp.ready = true;
    }
    const int rowtype = sizeof(R) == 8 ? MDSP_C64 : MDSP_C32;
    const int64_t per = (int64_t)sizeof(cx<R>) * nfft * nch;   // work bytes per unit
    const int mib = std::max(16, tunables().big_chunk_mib);
    const int64_t C = std::max<int64_t>(1, std::min<int64_t>(units, ((int64_t)mib << 20) / per));
    MDSP_TRY(rp.work.reserve((size_t)(per * C)));
    int64_t nslots0 = 0;
    for (int64_t c0 = 0; c0 < units; c0 += C) {
        const int64_t cnt = std::min<int64_t>(C, units - c0);
        RowsColArgs a{};
        a.s = s; a.work = rp.work.p; a.win = rp.win.p; a.rootsN = rp.rootsN.p;
        a.lds_ = lds_; a.K = K; a.hop = hop; a.u0 = c0; a.cnt = cnt; a.n = n; a.nfft = (int)nfft; a.S = (int)S;
        MDSP_TRY(cplx ? (launch_cols<true, R>(R0, a, nch, st)) : (launch_cols<false, R>(R0, a, nch, st)));
        int64_t nslots = 0;
        // rows: (channel, k1) are the row kernel's channels, `cnt` frames of S points S apart; its partial rows persist from chunk to chunk (the first chunk is the
        // largest: it writes every slot a later one adds to)
        MDSP_TRY(ctbig_welch(rp.rows, rowtype, rp.work.p, cnt * S, cnt, S, nch * R0, (int)S, S, nullptr, st, &nslots, &rp.partial, c0 > 0 ? 1 : 0));
        if (c0 == 0) nslots0 = nslots;
    }
    const dim3 grid((unsigned)cdiv(S, 32), (unsigned)cdiv(R0, 32), (unsigned)nch);
    hipLaunchKernelGGL(rows_reduce_kernel, grid, dim3(256), 0, st, rp.partial.as<double>(), acc, R0, (int)S, (int)nslots0, nch * R0, fresh ? 0 : 1);
    MDSP_LAUNCH_CHECK();
    return MDSP_OK;
}

int ctrows_welch(CtRowsPlan& rp, int dtype, int R0, const void* s, int64_t lds_, int64_t K, int64_t hop, int64_t nch, int n, int64_t nfft, const double* win_dev,
                 double* acc, bool fresh, hipStream_t st) {
    const bool dbl = dtype_is_double(dtype), cplx = dtype_is_complex(dtype);
    if (nfft % R0 || !radix_ok(R0) || !row_ok(dbl, nfft / R0)) MDSP_FAIL(MDSP_ERR_ASSERTION, "nfft=%lld is not %d rows of a single-workgroup size", (long long)nfft, R0);
    return dbl ? ctrows_run<double>(rp, cplx, R0, s, lds_, K, hop, nch, n, nfft, win_dev, acc, fresh, st)
               : ctrows_run<float>(rp, cplx, R0, s, lds_, K, hop, nch, n, nfft, win_dev, acc, fresh, st);
}
}  // namespace mdsp
