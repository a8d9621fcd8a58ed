// Multi-pass fused spectral engine: Welch sums, STFT / spectrogram / periodogram columns for transforms of 2^13 .. 2^30 points (bigfft.h).
//
// Reference loops being replaced (src/periodograms.jl) -- the same four as spectral.hip, at the sizes the reference's DEFAULT arguments ask for
// (n = length(s) >> 3, nfft = nextfastfft(n): :560, :647, :828, :872):
//   K4  ArraySplit getindex :57-69     buf[i] = s[offset+i] * window[i], zero tail up to nfft   -> the loads of pass 0
//   F3  mul!(outbuf, plan, sig) :754, :888                                                       -> P passes of bigfft_pass.h, one launch each
//   K5  fft2pow! :142-172              out[i] = muladd(abs2(X[i]), m, out[i])                    -> the stores of the last pass
//   K6  fft2oneortwosided! :234-244    raw column, conjugate mirror for real -> two-sided       -> the stores of the last pass / big_untangle_kernel
//
// HBM traffic per transform of N complex points (Float32: 8 N bytes): pass 0 reads the frames (and the window) and writes 8 N, every middle
// pass reads and writes 8 N in place, the last pass reads 8 N -- against window + three rocFFT kernels + abs2 over separate buffers before.
// Transforms are processed in groups of MDSP_BIG_CHUNK_MIB of work buffer (default 1 GiB: the default call's 8 transforms of 2^24 points in one
// group).  Measured on the default Welch call at 2^27 samples (profiles/r05_bigfft_sessions.json): 1.77 ms with everything in one group against
// 2.7 ms with a 128 MiB group -- fewer launches and one accumulator update instead of eight; no Infinity Cache effect was visible at any size.
// The same profile has the bound of the three-pass form: loads 1.0 ms + stores 0.5 ms + butterflies 0.5 ms when each runs alone, 1.8 ms together.
//
// Real signals ride two frames per transform (z = w (a + i b), as in the single-workgroup kernels): Welch needs no untangling, columns are
// untangled by big_untangle_kernel from the natural-order spectrum the last pass leaves.
#include <algorithm>
#include <memory>
#include <vector>

#include "bigfft.h"
#include "bigfft_plan.h"
#include "welch_plan.h"

using namespace mdsp;
using mdsp::fft::cx;

namespace mdsp {
// ols.hip: the row kernel of the long-filter convolution and the root table of its transforms
int ols_rows(int dbl, void* work, int64_t rows, int hrows, const void* Hrows, const void* table, const void* rt0, const void* rt1, int rlogS, hipStream_t st);
int ols_rows_table(int dbl, DevBuf& buf);
// spectral.hip: |FFT_S|^2 of `nch` rows, K frames `hop` apart, summed into the row plan's Float64 accumulator
int welch_rows_accumulate(mdsp_welch_plan_s* pl, const void* work, int64_t K, int64_t hop, int64_t nch, hipStream_t st);
namespace big {

struct Engine {
    int dtype = MDSP_F32;
    int64_t n = 0, nfft = 0;
    int P = 0;
    Pass pass[MAXP];
    DevBuf tables, work, nat, partial, winR;
    const double* win_src = nullptr;   // the Float64 window winR was converted from
    // overlap-save, rows form: nfft = R0 x S with S the longest single-workgroup transform (8192 Float32 / 4096 Float64).  pass[0] is the forward
    // column pass (twiddles behind it), inv0 the same pass with tables of ones (the row kernel has applied the inverse twiddle already)
    int rowsR0 = 0;
    Pass inv0;
    DevBuf rowtab;                     // the S forward roots of the row transforms (ols.hip)
    mdsp_welch_plan rowplan = nullptr; // Welch, rows form: the S-point complex plan whose single-workgroup kernel sums |Z|^2 over the rows (spectral.hip)
    ~Engine() {
        if (rowplan) mdsp_welch_plan_destroy(rowplan);
    }
};
EngineHolder::~EngineHolder() {
    delete p;
    delete rows;
}

namespace {

template <typename R> struct BigArgs {
    Pass p;
    int in_mode;        // 0: the work buffer; 1: two real frames per transform, windowed; 2: one complex frame, windowed
    const void* s;      // the channel's first sample
    const R* win;       // n window values or nullptr
    cx<R>* buf;         // work buffer [ntrans][N]
    void* out;          // OUT 1: Float64 rows [groups][N];  OUT 2: the channel's output matrix;  OUT 3: natural-order spectra [ntrans][N]
    int64_t t0, ntrans; // first transform of this launch (unit index inside the channel), transforms in this launch
    int64_t K, hop, ldo;
    int n, psd, accumulate, acc_add;
    int64_t nout;
    R m1;
    // overlap-save on the same passes (in_mode 3 / 4, OUT 4): two real blocks per transform, x -> FFT -> . H -> FFT of the conjugate -> y
    const void* H;      // the filter's spectrum, natural order, 1 / N folded in
    const void* src2;   // in_mode 4: natural-order spectra [ntrans][N] (what OUT 3 left)
    int64_t nx, nb1, L; // signal length, nb - 1, outputs per block (N - nb + 1); K = number of blocks, nout = outputs of the column
    int ols_cplx;       // complex signal and taps: one block per transform
    int out_mode;       // the two-stage kernel: OUT of big_pass_kernel as a run-time value (0, 2, 3, 4)
    int ablate;         // MDSP_BIG_ABLATE (profiling; results are garbage): 1 no sub-transforms, 2 no stores, 4 no loads after the first item, 8 no tile twiddles
};

// OUT 0: twiddled results back into the work buffer (every pass but the last)
// OUT 1: Welch -- |Z|^2 summed over the launch's transforms of this group in registers (Float64), one row of N per group
// OUT 2: complex signal -- the frame's column: raw spectrum or |Z|^2 / r
// OUT 3: natural-order spectrum into `out` (real signals' columns: untangled by big_untangle_kernel)
// E: elements per thread (R_p B <= E TPB).  grid = (tile lanes, transform groups): a workgroup walks tiles blockIdx.x, + gridDim.x, ... and for each
// the transforms blockIdx.y, + gridDim.y, ... of the launch; OUT 1 is launched with one tile per workgroup (its sums belong to the tile).
// The NEXT item's elements are loaded into registers before this item's sub-transform starts: first measured without that (and with two table
// reads per element for the twiddles), a workgroup spent 21 us per tile, two thirds of it waiting (profiles/r05_bigfft_first.txt).
// (sub-transforms above 256 points fill the LDS with ONE workgroup per CU: that form may use the whole register file)
template <typename R, int OUT, int E> __global__ __launch_bounds__(TPB, (E < elems<R>() ? 2 : 1)) void big_pass_kernel(const BigArgs<R> a) {
    constexpr int B = cols<R>(), Bp = B + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char big_smem[];
    const Pass& p = a.p;
    cx<R>* bufA = reinterpret_cast<cx<R>*>(big_smem);
    cx<R>* bufB = bufA + p.Rp * Bp;
    cx<R>* rootsL = bufB + p.Rp * Bp;
    cx<R>* twc = rootsL + p.Rp;
    const int tid = threadIdx.x;
    {   // the sub-transform's roots: R_p entries, read by every butterfly of every sub-pass
        const cx<R>* g = static_cast<const cx<R>*>(p.roots);
        for (int i = tid; i < p.Rp; i += TPB) fft::st2(rootsL + i, g[i]);
    }
    const int64_t N = p.N;
    cx<R> twb[E];
    if (!p.last) load_twb<R, E>(p, tid, twb);
    [[maybe_unused]] double acc[OUT == 1 ? E : 1];
    if constexpr (OUT == 1) {
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = 0.0;
    }
    cx<R> pre[E];
    auto fetch = [&](int64_t tile, int64_t t) __attribute__((always_inline)) {
        const Tile tc = tile_of<R>(p, tile);
        const int64_t u = a.t0 + t;
        if (a.in_mode == 0) {
            const cx<R>* src = a.buf + t * N;
            load_regs<R, E>(p, tc, tid, pre, [&](int64_t pos) { return src[pos]; });
        } else if (a.in_mode == 1) {   // K4, two frames: z = w (a + i b); frame b may not exist (odd K), the tail past n is zero
            const R* fa = static_cast<const R*>(a.s) + 2 * u * a.hop;
            const bool haveB = 2 * u + 1 < a.K;
            load_regs<R, E>(p, tc, tid, pre, [&](int64_t pos) {
                cx<R> z = {(R)0, (R)0};
                if (pos < a.n) {
                    const R w = a.win ? a.win[pos] : (R)1;
                    z.x = fa[pos] * w;
                    if (haveB) z.y = fa[pos + a.hop] * w;
                }
                return z;
            });
        } else if (a.in_mode == 2) {
            const cx<R>* fa = static_cast<const cx<R>*>(a.s) + u * a.hop;
            load_regs<R, E>(p, tc, tid, pre, [&](int64_t pos) {
                cx<R> z = {(R)0, (R)0};
                if (pos < a.n) {
                    const R w = a.win ? a.win[pos] : (R)1;
                    const cx<R> v = fa[pos];
                    z = {v.x * w, v.y * w};
                }
                return z;
            });
        } else if (a.in_mode == 3) {   // overlap-save: blocks 2u and 2u + 1 of the column, nb - 1 samples of history in front (zeros before the signal and behind it)
            if (a.ols_cplx) {
                const cx<R>* xs = static_cast<const cx<R>*>(a.s);
                const int64_t ia = u * a.L - a.nb1;
                load_regs<R, E>(p, tc, tid, pre, [&](int64_t pos) {
                    const int64_t i = ia + pos;
                    return (i >= 0 && i < a.nx) ? xs[i] : cx<R>{(R)0, (R)0};
                });
            } else {
                const R* xs = static_cast<const R*>(a.s);
                const int64_t ia = 2 * u * a.L - a.nb1, ib = ia + a.L;
                const bool haveB = 2 * u + 1 < a.K;
                load_regs<R, E>(p, tc, tid, pre, [&](int64_t pos) {
                    cx<R> z = {(R)0, (R)0};
                    const int64_t i = ia + pos, j = ib + pos;
                    if (i >= 0 && i < a.nx) z.x = xs[i];
                    if (haveB && j >= 0 && j < a.nx) z.y = xs[j];
                    return z;
                });
            }
        } else if (a.in_mode == 5) {   // overlap-save, rows form: the way back starts from the conjugate of what the row kernel left in the work buffer
            const cx<R>* zs = a.buf + t * N;
            load_regs<R, E>(p, tc, tid, pre, [&](int64_t pos) {
                const cx<R> v = zs[pos];
                return cx<R>{v.x, -v.y};
            });
        } else {                        // overlap-save, the way back: conj(Z H) -- the inverse transform is the forward one of the conjugate, conjugated
            const cx<R>* zs = static_cast<const cx<R>*>(a.src2) + t * N;
            const cx<R>* Hs = static_cast<const cx<R>*>(a.H);
            load_regs<R, E>(p, tc, tid, pre, [&](int64_t pos) {
                const cx<R> v = fft::cmul(zs[pos], Hs[pos]);
                return cx<R>{v.x, -v.y};
            });
        }
    };
    int64_t tile = blockIdx.x, t = blockIdx.y;
    bool have = tile < p.ntiles && t < a.ntrans;
    if (have) fetch(tile, t);
    int64_t twc_tile = -1;
    __syncthreads();
    while (have) {
        const Tile tc = tile_of<R>(p, tile);
        const int64_t u = a.t0 + t;
        regs_to_lds<R, E>(p, tid, pre, bufA);
        if (!p.last && tile != twc_tile && !(a.ablate & 8)) {   // (rewritten only behind the barrier that ends the previous item's stores)
            fill_twc<R>(p, tc, tid, twc);
            twc_tile = tile;
        }
        int64_t nt = t + gridDim.y, ntile = tile;
        if (nt >= a.ntrans) {
            nt = blockIdx.y;
            ntile = tile + gridDim.x;
        }
        const bool nhave = ntile < p.ntiles;
        if (nhave && !(a.ablate & 4)) fetch(ntile, nt);   // in flight while this item is transformed
        __syncthreads();
        cx<R>*src = bufA, *dst = bufB;
        for (int sp = 0; sp < ((a.ablate & 1) ? 0 : p.nsub); ++sp) {
            phase_sub<R>(p, sp, tid, src, dst, rootsL);
            __syncthreads();
            cx<R>* tmp = src;
            src = dst;
            dst = tmp;
        }
        if (a.ablate & 2) {
        } else if constexpr (OUT == 0) {
            cx<R>* o = a.buf + t * N;
            phase_store<R, E>(p, tc, tid, src, twc, twb, [&](int, int64_t pos, cx<R> z) { o[pos] = z; });
        } else if constexpr (OUT == 1) {   // K5: |Z|^2 in the working precision (one rounding per term), accumulated over frames in double
            phase_store<R, E>(p, tc, tid, src, twc, twb, [&](int e, int64_t, cx<R> z) { acc[e] += (double)(z.x * z.x + z.y * z.y); });
        } else if constexpr (OUT == 2) {
            const int64_t o0 = u * a.ldo;
            phase_store<R, E>(p, tc, tid, src, twc, twb, [&](int, int64_t k, cx<R> z) {
                if (k < a.nout) {
                    if (a.psd) {
                        R* o = static_cast<R*>(a.out) + o0 + k;
                        const R pw = z.x * z.x + z.y * z.y;
                        *o = a.accumulate ? fma(pw, a.m1, *o) : pw * a.m1;   // fft2pow!: out = muladd(abs2, m, out)
                    } else static_cast<cx<R>*>(a.out)[o0 + k] = z;
                }
            });
        } else if constexpr (OUT == 3) {
            cx<R>* o = static_cast<cx<R>*>(a.out) + t * N;
            phase_store<R, E>(p, tc, tid, src, twc, twb, [&](int, int64_t k, cx<R> z) { o[k] = z; });
        } else {   // OUT 4, overlap-save: the L valid samples of both blocks (y = conj of what the transform of the conjugate left)
            if (a.ols_cplx) {
                cx<R>* yo = static_cast<cx<R>*>(a.out);
                phase_store<R, E>(p, tc, tid, src, twc, twb, [&](int, int64_t k, cx<R> z) {
                    const int64_t o = u * a.L + (k - a.nb1);
                    if (k >= a.nb1 && o < a.nout) yo[o] = cx<R>{z.x, -z.y};
                });
            } else {
                R* yo = static_cast<R*>(a.out);
                const bool haveB = 2 * u + 1 < a.K;
                phase_store<R, E>(p, tc, tid, src, twc, twb, [&](int, int64_t k, cx<R> z) {
                    if (k >= a.nb1) {
                        const int64_t o = 2 * u * a.L + (k - a.nb1);
                        if (o < a.nout) yo[o] = z.x;
                        if (haveB && o + a.L < a.nout) yo[o + a.L] = -z.y;
                    }
                });
            }
        }
        __syncthreads();   // the buffer the stores read is the one the next item's samples or first sub-pass write
        tile = ntile;
        t = nt;
        have = nhave;
    }
    if constexpr (OUT == 1) {   // one tile per workgroup (gridDim.x == ntiles)
        if (blockIdx.x < p.ntiles) {
            const Tile tc = tile_of<R>(p, (int64_t)blockIdx.x);
            double* row = static_cast<double*>(a.out) + (int64_t)blockIdx.y * N;
            phase_store<R, E>(p, tc, tid, bufA, twc, twb, [&](int e, int64_t k, cx<R>) { row[k] = a.acc_add ? row[k] + acc[e] : acc[e]; });
        }
    }
}

// The same pass with the sub-transform as two register stages around ONE LDS exchange (bigfft_pass.h "two-stage sub-transforms"): R_p = RA x TJ,
// tiles of 256 / TJ columns.  Two exchange buffers alternate from item to item, so an item costs one barrier (the last pass: two -- its elements
// reach the owners of their rows through LDS first).  out_mode: 0 / 2 / 3 as OUT of big_pass_kernel; WELCH is its OUT 1.
template <typename R, bool WELCH, int RA, int TJ>
__global__ __launch_bounds__(TPB, (sizeof(R) == 4 ? 2 : 1)) void big_fast_kernel(const BigArgs<R> a) {
    using G = FastGeo<RA, TJ>;
    extern __shared__ __attribute__((aligned(16))) unsigned char big_smem[];
    const Pass& p = a.p;
    cx<R>* buf0 = reinterpret_cast<cx<R>*>(big_smem);
    cx<R>* buf1 = buf0 + G::Rp * G::Bp;
    cx<R>* twc0 = buf1 + G::Rp * G::Bp;
    cx<R>* twc1 = twc0 + G::Rp;
    const int tid = threadIdx.x;
    const int64_t N = p.N;
    cx<R> rt[G::NO], twb[G::NO];
    fast_roots<R, RA, TJ>(p, tid, rt);
    if (!p.last) fast_twb<R, RA, TJ>(p, tid, twb);
    [[maybe_unused]] double acc[WELCH ? G::NO : 1];
    if constexpr (WELCH) {
#pragma unroll
        for (int e = 0; e < G::NO; ++e) acc[e] = 0.0;
    }
    cx<R> pre[RA];
    // MDSP_BIG_ABLATE (profiling; garbage results): 1 no butterflies, 2 no stores, 4 no loads after the first item, 8 no tile twiddles,
    // 16 every tile a contiguous block of the buffer (the same bytes without the strided access pattern; passes other than the last)
    auto tile_at = [&](int64_t tile) __attribute__((always_inline)) {
        Tile tc = tile_of_b<R>(p, tile, G::B);
        if ((a.ablate & 16) && !p.last) {
            tc.base = tile * (G::Rp * G::B);
            tc.row_stride = G::B;
        }
        return tc;
    };
    auto fetch = [&](int64_t tile, int64_t t) __attribute__((always_inline)) {
        const Tile tc = tile_at(tile);
        const int64_t u = a.t0 + t;
        if (a.in_mode == 0) {
            const cx<R>* src = a.buf + t * N;
            fast_load<R, RA, TJ>(p, tc, tid, pre, [&](int64_t pos) { return src[pos]; });
        } else if (a.in_mode == 1) {   // K4, two frames: z = w (a + i b); frame b may not exist (odd K), the tail past n is zero
            const R* fa = static_cast<const R*>(a.s) + 2 * u * a.hop;
            const bool haveB = 2 * u + 1 < a.K;
            fast_load<R, RA, TJ>(p, tc, tid, pre, [&](int64_t pos) {
                cx<R> z = {(R)0, (R)0};
                if (pos < a.n) {
                    const R w = a.win ? a.win[pos] : (R)1;
                    z.x = fa[pos] * w;
                    if (haveB) z.y = fa[pos + a.hop] * w;
                }
                return z;
            });
        } else if (a.in_mode == 2) {
            const cx<R>* fa = static_cast<const cx<R>*>(a.s) + u * a.hop;
            fast_load<R, RA, TJ>(p, tc, tid, pre, [&](int64_t pos) {
                cx<R> z = {(R)0, (R)0};
                if (pos < a.n) {
                    const R w = a.win ? a.win[pos] : (R)1;
                    const cx<R> v = fa[pos];
                    z = {v.x * w, v.y * w};
                }
                return z;
            });
        } else if (a.in_mode == 3) {   // overlap-save: blocks 2u and 2u + 1 of the column, nb - 1 samples of history in front (zeros before the signal and behind it)
            if (a.ols_cplx) {
                const cx<R>* xs = static_cast<const cx<R>*>(a.s);
                const int64_t ia = u * a.L - a.nb1;
                fast_load<R, RA, TJ>(p, tc, tid, pre, [&](int64_t pos) {
                    const int64_t i = ia + pos;
                    return (i >= 0 && i < a.nx) ? xs[i] : cx<R>{(R)0, (R)0};
                });
            } else {
                const R* xs = static_cast<const R*>(a.s);
                const int64_t ia = 2 * u * a.L - a.nb1, ib = ia + a.L;
                const bool haveB = 2 * u + 1 < a.K;
                fast_load<R, RA, TJ>(p, tc, tid, pre, [&](int64_t pos) {
                    cx<R> z = {(R)0, (R)0};
                    const int64_t i = ia + pos, j = ib + pos;
                    if (i >= 0 && i < a.nx) z.x = xs[i];
                    if (haveB && j >= 0 && j < a.nx) z.y = xs[j];
                    return z;
                });
            }
        } else if (a.in_mode == 5) {   // overlap-save, rows form: the way back starts from the conjugate of what the row kernel left in the work buffer
            const cx<R>* zs = a.buf + t * N;
            fast_load<R, RA, TJ>(p, tc, tid, pre, [&](int64_t pos) {
                const cx<R> v = zs[pos];
                return cx<R>{v.x, -v.y};
            });
        } else {                        // overlap-save, the way back: conj(Z H) -- the inverse transform is the forward one of the conjugate, conjugated
            const cx<R>* zs = static_cast<const cx<R>*>(a.src2) + t * N;
            const cx<R>* Hs = static_cast<const cx<R>*>(a.H);
            fast_load<R, RA, TJ>(p, tc, tid, pre, [&](int64_t pos) {
                const cx<R> v = fft::cmul(zs[pos], Hs[pos]);
                return cx<R>{v.x, -v.y};
            });
        }
    };
    int64_t tile = blockIdx.x, t = blockIdx.y;
    bool have = tile < p.ntiles && t < a.ntrans;
    if (have) fetch(tile, t);
    for (int it = 0; have; ++it) {
        const Tile tc = tile_at(tile);
        const int64_t u = a.t0 + t;
        cx<R>* X = p.last ? buf1 : ((it & 1) ? buf1 : buf0);
        cx<R>* twc = (it & 1) ? twc1 : twc0;
        if (!p.last && !(a.ablate & 8)) fill_twc<R>(p, tc, tid, twc);
        int64_t nt = t + gridDim.y, ntile = tile;
        if (nt >= a.ntrans) {
            nt = blockIdx.y;
            ntile = tile + gridDim.x;
        }
        const bool nhave = ntile < p.ntiles;
        if (p.last) {
            fast_stage_put<R, RA, TJ>(tid, pre, buf0);
            __syncthreads();
            fast_stage_get<R, RA, TJ>(tid, pre, buf0);
        }
        fast_stage1<R, RA, TJ>(tid, pre, X);
        if (nhave && !(a.ablate & 4)) fetch(ntile, nt);   // the next item's samples take over the registers stage 1 has just emptied: in flight through the barrier, stage 2 and the stores
        __syncthreads();
        cx<R> y[G::NO];
        fast_stage2<R, RA, TJ>(tid, X, rt, y);
        if (a.ablate & 2) {
        } else if constexpr (WELCH) {   // K5: |Z|^2 in the working precision (one rounding per term), accumulated over frames in double
            fast_store<R, RA, TJ>(p, tc, tid, y, twc, twb, [&](int e, int64_t, cx<R> z) { acc[e] += (double)(z.x * z.x + z.y * z.y); });
        } else if (a.out_mode == 0) {
            cx<R>* o = a.buf + t * N;
            fast_store<R, RA, TJ>(p, tc, tid, y, twc, twb, [&](int, int64_t pos, cx<R> z) { o[pos] = z; });
        } else if (a.out_mode == 2) {
            const int64_t o0 = u * a.ldo;
            fast_store<R, RA, TJ>(p, tc, tid, y, twc, twb, [&](int, int64_t k, cx<R> z) {
                if (k < a.nout) {
                    if (a.psd) {
                        R* o = static_cast<R*>(a.out) + o0 + k;
                        const R pw = z.x * z.x + z.y * z.y;
                        *o = a.accumulate ? fma(pw, a.m1, *o) : pw * a.m1;   // fft2pow!: out = muladd(abs2, m, out)
                    } else static_cast<cx<R>*>(a.out)[o0 + k] = z;
                }
            });
        } else if (a.out_mode == 3) {
            cx<R>* o = static_cast<cx<R>*>(a.out) + t * N;
            fast_store<R, RA, TJ>(p, tc, tid, y, twc, twb, [&](int, int64_t k, cx<R> z) { o[k] = z; });
        } else {   // 4, overlap-save: the L valid samples of both blocks
            if (a.ols_cplx) {
                cx<R>* yo = static_cast<cx<R>*>(a.out);
                fast_store<R, RA, TJ>(p, tc, tid, y, twc, twb, [&](int, int64_t k, cx<R> z) {
                    const int64_t o = u * a.L + (k - a.nb1);
                    if (k >= a.nb1 && o < a.nout) yo[o] = cx<R>{z.x, -z.y};
                });
            } else {
                R* yo = static_cast<R*>(a.out);
                const bool haveB = 2 * u + 1 < a.K;
                fast_store<R, RA, TJ>(p, tc, tid, y, twc, twb, [&](int, int64_t k, cx<R> z) {
                    if (k >= a.nb1) {
                        const int64_t o = 2 * u * a.L + (k - a.nb1);
                        if (o < a.nout) yo[o] = z.x;
                        if (haveB && o + a.L < a.nout) yo[o + a.L] = -z.y;
                    }
                });
            }
        }
        tile = ntile;
        t = nt;
        have = nhave;
    }
    if constexpr (WELCH) {   // one tile per workgroup (gridDim.x == ntiles)
        if (blockIdx.x < p.ntiles) {
            const Tile tc = tile_of_b<R>(p, (int64_t)blockIdx.x, G::B);
            double* row = static_cast<double*>(a.out) + (int64_t)blockIdx.y * N;
            cx<R> y[G::NO];
#pragma unroll
            for (int e = 0; e < G::NO; ++e) y[e] = {(R)0, (R)0};
            fast_store<R, RA, TJ>(p, tc, tid, y, twc0, twb, [&](int e, int64_t k, cx<R>) { row[k] = a.acc_add ? row[k] + acc[e] : acc[e]; });
        }
    }
}

// acc[k] = (add ? acc[k] : 0) + sum_g rows[g][k], fixed order
__global__ __launch_bounds__(256) void big_reduce_kernel(const double* __restrict__ rows, double* __restrict__ acc, int64_t N, int groups, int add) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    double v = add ? acc[k] : 0.0;
    for (int g = 0; g < groups; ++g) v += rows[(int64_t)g * N + k];
    acc[k] = v;
}

template <typename R> __global__ __launch_bounds__(256) void big_window_kernel(const double* __restrict__ win, R* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (R)win[i];
}

// Columns of the two real frames of a transform from its natural-order spectrum: A[k] = (Z[k] + conj Z[N-k]) / 2, B[k] = (Z[k] - conj Z[N-k]) / (2i)
// (the same arithmetic as the single-workgroup kernels, spectral_gen.h); grid (bins / 256, transforms)
template <typename R>
__global__ __launch_bounds__(256) void big_untangle_kernel(const cx<R>* __restrict__ nat, void* __restrict__ out, int64_t N, int64_t t0, int64_t K, int64_t ldo,
                                                            int64_t nout, int psd, int accumulate, int onesided, R m1, R m2) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nout) return;
    const int64_t t = blockIdx.y, f0 = 2 * (t0 + t);
    const bool haveB = f0 + 1 < K;
    const cx<R>* z = nat + t * N;
    const bool mirror = j > N / 2;                     // real -> two-sided: X[N-k] = conj(X[k]) (fft2oneortwosided!, :234-244)
    const int64_t k = mirror ? N - j : j;
    const cx<R> zk = z[k], zm = z[k == 0 ? 0 : N - k];
    cx<R> A = {(R)0.5 * (zk.x + zm.x), (R)0.5 * (zk.y - zm.y)};
    cx<R> Bv = {(R)0.5 * (zk.y + zm.y), (R)0.5 * (zm.x - zk.x)};
    const int64_t o0 = f0 * ldo + j;
    if (psd) {
        R m = m1;
        if (onesided && !(j == 0 || (j == nout - 1 && N % 2 == 0))) m = m2;
        R* o = static_cast<R*>(out) + o0;
        const R pa = A.x * A.x + A.y * A.y;
        *o = accumulate ? fma(pa, m, *o) : pa * m;
        if (haveB) {
            const R pb = Bv.x * Bv.x + Bv.y * Bv.y;
            o[ldo] = accumulate ? fma(pb, m, o[ldo]) : pb * m;
        }
    } else {
        if (mirror) {
            A.y = -A.y;
            Bv.y = -Bv.y;
        }
        cx<R>* o = static_cast<cx<R>*>(out) + o0;
        *o = A;
        if (haveB) o[ldo] = Bv;
    }
}

template <typename R> int build(Engine* e) {
    HostPlan<R> hp;
    if (!make_plan<R>(e->nfft, hp, tunables().big_rmax, tunables().big_fast)) MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "nfft=%lld does not split into 2..4 factors of at most %d", (long long)e->nfft, RMAX);
    size_t total = 0;
    auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
    for (int p = 0; p < hp.P; ++p) total += al(hp.roots[p].size()) + al(hp.T0[p].size()) + al(hp.T1[p].size());
    std::vector<cx<R>> all(total);
    MDSP_TRY(e->tables.reserve(sizeof(cx<R>) * total));
    size_t off = 0;
    auto put = [&](const std::vector<cx<R>>& v) -> const void* {
        const void* dev = e->tables.as<cx<R>>() + off;
        std::copy(v.begin(), v.end(), all.begin() + (long)off);
        off += al(v.size());
        return dev;
    };
    e->P = hp.P;
    for (int p = 0; p < hp.P; ++p) {
        e->pass[p] = hp.pass[p];
        e->pass[p].roots = put(hp.roots[p]);
        e->pass[p].T0 = put(hp.T0[p]);
        e->pass[p].T1 = put(hp.T1[p]);
    }
    MDSP_HIP(hipMemcpy(e->tables.p, all.data(), sizeof(cx<R>) * total, hipMemcpyHostToDevice));
    return MDSP_OK;
}

template <typename R> int build_rows(Engine* e, int R0) {
    HostPlan<R> hp;
    const int Rf[2] = {R0, (int)(e->nfft / R0)};
    if (!make_plan_factors<R>(e->nfft, hp, Rf, 2, tunables().big_fast, true)) MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "no column pass of %d points", R0);
    auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t nr = al(hp.roots[0].size()), n0 = al(hp.T0[0].size()), n1 = al(hp.T1[0].size());
    std::vector<cx<R>> all(nr + 2 * (n0 + n1), cx<R>{(R)1, (R)0});
    std::copy(hp.roots[0].begin(), hp.roots[0].end(), all.begin());
    std::copy(hp.T0[0].begin(), hp.T0[0].end(), all.begin() + (long)nr);
    std::copy(hp.T1[0].begin(), hp.T1[0].end(), all.begin() + (long)(nr + n0));
    MDSP_TRY(e->tables.reserve(sizeof(cx<R>) * all.size()));
    MDSP_HIP(hipMemcpy(e->tables.p, all.data(), sizeof(cx<R>) * all.size(), hipMemcpyHostToDevice));
    const cx<R>* dev = e->tables.as<cx<R>>();
    e->P = 1;
    e->pass[0] = hp.pass[0];
    e->pass[0].roots = dev;
    e->pass[0].T0 = dev + nr;
    e->pass[0].T1 = dev + nr + n0;
    e->inv0 = e->pass[0];
    e->inv0.T0 = dev + nr + n0 + n1;          // ones
    e->inv0.T1 = dev + nr + n0 + n1 + n0;
    e->rowsR0 = R0;
    return ols_rows_table(sizeof(R) == 8, e->rowtab);
}

int get_engine_rows(EngineHolder& h, int dtype, int64_t n, int64_t nfft, int R0, Engine** out) {
    if (h.rows && (h.rows->dtype != dtype || h.rows->n != n || h.rows->rowsR0 != R0 || h.rows->nfft != nfft)) {
        delete h.rows;
        h.rows = nullptr;
    }
    if (!h.rows) {
        std::unique_ptr<Engine> e(new Engine());
        e->dtype = dtype;
        e->n = n;
        e->nfft = nfft;
        MDSP_TRY(dtype_is_double(dtype) ? build_rows<double>(e.get(), R0) : build_rows<float>(e.get(), R0));
        h.rows = e.release();
    }
    *out = h.rows;
    return MDSP_OK;
}

int get_engine(EngineHolder& h, int dtype, int64_t n, int64_t nfft, Engine** out) {
    if (h.p && (h.p->dtype != dtype || h.p->n != n || h.p->nfft != nfft)) {
        delete h.p;
        h.p = nullptr;
    }
    if (!h.p) {
        std::unique_ptr<Engine> e(new Engine());
        e->dtype = dtype;
        e->n = n;
        e->nfft = nfft;
        MDSP_TRY(dtype_is_double(dtype) ? build<double>(e.get()) : build<float>(e.get()));
        h.p = e.release();
    }
    *out = h.p;
    return MDSP_OK;
}

template <typename R, int OUT, int E> int launch_pass_e(const BigArgs<R>& a, int lanes, int groups, hipStream_t st) {
    constexpr int Bp = cols<R>() + 1;
    auto kern = big_pass_kernel<R, OUT, E>;
    const size_t lds_bytes = sizeof(cx<R>) * ((size_t)2 * a.p.Rp * Bp + (size_t)2 * a.p.Rp);
    if (lds_bytes > 48 * 1024) MDSP_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)lanes, (unsigned)groups), dim3(TPB), lds_bytes, st, a);
    MDSP_LAUNCH_CHECK();
    return MDSP_OK;
}
template <typename R, bool WELCH, int RA, int TJ> int launch_fast(const BigArgs<R>& a, int lanes, int groups, hipStream_t st) {
    using G = FastGeo<RA, TJ>;
    auto kern = big_fast_kernel<R, WELCH, RA, TJ>;
    const size_t lds_bytes = sizeof(cx<R>) * ((size_t)2 * G::Rp * G::Bp + (size_t)2 * G::Rp);
    if (lds_bytes > 48 * 1024) MDSP_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)lanes, (unsigned)groups), dim3(TPB), lds_bytes, st, a);
    MDSP_LAUNCH_CHECK();
    return MDSP_OK;
}
// lanes: workgroups along the tiles (each walks tiles lane, lane + lanes, ...); OUT 1 takes one tile per workgroup
template <typename R, int OUT> int launch_pass(const BigArgs<R>& a0, int lanes, int groups, hipStream_t st) {
    constexpr int EH = elems<R>() / 2;   // sub-transforms up to 256 points: half the elements per thread (and half the registers of samples in flight)
    if (OUT == 1) lanes = (int)a0.p.ntiles;
    if (a0.p.fTJ) {   // the two-stage register form
        BigArgs<R> a = a0;
        a.out_mode = OUT;
        constexpr bool W = OUT == 1;
        if (a.p.fRA == 16 && a.p.fTJ == 4) {   // 64 = 16 x 4: 64-column tiles (MDSP_BIG_FAST=3)
            if constexpr (sizeof(R) == 4) return launch_fast<R, W, 16, 4>(a, lanes, groups, st);
            else MDSP_FAIL(MDSP_ERR_ASSERTION, "the 16 x 4 form is Float32 only");
        }
        if (a.p.fRA == 16 && a.p.fTJ == 8) {
            if constexpr (sizeof(R) == 4) return launch_fast<R, W, 16, 8>(a, lanes, groups, st);
            else MDSP_FAIL(MDSP_ERR_ASSERTION, "the 16 x 8 form is Float32 only");
        }
        if (a.p.fRA == 16) return launch_fast<R, W, 16, 16>(a, lanes, groups, st);
        if (a.p.fRA == 8 && a.p.fTJ == 16) return launch_fast<R, W, 8, 16>(a, lanes, groups, st);
        if (a.p.fRA == 8) return launch_fast<R, W, 8, 8>(a, lanes, groups, st);
        return launch_fast<R, W, 4, 8>(a, lanes, groups, st);
    }
    const BigArgs<R>& a = a0;
    if (a.p.Rp * cols<R>() <= EH * TPB) return launch_pass_e<R, OUT, EH>(a, lanes, groups, st);
    return launch_pass_e<R, OUT, elems<R>()>(a, lanes, groups, st);
}

// mode 0: Welch sums into acc; 1: columns
template <typename R, bool CPLX>
int run(Engine* e, int mode, const void* s, int64_t K, int64_t hop, const double* win_dev, double* acc, bool fresh, void* out, int64_t ldo, int64_t nout,
        int onesided, int psd, int accumulate, double r, hipStream_t st) {
    const int64_t N = e->nfft;
    const int64_t units = CPLX ? K : cdiv(K, 2);
    if (units == 0) return MDSP_OK;
    const R* win = nullptr;
    if (win_dev) {
        if constexpr (sizeof(R) == 8) win = reinterpret_cast<const R*>(win_dev);
        else {   // Float32 signals: the window rounded to Float32 first, as the single-workgroup kernels do -- and half the bytes per frame.
                 // Converted once per window (a plan's window never changes; a multitaper plan walks its tapers, one pointer each).
            if (e->win_src != win_dev) {
                MDSP_TRY(e->winR.reserve(sizeof(R) * (size_t)e->n));
                hipLaunchKernelGGL(big_window_kernel<R>, dim3((unsigned)cdiv(e->n, 256)), dim3(256), 0, st, win_dev, e->winR.as<R>(), e->n);
                MDSP_LAUNCH_CHECK();
                e->win_src = win_dev;
            }
            win = e->winR.as<R>();
        }
    }
    const int64_t per = (int64_t)sizeof(cx<R>) * N;
    const int64_t C = std::max<int64_t>(1, std::min<int64_t>(units, ((int64_t)tunables().big_chunk_mib << 20) / per));
    MDSP_TRY(e->work.reserve((size_t)(per * C)));
    const bool untangle = mode == 1 && !CPLX;
    if (untangle) MDSP_TRY(e->nat.reserve((size_t)(per * C)));
    const int cus = device_cu_count();
    bool add = !fresh;
    for (int64_t c0 = 0; c0 < units; c0 += C) {
        const int64_t cnt = std::min<int64_t>(C, units - c0);
        for (int p = 0; p < e->P; ++p) {
            BigArgs<R> a{};
            a.p = e->pass[p];
            a.in_mode = p == 0 ? (CPLX ? 2 : 1) : 0;
            a.s = s;
            a.win = win;
            a.buf = e->work.as<cx<R>>();
            a.t0 = c0;
            a.ntrans = cnt;
            a.K = K;
            a.hop = hop;
            a.ldo = ldo;
            a.n = (int)e->n;
            a.nout = nout;
            a.psd = psd;
            a.accumulate = accumulate;
            a.m1 = (R)(1.0 / r);
            a.ablate = tunables().big_ablate;
            // workgroups: `wgs` per CU resident (two fit: LDS), each walking several (tile, transform) items with the next one's samples in flight
            // (Float64 two-stage passes: one, but two for the 64-point pass -- Welch at 2^22 points 0.30 -> 0.38 TB/s, profiles/r05_bigfft_sessions.json "f64_wgs")
            const int64_t wgs = (int64_t)cus * (tunables().big_wgs > 0 ? tunables().big_wgs
                                                : (a.p.Rp > RMAX / 2 ? 1 : (a.p.fTJ && sizeof(R) == 8) ? (a.p.Rp <= 64 ? 2 : 1) : (a.p.fTJ && a.p.Rp <= 64 ? 4 : 2)));
            int groups = tunables().big_groups > 0 ? tunables().big_groups : (int)cdiv(wgs, a.p.ntiles);
            groups = (int)std::max<int64_t>(1, std::min<int64_t>(groups, cnt));
            const int lanes = (int)std::max<int64_t>(1, std::min<int64_t>(a.p.ntiles, wgs / groups));
            if (p < e->P - 1) {
                MDSP_TRY((launch_pass<R, 0>(a, lanes, groups, st)));
            } else if (mode == 0) {
                groups = std::min(groups, 32);
                if (groups == 1) {
                    a.out = acc;
                    a.acc_add = add ? 1 : 0;
                    MDSP_TRY((launch_pass<R, 1>(a, lanes, 1, st)));
                } else {
                    MDSP_TRY(e->partial.reserve(sizeof(double) * (size_t)groups * (size_t)N));
                    a.out = e->partial.p;
                    a.acc_add = 0;
                    MDSP_TRY((launch_pass<R, 1>(a, lanes, groups, st)));
                    hipLaunchKernelGGL(big_reduce_kernel, dim3((unsigned)cdiv(N, 256)), dim3(256), 0, st, e->partial.as<double>(), acc, N, groups, add ? 1 : 0);
                    MDSP_LAUNCH_CHECK();
                }
                add = true;
            } else if (!untangle) {
                a.out = out;
                MDSP_TRY((launch_pass<R, 2>(a, lanes, groups, st)));
            } else {
                a.out = e->nat.p;
                MDSP_TRY((launch_pass<R, 3>(a, lanes, groups, st)));
                hipLaunchKernelGGL(big_untangle_kernel<R>, dim3((unsigned)cdiv(nout, 256), (unsigned)cnt), dim3(256), 0, st, e->nat.as<cx<R>>(), out, N, c0, K, ldo,
                                   nout, psd, accumulate, onesided, (R)(1.0 / r), (R)(2.0 / r));
                MDSP_LAUNCH_CHECK();
            }
        }
    }
    return MDSP_OK;
}

// Overlap-save of one column on the engine's passes: blocks [g0, g1) of the column's grid (g0 even for real signals).  Forward passes read the
// blocks straight from the signal (in_mode 3), the last one leaves natural-order spectra (OUT 3); the way back starts from conj(Z H) (in_mode 4)
// and its last pass stores the L valid samples of each block (OUT 4).
template <typename R> int run_ols(Engine* e, bool cplx, const void* x, int64_t nx, const void* H, int64_t nb, void* y, int64_t nout, int64_t g0, int64_t g1, hipStream_t st) {
    const int64_t N = e->nfft, L = N - (nb - 1);
    const int64_t u0 = cplx ? g0 : g0 / 2, u1 = cplx ? g1 : cdiv(g1, 2);
    if (u1 <= u0) return MDSP_OK;
    const int64_t per = (int64_t)sizeof(cx<R>) * N;
    const int64_t C = std::max<int64_t>(1, std::min<int64_t>(u1 - u0, ((int64_t)tunables().big_chunk_mib << 20) / (2 * per)));
    MDSP_TRY(e->work.reserve((size_t)(per * C)));
    MDSP_TRY(e->nat.reserve((size_t)(per * C)));
    const int cus = device_cu_count();
    for (int64_t c0 = u0; c0 < u1; c0 += C) {
        const int64_t cnt = std::min<int64_t>(C, u1 - c0);
        for (int dir = 0; dir < 2; ++dir) {
            for (int p = 0; p < e->P; ++p) {
                BigArgs<R> a{};
                a.p = e->pass[p];
                a.in_mode = p == 0 ? (dir == 0 ? 3 : 4) : 0;
                a.s = x;
                a.buf = e->work.as<cx<R>>();
                a.src2 = e->nat.p;
                a.H = H;
                a.t0 = c0;
                a.ntrans = cnt;
                a.K = g1;          // blocks past the range do not exist for this call (second block of the last pair)
                a.nx = nx;
                a.nb1 = nb - 1;
                a.L = L;
                a.nout = nout;
                a.ols_cplx = cplx ? 1 : 0;
                a.ablate = tunables().big_ablate;
                const int64_t wgs = (int64_t)cus * (tunables().big_wgs > 0 ? tunables().big_wgs : ((a.p.Rp > RMAX / 2 || (a.p.fTJ && sizeof(R) == 8)) ? 1 : (a.p.fTJ && a.p.Rp <= 64 ? 4 : 2)));
                int groups = tunables().big_groups > 0 ? tunables().big_groups : (int)cdiv(wgs, a.p.ntiles);
                groups = (int)std::max<int64_t>(1, std::min<int64_t>(groups, cnt));
                const int lanes = (int)std::max<int64_t>(1, std::min<int64_t>(a.p.ntiles, wgs / groups));
                if (p < e->P - 1) {
                    MDSP_TRY((launch_pass<R, 0>(a, lanes, groups, st)));
                } else if (dir == 0) {
                    a.out = e->nat.p;
                    MDSP_TRY((launch_pass<R, 3>(a, lanes, groups, st)));
                } else {
                    a.out = y;
                    MDSP_TRY((launch_pass<R, 4>(a, lanes, groups, st)));
                }
            }
        }
    }
    return MDSP_OK;
}

// The rows form: nfft = R0 x S.  Column pass (R0-point transforms along the stride-S dimension, the inter-pass twiddle behind them), then ONE kernel per
// row that does the S-point transform, the row of the filter's spectrum, the inverse S-point transform and the inverse twiddle (ols.hip: the
// single-workgroup transforms, in place), then the inverse column pass straight into y: 7 trips of 8 nfft bytes through HBM per transform instead of 13.
template <typename R> int run_ols_rows(Engine* e, bool cplx, const void* x, int64_t nx, const void* H, int64_t nb, void* y, int64_t nout, int64_t g0, int64_t g1, hipStream_t st) {
    const int64_t N = e->nfft, L = N - (nb - 1);
    const int64_t u0 = cplx ? g0 : g0 / 2, u1 = cplx ? g1 : cdiv(g1, 2);
    if (u1 <= u0) return MDSP_OK;
    const int64_t per = (int64_t)sizeof(cx<R>) * N;
    const int64_t C = std::max<int64_t>(1, std::min<int64_t>(u1 - u0, ((int64_t)tunables().big_chunk_mib << 20) / per));
    MDSP_TRY(e->work.reserve((size_t)(per * C)));
    const int cus = device_cu_count();
    for (int64_t c0 = u0; c0 < u1; c0 += C) {
        const int64_t cnt = std::min<int64_t>(C, u1 - c0);
        for (int dir = 0; dir < 2; ++dir) {
            BigArgs<R> a{};
            a.p = dir == 0 ? e->pass[0] : e->inv0;
            a.in_mode = dir == 0 ? 3 : 5;
            a.s = x;
            a.buf = e->work.as<cx<R>>();
            a.t0 = c0;
            a.ntrans = cnt;
            a.K = g1;
            a.nx = nx;
            a.nb1 = nb - 1;
            a.L = L;
            a.nout = nout;
            a.ols_cplx = cplx ? 1 : 0;
            a.ablate = tunables().big_ablate;
            // workgroups per CU of the column passes, measured 1 .. 6 (r05_s26.sh): four for the 64-point pass in either precision (Float64: 0.79 TB/s with one,
            // 1.00 with two or four), two otherwise
            const int64_t wgs = (int64_t)cus * (tunables().big_wgs > 0 ? tunables().big_wgs : (a.p.fTJ && a.p.Rp <= 64 ? 4 : 2));
            int groups = tunables().big_groups > 0 ? tunables().big_groups : (int)cdiv(wgs, a.p.ntiles);
            groups = (int)std::max<int64_t>(1, std::min<int64_t>(groups, cnt));
            const int lanes = (int)std::max<int64_t>(1, std::min<int64_t>(a.p.ntiles, wgs / groups));
            if (dir == 0) {
                MDSP_TRY((launch_pass<R, 0>(a, lanes, groups, st)));
                MDSP_TRY(ols_rows(sizeof(R) == 8, e->work.p, cnt * e->rowsR0, e->rowsR0, H, e->rowtab.p, e->pass[0].T0, e->pass[0].T1, e->pass[0].logS, st));
            } else {
                a.out = y;
                MDSP_TRY((launch_pass<R, 4>(a, lanes, groups, st)));
            }
        }
    }
    return MDSP_OK;
}

// acc[k1 + R0 k2] (+)= red[k1 S + k2]: the rows' sums into the natural-order accumulator, 32 x 32 tiles through LDS (256-byte runs on both sides)
__global__ __launch_bounds__(256) void big_rows_to_natural_kernel(const double* __restrict__ red, double* __restrict__ acc, int R0, int64_t S, int add) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t k2_0 = (int64_t)blockIdx.x * 32;
    const int k1_0 = blockIdx.y * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[ty + 8 * j][tx] = red[(int64_t)(k1_0 + ty + 8 * j) * S + k2_0 + tx];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t o = (int64_t)(k1_0 + tx) + (int64_t)R0 * (k2_0 + ty + 8 * j);
        const double v = tile[tx][ty + 8 * j];
        acc[o] = add ? acc[o] + v : v;
    }
}

// Welch in the rows form: nfft = R0 x S.  The column pass reads the windowed frames (two real frames per transform) and leaves the twiddled columns in the
// work buffer; the rows are then complex frames of S points, R0 channels of them, K transforms `nfft` elements apart, for the single-workgroup Welch
// kernel (spectral.hip): two trips through HBM per transform instead of three, and no natural-order pass -- |Z|^2 is summed row-major and transposed once.
template <typename R, bool CPLX>
int run_welch_rows(Engine* e, const void* s, int64_t K, int64_t hop, const double* win_dev, double* acc, bool fresh, hipStream_t st) {
    const int64_t N = e->nfft, S = N / e->rowsR0;
    const int64_t units = CPLX ? K : cdiv(K, 2);
    if (units == 0) return MDSP_OK;
    const R* win = nullptr;
    if (win_dev) {
        if constexpr (sizeof(R) == 8) win = reinterpret_cast<const R*>(win_dev);
        else {
            if (e->win_src != win_dev) {
                MDSP_TRY(e->winR.reserve(sizeof(R) * (size_t)e->n));
                hipLaunchKernelGGL(big_window_kernel<R>, dim3((unsigned)cdiv(e->n, 256)), dim3(256), 0, st, win_dev, e->winR.as<R>(), e->n);
                MDSP_LAUNCH_CHECK();
                e->win_src = win_dev;
            }
            win = e->winR.as<R>();
        }
    }
    if (!e->rowplan) {
        const int cd = sizeof(R) == 8 ? MDSP_C64 : MDSP_C32;
        MDSP_TRY(mdsp_welch_plan_create(&e->rowplan, S, 0, S, nullptr, 1.0, 0, cd, MDSP_ENGINE_FUSED));
    }
    MDSP_TRY(mdsp_welch_reset(e->rowplan));
    const int64_t per = (int64_t)sizeof(cx<R>) * N;
    const int64_t C = std::max<int64_t>(1, std::min<int64_t>(units, ((int64_t)tunables().big_chunk_mib << 20) / per));
    MDSP_TRY(e->work.reserve((size_t)(per * C)));
    const int cus = device_cu_count();
    for (int64_t c0 = 0; c0 < units; c0 += C) {
        const int64_t cnt = std::min<int64_t>(C, units - c0);
        BigArgs<R> a{};
        a.p = e->pass[0];
        a.in_mode = CPLX ? 2 : 1;
        a.s = s;
        a.win = win;
        a.buf = e->work.as<cx<R>>();
        a.t0 = c0;
        a.ntrans = cnt;
        a.K = K;
        a.hop = hop;
        a.n = (int)e->n;
        a.ablate = tunables().big_ablate;
        const int64_t wgs = (int64_t)cus * (tunables().big_wgs > 0 ? tunables().big_wgs : (a.p.fTJ && a.p.Rp <= 64 ? 4 : 2));
        int groups = tunables().big_groups > 0 ? tunables().big_groups : (int)cdiv(wgs, a.p.ntiles);
        groups = (int)std::max<int64_t>(1, std::min<int64_t>(groups, cnt));
        const int lanes = (int)std::max<int64_t>(1, std::min<int64_t>(a.p.ntiles, wgs / groups));
        MDSP_TRY((launch_pass<R, 0>(a, lanes, groups, st)));
        MDSP_TRY(welch_rows_accumulate(e->rowplan, e->work.p, cnt, N, e->rowsR0, st));
    }
    hipLaunchKernelGGL(big_rows_to_natural_kernel, dim3((unsigned)(S / 32), (unsigned)(e->rowsR0 / 32)), dim3(256), 0, st, e->rowplan->reduced.as<double>(), acc, e->rowsR0, S,
                       fresh ? 0 : 1);
    MDSP_LAUNCH_CHECK();
    return MDSP_OK;
}

}  // namespace

static int rows_r0(int dtype, int64_t N) {
    const int64_t S = dtype_is_double(dtype) ? 4096 : 8192;
    if (N % S) return 0;
    const int64_t r = N / S;
    return (r == 32 || r == 64 || r == 128 || r == 256) ? (int)r : 0;
}
int ols_rows_r0(int dtype, int64_t N) { return tunables().big_ols_rows == 0 ? 0 : rows_r0(dtype, N); }

int64_t ols_size(int dtype, int64_t nb, int64_t nout_hint) {
    if (!tunables().bigfft || nb < 2) return 0;
    const bool dbl = dtype_is_double(dtype);
    // The rows form first (profiles/r05_big_ols.json): R0 = 64 rows -- the column pass whose tiles are 32 columns wide, 128-byte runs of a real Float32 signal -- is
    // the fastest while the filter stays under a quarter of the block (Float32 2^19: 0.90 / 0.86 / 0.75 TB/s at 32768 / 65536 / 131072 taps; Float64 2^18: 1.00
    // at 32768), then R0 = 256 (Float32 2^21: 0.73; Float64 2^20: 0.83 / 0.79 at 65536 / 131072); R0 = 128 measured slowest at every length (0.62 / 0.71).
    // Beyond those, three passes each way on natural-order spectra: 2^20 points, more while the filter covers over an eighth of the block (0.41 - 0.58).
    const int64_t S = dbl ? 4096 : 8192;
    int64_t N = 0;
    if (tunables().big_ols_rows != 0) {
        if (4 * nb <= 64 * S) N = 64 * S;
        else if (4 * nb <= 256 * S) N = 256 * S;
    }
    if (N == 0) {
        N = (int64_t)1 << 20;
        while (N < 8 * nb) N <<= 1;
    }
    // less when one block already holds the whole signal
    while (nout_hint > 0 && N / 2 >= 2 * nb && N / 2 - (nb - 1) >= nout_hint) N >>= 1;
    if (tunables().big_ols_log2n > 0) N = (int64_t)1 << tunables().big_ols_log2n;
    if (N > ((int64_t)1 << 26) || N < 2 * nb || N <= 4096) return 0;
    if (ols_rows_r0(dtype, N)) return N;
    int R[MAXP];
    return factorise(N, R, tunables().big_rmax) >= 2 ? N : 0;
}

int ols(EngineHolder& h, int dtype, int64_t N, int R0, const void* x, int64_t nx, const void* H, int64_t nb, void* y, int64_t nout, int64_t g0, int64_t g1, hipStream_t st) {
    Engine* e = nullptr;
    if (R0) {
        MDSP_TRY(get_engine_rows(h, dtype, 0, N, R0, &e));
        const bool cplx = dtype_is_complex(dtype);
        return dtype_is_double(dtype) ? run_ols_rows<double>(e, cplx, x, nx, H, nb, y, nout, g0, g1, st) : run_ols_rows<float>(e, cplx, x, nx, H, nb, y, nout, g0, g1, st);
    }
    MDSP_TRY(get_engine(h, dtype, 0, N, &e));
    const bool cplx = dtype_is_complex(dtype);
    return dtype_is_double(dtype) ? run_ols<double>(e, cplx, x, nx, H, nb, y, nout, g0, g1, st) : run_ols<float>(e, cplx, x, nx, H, nb, y, nout, g0, g1, st);
}

bool size_ok(int dtype, int64_t nfft) {
    if (!tunables().bigfft) return false;
    // above the single-workgroup kernels of spectral.hip (which keep precedence where they exist: power-of-two register forms to 8192 / 4096,
    // mixed-radix LDS forms to 8192 / 8000 -- spectral.hip use_big)
    (void)dtype;
    if (nfft <= 4096 || nfft >= ((int64_t)1 << 31) || !seven_smooth(nfft)) return false;
    int R[MAXP];
    return factorise(nfft, R, tunables().big_rmax) >= 2;
}

int welch(EngineHolder& h, int dtype, int64_t n, int64_t nfft, const void* s, int64_t K, int64_t hop, const double* win_dev, double* acc, bool fresh,
          hipStream_t st) {
    Engine* e = nullptr;
    const bool cplx = dtype_is_complex(dtype), dbl = dtype_is_double(dtype);
    if (K == 0) {
        if (fresh) MDSP_HIP(hipMemsetAsync(acc, 0, sizeof(double) * (size_t)nfft, st));
        return MDSP_OK;
    }
    // nfft = R0 x S, R0 = 32 .. 256: column pass + the single-workgroup Welch kernel over the rows.  Measured against three passes (profiles/r05_welch_rows.json):
    // 255 .. 2047 frames of a 2^27-sample stream 1.14 - 1.92x faster (Float32 at R0 = 128: 0.99x with the 8 x 16 column pass, 1.17x with the 16 x 8 one that
    // replaced it), the 15 frames of the default call 0.9 - 1.3x.
    // With few transforms in the call (the 15 frames of the reference's default n = length >> 3 are 8 of them) the 128- and 256-row forms in Float32 lose 6 - 11 % to
    // three passes -- 2^20 / 2^21 points 0.28 / 0.34 against 0.32 / 0.36 TB/s: they keep three passes below 32 transforms.
    int R0w = tunables().big_welch_rows ? rows_r0(dtype, nfft) : 0;
    if (R0w >= 128 && !dbl && tunables().big_welch_rows != 2 && (cplx ? K : (K + 1) / 2) < 32) R0w = 0;
    if (const int R0 = R0w) {
        MDSP_TRY(get_engine_rows(h, dtype, n, nfft, R0, &e));
        if (cplx) return dbl ? run_welch_rows<double, true>(e, s, K, hop, win_dev, acc, fresh, st) : run_welch_rows<float, true>(e, s, K, hop, win_dev, acc, fresh, st);
        return dbl ? run_welch_rows<double, false>(e, s, K, hop, win_dev, acc, fresh, st) : run_welch_rows<float, false>(e, s, K, hop, win_dev, acc, fresh, st);
    }
    MDSP_TRY(get_engine(h, dtype, n, nfft, &e));
    if (cplx) return dbl ? run<double, true>(e, 0, s, K, hop, win_dev, acc, fresh, nullptr, 0, 0, 0, 0, 0, 1.0, st)
                         : run<float, true>(e, 0, s, K, hop, win_dev, acc, fresh, nullptr, 0, 0, 0, 0, 0, 1.0, st);
    return dbl ? run<double, false>(e, 0, s, K, hop, win_dev, acc, fresh, nullptr, 0, 0, 0, 0, 0, 1.0, st)
               : run<float, false>(e, 0, s, K, hop, win_dev, acc, fresh, nullptr, 0, 0, 0, 0, 0, 1.0, st);
}

int stft(EngineHolder& h, int dtype, int64_t n, int64_t nfft, const void* s, int64_t K, int64_t hop, const double* win_dev, void* out, int64_t ldo,
         int64_t nout, int onesided, int psd, int accumulate, double r, hipStream_t st) {
    Engine* e = nullptr;
    MDSP_TRY(get_engine(h, dtype, n, nfft, &e));
    const bool cplx = dtype_is_complex(dtype), dbl = dtype_is_double(dtype);
    if (K == 0) return MDSP_OK;
    if (cplx) return dbl ? run<double, true>(e, 1, s, K, hop, win_dev, nullptr, false, out, ldo, nout, onesided, psd, accumulate, r, st)
                         : run<float, true>(e, 1, s, K, hop, win_dev, nullptr, false, out, ldo, nout, onesided, psd, accumulate, r, st);
    return dbl ? run<double, false>(e, 1, s, K, hop, win_dev, nullptr, false, out, ldo, nout, onesided, psd, accumulate, r, st)
               : run<float, false>(e, 1, s, K, hop, win_dev, nullptr, false, out, ldo, nout, onesided, psd, accumulate, r, st);
}

}  // namespace big
}  // namespace mdsp
