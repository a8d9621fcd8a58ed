// N-dimensional convolution on the device: conv(u, v) for arrays (dspbase.jl:709-792).
//
//   * mdsp_convnd_fft    -- _conv_kern_fft! (dspbase.jl:611-644): zero-pad both operands to nextfastfft(outsize) per
//                           dimension, one N-d transform each (rfft for real eltypes), spectrum product, inverse, crop.
//                           rocFFT N-d plans (N <= 3); the reference's overlap-save variant (:490-609) computes the same
//                           sums block-wise on the CPU to bound FFTW plan sizes -- with 288 GB of HBM a single transform
//                           of the padded output is the simpler, faster shape, and the results agree to rounding.
//   * mdsp_convnd_direct -- _conv_td! (dspbase.jl:646-660): the convolution sum itself, one thread per output element
//                           (gather form), any N <= 8.  Used for the reference's :direct choices (small operands, integer
//                           eltypes computed exactly in Float64).
//
// Arrays are column-major (first dimension fastest), as Julia stores them.
#include <algorithm>
#include <mutex>
#include <vector>

#include "common.h"
#include "fft_lds.h"
#include "rocfft_wrap.h"

using namespace mdsp;
using mdsp::fft::cx;

extern "C" int64_t mdsp_nextfastfft(int64_t n);

namespace {

constexpr int MAXD = 8;

struct Dims {
    int nd;
    int64_t d[MAXD];
    __host__ __device__ int64_t count() const {
        int64_t c = 1;
        for (int i = 0; i < nd; ++i) c *= d[i];
        return c;
    }
};

template <typename T> __device__ __forceinline__ T zero_of() { return T{}; }

// dst (dims n) = src (dims s) in the corner, zero elsewhere; ld0 = element stride between dst's columns along
// dimension 0 (n0 for the contiguous layouts used here)
template <typename T> __global__ __launch_bounds__(256) void pad_nd_kernel(const T* __restrict__ src, Dims s, T* __restrict__ dst, Dims n, int64_t ld0) {
    const int64_t total = n.count();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i, si = 0, sstride = 1;
        bool inside = true;
        int64_t dsti = 0, dstride = 1;
        for (int d = 0; d < n.nd; ++d) {
            const int64_t c = r % n.d[d];
            r /= n.d[d];
            inside = inside && c < s.d[d];
            si += c * sstride;
            sstride *= s.d[d];
            dsti += c * dstride;
            dstride *= d == 0 ? ld0 : n.d[d];
        }
        dst[dsti] = inside ? src[si] : zero_of<T>();
    }
}

// dst (dims o, contiguous) = src[0:o] (dims n, first-dimension stride ld0)
template <typename T> __global__ __launch_bounds__(256) void crop_nd_kernel(const T* __restrict__ src, Dims n, int64_t ld0, T* __restrict__ dst, Dims o) {
    const int64_t total = o.count();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i, si = 0, sstride = 1;
        for (int d = 0; d < o.nd; ++d) {
            const int64_t c = r % o.d[d];
            r /= o.d[d];
            si += c * sstride;
            sstride *= d == 0 ? ld0 : n.d[d];
        }
        dst[i] = src[si];
    }
}

template <typename R> __global__ __launch_bounds__(256) void spectrum_product_kernel(cx<R>* __restrict__ a, const cx<R>* __restrict__ b, int64_t count, R scale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const cx<R> x = a[i], y = b[i];
        a[i] = {(x.x * y.x - x.y * y.y) * scale, (x.x * y.y + x.y * y.x) * scale};
    }
}

template <typename T> __device__ __forceinline__ void mac(T& acc, T a, T b) { acc = fma(a, b, acc); }
template <typename R> __device__ __forceinline__ void mac(cx<R>& acc, cx<R> a, cx<R> b) {
    acc.x = fma(a.x, b.x, fma(-a.y, b.y, acc.x));
    acc.y = fma(a.x, b.y, fma(a.y, b.x, acc.y));
}

// out[k] = sum_m small[m] big[k - m]: one thread per output element, the smaller operand is the loop
template <typename T>
__global__ __launch_bounds__(256) void direct_nd_kernel(const T* __restrict__ big, Dims sb, const T* __restrict__ small, Dims ss, T* __restrict__ out, Dims so) {
    const int64_t total = so.count(), nsmall = ss.count();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t k[MAXD];
        int64_t r = i;
        for (int d = 0; d < so.nd; ++d) {
            k[d] = r % so.d[d];
            r /= so.d[d];
        }
        T acc = zero_of<T>();
        for (int64_t j = 0; j < nsmall; ++j) {
            int64_t q = j, bi = 0, bstride = 1;
            bool inside = true;
            for (int d = 0; d < so.nd; ++d) {
                const int64_t m = q % ss.d[d];
                q /= ss.d[d];
                const int64_t c = k[d] - m;
                inside = inside && c >= 0 && c < sb.d[d];
                bi += c * bstride;
                bstride *= sb.d[d];
            }
            if (inside) mac(acc, small[j], big[bi]);
        }
        out[i] = acc;
    }
}

dim3 grid_for(int64_t n) { return dim3((unsigned)std::min<int64_t>(cdiv(n, (int64_t)256), (int64_t)device_cu_count() * 32)); }

// Transforms and work buffers of the most recent padded sizes, per dtype: repeated convolutions of same-sized operands (image
// batches, sliding kernels) pay plan creation and allocation once.  One entry per dtype, guarded by a mutex; a call holds
// the lock for its duration (the buffers are shared state), so concurrent callers of one dtype serialise.
struct ConvCache {
    std::mutex mu;
    Dims n{};
    bool valid = false;
    RocPlan fwd, inv;
    DevBuf pad, a, b;
    hipStream_t last = nullptr;   // stream of the previous call: a call on another stream waits for it before reusing the buffers
    bool used = false;
    int enter(hipStream_t st) {
        if (used && last != st) MDSP_HIP(hipStreamSynchronize(last));
        last = st;
        used = true;
        return MDSP_OK;
    }
};
ConvCache& conv_cache(int dtype) {
    static ConvCache c[4];
    return c[dtype - MDSP_F32];
}
bool same_dims(const Dims& x, const Dims& y) {
    if (x.nd != y.nd) return false;
    for (int d = 0; d < x.nd; ++d)
        if (x.d[d] != y.d[d]) return false;
    return true;
}

template <typename R> int conv_fft_real(int dtype, const R* u, const Dims& su, const R* v, const Dims& sv, R* out, const Dims& so, const Dims& n, hipStream_t st) {
    // out-of-place real transforms with rocFFT's default contiguous layouts: real (n0, n1, ..), Hermitian (n0/2 + 1, n1, ..)
    const int64_t h0 = n.d[0] / 2 + 1;
    int64_t rest = 1;
    for (int d = 1; d < n.nd; ++d) rest *= n.d[d];
    const int64_t nspec = h0 * rest, total = n.count();
    ConvCache& c = conv_cache(dtype);
    std::lock_guard<std::mutex> lk(c.mu);
    MDSP_TRY(c.enter(st));
    if (!c.valid || !same_dims(c.n, n)) {
        c.valid = false;
        MDSP_HIP(hipStreamSynchronize(st));   // a previous call on this stream may still use the buffers about to be replaced
        MDSP_TRY(c.pad.reserve(sizeof(R) * (size_t)total));
        MDSP_TRY(c.a.reserve(sizeof(cx<R>) * (size_t)nspec));
        MDSP_TRY(c.b.reserve(sizeof(cx<R>) * (size_t)nspec));
        MDSP_TRY(c.fwd.create_nd(FftKind::R2C, sizeof(R) == 8, n.nd, n.d, false));
        MDSP_TRY(c.inv.create_nd(FftKind::C2R, sizeof(R) == 8, n.nd, n.d, false));
        c.n = n;
        c.valid = true;
    }
    hipLaunchKernelGGL(pad_nd_kernel<R>, grid_for(total), dim3(256), 0, st, u, su, c.pad.as<R>(), n, n.d[0]);
    MDSP_LAUNCH_CHECK();
    MDSP_TRY(c.fwd.exec(c.pad.p, c.a.p, st));
    hipLaunchKernelGGL(pad_nd_kernel<R>, grid_for(total), dim3(256), 0, st, v, sv, c.pad.as<R>(), n, n.d[0]);
    MDSP_LAUNCH_CHECK();
    MDSP_TRY(c.fwd.exec(c.pad.p, c.b.p, st));
    hipLaunchKernelGGL(spectrum_product_kernel<R>, grid_for(nspec), dim3(256), 0, st, c.a.as<cx<R>>(), c.b.as<cx<R>>(), nspec, (R)(1.0 / (double)total));
    MDSP_LAUNCH_CHECK();
    MDSP_TRY(c.inv.exec(c.a.p, c.pad.p, st));
    hipLaunchKernelGGL(crop_nd_kernel<R>, grid_for(so.count()), dim3(256), 0, st, c.pad.as<R>(), n, n.d[0], out, so);
    MDSP_LAUNCH_CHECK();
    return MDSP_OK;   // stream-ordered: the cached buffers are only reused by later launches on a stream the next call synchronises with
}

template <typename R> int conv_fft_complex(int dtype, const cx<R>* u, const Dims& su, const cx<R>* v, const Dims& sv, cx<R>* out, const Dims& so, const Dims& n, hipStream_t st) {
    const int64_t total = n.count();
    ConvCache& c = conv_cache(dtype);
    std::lock_guard<std::mutex> lk(c.mu);
    MDSP_TRY(c.enter(st));
    if (!c.valid || !same_dims(c.n, n)) {
        c.valid = false;
        MDSP_HIP(hipStreamSynchronize(st));
        MDSP_TRY(c.a.reserve(sizeof(cx<R>) * (size_t)total));
        MDSP_TRY(c.b.reserve(sizeof(cx<R>) * (size_t)total));
        MDSP_TRY(c.fwd.create_nd(FftKind::C2C_FWD, sizeof(R) == 8, n.nd, n.d, true));
        MDSP_TRY(c.inv.create_nd(FftKind::C2C_INV, sizeof(R) == 8, n.nd, n.d, true));
        c.n = n;
        c.valid = true;
    }
    hipLaunchKernelGGL(pad_nd_kernel<cx<R>>, grid_for(total), dim3(256), 0, st, u, su, c.a.as<cx<R>>(), n, n.d[0]);
    hipLaunchKernelGGL(pad_nd_kernel<cx<R>>, grid_for(total), dim3(256), 0, st, v, sv, c.b.as<cx<R>>(), n, n.d[0]);
    MDSP_LAUNCH_CHECK();
    MDSP_TRY(c.fwd.exec(c.a.p, c.a.p, st));
    MDSP_TRY(c.fwd.exec(c.b.p, c.b.p, st));
    hipLaunchKernelGGL(spectrum_product_kernel<R>, grid_for(total), dim3(256), 0, st, c.a.as<cx<R>>(), c.b.as<cx<R>>(), total, (R)(1.0 / (double)total));
    MDSP_LAUNCH_CHECK();
    MDSP_TRY(c.inv.exec(c.a.p, c.a.p, st));
    hipLaunchKernelGGL(crop_nd_kernel<cx<R>>, grid_for(so.count()), dim3(256), 0, st, c.a.as<cx<R>>(), n, n.d[0], out, so);
    MDSP_LAUNCH_CHECK();
    return MDSP_OK;
}

int read_dims(const int64_t* su, const int64_t* sv, int ndim, Dims& u, Dims& v, Dims& o) {
    if (ndim < 1 || ndim > MAXD) MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "convolution of %d-dimensional arrays (1..%d supported)", ndim, MAXD);
    if (!su || !sv) MDSP_FAIL(MDSP_ERR_ARGUMENT, "NULL size array");
    u.nd = v.nd = o.nd = ndim;
    for (int d = 0; d < ndim; ++d) {
        if (su[d] < 1 || sv[d] < 1) MDSP_FAIL(MDSP_ERR_ARGUMENT, "empty operand: the caller zero-fills the output (dspbase.jl:730)");
        u.d[d] = su[d];
        v.d[d] = sv[d];
        o.d[d] = su[d] + sv[d] - 1;
    }
    return MDSP_OK;
}

}  // namespace

extern "C" {

int mdsp_convnd_fft(const void* u_dev, const int64_t* su, const void* v_dev, const int64_t* sv, int ndim, int dtype, void* out_dev, void* stream) {
    if (!dtype_valid(dtype)) MDSP_FAIL(MDSP_ERR_ARGUMENT, "invalid dtype");
    Dims u, v, o, n;
    MDSP_TRY(read_dims(su, sv, ndim, u, v, o));
    if (!u_dev || !v_dev || !out_dev) MDSP_FAIL(MDSP_ERR_ARGUMENT, "NULL buffer");
    // drop the dimensions of extent 1 in both operands (a trailing singleton promoted by conv(A::M-d, B::N-d), :784-792): they
    // do not change the memory layout, and rocFFT plans have at most three
    Dims uc{}, vc{}, oc{};
    for (int d = 0; d < ndim; ++d) {
        if (u.d[d] == 1 && v.d[d] == 1) continue;
        uc.d[uc.nd++] = u.d[d];
        vc.d[vc.nd++] = v.d[d];
        oc.d[oc.nd++] = o.d[d];
    }
    if (uc.nd == 0) {
        uc.nd = vc.nd = oc.nd = 1;
        uc.d[0] = vc.d[0] = oc.d[0] = 1;
    }
    if (uc.nd > 3) MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "FFT convolution over %d non-trivial dimensions (rocFFT plans have at most 3); use the direct algorithm", uc.nd);
    n.nd = uc.nd;
    for (int d = 0; d < uc.nd; ++d) n.d[d] = mdsp_nextfastfft(oc.d[d]);
    MDSP_TRY(rocfft_ensure_setup());
    hipStream_t st = as_stream(stream);
    switch (dtype) {
        case MDSP_F32: return conv_fft_real<float>(dtype, (const float*)u_dev, uc, (const float*)v_dev, vc, (float*)out_dev, oc, n, st);
        case MDSP_F64: return conv_fft_real<double>(dtype, (const double*)u_dev, uc, (const double*)v_dev, vc, (double*)out_dev, oc, n, st);
        case MDSP_C32: return conv_fft_complex<float>(dtype, (const cx<float>*)u_dev, uc, (const cx<float>*)v_dev, vc, (cx<float>*)out_dev, oc, n, st);
        default: return conv_fft_complex<double>(dtype, (const cx<double>*)u_dev, uc, (const cx<double>*)v_dev, vc, (cx<double>*)out_dev, oc, n, st);
    }
}

int mdsp_convnd_direct(const void* u_dev, const int64_t* su, const void* v_dev, const int64_t* sv, int ndim, int dtype, void* out_dev, void* stream) {
    if (!dtype_valid(dtype)) MDSP_FAIL(MDSP_ERR_ARGUMENT, "invalid dtype");
    Dims u, v, o;
    MDSP_TRY(read_dims(su, sv, ndim, u, v, o));
    if (!u_dev || !v_dev || !out_dev) MDSP_FAIL(MDSP_ERR_ARGUMENT, "NULL buffer");
    const bool u_big = u.count() >= v.count();
    const Dims& sb = u_big ? u : v;
    const Dims& ss = u_big ? v : u;
    const void* big = u_big ? u_dev : v_dev;
    const void* small = u_big ? v_dev : u_dev;
    hipStream_t st = as_stream(stream);
    const dim3 g = grid_for(o.count());
    switch (dtype) {
        case MDSP_F32: hipLaunchKernelGGL(direct_nd_kernel<float>, g, dim3(256), 0, st, (const float*)big, sb, (const float*)small, ss, (float*)out_dev, o); break;
        case MDSP_F64: hipLaunchKernelGGL(direct_nd_kernel<double>, g, dim3(256), 0, st, (const double*)big, sb, (const double*)small, ss, (double*)out_dev, o); break;
        case MDSP_C32:
            hipLaunchKernelGGL(direct_nd_kernel<cx<float>>, g, dim3(256), 0, st, (const cx<float>*)big, sb, (const cx<float>*)small, ss, (cx<float>*)out_dev, o);
            break;
        default:
            hipLaunchKernelGGL(direct_nd_kernel<cx<double>>, g, dim3(256), 0, st, (const cx<double>*)big, sb, (const cx<double>*)small, ss, (cx<double>*)out_dev, o);
            break;
    }
    MDSP_LAUNCH_CHECK();
    return MDSP_OK;
}

}  // extern "C"
