// Small host-side double-precision FFT, used once per plan to transform the FIR taps into the filter
// spectrum (the role of `mul!(filterft, p1, tmp1)` Filters/filt.jl:501 and os_filter_transform!
// dspbase.jl:324-330).  Recursive mixed-radix Cooley-Tukey; O(N * sum(prime factors)).
#pragma once

#include <cmath>
#include <complex>
#include <vector>

namespace mdsp {

using zd = std::complex<double>;

inline zd unit_root(int64_t k, int64_t n, int sign) {
    // exp(sign * 2*pi*i * k / n) with exact octant reduction of k/n
    k %= n;
    if (k < 0) k += n;
    const long double t = 2.0L * 3.141592653589793238462643383279502884L * (long double)k / (long double)n;
    return zd((double)cosl(t), sign * (double)sinl(t));
}

inline void host_fft_rec(const zd* in, int64_t stride, zd* out, int64_t n, int sign, int64_t nroot,
                         const std::vector<zd>& roots, std::vector<zd>& scratch_pool) {
    if (n == 1) {
        out[0] = in[0];
        return;
    }
    int64_t p = 0;
    for (int64_t f = 2; f * f <= n; ++f)
        if (n % f == 0) {
            p = f;
            break;
        }
    if (p == 0) p = n;  // prime
    const int64_t m = n / p;
    // p sub-transforms of length m over the decimated inputs
    for (int64_t r = 0; r < p; ++r)
        host_fft_rec(in + r * stride, stride * p, out + r * m, m, sign, nroot, roots, scratch_pool);
    // combine: X[k + q*m] = sum_r W_n^{r(k+q m)} * Y_r[k]
    std::vector<zd> tmp(p);
    const int64_t step = nroot / n;
    for (int64_t k = 0; k < m; ++k) {
        for (int64_t r = 0; r < p; ++r) tmp[r] = out[r * m + k] * roots[(size_t)((r * k * step) % nroot)];
        for (int64_t q = 0; q < p; ++q) {
            zd acc = tmp[0];
            for (int64_t r = 1; r < p; ++r) acc += tmp[r] * roots[(size_t)(((r * q * m) % n) * step)];
            scratch_pool[(size_t)(q)] = acc;
        }
        for (int64_t q = 0; q < p; ++q) out[q * m + k] = scratch_pool[(size_t)q];
    }
}

// out[k] = sum_j in[j] * exp(sign*2*pi*i*j*k/n)
inline std::vector<zd> host_fft(const std::vector<zd>& in, int sign) {
    const int64_t n = (int64_t)in.size();
    std::vector<zd> out((size_t)n);
    if (n == 0) return out;
    std::vector<zd> roots((size_t)n);
    for (int64_t k = 0; k < n; ++k) roots[(size_t)k] = unit_root(k, n, sign);
    std::vector<zd> pool((size_t)n);
    host_fft_rec(in.data(), 1, out.data(), n, sign, n, roots, pool);
    return out;
}

}  // namespace mdsp
