// Host run of every compile-time schedule of the mixed-radix spectral kernels (dsp.jl_amd/csrc/ct_sched.h: MDSP_GEN_CT_SIZES, MDSP_GEN_CT_WIDE_SIZES) with
// the schedule type's OWN index arithmetic -- radices, butterflies per pass, twiddle indices, group padding (CtSched::padded / gin / rstride / NP), the
// single-buffer form -- and the butterflies of fft_lds.h (composite radices included), against a Float64 DFT.  The device code (spectral_gen.h ct_passes,
// ct_passes_inplace, ct_pass0_compute, ct_last_pass_regs) uses the same expressions; a table entry whose radices, padding or buffer size do not fit
// together fails here, without a GPU.
//     g++ -O2 -std=c++17 tests/cpu_harness/ct_layout.cpp -o ct_layout && ./ct_layout
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <vector>

#include "../../dsp.jl_amd/csrc/fft_lds.h"
#include "../../dsp.jl_amd/csrc/ct_sched.h"
#ifdef CT_BIG   // the single-workgroup tables of round 6 (ctbig_sizes.h: 177 schedules up to 16384 points); the reference DFT on every CT_BIG-th bin
#include "../../dsp.jl_amd/csrc/ctbig_sizes.h"
#endif

using namespace mdsp::fft;

template <typename S, typename R, int p>
void run_passes(std::vector<cx<R>>& a, std::vector<cx<R>>& b, const std::vector<cx<R>>& roots, std::vector<cx<R>>& result, bool inplace) {
    if constexpr (p < S::P) {
        constexpr int Rdx = S::radix(p), Ns = S::ns(p), nbf = S::nbf(p), N = S::N;
        constexpr int stride = N / (Ns * Rdx);
        std::vector<cx<R>>& in = a;
        std::vector<cx<R>>& out = inplace ? a : b;
        std::vector<cx<R>> vals((size_t)nbf * Rdx);
        for (int j = 0; j < nbf; ++j) {                      // every butterfly of the pass: all of its reads first (the single-buffer form)
            const unsigned jb = S::gin(p) ? (unsigned)j + (unsigned)j / (unsigned)(S::gin(p) ? S::gin(p) : 1) : (unsigned)j;
            for (int q = 0; q < Rdx; ++q) {
                const size_t idx = (size_t)jb + (size_t)S::rstride(p) * q;
                if (idx >= in.size()) { printf("read out of range: N %d pass %d\n", N, p); exit(1); }
                vals[(size_t)j * Rdx + q] = in[idx];
            }
        }
        if (inplace) std::fill(out.begin(), out.end(), cx<R>{std::numeric_limits<R>::quiet_NaN(), std::numeric_limits<R>::quiet_NaN()});
        for (int j = 0; j < nbf; ++j) {
            cx<R> v[Rdx];
            for (int q = 0; q < Rdx; ++q) v[q] = vals[(size_t)j * Rdx + q];
            const unsigned k = (unsigned)j % (unsigned)Ns, hi = (unsigned)j / (unsigned)Ns;
            if constexpr (p > 0)
                for (int q = 1; q < Rdx; ++q) v[q] = cmul(v[q], roots[((unsigned)q * k * (unsigned)stride) % (unsigned)N]);
            gen_bfly<Rdx>(v);
            if constexpr (p == S::P - 1) {
                for (int q = 0; q < Rdx; ++q) result[j + nbf * q] = v[q];        // natural order, from registers
            } else {
                const size_t o = (size_t)hi * (size_t)(Ns * Rdx + (S::padded(p) ? 1 : 0)) + k;
                for (int q = 0; q < Rdx; ++q) {
                    if (o + (size_t)Ns * q >= out.size()) { printf("write out of range: N %d pass %d\n", N, p); exit(1); }
                    out[o + (size_t)Ns * q] = v[q];
                }
            }
        }
        if (inplace) run_passes<S, R, p + 1>(a, b, roots, result, inplace);
        else run_passes<S, R, p + 1>(b, a, roots, result, inplace);
    }
}

template <typename S, typename R> double check_schedule(bool inplace) {
    constexpr int N = S::N;
    static_assert(S::NP >= N, "buffer");
    std::vector<cx<R>> roots(N), a(S::NP, cx<R>{std::numeric_limits<R>::quiet_NaN(), 0}), b(S::NP, cx<R>{std::numeric_limits<R>::quiet_NaN(), 0}), result(N);
    std::vector<std::complex<double>> x(N), w(N);
    srand(N + (inplace ? 1 : 0));
    for (int i = 0; i < N; ++i) {
        const double ang = -2.0 * M_PI * i / N;
        w[i] = {cos(ang), sin(ang)};
        roots[i] = {(R)w[i].real(), (R)w[i].imag()};
        x[i] = {rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5};
    }
    // pass 0 reads the frame in natural order (gin(0) = 0): element i at a[i]
    for (int i = 0; i < N; ++i) a[i] = {(R)x[i].real(), (R)x[i].imag()};
    run_passes<S, R, 0>(a, b, roots, result, inplace);
    double err2 = 0, norm = 0;
#ifdef CT_BIG
    constexpr int KSTEP = N > 4096 ? CT_BIG : 7;
#else
    constexpr int KSTEP = 1;
#endif
    for (int k = 0; k < N; k += KSTEP) {
        std::complex<double> acc = 0;
        for (int n = 0; n < N; ++n) acc += std::complex<double>((double)(R)x[n].real(), (double)(R)x[n].imag()) * w[(int)(((long long)n * k) % N)];
        err2 += std::norm(std::complex<double>(result[k].x, result[k].y) - acc);
        norm += std::norm(acc);
    }
    return sqrt(err2 / norm);
}

int main() {
    int bad = 0, count = 0;
#define MDSP_X(N, T, F, ...)                                                                                                         \
    {                                                                                                                                \
        using S = CtSched<N, T, F, __VA_ARGS__>;                                                                                     \
        static_assert(sizeof(cx<float>) * ((S::INPLACE ? 1 : 2) * (size_t)S::NP + (S::TW2L ? S::TWS + S::NTWHI : 0)) <= 160 * 1024, "LDS");   \
        const double e2 = S::INPLACE ? 0.0 : check_schedule<S, float>(false), e1 = check_schedule<S, float>(true);                 \
        const bool ok = e1 < 3e-6 && e2 < 3e-6;                                                                                      \
        printf("N %5d T %3d flags %4d passes %d NP %5d: two buffers %.2e, one buffer %.2e%s\n", N, T, F, S::P, S::NP, e2, e1, ok ? "" : "  FAIL"); \
        bad |= !ok;                                                                                                                  \
        ++count;                                                                                                                     \
    }
#ifdef CT_BIG
    MDSP_CTBIG_SIZES(MDSP_X)
    MDSP_CTBIG_LEAN_SIZES(MDSP_X)
    MDSP_CTBIG_SMALL_SIZES(MDSP_X)
    MDSP_CTBIG_PREF_SIZES(MDSP_X)
#else
    MDSP_GEN_CT_SIZES(MDSP_X)
    MDSP_GEN_CT_WIDE_SIZES(MDSP_X)
#endif
#undef MDSP_X
    {   // Float64 on the small-radix list with the padding bits stripped, as gen_ct_dispatch instantiates it
        using S = CtSched<3000, 384, 4 & ~1536, 3, 5, 5, 5, 8>;
        const double e = check_schedule<S, double>(true);
        printf("N 3000 Float64, one buffer: %.2e\n", e);
        bad |= !(e < 1e-14);
    }
    printf("%d schedules: %s\n", count, bad ? "FAIL" : "OK");
    return bad;
}
