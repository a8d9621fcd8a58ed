// Host run of the run-time-schedule spectral kernel's planner and pass code (dsp.jl_amd/csrc/gx_sched.h, gx_pass.h): for a list of 7-smooth sizes --
// or every size given on the command line -- plan the schedule, run the passes thread by thread in the kernel's phase order (pass 0 from "registers",
// read | barrier | butterflies + write | barrier, last pass consumed from registers) on one buffer of exactly Sched::np elements, and compare with a
// Float64 DFT.  Out-of-range LDS indices, overlapping writes and unplannable sizes fail here, without a GPU.
//     g++ -O2 -std=c++17 tests/cpu_harness/gx_emul.cpp -o gx_emul && ./gx_emul [f32|f64] [N ...]
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "../../dsp.jl_amd/csrc/gx_pass.h"

using namespace mdsp::fft;
using namespace mdsp::gx;

template <typename R, int EMAX> struct Emu {
    Sched s;
    std::vector<cx<R>> lds, lo1, hi;
    std::vector<std::vector<cx<R>>> regs;   // per thread v[EMAX]
    std::vector<char> written;

    template <int RR> void pass0(const std::vector<cx<R>>& x) {
        for (int t = 0; t < s.T; ++t) {
            cx<R> v[EMAX];
            for (int i = 0; i < EMAX; ++i) v[i] = cx<R>{std::numeric_limits<R>::quiet_NaN(), 0};
            constexpr int MMAX = EMAX / RR;
            for (int m = 0; m < MMAX; ++m)
                if (m < s.M[0]) {
                    const Bf b = bf_of(pass_of(s, 0), t, m);
                    for (int q = 0; q < RR; ++q) v[m * RR + q] = x[b.j + (unsigned)s.nbf[0] * q];
                }
            pass_butterflies<RR, EMAX>(pass_of(s, 0), t, v, lo1.data(), hi.data());   // p == 0: no twiddles
            pass_write_checked<RR>(0, t, v);
        }
    }
    template <int RR> void pass_write_checked(int p, int t, cx<R> (&v)[EMAX]) {
        // the same indices pass_write uses, with range and collision checks
        constexpr int MMAX = EMAX / RR;
        for (int m = 0; m < MMAX; ++m)
            if (m < s.M[p]) {
                const Bf b = bf_of(pass_of(s, p), t, m);
                if (!b.on) continue;
                for (int q = 0; q < RR; ++q) {
                    const size_t o = (size_t)b.hi * s.gstride[p] + b.k + (size_t)s.ns[p] * q;
                    if (o >= lds.size()) { printf("write out of range N %d pass %d\n", s.N, p); exit(1); }
                    if (written[o]) { printf("write collision N %d pass %d\n", s.N, p); exit(1); }
                    written[o] = 1;
                }
            }
        pass_write<RR, EMAX>(pass_of(s, p), t, lds.data(), v);
    }
    template <int RR> void read_all(int p) {
        for (int t = 0; t < s.T; ++t) {
            cx<R> v[EMAX];
            for (int i = 0; i < EMAX; ++i) v[i] = cx<R>{std::numeric_limits<R>::quiet_NaN(), 0};
            constexpr int MMAX = EMAX / RR;
            for (int m = 0; m < MMAX; ++m)
                if (m < s.M[p]) {
                    const Bf b = bf_of(pass_of(s, p), t, m);
                    const size_t last = (size_t)b.j + (s.gin[p] ? b.hi : 0u) + (size_t)s.rstride[p] * (RR - 1);
                    if (last >= lds.size()) { printf("read out of range N %d pass %d\n", s.N, p); exit(1); }
                }
            pass_read<RR, EMAX>(pass_of(s, p), t, lds.data(), v);
            for (int i = 0; i < EMAX; ++i) regs[t][i] = v[i];
        }
    }
    template <int RR> void mid(int p) {
        read_all<RR>(p);
        std::fill(lds.begin(), lds.end(), cx<R>{std::numeric_limits<R>::quiet_NaN(), 0});   // barrier: everything was read before anything is written
        std::fill(written.begin(), written.end(), 0);
        for (int t = 0; t < s.T; ++t) {
            cx<R> v[EMAX];
            for (int i = 0; i < EMAX; ++i) v[i] = regs[t][i];
            pass_butterflies<RR, EMAX>(pass_of(s, p), t, v, lo1.data(), hi.data());
            pass_write_checked<RR>(p, t, v);
        }
    }
    template <int RR> void last(std::vector<cx<R>>& out, std::vector<char>& got) {
        const int p = s.P - 1;
        read_all<RR>(p);
        for (int t = 0; t < s.T; ++t) {
            cx<R> v[EMAX];
            for (int i = 0; i < EMAX; ++i) v[i] = regs[t][i];
            pass_butterflies<RR, EMAX>(pass_of(s, p), t, v, lo1.data(), hi.data());
            last_consume<RR, EMAX>(pass_of(s, p), t, v, [&](int, unsigned bin, cx<R> z) {
                if (bin >= (unsigned)s.N || got[bin]) { printf("bin %u twice or out of range, N %d\n", bin, s.N); exit(1); }
                got[bin] = 1;
                out[bin] = z;
            });
        }
    }
#define GX_DISPATCH(r, CALL)                                             \
    switch (r) {                                                         \
        MDSP_GX_RADIX_CASES_16(CALL)                                     \
        default:                                                         \
            if constexpr (EMAX >= 32) {                                  \
                switch (r) {                                             \
                    MDSP_GX_RADIX_CASES_32(CALL)                         \
                    default: printf("radix %d\n", r); exit(1);           \
                }                                                        \
            } else { printf("radix %d\n", r); exit(1); }                 \
    }
    // plan N points and run the passes on xin; out in natural order.  < 0: no schedule
    int transform(int N, const std::vector<cx<R>>& xin, std::vector<cx<R>>& out) {
        const int esz = (int)sizeof(cx<R>);
        s = plan(N, EMAX, 512, sizeof(R) == 8 ? 512 : 1024, 160 * 1024, esz, esz * (TWS + tw_hi_entries(N)), esz / 2);
        if (s.P == 0) return -1;
        lds.assign(s.np, cx<R>{std::numeric_limits<R>::quiet_NaN(), 0});
        written.assign(s.np, 0);
        regs.assign(s.T, std::vector<cx<R>>(EMAX));
        lo1.resize(TWS);
        hi.resize(tw_hi_entries(N));
        const long double PI = 3.141592653589793238462643383279502884L;
        for (int i = 0; i < TWS; ++i) {
            const long double sh = sinl(PI * (i % N) / N);
            lo1[i] = {(R)(double)(-2.0L * sh * sh), (R)(double)sinl(-2.0L * PI * (i % N) / N)};
        }
        for (int i = 0; i < (int)hi.size(); ++i) hi[i] = {(R)(double)cosl(-2.0L * PI * ((long long)i * TWS) / N), (R)(double)sinl(-2.0L * PI * ((long long)i * TWS) / N)};
#define P0T(RR) pass0<RR>(xin)
        GX_DISPATCH(s.radix[0], P0T)
        for (int p = 1; p < s.P - 1; ++p) {
#define PMT(RR) mid<RR>(p)
            GX_DISPATCH(s.radix[p], PMT)
        }
        std::vector<char> got(N, 0);
#define PLT(RR) last<RR>(out, got)
        GX_DISPATCH(s.radix[s.P - 1], PLT)
        for (int k = 0; k < N; ++k)
            if (!got[k]) { printf("bin %d never produced, N %d\n", k, N); exit(1); }
        return 0;
    }
    double run(int N) {
        const int esz = (int)sizeof(cx<R>);
        // the device's own planning parameters (spectral_gx.h gx_choose): 512 threads at most, a CU holds 1024 threads of the Float32 kernels (128 registers) and
        // 512 of the Float64 ones, 160 KiB of LDS, the Welch sums (one real per bin) next to the buffers
        s = plan(N, EMAX, 512, sizeof(R) == 8 ? 512 : 1024, 160 * 1024, esz, esz * (TWS + tw_hi_entries(N)), esz / 2);
        if (s.P == 0) return -1;
        lds.assign(s.np, cx<R>{std::numeric_limits<R>::quiet_NaN(), 0});
        written.assign(s.np, 0);
        regs.assign(s.T, std::vector<cx<R>>(EMAX));
        lo1.resize(TWS);
        hi.resize(tw_hi_entries(N));
        // (the tables exist before pass 0 runs: its butterflies take them as arguments and never read them)
        std::vector<std::complex<double>> w(N), x(N);
        for (int i = 0; i < N; ++i) {
            const long double ang = -2.0L * 3.141592653589793238462643383279502884L * i / N;
            w[i] = {(double)cosl(ang), (double)sinl(ang)};
        }
        for (int i = 0; i < TWS && i < N; ++i) lo1[i] = {(R)(double)(cosl(-2.0L * 3.141592653589793238462643383279502884L * i / N) - 1.0L), (R)w[i].imag()};
        for (int i = 0; i < (int)hi.size(); ++i) hi[i] = {(R)w[(size_t)i * TWS].real(), (R)w[(size_t)i * TWS].imag()};
        srand(N);
        std::vector<cx<R>> xin(N), out(N);
        for (int i = 0; i < N; ++i) {
            xin[i] = {(R)(rand() / (double)RAND_MAX - 0.5), (R)(rand() / (double)RAND_MAX - 0.5)};
            x[i] = {(double)xin[i].x, (double)xin[i].y};
        }
#define P0(RR) pass0<RR>(xin)
        GX_DISPATCH(s.radix[0], P0)
        for (int p = 1; p < s.P - 1; ++p) {
#define PM(RR) mid<RR>(p)
            GX_DISPATCH(s.radix[p], PM)
        }
        std::vector<char> got(N, 0);
#define PL(RR) last<RR>(out, got)
        GX_DISPATCH(s.radix[s.P - 1], PL)
        for (int k = 0; k < N; ++k)
            if (!got[k]) { printf("bin %d never produced, N %d\n", k, N); exit(1); }
        double err2 = 0, norm = 0;
        const int step = N > 4096 ? 7 : 1;   // every 7th bin of the long transforms (O(N^2) reference)
        for (int k = 0; k < N; k += step) {
            std::complex<double> acc = 0;
            for (int n = 0; n < N; ++n) acc += x[n] * w[(size_t)(((long long)n * k) % N)];
            err2 += std::norm(std::complex<double>(out[k].x, out[k].y) - acc);
            norm += std::norm(acc);
        }
        return sqrt(err2 / norm);
    }
};

// nfft = R0 S: the decimation-in-frequency step the kernel fuses into its loads (gx_kernels.h), with the kernel's own tables and arithmetic --
//   y_k1[i] = W_nfft^{i k1} sum_n1 x[S n1 + i] W_R0^{n1 k1}   (two-level table for W_nfft, W_R0 table),   X[k1 + R0 k2] = FFT_S(y_k1)[k2]
// -- every row k1 through the planned S-point passes, against a Float64 DFT of nfft points (every 11th bin).
template <typename R, int EMAX> double run_columns(int nfft, int R0) {
    const int S = nfft / R0;
    const long double PI = 3.141592653589793238462643383279502884L;
    std::vector<std::complex<double>> x(nfft);
    std::vector<cx<R>> xin(nfft);
    srand(nfft + R0);
    for (int i = 0; i < nfft; ++i) {
        xin[i] = {(R)(rand() / (double)RAND_MAX - 0.5), (R)(rand() / (double)RAND_MAX - 0.5)};
        x[i] = {(double)xin[i].x, (double)xin[i].y};
    }
    std::vector<cx<R>> lo1N(TWS), hiN(tw_hi_entries(nfft)), cw(R0);
    for (int i = 0; i < TWS; ++i) {
        const long double sh = sinl(PI * i / nfft);
        lo1N[i] = {(R)(double)(-2.0L * sh * sh), (R)(double)sinl(-2.0L * PI * i / nfft)};
    }
    for (int i = 0; i < (int)hiN.size(); ++i) hiN[i] = {(R)(double)cosl(-2.0L * PI * ((long long)i * TWS) / nfft), (R)(double)sinl(-2.0L * PI * ((long long)i * TWS) / nfft)};
    for (int i = 0; i < R0; ++i) cw[i] = {(R)(double)cosl(-2.0L * PI * i / R0), (R)(double)sinl(-2.0L * PI * i / R0)};
    std::vector<std::complex<double>> got(nfft);
    for (int k1 = 0; k1 < R0; ++k1) {
        std::vector<cx<R>> y(S);
        for (int i = 0; i < S; ++i) {
            cx<R> acc = {(R)0, (R)0};
            unsigned cidx = 0;
            for (int n1 = 0; n1 < R0; ++n1) {
                acc = cadd(acc, cmul(xin[(size_t)S * n1 + i], cw[cidx]));
                cidx += (unsigned)k1;
                if (cidx >= (unsigned)R0) cidx -= (unsigned)R0;
            }
            y[i] = k1 == 0 ? acc : cmul(acc, tw2(lo1N.data(), hiN.data(), (unsigned)i * (unsigned)k1));
        }
        Emu<R, EMAX> emu;
        std::vector<cx<R>> out(S);
        if (emu.transform(S, y, out) < 0) return -1;
        for (int k2 = 0; k2 < S; ++k2) got[k1 + (size_t)R0 * k2] = {(double)out[k2].x, (double)out[k2].y};
    }
    double err2 = 0, norm = 0;
    std::vector<std::complex<double>> wn(nfft);
    for (int i = 0; i < nfft; ++i) wn[i] = {(double)cosl(-2.0L * PI * i / nfft), (double)sinl(-2.0L * PI * i / nfft)};
    for (int k = 0; k < nfft; k += 11) {
        std::complex<double> acc = 0;
        for (int n = 0; n < nfft; ++n) acc += x[n] * wn[(size_t)(((long long)n * k) % nfft)];
        err2 += std::norm(got[k] - acc);
        norm += std::norm(acc);
    }
    return sqrt(err2 / norm);
}

int main(int argc, char** argv) {
    bool f64 = argc > 1 && !strcmp(argv[1], "f64");
    std::vector<int> sizes;
    for (int i = 2; i < argc; ++i) sizes.push_back(atoi(argv[i]));
    if (sizes.empty()) {
        // sizes one workgroup holds (the others run as R0 x S: the column cases below)
        if (f64) sizes = {256, 1000, 1125, 2187, 2401, 3125, 4096, 4375, 4800, 5000, 6000, 6144};
        else sizes = {256, 1000, 1125, 2187, 2401, 3125, 4096, 4802, 5000, 6000, 6144, 7000, 8192};
    }
    int bad = 0;
    for (int N : sizes) {
        double e;
        Sched s;
        if (f64) {
            Emu<double, 16> emu;
            e = emu.run(N);
            s = emu.s;
        } else {
            Emu<float, 16> emu;
            e = emu.run(N);
            s = emu.s;
        }
        if (e < 0) { printf("N %6d: no schedule\n", N); if (argc <= 2) ++bad; continue; }
        printf("N %6d T %3d np %6d radices", N, s.T, s.np);
        for (int p = 0; p < s.P; ++p) printf(" %d", s.radix[p]);
        printf("  rel err %.3g\n", e);
        if (!(e < (f64 ? 1e-14 : 6e-7))) { ++bad; printf("   ^^^ too large\n"); }
    }
    // the fused column step (nfft = R0 S)
    const int cols[][2] = {{16384, 2}, {16384, 4}, {12500, 5}, {8400, 2}, {9261, 3}, {40000, 8}, {19683, 9}};
    for (auto& c : cols) {
        if (f64 ? (c[0] > 20000 || c[0] / c[1] > 6144) : (c[0] == 16384 && c[1] == 4)) continue;
        const double e = f64 ? run_columns<double, 16>(c[0], c[1]) : run_columns<float, 16>(c[0], c[1]);
        printf("columns nfft %6d = %d x %d  rel err %.3g\n", c[0], c[1], c[0] / c[1], e);
        if (!(e >= 0 && e < (f64 ? 2e-14 : 8e-7))) { ++bad; printf("   ^^^ too large\n"); }
    }
    printf(bad ? "FAILED\n" : "OK\n");
    return bad != 0;
}
