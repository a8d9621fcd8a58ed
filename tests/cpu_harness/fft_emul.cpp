// Host emulation of the workgroup FFT in dsp.jl_amd/csrc/fft_lds.h: runs the SAME pass_compute / pass_reload
// code thread-by-thread (a barrier = the end of a loop over threads) and checks it against a long-double DFT.
// Built and run by tests/test_fft_core_cpu.py with g++ (no GPU needed).
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../dsp.jl_amd/csrc/fft_lds.h"
#include "../../dsp.jl_amd/csrc/fft_w64.h"

using namespace mdsp::fft;

template <typename C, typename R, int DIR, int TWREG, int PADSHIFT, int PASS, int PERMUTE = false>
static void run_passes(std::vector<cx<R>>& regs, std::vector<cx<R>>& tw, const std::vector<cx<R>>& table, std::vector<cx<R>>& lds) {
    if constexpr (PASS < C::P) {
        constexpr int NTWA = C::NTW > 0 ? C::NTW : 1;
        for (int t = 0; t < C::T; ++t) {
            auto& x = *reinterpret_cast<cx<R>(*)[C::E]>(&regs[(size_t)t * C::E]);
            auto& w = *reinterpret_cast<cx<R>(*)[NTWA]>(&tw[(size_t)t * NTWA]);
            pass_compute<C, DIR, PASS, TWREG, PADSHIFT, PERMUTE>(x, t, w, table.data(), lds.data());
        }
        if constexpr (PASS < C::P - 1) {
            for (int t = 0; t < C::T; ++t) {
                auto& x = *reinterpret_cast<cx<R>(*)[C::E]>(&regs[(size_t)t * C::E]);
                pass_reload<C, PADSHIFT, PASS + 1, PERMUTE>(x, t, lds.data());
            }
        }
        run_passes<C, R, DIR, TWREG, PADSHIFT, PASS + 1, PERMUTE>(regs, tw, table, lds);
    }
}

template <int N, int E, typename R, int DIR, int TWREG, int PADSHIFT, int PERMUTE = false> static double check() {
    using C = Cfg<N, E>;
    constexpr int NTWA = C::NTW > 0 ? C::NTW : 1;
    std::vector<cx<R>> table(N), regs((size_t)C::T * E), tw((size_t)C::T * NTWA), lds(lds_elems<N, PADSHIFT>());
    for (int k = 0; k < N; ++k) {
        const long double a = -2.0L * 3.141592653589793238462643383279502884L * k / N;
        table[k] = {(R)cosl(a), (R)sinl(a)};
    }
    std::vector<std::complex<long double>> in(N), ref(N);
    srand(1776 + N + E);
    for (int i = 0; i < N; ++i) in[i] = {(long double)rand() / RAND_MAX - 0.5L, (long double)rand() / RAND_MAX - 0.5L};
    for (int t = 0; t < C::T; ++t) {
        const int ti = io_lane<C, PERMUTE>(t);   // thread t owns X[io_lane(t) + T*e] before the first and after the last pass
        for (int e = 0; e < E; ++e) regs[(size_t)t * E + e] = {(R)in[ti + C::T * e].real(), (R)in[ti + C::T * e].imag()};
        auto& w = *reinterpret_cast<cx<R>(*)[NTWA]>(&tw[(size_t)t * NTWA]);
        load_twiddles<C, R, 1, TW_REG, PERMUTE>(w, t, table.data());
    }
    std::vector<cx<R>> twl(C::NTWLDS);
    for (int t = 0; t < C::T; ++t) fill_lds_twiddles<C, R>(twl.data(), t, table.data());
    run_passes<C, R, DIR, TWREG, PADSHIFT, 0, PERMUTE>(regs, tw, TWREG == TW_LDS ? twl : table, lds);
    // reference DFT (O(N^2), long double) on the rounded inputs
    long double maxerr = 0, norm = 0;
    std::vector<std::complex<long double>> root(N);
    for (int k = 0; k < N; ++k) {
        const long double a = (DIR < 0 ? -2.0L : 2.0L) * 3.141592653589793238462643383279502884L * k / N;
        root[k] = {cosl(a), sinl(a)};
    }
    for (int k = 0; k < N; ++k) {
        std::complex<long double> acc = 0;
        for (int n = 0; n < N; ++n)
            acc += std::complex<long double>((R)in[n].real(), (R)in[n].imag()) * root[(size_t)(((long long)n * k) % N)];
        ref[k] = acc;
        norm += std::norm(acc);
    }
    long double err2 = 0;
    for (int t = 0; t < C::T; ++t)
        for (int e = 0; e < E; ++e) {
            const auto got = std::complex<long double>(regs[(size_t)t * E + e].x, regs[(size_t)t * E + e].y);
            err2 += std::norm(got - ref[out_lane<C, PERMUTE>(t) + C::T * e]);   // == io_lane except in the wave-private mode
        }
    (void)maxerr;
    return (double)sqrtl(err2 / norm);
}

template <int N, int E> static int check_all() {
    int bad = 0;
    const double e1 = std::max(check<N, E, float, -1, 1, 4>(), check<N, E, float, -1, 2, 4>());
    const double e2 = check<N, E, float, +1, 0, 5>();
    const double e3 = std::max(check<N, E, double, -1, 0, 31>(), check<N, E, double, +1, 2, 3>());
    double e4 = check<N, E, double, +1, 1, 4>();
    if constexpr (LanePerm<N, E>::any) {   // the lane-permuted schedule (register / global twiddles, the geometry's own padding)
        constexpr int PS = LanePerm<N, E>::padshift;
        e4 = std::max(e4, std::max(check<N, E, double, +1, 1, PS, true>(), check<N, E, double, -1, 0, PS, true>()));
        const double ep = std::max(check<N, E, float, -1, 1, PS, true>(), check<N, E, float, +1, 1, PS, true>());
        if (!(ep < 2e-6)) bad = 1;
        printf("N=%5d E=%2d lane-permuted schedule, pad shift %d: f32 %.2e\n", N, E, PS, ep);
    }
    if constexpr (wave_private_ok<Cfg<N, E>>()) {   // PERMUTE = 2: the last exchange stays inside each wavefront (pad shift 4: region = 4 x 4 x 272)
        const double ew = std::max(std::max(check<N, E, float, -1, 1, 4, 2>(), check<N, E, float, +1, 1, 4, 2>()), check<N, E, float, -1, 0, 4, 2>());
        const double ed = std::max(check<N, E, double, -1, 1, 4, 2>(), check<N, E, double, +1, 0, 4, 2>());
        if (!(ew < 2e-6) || !(ed < 1e-14)) bad = 1;
        printf("N=%5d E=%2d wave-private last exchange: f32 %.2e f64 %.2e\n", N, E, ew, ed);
    }
    printf("N=%5d E=%2d P=%d radices:", N, E, Cfg<N, E>::P);
    for (int p = 0; p < Cfg<N, E>::P; ++p) printf(" %d", Cfg<N, E>::radix(p));
    printf("  relerr f32 fwd %.2e inv %.2e  f64 fwd %.2e inv %.2e\n", e1, e2, e3, e4);
    if (!(e1 < 2e-6) || !(e2 < 2e-6) || !(e3 < 1e-14) || !(e4 < 1e-14)) bad = 1;
    return bad;
}

// mixed-radix transforms (runtime N, everything through LDS): gen_schedule + gen_pass thread by thread against the long-double DFT, and the
// multiply-shift division of gen_pass checked for every butterfly index
template <typename R> static int check_gen(int N, int T, double tol) {
    int radix[MDSP_GEN_MAXP], ns[MDSP_GEN_MAXP];
    const int P = gen_schedule(N, radix, ns);
    if (P == 0) { printf("gen N=%d: no schedule\n", N); return 1; }
    std::vector<cx<R>> roots(N), a(gen_lds_elems(N)), b(gen_lds_elems(N));
    for (int k = 0; k < N; ++k) {
        const long double ang = -2.0L * 3.141592653589793238462643383279502884L * k / N;
        roots[k] = {(R)cosl(ang), (R)sinl(ang)};
    }
    std::vector<std::complex<long double>> in(N);
    srand(99 + N);
    for (int i = 0; i < N; ++i) {
        in[i] = {(long double)(R)((double)rand() / RAND_MAX - 0.5), (long double)(R)((double)rand() / RAND_MAX - 0.5)};
        a[gen_pad(i)] = {(R)in[i].real(), (R)in[i].imag()};
    }
    cx<R>*src = a.data(), *dst = b.data();
    for (int p = 0; p < P; ++p) {
        const unsigned divm = (unsigned)(((1u << 24) + ns[p] - 1) / ns[p]);
        for (int j = 0; j < N / radix[p]; ++j)
            if (ns[p] > 1 && (int)(((unsigned long long)(unsigned)j * divm) >> 24) != j / ns[p]) { printf("gen N=%d pass %d: division by %d wrong at j=%d\n", N, p, ns[p], j); return 1; }
        for (int t = 0; t < T; ++t) gen_pass_dispatch(radix[p], src, dst, roots.data(), N, ns[p], divm, t, T);
        std::swap(src, dst);
    }
    long double err2 = 0, norm = 0;
    for (int k = 0; k < N; ++k) {
        std::complex<long double> acc = 0;
        for (int n = 0; n < N; ++n) {
            const long double ang = -2.0L * 3.141592653589793238462643383279502884L * (long double)(((long long)n * k) % N) / N;
            acc += in[n] * std::complex<long double>(cosl(ang), sinl(ang));
        }
        err2 += std::norm(std::complex<long double>(src[gen_pad(k)].x, src[gen_pad(k)].y) - acc);
        norm += std::norm(acc);
    }
    const double e = (double)sqrtl(err2 / norm);
    printf("gen N=%5d T=%3d passes %d radices:", N, T, P);
    for (int p = 0; p < P; ++p) printf(" %d", radix[p]);
    printf("  relerr %.2e\n", e);
    return e < tol ? 0 : 1;
}

// Windowed first pass (pass0_windowed / bfly16_win: the window folded into the butterfly's first stage) followed by the ordinary passes, with the
// sample pairing of welch_half3_kernel: q[e] = (frame A, frame B)[t + T e] for e < 8, f[e] = the same for e + 8, wp[e] = {w[e], w[e + 8]};
// `swap` exchanges the two components (z = b + i a).  Checked against the long-double DFT of w (a + i b).
template <int N, typename R, int PADSHIFT> static double check_windowed(bool swap) {
    using C = Cfg<N, 16>;
    constexpr int E = 16, T = C::T, NTWA = C::NTW > 0 ? C::NTW : 1;
    std::vector<cx<R>> table(N), regs((size_t)T * E), tw((size_t)T * NTWA), lds(lds_elems<N, PADSHIFT>());
    for (int k = 0; k < N; ++k) {
        const long double a = -2.0L * 3.141592653589793238462643383279502884L * k / N;
        table[k] = {(R)cosl(a), (R)sinl(a)};
    }
    std::vector<R> fa(N), fb(N), w(N);
    srand(4242 + N);
    for (int i = 0; i < N; ++i) {
        fa[i] = (R)((double)rand() / RAND_MAX - 0.5);
        fb[i] = (R)((double)rand() / RAND_MAX - 0.5);
        w[i] = (R)(0.5 - 0.5 * cos(2.0 * 3.14159265358979323846 * i / (N - 1)));
    }
    for (int t = 0; t < T; ++t) {
        cx<R> q[8], f[8], wp[8];
        for (int e = 0; e < 8; ++e) {
            const int i0 = t + T * e, i1 = t + T * (e + 8);
            q[e] = swap ? cx<R>{fb[i0], fa[i0]} : cx<R>{fa[i0], fb[i0]};
            f[e] = swap ? cx<R>{fb[i1], fa[i1]} : cx<R>{fa[i1], fb[i1]};
            wp[e] = {w[i0], w[i1]};
        }
        pass0_windowed<C, PADSHIFT>(q, f, wp, t, lds.data());
        auto& ww = *reinterpret_cast<cx<R>(*)[NTWA]>(&tw[(size_t)t * NTWA]);
        load_twiddles<C, R, 1, TW_REG, false>(ww, t, table.data());
    }
    for (int t = 0; t < T; ++t) {
        auto& x = *reinterpret_cast<cx<R>(*)[E]>(&regs[(size_t)t * E]);
        pass_reload<C, PADSHIFT, 1, false>(x, t, lds.data());
    }
    run_passes<C, R, -1, TW_REG, PADSHIFT, 1, false>(regs, tw, table, lds);
    long double err2 = 0, norm = 0;
    for (int k = 0; k < N; ++k) {
        std::complex<long double> acc = 0;
        for (int n = 0; n < N; ++n) {
            const long double ang = -2.0L * 3.141592653589793238462643383279502884L * (long double)(((long long)n * k) % N) / N;
            const std::complex<long double> z = swap ? std::complex<long double>(fb[n], fa[n]) : std::complex<long double>(fa[n], fb[n]);
            acc += (long double)w[n] * z * std::complex<long double>(cosl(ang), sinl(ang));
        }
        const cx<R> g = regs[(size_t)(k % T) * E + k / T];
        err2 += std::norm(std::complex<long double>(g.x, g.y) - acc);
        norm += std::norm(acc);
    }
    return (double)sqrtl(err2 / norm);
}

// One wavefront per 4096-point transform (fft_w64.h): bfly64 alone against the 64-point DFT, then a whole Welch unit -- passA_welch lane by lane,
// the 64 x 64 transposition (the half exchange v_permlane32_swap performs + two rounds through the padded buffer, with the kernel's index
// functions), the W4096^{lane * register} twiddles on the reader side and bfly64 -- against the long-double DFT of w (a + i b).
template <typename R> static double check_bfly64() {
    cx<R> v[64];
    std::complex<long double> in[64];
    srand(640);
    for (int i = 0; i < 64; ++i) {
        v[i] = {(R)((double)rand() / RAND_MAX - 0.5), (R)((double)rand() / RAND_MAX - 0.5)};
        in[i] = {v[i].x, v[i].y};
    }
    bfly64<-1>(v);
    long double err2 = 0, norm = 0;
    for (int k = 0; k < 64; ++k) {
        std::complex<long double> acc = 0;
        for (int n = 0; n < 64; ++n) {
            const long double ang = -2.0L * 3.141592653589793238462643383279502884L * (long double)((n * k) % 64) / 64;
            acc += in[n] * std::complex<long double>(cosl(ang), sinl(ang));
        }
        err2 += std::norm(std::complex<long double>(v[slot64(k)].x, v[slot64(k)].y) - acc);
        norm += std::norm(acc);
    }
    return (double)sqrtl(err2 / norm);
}
template <typename R> static double check_wave64(bool frame_b) {
    constexpr int N = 4096, HALF = N / 2;
    std::vector<cx<R>> table(N);
    for (int k = 0; k < N; ++k) {
        const long double a = -2.0L * 3.141592653589793238462643383279502884L * k / N;
        table[k] = {(R)cosl(a), (R)sinl(a)};
    }
    std::vector<R> h0(HALF), h1(HALF), h2(HALF), w(N);
    srand(6464);
    for (int i = 0; i < HALF; ++i) {
        h0[i] = (R)((double)rand() / RAND_MAX - 0.5);
        h1[i] = (R)((double)rand() / RAND_MAX - 0.5);
        h2[i] = (R)((double)rand() / RAND_MAX - 0.5);
    }
    for (int i = 0; i < N; ++i) w[i] = (R)(0.5 - 0.5 * cos(2.0 * 3.14159265358979323846 * i / (N - 1)));
    std::vector<cx<R>> reg((size_t)64 * 64), lds(XP64_ELEMS);   // reg[lane * 64 + logical register]
    for (int t = 0; t < 64; ++t) {                             // pass A
        cx<R> xp[32], wp[32], v[64];
        R hh[32];
        for (int e = 0; e < 32; ++e) {
            const int p = t + 64 * e;
            xp[e] = {h0[p], h2[p]};
            hh[e] = h1[p];
            wp[e] = {w[p], w[p + HALF]};
        }
        if (frame_b) passA_welch<true>(xp, hh, wp, v);
        else passA_welch<false>(xp, hh, wp, v);
        for (int ke = 0; ke < 64; ++ke) reg[(size_t)t * 64 + ke] = v[slot64(ke)];
    }
    for (int r = 0; r < 32; ++r)                               // step 1: lanes 32..63 of register r <-> lanes 0..31 of register r + 32
        for (int l = 0; l < 32; ++l) std::swap(reg[(size_t)(l + 32) * 64 + r], reg[(size_t)l * 64 + r + 32]);
    for (int round = 0; round < 2; ++round) {                  // step 2: two rounds of 32 registers through the padded buffer
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 32; ++r) lds[xp64_write_index(l, r)] = reg[(size_t)l * 64 + 32 * round + r];
        for (int l = 0; l < 64; ++l)
            for (int T = 0; T < 32; ++T) reg[(size_t)l * 64 + 32 * round + T] = lds[xp64_read_index(l, T)];
    }
    long double err2 = 0, norm = 0;
    std::vector<std::complex<long double>> got(N);
    for (int ke = 0; ke < 64; ++ke) {                          // reader side: twiddle W^{lane * register}, pass B
        cx<R> v[64];
        for (int t = 0; t < 64; ++t) v[t] = t == 0 ? reg[(size_t)ke * 64] : twmul<-1>(reg[(size_t)ke * 64 + t], table[(ke * t) & (N - 1)]);
        bfly64<-1>(v);
        for (int kt = 0; kt < 64; ++kt) got[ke + 64 * kt] = {v[slot64(kt)].x, v[slot64(kt)].y};
    }
    for (int k = 0; k < N; ++k) {
        std::complex<long double> acc = 0;
        for (int n = 0; n < N; ++n) {
            const long double ang = -2.0L * 3.141592653589793238462643383279502884L * (long double)(((long long)n * k) % N) / N;
            const long double a = n < HALF ? h0[n] : h1[n - HALF], b = frame_b ? (n < HALF ? h1[n] : h2[n - HALF]) : 0.0L;
            acc += (long double)w[n] * std::complex<long double>(a, b) * std::complex<long double>(cosl(ang), sinl(ang));
        }
        err2 += std::norm(got[k] - acc);
        norm += std::norm(acc);
    }
    return (double)sqrtl(err2 / norm);
}

// composite butterflies (radix 6 ... 25 inside one thread's registers, fft_lds.h bfly_comp) against the long-double DFT
template <int RDX, typename R> double check_comp() {
    cx<R> v[RDX];
    std::complex<long double> x[RDX];
    srand(17 + RDX);
    for (int i = 0; i < RDX; ++i) {
        const R a = (R)(rand() / (double)RAND_MAX - 0.5), b = (R)(rand() / (double)RAND_MAX - 0.5);
        v[i] = {a, b};
        x[i] = {(long double)a, (long double)b};
    }
    gen_bfly<RDX>(v);
    long double err2 = 0, norm = 0;
    for (int k = 0; k < RDX; ++k) {
        std::complex<long double> acc = 0;
        for (int n = 0; n < RDX; ++n) {
            const long double ang = -2.0L * 3.141592653589793238462643383279502884L * (long double)((n * k) % RDX) / RDX;
            acc += x[n] * std::complex<long double>(cosl(ang), sinl(ang));
        }
        err2 += std::norm(std::complex<long double>(v[k].x, v[k].y) - acc);
        norm += std::norm(acc);
    }
    return (double)sqrtl(err2 / norm);
}
template <int... RS> int check_comps() {
    int bad = 0;
    const double e32[] = {check_comp<RS, float>()...}, e64[] = {check_comp<RS, double>()...};
    const int rs[] = {RS...};
    printf("composite butterflies:");
    for (size_t i = 0; i < sizeof...(RS); ++i) {
        printf(" %d: %.1e / %.1e", rs[i], e32[i], e64[i]);
        if (!(e32[i] < 4e-7) || !(e64[i] < 1e-15)) bad = 1;
    }
    printf("\n");
    return bad;
}

int main() {
    int bad = 0;
    bad |= check_comps<6, 10, 12, 15, 20, 24, 25>();
    {
        const double b32 = check_bfly64<float>(), b64 = check_bfly64<double>();
        const double e32 = std::max(check_wave64<float>(true), check_wave64<float>(false)), e64 = std::max(check_wave64<double>(true), check_wave64<double>(false));
        printf("one wavefront per transform (64 x 64): bfly64 f32 %.2e f64 %.2e, Welch unit f32 %.2e f64 %.2e\n", b32, b64, e32, e64);
        if (!(b32 < 1e-6) || !(b64 < 1e-15) || !(e32 < 2e-6) || !(e64 < 1e-14)) bad = 1;
    }
    {
        const double e1 = std::max(check_windowed<4096, float, 5>(false), check_windowed<4096, float, 5>(true));
        const double e2 = std::max(check_windowed<2048, float, 5>(false), check_windowed<1024, float, 4>(true));
        const double e3 = check_windowed<4096, double, 5>(false);
        printf("windowed first pass (bfly16_win): f32 4096 %.2e, 2048 / 1024 %.2e, f64 %.2e\n", e1, e2, e3);
        if (!(e1 < 2e-6) || !(e2 < 2e-6) || !(e3 < 1e-14)) bad = 1;
    }
    for (int N : {18, 30, 100, 120, 1000, 1536, 3000, 768, 1500, 2000, 2401, 625, 6561, 7000, 8000, 6, 7, 5, 3, 2, 64, 4096, 7680})
        bad |= check_gen<float>(N, N > 2000 ? 256 : 64, 3e-6) | check_gen<double>(N, 128, 2e-14);
    bad |= check_all<16, 16>();
    bad |= check_all<64, 8>();
    bad |= check_all<128, 16>();
    bad |= check_all<256, 16>();
    bad |= check_all<512, 8>();
    bad |= check_all<512, 16>();
    bad |= check_all<1024, 16>();
    bad |= check_all<2048, 16>();
    bad |= check_all<2048, 8>();
    bad |= check_all<4096, 16>();
    bad |= check_all<8192, 16>();
    bad |= check_all<4096, 8>();
    bad |= check_all<1024, 8>();
    bad |= check_all<256, 4>();
    bad |= check_all<8192, 8>();
    printf(bad ? "FAIL\n" : "OK\n");
    return bad;
}
