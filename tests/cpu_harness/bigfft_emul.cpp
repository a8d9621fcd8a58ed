// Host emulation of the large-transform passes (dsp.jl_amd/csrc/bigfft_pass.h + bigfft_plan.h): the SAME tile / sub-pass / store code the
// kernels of bigfft.hip run, executed thread by thread, against a Float64 mixed-radix DFT (hostfft.h).
//   g++ -O2 -std=c++17 tests/cpu_harness/bigfft_emul.cpp -o bigfft_emul && ./bigfft_emul
#include <cstdio>
#include <cstdlib>
#include <array>
#include <random>

#include "../../dsp.jl_amd/csrc/bigfft_plan.h"

using namespace mdsp;
using namespace mdsp::big;
using fft::cx;


// one tile of a two-stage pass, thread by thread (the barriers of the kernel are the boundaries between the loops)
template <typename R, int RA, int TJ>
void fast_tile(const Pass& q, int64_t tile, std::vector<cx<R>>& buf, std::vector<cx<R>>& out, int64_t N) {
    using G = FastGeo<RA, TJ>;
    const Tile t = tile_of_b<R>(q, tile, G::B);
    std::vector<cx<R>> lds((size_t)G::Rp * G::Bp), stage((size_t)G::Rp * G::Bp), twc((size_t)G::Rp);
    std::vector<std::array<cx<R>, RA>> x(TPB);
    std::vector<std::array<cx<R>, G::NO>> y(TPB), rt(TPB), twb(TPB);
#define ARR(T, n, a) reinterpret_cast<T(&)[n]>(*(a).data())
    for (int tid = 0; tid < TPB; ++tid) {
        fast_roots<R, RA, TJ>(q, tid, ARR(cx<R>, G::NO, rt[tid]));
        if (!q.last) fast_twb<R, RA, TJ>(q, tid, ARR(cx<R>, G::NO, twb[tid]));
        fast_load<R, RA, TJ>(q, t, tid, ARR(cx<R>, RA, x[tid]), [&](int64_t pos) { return buf[(size_t)pos]; });
    }
    if (q.last) {
        for (int tid = 0; tid < TPB; ++tid) fast_stage_put<R, RA, TJ>(tid, ARR(const cx<R>, RA, x[tid]), stage.data());
        for (int tid = 0; tid < TPB; ++tid) fast_stage_get<R, RA, TJ>(tid, ARR(cx<R>, RA, x[tid]), stage.data());
    } else
        for (int tid = 0; tid < TPB; ++tid) fill_twc<R>(q, t, tid, twc.data());
    for (int tid = 0; tid < TPB; ++tid) fast_stage1<R, RA, TJ>(tid, ARR(cx<R>, RA, x[tid]), lds.data());
    for (int tid = 0; tid < TPB; ++tid) fast_stage2<R, RA, TJ>(tid, lds.data(), ARR(const cx<R>, G::NO, rt[tid]), ARR(cx<R>, G::NO, y[tid]));
    for (int tid = 0; tid < TPB; ++tid)
        fast_store<R, RA, TJ>(q, t, tid, ARR(const cx<R>, G::NO, y[tid]), twc.data(), ARR(const cx<R>, G::NO, twb[tid]), [&](int, int64_t pos, cx<R> z) {
            if (pos < 0 || pos >= N) {
                std::printf(" index %lld out of range (two-stage form)\n", (long long)pos);
                std::exit(1);
            }
            (q.last ? out : buf)[(size_t)pos] = z;
        });
#undef ARR
}

template <typename R> double run(int64_t N, int rmax, bool verbose, int fast = 1) {
    HostPlan<R> hp;
    if (!make_plan<R>(N, hp, rmax, fast)) {
        std::printf("N=%lld: no plan\n", (long long)N);
        return 1e9;
    }
    constexpr int B = cols<R>(), Bp = B + 1;
    std::mt19937_64 rng(1776 + (unsigned)N);
    std::normal_distribution<double> nd;
    std::vector<zd> xin((size_t)N);
    std::vector<cx<R>> buf((size_t)N), out((size_t)N);
    for (int64_t i = 0; i < N; ++i) {
        buf[(size_t)i] = {(R)nd(rng), (R)nd(rng)};
        xin[(size_t)i] = zd((double)buf[(size_t)i].x, (double)buf[(size_t)i].y);
    }
    if (verbose) {
        std::printf("N=%9lld %s P=%d :", (long long)N, sizeof(R) == 4 ? "f32" : "f64", hp.P);
        for (int p = 0; p < hp.P; ++p) {
            if (hp.pass[p].fTJ) std::printf(" %d[%dx%d two-stage, %d columns]", hp.pass[p].Rp, hp.pass[p].fRA, hp.pass[p].fTJ, hp.pass[p].B);
            else {
                std::printf(" %d[", hp.pass[p].Rp);
                for (int s = 0; s < hp.pass[p].nsub; ++s) std::printf("%s%d", s ? "," : "", hp.pass[p].radix[s]);
                std::printf("]");
            }
        }
    }
    for (int p = 0; p < hp.P; ++p) {
        Pass& q = hp.pass[p];
        q.roots = hp.roots[p].data();
        q.T0 = hp.T0[p].data();
        q.T1 = hp.T1[p].data();
        if (q.fTJ) {
            for (int64_t tile = 0; tile < q.ntiles; ++tile) {
                if (q.fRA == 16 && q.fTJ == 4) fast_tile<R, 16, 4>(q, tile, buf, out, N);
                else if (q.fRA == 16 && q.fTJ == 8) fast_tile<R, 16, 8>(q, tile, buf, out, N);
                else if (q.fRA == 16) fast_tile<R, 16, 16>(q, tile, buf, out, N);
                else if (q.fRA == 8 && q.fTJ == 16) fast_tile<R, 8, 16>(q, tile, buf, out, N);
                else if (q.fRA == 8) fast_tile<R, 8, 8>(q, tile, buf, out, N);
                else fast_tile<R, 4, 8>(q, tile, buf, out, N);
            }
            continue;
        }
        std::vector<cx<R>> A((size_t)q.Rp * Bp), Bf((size_t)q.Rp * Bp), twc((size_t)q.Rp);
        constexpr int E = elems<R>();
        std::vector<std::array<cx<R>, E>> twb(TPB), regs(TPB);
        if (!q.last)
            for (int tid = 0; tid < TPB; ++tid) load_twb<R, E>(q, tid, reinterpret_cast<cx<R>(&)[E]>(*twb[tid].data()));
        for (int64_t tile = 0; tile < q.ntiles; ++tile) {
            const Tile t = tile_of<R>(q, tile);
            for (int tid = 0; tid < TPB; ++tid)
                load_regs<R, E>(q, t, tid, reinterpret_cast<cx<R>(&)[E]>(*regs[tid].data()), [&](int64_t pos) { return buf[(size_t)pos]; });
            for (int tid = 0; tid < TPB; ++tid) regs_to_lds<R, E>(q, tid, reinterpret_cast<const cx<R>(&)[E]>(*regs[tid].data()), A.data());
            if (!q.last)
                for (int tid = 0; tid < TPB; ++tid) fill_twc<R>(q, t, tid, twc.data());
            cx<R>*src = A.data(), *dst = Bf.data();
            for (int sp = 0; sp < q.nsub; ++sp) {
                for (int tid = 0; tid < TPB; ++tid) phase_sub<R>(q, sp, tid, src, dst, hp.roots[p].data());
                std::swap(src, dst);
            }
            auto sink = [&](int, int64_t pos, cx<R> z) {
                if (pos < 0 || pos >= N) {
                    std::printf(" index %lld out of range in pass %d\n", (long long)pos, p);
                    std::exit(1);
                }
                (q.last ? out : buf)[(size_t)pos] = z;
            };
            for (int tid = 0; tid < TPB; ++tid)
                phase_store<R, E>(q, t, tid, src, twc.data(), reinterpret_cast<const cx<R>(&)[E]>(*twb[tid].data()), sink);
        }
    }
    const std::vector<zd> ref = host_fft(xin, -1);
    double num = 0, den = 0;
    for (int64_t i = 0; i < N; ++i) {
        const zd d = zd((double)out[(size_t)i].x, (double)out[(size_t)i].y) - ref[(size_t)i];
        num += std::norm(d);
        den += std::norm(ref[(size_t)i]);
    }
    const double err = std::sqrt(num / den);
    if (verbose) std::printf("  rel err %.3g\n", err);
    return err;
}

// The rows form of the long-filter convolution (bigfft.hip run_ols_rows, ols.hip ROWS), on the host: column pass (pass 0 of a two-factor plan, through the
// tile code above), per row the S-point transform x the row of H' x the inverse transform x the inverse inter-pass twiddle (a Float64 DFT stands in for
// the single-workgroup kernel), then the column pass back on the conjugate with tables of ones -- against the circular convolution computed directly.
template <typename R, int RA, int TJ> double run_rows(int R0, int S, bool verbose) {
    const int64_t N = (int64_t)R0 * S;
    HostPlan<R> hp;
    const int Rf[2] = {R0, S};
    if (!make_plan_factors<R>(N, hp, Rf, 2, 1, true) || hp.pass[0].fRA != RA || hp.pass[0].fTJ != TJ) {
        std::printf("rows %d x %d: no column pass of that geometry\n", R0, S);
        return 1e9;
    }
    Pass fwd = hp.pass[0];
    fwd.roots = hp.roots[0].data();
    fwd.T0 = hp.T0[0].data();
    fwd.T1 = hp.T1[0].data();
    std::vector<cx<R>> one0(hp.T0[0].size(), cx<R>{(R)1, (R)0}), one1(hp.T1[0].size(), cx<R>{(R)1, (R)0});
    Pass inv = fwd;
    inv.T0 = one0.data();
    inv.T1 = one1.data();
    std::mt19937_64 rng(77 + (unsigned)N);
    std::normal_distribution<double> nd;
    const int nb = S / 3 + 5;                      // a filter shorter than the block
    std::vector<zd> x((size_t)N), h((size_t)N, zd(0, 0));
    std::vector<cx<R>> buf((size_t)N), dummy((size_t)N);
    for (int64_t i = 0; i < N; ++i) {
        x[(size_t)i] = zd(nd(rng), nd(rng));
        buf[(size_t)i] = {(R)x[(size_t)i].real(), (R)x[(size_t)i].imag()};
        x[(size_t)i] = zd((double)buf[(size_t)i].x, (double)buf[(size_t)i].y);
    }
    for (int i = 0; i < nb; ++i) h[(size_t)i] = zd(nd(rng), 0) / (double)N;
    const std::vector<zd> H = host_fft(h, -1);
    for (int64_t tile = 0; tile < fwd.ntiles; ++tile) fast_tile<R, RA, TJ>(fwd, tile, buf, dummy, N);          // column pass, twiddled
    for (int k1 = 0; k1 < R0; ++k1) {                                                                           // the row kernel
        std::vector<zd> row((size_t)S);
        for (int c = 0; c < S; ++c) row[(size_t)c] = zd((double)buf[(size_t)(k1 * (int64_t)S + c)].x, (double)buf[(size_t)(k1 * (int64_t)S + c)].y);
        std::vector<zd> Z = host_fft(row, -1);
        for (int k2 = 0; k2 < S; ++k2) Z[(size_t)k2] *= H[(size_t)(k1 + (int64_t)R0 * k2)];                     // H'[k1 S + k2] = H[k1 + R0 k2]
        const std::vector<zd> z = host_fft(Z, +1);
        for (int c = 0; c < S; ++c) {
            const zd w = std::conj(unit_root((int64_t)k1 * c, N, -1));
            const zd v = z[(size_t)c] * w;
            buf[(size_t)(k1 * (int64_t)S + c)] = {(R)v.real(), -(R)v.imag()};                                    // the column pass back starts from the conjugate
        }
    }
    for (int64_t tile = 0; tile < inv.ntiles; ++tile) fast_tile<R, RA, TJ>(inv, tile, buf, dummy, N);          // ... and its results are conjugated on the way out
    // reference: circular convolution through Float64 transforms of the whole block
    std::vector<zd> X = host_fft(x, -1);
    for (int64_t k = 0; k < N; ++k) X[(size_t)k] *= H[(size_t)k];
    const std::vector<zd> y = host_fft(X, +1);
    double num = 0, den = 0;
    for (int64_t i = 0; i < N; ++i) {
        const zd d = zd((double)buf[(size_t)i].x, -(double)buf[(size_t)i].y) - y[(size_t)i];
        num += std::norm(d);
        den += std::norm(y[(size_t)i]);
    }
    const double err = std::sqrt(num / den);
    if (verbose) std::printf("rows form %3d x %4d (%dx%d column pass): rel err %.3g\n", R0, S, RA, TJ, err);
    return err;
}

int main() {
    int bad = 0;
    auto chk = [&](double e, double tol) {
        if (!(e < tol)) {
            ++bad;
            std::printf("FAIL (%.3g >= %.3g)\n", e, tol);
        }
    };
    // two passes, powers of two and the nextfastfft neighbours of the defaults
    for (int64_t N : {16384, 32768, 65536, 100000, 125000, 262144, 49152, 30375, 16807 * 2, 9000, 12500, 8232})
        chk(run<float>(N, RMAX, true), 2e-6);
    for (int64_t N : {5000, 8192, 10000, 65536, 125000, 30375, 9604})
        chk(run<double>(N, RMAX, true), 2e-15);
    // the same powers of two through the generic phases only (what the two-stage form replaces)
    for (int64_t N : {16384, 65536, 262144}) chk(run<float>(N, RMAX, true, 0), 2e-6);
    // the other two-stage geometries of 128 (8 x 16) and 64 (16 x 4)
    for (int64_t N : {16384, 32768, 262144}) chk(run<float>(N, RMAX, true, 2), 2e-6);
    for (int64_t N : {16384, 262144}) chk(run<float>(N, RMAX, true, 3), 2e-6);
    chk(run<double>(65536, RMAX, true, 0), 2e-15);
    // sub-transforms capped at 256 / 128 / 64: the other two-stage geometries
    for (int64_t N : {131072, 2097152}) chk(run<float>(N, 256, true), 3e-6);
    chk(run<float>(16384, 128, true), 2e-6);
    chk(run<float>(65536, 64, true), 2e-6);
    chk(run<double>(32768, 64, true), 2e-15);
    // three passes
    for (int64_t N : {524288, 1048576, 1250000, 2097152}) chk(run<float>(N, RMAX, true), 3e-6);
    chk(run<double>(1048576, RMAX, true), 3e-15);
    // three and four passes of short sub-transforms (the digit walk of the last pass): factors capped at 16 / 12
    for (int64_t N : {4096, 65536, 20736, 50625}) chk(run<float>(N, N == 20736 ? 12 : (N == 50625 ? 15 : 16), true), 2e-6);
    chk(run<double>(65536, 16, true), 2e-15);
    // the rows form of the long-filter convolution: 64 / 128 / 256 rows
    chk(run_rows<float, 8, 8>(64, 256, true), 2e-6);
    chk(run_rows<float, 16, 8>(128, 128, true), 2e-6);
    chk(run_rows<float, 16, 16>(256, 64, true), 2e-6);
    chk(run_rows<double, 8, 8>(64, 128, true), 2e-15);
    std::printf(bad ? "%d FAILED\n" : "OK\n", bad);
    return bad ? 1 : 0;
}
