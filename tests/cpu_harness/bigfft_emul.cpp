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
    std::printf(bad ? "%d FAILED\n" : "OK\n", bad);
    return bad ? 1 : 0;
}
