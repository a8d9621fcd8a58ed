// Host emulation of the large-transform passes (dsp.jl_amd/csrc/bigfft_pass.h + bigfft_plan.h): the SAME tile / sub-pass / store code the
// kernels of bigfft.hip run, executed thread by thread, against a Float64 mixed-radix DFT (hostfft.h).
//   g++ -O2 -std=c++17 tests/cpu_harness/bigfft_emul.cpp -o bigfft_emul && ./bigfft_emul
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../dsp.jl_amd/csrc/bigfft_plan.h"

using namespace mdsp;
using namespace mdsp::big;
using fft::cx;

template <typename R> double run(int64_t N, int rmax, bool verbose) {
    HostPlan<R> hp;
    if (!make_plan<R>(N, hp, rmax)) {
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
            std::printf(" %d[", hp.pass[p].Rp);
            for (int s = 0; s < hp.pass[p].nsub; ++s) std::printf("%s%d", s ? "," : "", hp.pass[p].radix[s]);
            std::printf("]");
        }
    }
    for (int p = 0; p < hp.P; ++p) {
        Pass& q = hp.pass[p];
        q.roots = hp.roots[p].data();
        q.T0 = hp.T0[p].data();
        q.T1 = hp.T1[p].data();
        std::vector<cx<R>> A((size_t)q.Rp * Bp), Bf((size_t)q.Rp * Bp);
        for (int64_t tile = 0; tile < q.ntiles; ++tile) {
            const Tile t = tile_of<R>(q, tile);
            for (int tid = 0; tid < TPB; ++tid) phase_load<R>(q, t, tid, A.data(), [&](int64_t pos) { return buf[(size_t)pos]; });
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
            for (int tid = 0; tid < TPB; ++tid) {   // both loop forms of the store phase (the Welch kernel uses the fully unrolled one)
                if (tile & 1) phase_store<R, true>(q, t, tid, src, sink);
                else phase_store<R, false>(q, t, tid, src, sink);
            }
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
    // three passes
    for (int64_t N : {524288, 1048576, 1250000, 2097152}) chk(run<float>(N, RMAX, true), 3e-6);
    chk(run<double>(1048576, RMAX, true), 3e-15);
    // three and four passes of short sub-transforms (the digit walk of the last pass): factors capped at 16 / 12
    for (int64_t N : {4096, 65536, 20736, 50625}) chk(run<float>(N, N == 20736 ? 12 : (N == 50625 ? 15 : 16), true), 2e-6);
    chk(run<double>(65536, 16, true), 2e-15);
    std::printf(bad ? "%d FAILED\n" : "OK\n", bad);
    return bad ? 1 : 0;
}
