// Host side of the run-time-schedule spectral kernel (gx_kernels.h; included by spectral.hip inside its anonymous namespace): which sizes it takes, the
// split nfft = R0 x S, root tables, launch geometry.  The kernels themselves live in their own translation units (gx_inst_*.hip) behind mdsp::gx_run.
#pragma once

#include "gx_kernels.h"

// ---- host side ------------------------------------------------------------------------------------------------------------------------
template <typename R> __global__ __launch_bounds__(256) void gx_window_kernel(const double* __restrict__ win, R* __restrict__ out, int n, int nfft) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nfft) out[i] = i < n ? (win ? (R)win[i] : (R)1) : (R)0;
}

constexpr int GX_LDS_BYTES = 160 * 1024;
constexpr int GX_R0_MAX = 8;    // column factor at most (R0 loads per point, R0 - 1 of them from the L2: 125000 = 20 x 6250 measured 3.0 ms against the multi-pass engine's 1.2)
constexpr int GX_R0_LAST = 32;  // ... and for the few sizes below 50000 points that no R0 <= 8 splits (19683 = 9 x 2187, 46875 = 15 x 3125): still ahead of the rocFFT pipeline
inline int gx_emax(int) { return 16; }
inline int gx_tmax(int) { return 512; }         // GxGeo<R>::LB
inline int gx_cu_threads(int dtype) { return dtype_is_double(dtype) ? 512 : 1024; }  // 256 / 128 registers per thread (GxGeo<R>::MINW)
// table entries behind the buffer for a split nfft = R0 S
inline int gx_table_entries(int64_t nfft, int64_t S, int R0) {
    return mdsp::gx::TWS + mdsp::gx::tw_hi_entries(S) + (R0 > 1 ? mdsp::gx::TWS + mdsp::gx::tw_hi_entries(nfft) + R0 : 0);
}
// the split the kernel runs nfft with: R0 = 1 -- the whole transform in one workgroup -- where it has a schedule; else the R0 <= GX_R0_MAX that costs least: R0
// workgroup-units per frame, each the schedule of S = nfft / R0 points plus the column step (R0 loads and a complex multiply-add per point, a twiddle)
inline bool gx_choose(int dtype, int64_t nfft, bool welch, int* R0_out, mdsp::gx::Sched* sc_out) {
    const int esz = dtype_is_double(dtype) ? 16 : 8, emax = gx_emax(dtype);
    const int tmax = gx_tmax(dtype);
    if (nfft < 64 || nfft > (int64_t)GX_R0_MAX * tmax * emax || !mdsp::gx::smooth7(nfft)) return false;
    bool found = false;
    double best = 0;
    for (int R0 = 1; R0 <= (nfft < 50000 ? GX_R0_LAST : GX_R0_MAX); ++R0) {
        if (R0 > GX_R0_MAX && found) break;
        if (nfft % R0 != 0) continue;
        const int64_t S = nfft / R0;
        if (S > (int64_t)tmax * emax || S < 16) continue;
        const mdsp::gx::Sched sc = mdsp::gx::plan((int)S, emax, tmax, gx_cu_threads(dtype), GX_LDS_BYTES, esz, esz * gx_table_entries(nfft, S, R0),
                                                       welch ? esz / 2 : 0);   // Welch: the sums live in LDS, one real per bin
        if (sc.P < 2) continue;
        const double pts = (double)((S + sc.T - 1) / sc.T);
        // (the column step's weight: 12500 = 2 x 6250 -- six passes -- measured 0.49 TB/s, 5 x 2500 -- four -- 0.39: profiles/r06_gx_sessions.json)
        const double score = (double)R0 * (sc.cost + (R0 > 1 ? pts * (16.0 * R0 + 9.0) : 0.0)) / sc.wgs;
        if (!found || score < best) {
            found = true;
            best = score;
            if (R0_out) *R0_out = R0;
            if (sc_out) *sc_out = sc;
        }
        if (R0 == 1) break;   // one workgroup per transform wherever that is possible
    }
    return found;
}
inline bool gx_size_ok(int dtype, int64_t nfft) {
    static std::mutex mu;
    static std::vector<std::pair<int64_t, bool>> memo;   // (nfft << 1 | double) -> plannable
    const int64_t key = (nfft << 1) | (dtype_is_double(dtype) ? 1 : 0);
    {
        std::lock_guard<std::mutex> lk(mu);
        for (auto& e : memo)
            if (e.first == key) return e.second;
    }
    const bool ok = gx_choose(dtype, nfft, true, nullptr, nullptr);
    std::lock_guard<std::mutex> lk(mu);
    memo.emplace_back(key, ok);
    return ok;
}

// the column factor of the split (1: one workgroup per transform; 0: no schedule)
inline int gx_split_r0(int dtype, int64_t nfft) {
    int R0 = 0;
    return gx_choose(dtype, nfft, true, &R0, nullptr) ? R0 : 0;
}

template <typename R> int gx_prepare(GxPlan& gp, int dtype, int64_t nfft, bool welch) {
    if (gp.ready) return MDSP_OK;
    if (!gx_choose(dtype, nfft, welch, &gp.R0, &gp.sc)) MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "nfft=%lld has no single-workgroup schedule", (long long)nfft);
    const int64_t S = gp.sc.N;
    gp.nhs = mdsp::gx::tw_hi_entries(S);
    gp.nhn = gp.R0 > 1 ? mdsp::gx::tw_hi_entries(nfft) : 0;
    const int ntab = gx_table_entries(nfft, S, gp.R0);
    std::vector<cx<R>> tab((size_t)ntab);
    size_t o = 0;
    auto root = [](int64_t k, int64_t n) { return unit_root(k % n, n, -1); };
    for (int i = 0; i < mdsp::gx::TWS; ++i) {   // W_S^i - 1: the real part as -2 sin^2 (no cancellation)
        const long double ang = 3.141592653589793238462643383279502884L * (long double)(i % S) / (long double)S;
        const long double sh = sinl(ang);
        tab[o++] = {(R)(double)(-2.0L * sh * sh), (R)root(i, S).imag()};
    }
    for (int i = 0; i < gp.nhs; ++i) {
        const zd w = root((int64_t)i * mdsp::gx::TWS, S);
        tab[o++] = {(R)w.real(), (R)w.imag()};
    }
    if (gp.R0 > 1) {
        for (int i = 0; i < mdsp::gx::TWS; ++i) {
            const long double ang = 3.141592653589793238462643383279502884L * (long double)i / (long double)nfft;
            const long double sh = sinl(ang);
            tab[o++] = {(R)(double)(-2.0L * sh * sh), (R)root(i, nfft).imag()};
        }
        for (int i = 0; i < gp.nhn; ++i) {
            const zd w = root((int64_t)i * mdsp::gx::TWS, nfft);
            tab[o++] = {(R)w.real(), (R)w.imag()};
        }
        for (int i = 0; i < gp.R0; ++i) {
            const zd w = root(i, gp.R0);
            tab[o++] = {(R)w.real(), (R)w.imag()};
        }
    }
    MDSP_TRY(gp.tw.reserve(sizeof(cx<R>) * tab.size()));
    MDSP_HIP(hipMemcpy(gp.tw.p, tab.data(), sizeof(cx<R>) * tab.size(), hipMemcpyHostToDevice));
    MDSP_TRY(gp.win.reserve(sizeof(R) * (size_t)nfft));
    gp.lds_bytes = sizeof(cx<R>) * ((size_t)gp.sc.np * (size_t)gp.sc.nbuf + (size_t)ntab) + (welch ? sizeof(R) * (size_t)S : 0);
    gp.ready = true;
    return MDSP_OK;
}

// a.s, lds_, K, hop, nch, n, nfft, nout, onesided, psd, accumulate, r, ldo, chs, out (columns) filled by the caller
template <typename R, bool CPLX, int MODE>
int gx_launch(GxPlan& gp, GxArgs& a, const double* win_dev, int dtype, hipStream_t st, int64_t* ngroups_out, DevBuf* partial) {
    MDSP_TRY(gx_prepare<R>(gp, dtype, a.nfft, MODE == 0));
    hipLaunchKernelGGL(gx_window_kernel<R>, dim3((unsigned)cdiv(a.nfft, 256)), dim3(256), 0, st, win_dev, gp.win.as<R>(), a.n, a.nfft);
    MDSP_LAUNCH_CHECK();
    a.sc = gp.sc;
    a.nbuf = gp.sc.nbuf;
    a.R0 = gp.R0;
    a.nhs = gp.nhs;
    a.nhn = gp.nhn;
    a.tw = gp.tw.p;
    a.win = gp.win.p;
    const bool pair = !CPLX && (MODE == 0 || gp.R0 == 1);
    a.pairs = pair ? 1 : 0;
    a.units_per_ch = pair ? cdiv(a.K, 2) : a.K;
    a.flush = 64;
    // resident workgroups: what the registers (gx_cu_threads) and the LDS admit
    const int per_cu = std::max<int>(1, std::min<int>(gx_cu_threads(dtype) / gp.sc.T, (int)((size_t)GX_LDS_BYTES / gp.lds_bytes)));
    int64_t resident = std::max<int64_t>(1, (int64_t)device_cu_count() * (tunables().wg_per_cu > 0 ? tunables().wg_per_cu : per_cu) / std::max<int64_t>(1, a.nch));
    int64_t groups = std::max<int64_t>(1, std::min<int64_t>(a.units_per_ch, resident / gp.R0));
    if (gp.R0 > 1) groups = std::max<int64_t>(8, groups / 8 * 8);   // the kernel's XCD mapping walks groups in eights (rounded down: no second round of workgroups for a few)
    a.per_slot = cdiv(a.units_per_ch, groups);
    if (gp.R0 == 1) groups = cdiv(a.units_per_ch, a.per_slot);
    *ngroups_out = groups;
    if (MODE == 0) {
        MDSP_TRY(partial->reserve(sizeof(double) * (size_t)groups * (size_t)a.nch * (size_t)a.nfft));
        a.out = partial->p;
    }
    if (!CPLX && MODE == 0 && !pair) MDSP_FAIL(MDSP_ERR_ASSERTION, "Welch sums always pair real frames");
    const int id = (gp.R0 > 1 ? 10 : 0) + (sizeof(R) == 8 ? 5 : 0) + (MODE == 0 ? (CPLX ? 1 : 0) : (CPLX ? 2 : (pair ? 3 : 4)));
    return mdsp::gx_run(id, a, (unsigned)(groups * gp.R0), (unsigned)a.nch, gp.sc.T, gp.lds_bytes, st);
}
