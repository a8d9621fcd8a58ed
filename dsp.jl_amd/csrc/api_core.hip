// Library core: error state, device selection, memory helpers, pure index arithmetic, measurement helpers.
#include <atomic>
#include <cmath>

#include "common.h"
#include "hostpipe.h"

namespace mdsp {

static thread_local std::string g_err;

int set_error(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
void clear_error() { g_err.clear(); }

int device_cu_count() {
    static int cus[64] = {0};          // per device: a process may drive several GPUs
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cus[dev] == 0) {
        hipDeviceProp_t prop;
        int n = 0;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        cus[dev] = n > 0 ? n : 256;
    }
    return cus[dev];
}

// ---- tunables: the environment is read here and nowhere else ---------------------------------------------
static Tunables g_tun;
static std::once_flag g_tun_once;
static std::mutex g_tun_mu;

static const char* const kFirChoiceNames[] = {"MDSP_FIR_MM", "MDSP_FIR_DEC", "MDSP_FIR_P", "MDSP_FIR_MM_ROWS", "MDSP_FIR_MM_NG", "MDSP_FIR_MM_CH", "MDSP_FIR_MM_T64", "MDSP_FIR_MM_PRIO",
                                               "MDSP_FIR_MM_TIGHT", "MDSP_FIR_MM_RPX", "MDSP_FIR_MM_TIEWAVES", "MDSP_FIR_MM_NBLK", "MDSP_FIR_MM_ND", "MDSP_FIR_MM_NS", "MDSP_FIR_MM_PAD"};
int fir_choice_index(const char* name) {
    for (size_t i = 0; i < sizeof(kFirChoiceNames) / sizeof(kFirChoiceNames[0]); ++i)
        if (!strcmp(name, kFirChoiceNames[i])) return (int)i;
    return -1;
}
int* fir_choice_field(Tunables& t, int index) {
    switch (index) {
        case 0: return &t.fir_mm;
        case 1: return &t.fir_dec;
        case 2: return &t.fir_p;
        case 3: return &t.fir_mm_rows;
        case 4: return &t.fir_mm_ng;
        case 5: return &t.fir_mm_ch;
        case 6: return &t.fir_mm_t64;
        case 7: return &t.fir_mm_prio;
        case 8: return &t.fir_mm_tight;
        case 9: return &t.fir_mm_rpx;
        case 10: return &t.fir_mm_tiewaves;
        case 11: return &t.fir_mm_nblk;
        case 12: return &t.fir_mm_nd;
        case 13: return &t.fir_mm_ns;
        case 14: return &t.fir_mm_pad;
        default: return nullptr;
    }
}
// The choice file: OPT-IN (round 6) -- read only when MDSP_FIR_CHOICE_FILE names it; a drop-in keeps no hidden per-user state (up to round 5 the library
// also looked under ~/.cache).  Its first line must be the key  "# mi355dsp-fir-choices v<library version> gfx950"  (a file measured with another library
// version or on another architecture is ignored as a whole); malformed lines are skipped, '#' starts a comment, knob values are clamped to [-1, 64].
static void read_fir_choices(Tunables& t) {
    const char* path = getenv("MDSP_FIR_CHOICE_FILE");
    if (!path || !*path) return;
    FILE* f = fopen(path, "r");
    if (!f) return;
    char line[1024], key[96];
    snprintf(key, sizeof key, "# mi355dsp-fir-choices v%d gfx950", mdsp_version());
    if (!fgets(line, sizeof line, f) || strncmp(line, key, strlen(key)) != 0) {
        fclose(f);
        return;
    }
    while (fgets(line, sizeof line, f)) {
        if (line[0] == '#') continue;
        FirChoice c;
        long long L, M, n;
        int td, xd, used = 0;
        if (sscanf(line, "%lld %lld %lld %d %d %n", &L, &M, &n, &td, &xd, &used) < 5 || L < 1 || M < 1 || n < 1) continue;
        c.L = L; c.M = M; c.hlen = n; c.taps_dtype = td; c.x_dtype = xd;
        char* p = line + used;
        while (*p && c.nset < 12) {
            char name[64];
            int v = 0, adv = 0;
            if (sscanf(p, " %63[A-Z0-9_]=%d%n", name, &v, &adv) < 2) break;
            const int idx = fir_choice_index(name);
            if (idx >= 0) {
                c.field[c.nset] = idx;
                c.value[c.nset] = v < -1 ? -1 : (v > 64 ? 64 : v);
                ++c.nset;
            }
            p += adv;
            while (*p == ',' || *p == ' ') ++p;
        }
        if (c.nset > 0 && t.fir_choices.size() < 4096) t.fir_choices.push_back(c);
    }
    fclose(f);
}

// Round 6 (VERDICT r5 item 8 iv): the ENVIRONMENT carries fifteen documented variables -- engine choice, cache sizes, the switches a
// deployment may need.  Everything that only steers an experiment (kernel variants, tile shapes, priorities, ablations: ~50 names up to round 5) is a KNOB:
// set through mdsp_set_knob(name, value) by the tuning tools and the tests, never read from the environment of a product build.  Builds with
// -DMDSP_DEBUG_KNOBS also accept every knob as an environment variable of the same name (the tools' sweeps of round 1 - 5 run unchanged there).
struct KnobDef {
    const char* name;
    int Tunables::*field;
    int def, lo, hi;
};
static const KnobDef kKnobs[] = {
    {"MDSP_RUNS_PER_SLOT", &Tunables::runs_per_slot, 1, 1, 1 << 20},
    {"MDSP_OLS_VARIANT", &Tunables::ols_variant, 0, -1 << 20, 1 << 20},
    {"MDSP_OLS_PRIO", &Tunables::ols_prio, 1, 0, 3},
    {"MDSP_SPEC_PRIO", &Tunables::spec_prio, 1, 0, 3},
    {"MDSP_WELCH_VARIANT", &Tunables::welch_variant, 0, -1 << 20, 1 << 20},
    {"MDSP_STFT_VARIANT", &Tunables::stft_variant, 1, -1 << 20, 1 << 20},
    {"MDSP_FIR_LDS_KIB", &Tunables::fir_lds_kib, 20, 4, 150},
    {"MDSP_ARB_NCH", &Tunables::arb_nch, 4, -1 << 20, 1 << 20},
    {"MDSP_ARB_PRIO", &Tunables::arb_prio, 0, 0, 3},
    {"MDSP_ARB_TILE", &Tunables::arb_tile, 0, 0, 1 << 20},
    {"MDSP_OLS_PREFETCH", &Tunables::ols_prefetch, 0, 0, 1},
    {"MDSP_GEN_WIDE", &Tunables::gen_wide, 1, 0, 1},
    {"MDSP_GEN_CT_F64_MAX", &Tunables::gen_ct_f64_max, 8000, 0, 1 << 20},
    {"MDSP_BIG_GROUPS", &Tunables::big_groups, 0, 0, 1 << 20},
    {"MDSP_BIG_WGS", &Tunables::big_wgs, 0, 0, 64},
    {"MDSP_BIG_ABLATE", &Tunables::big_ablate, 0, 0, 1 << 20},
    {"MDSP_BIG_RMAX", &Tunables::big_rmax, 512, 16, 512},
    {"MDSP_BIG_FAST", &Tunables::big_fast, 1, 0, 3},
    {"MDSP_BIG_OLS_LOG2N", &Tunables::big_ols_log2n, 0, 0, 31},
    {"MDSP_BIG_OLS_ROWS", &Tunables::big_ols_rows, 1, 0, 1},
    {"MDSP_BIG_WELCH_ROWS", &Tunables::big_welch_rows, 1, 0, 1},
    {"MDSP_FIR_P", &Tunables::fir_p, 0, 0, 4},
    {"MDSP_FIR_DEC_WGS", &Tunables::fir_dec_wgs, 0, 0, 64},
    {"MDSP_FIR_DEC_ABLATE", &Tunables::fir_dec_ablate, 0, 0, 1 << 20},
    {"MDSP_FIR_DEC_NC", &Tunables::fir_dec_nc, 1, 0, 1},
    {"MDSP_FIR_MM_ROWS", &Tunables::fir_mm_rows, -1, -1, 2},
    {"MDSP_FIR_MM_NG", &Tunables::fir_mm_ng, 0, 0, 64},
    {"MDSP_FIR_MM_CH", &Tunables::fir_mm_ch, 0, 0, 64},
    {"MDSP_FIR_MM_PAD", &Tunables::fir_mm_pad, -1, -1, 1},
    {"MDSP_FIR_MM_VSTORE", &Tunables::fir_mm_vstore, 1, 0, 1},
    {"MDSP_FIR_MM_RPAD", &Tunables::fir_mm_rpad, 0, 0, 64},
    {"MDSP_FIR_MM_PRIO", &Tunables::fir_mm_prio, -1, -1, 1},
    {"MDSP_FIR_MM_T64", &Tunables::fir_mm_t64, 1, 0, 1},
    {"MDSP_FIR_MM_TIGHT", &Tunables::fir_mm_tight, 1, 0, 1},
    {"MDSP_FIR_MM_RPX", &Tunables::fir_mm_rpx, 0, 0, 1},
    {"MDSP_FIR_MM_TIEWAVES", &Tunables::fir_mm_tiewaves, 0, 0, 1},
    {"MDSP_FIR_MM_NBLK", &Tunables::fir_mm_nblk, 1, 0, 1},
    {"MDSP_FIR_MM_ND", &Tunables::fir_mm_nd, 0, 0, 16},
    {"MDSP_FIR_MM_NS", &Tunables::fir_mm_ns, 0, 0, 16},
};
static std::vector<std::pair<int, int>>& knob_overrides() {   // (index into kKnobs, value): what mdsp_set_knob put there, re-applied by every reload
    static std::vector<std::pair<int, int>> v;
    return v;
}

static void read_tunables_locked() {
    Tunables t;
    read_fir_choices(t);                                         // MDSP_FIR_CHOICE_FILE (opt-in)
    auto geti = [](const char* name, int def) {
        const char* e = getenv(name);
        return e && *e ? atoi(e) : def;
    };
    // ---- the environment of a product build: fifteen variables
    if (const char* e = getenv("MDSP_ENGINE")) {                 // engine of plans created with MDSP_ENGINE_AUTO
        if (!strcmp(e, "rocfft")) t.engine = MDSP_ENGINE_ROCFFT;
        else if (!strcmp(e, "fused")) t.engine = MDSP_ENGINE_FUSED;
    }
    t.wg_per_cu = std::max(0, geti("MDSP_WG_PER_CU", 0));
    t.plan_cache_total = std::max(1, geti("MDSP_PLAN_CACHE_TOTAL", 8 * MDSP_PLAN_CACHE_SIZE));
    t.plan_cache_idle = std::max(1, geti("MDSP_PLAN_CACHE_IDLE", 64));
    t.rocfft_chunk_mib = std::max(1, geti("MDSP_ROCFFT_CHUNK_MIB", 192));
    t.host_chunk_mib = std::max(1, geti("MDSP_HOST_CHUNK_MIB", 64));
    t.big_chunk_mib = std::max(1, geti("MDSP_BIG_CHUNK_MIB", 1024));
    t.bigfft = geti("MDSP_BIGFFT", 1);
    t.gx = geti("MDSP_GX", 1);
    t.fir_mm = geti("MDSP_FIR_MM", -1);
    t.fir_dec = geti("MDSP_FIR_DEC", 1);
    t.fir_exact = geti("MDSP_FIR_EXACT", 0);
    t.arb_scan = geti("MDSP_ARB_SCAN", 1);
    if (const char* e = getenv("MDSP_ARB_SCAN_MIN")) t.arb_scan_min = atoll(e);
    // ---- knobs: defaults, (debug builds: the environment,) then what mdsp_set_knob set
    for (const KnobDef& k : kKnobs) {
        int v = k.def;
#ifdef MDSP_DEBUG_KNOBS
        v = geti(k.name, k.def);
#endif
        t.*(k.field) = std::min(k.hi, std::max(k.lo, v));
    }
    for (const auto& o : knob_overrides()) t.*(kKnobs[o.first].field) = std::min(kKnobs[o.first].hi, std::max(kKnobs[o.first].lo, o.second));
#ifdef MDSP_DEBUG_KNOBS
    t.ablate = geti("MDSP_ABLATE", 0);
    t.welch_nohalf = getenv("MDSP_WELCH_NOHALF") != nullptr;
    t.stft_noshift = getenv("MDSP_STFT_NOSHIFT") != nullptr;
    t.stft_nopair = getenv("MDSP_STFT_NOPAIR") != nullptr;
    t.stft_nodirect = getenv("MDSP_STFT_NODIRECT") != nullptr;
    t.fir_generic = getenv("MDSP_FIR_GENERIC") != nullptr;
    t.fir_identity_lanes = getenv("MDSP_FIR_IDENTITY_LANES") != nullptr;
    t.mt_passes = getenv("MDSP_MT_PASSES") != nullptr;
    t.arb_prof = getenv("MDSP_ARB_PROF") != nullptr;
#endif
    g_tun = t;
}
static thread_local const Tunables* tl_tun_override = nullptr;
const Tunables& tunables() {
    if (tl_tun_override) return *tl_tun_override;
    std::call_once(g_tun_once, [] {
        std::lock_guard<std::mutex> lk(g_tun_mu);
        read_tunables_locked();
    });
    return g_tun;
}
const Tunables* tunables_override(const Tunables* t) {
    const Tunables* prev = tl_tun_override;
    tl_tun_override = t;
    return prev;
}
// Tuning tools only.  Readers (exec / plan paths) take no lock: a reload must not race with other threads' library calls (mi355dsp.h says so).
static std::atomic<uint64_t> g_tun_gen{1};
void reload_tunables() {
    (void)tunables();
    std::lock_guard<std::mutex> lk(g_tun_mu);
    read_tunables_locked();
    g_tun_gen.fetch_add(1, std::memory_order_release);
}
uint64_t tunables_generation() { return g_tun_gen.load(std::memory_order_acquire); }

// ---- pure index arithmetic ---------------------------------------------------------------------------
static int64_t nextprod2357(int64_t n) {
    if (n <= 1) return 1;
    int64_t best = INT64_MAX;
    for (int64_t p7 = 1; p7 < 2 * n && p7 > 0; p7 *= 7)
        for (int64_t p5 = p7; p5 < 2 * n && p5 > 0; p5 *= 5)
            for (int64_t p3 = p5; p3 < 2 * n && p3 > 0; p3 *= 3) {
                int64_t v = p3;
                while (v < n) v *= 2;
                if (v < best) best = v;
            }
    return best;
}

static double os_fft_complexity(double nfft, double nb) { return (nfft * std::log2(nfft) + nfft) / (nfft - nb + 1); }

static int64_t ceil_log2(int64_t v) {  // ceil(Int, log2(v)) for v >= 1, exact for powers of two
    int64_t p = 0;
    while ((int64_t(1) << p) < v) ++p;
    return p;
}

// 128-bit helpers so (len*L - phi + 1)/M style expressions never overflow
static int64_t ceil_div_i128(__int128 a, __int128 b) {
    __int128 q = a / b, r = a % b;
    if (r != 0 && ((r > 0) == (b > 0))) ++q;
    return (int64_t)q;
}
static int64_t floor_div_i128(__int128 a, __int128 b) {
    __int128 q = a / b, r = a % b;
    if (r != 0 && ((r > 0) != (b > 0))) --q;
    return (int64_t)q;
}

}  // namespace mdsp

using namespace mdsp;

extern "C" {

int mdsp_version(void) { return MDSP_VERSION; }

const char* mdsp_last_error_string(void) { return g_err.c_str(); }

int mdsp_device_count(int* count) {
    if (!count) MDSP_FAIL(MDSP_ERR_ARGUMENT, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return MDSP_OK;
}

int mdsp_init(int device) {
    int n = 0;
    MDSP_TRY(mdsp_device_count(&n));
    if (n <= 0) MDSP_FAIL(MDSP_ERR_DEVICE, "no HIP device visible: libmi355dsp has no CPU fallback");
    if (device < 0 || device >= n) MDSP_FAIL(MDSP_ERR_ARGUMENT, "device %d out of range [0,%d)", device, n);
    MDSP_HIP(hipSetDevice(device));
    (void)tunables();   // the environment is read once, here (or on first use)
    return MDSP_OK;
}

int mdsp_shutdown(void) {
    (void)mdsp_plan_cache_clear();   // borrowed plans die here: a host calls this once, after its last use of the library
    hostpipe::release_all();          // device and page-locked staging buffers of the host-array pipelines
    return MDSP_OK;
}

int mdsp_reload_tunables(void) {
    reload_tunables();
    return MDSP_OK;
}

int mdsp_set_knob(const char* name, int value, int unset) {
    if (!name) MDSP_FAIL(MDSP_ERR_ARGUMENT, "name is NULL");
    for (size_t i = 0; i < sizeof(kKnobs) / sizeof(kKnobs[0]); ++i) {
        if (strcmp(name, kKnobs[i].name)) continue;
        {
            (void)tunables();
            std::lock_guard<std::mutex> lk(g_tun_mu);
            auto& ov = knob_overrides();
            for (size_t j = 0; j < ov.size(); ++j)
                if (ov[j].first == (int)i) {
                    ov.erase(ov.begin() + (long)j);
                    break;
                }
            if (!unset) ov.emplace_back((int)i, value);
        }
        reload_tunables();
        return MDSP_OK;
    }
    MDSP_FAIL(MDSP_ERR_ARGUMENT, "%s is not a knob (environment variables of the product build are set in the environment: INTEGRATION.md)", name);
}

int mdsp_debug_knobs(void) {
#ifdef MDSP_DEBUG_KNOBS
    return 1;
#else
    return 0;
#endif
}

int mdsp_malloc(void** dev_ptr, size_t bytes) {
    if (!dev_ptr) MDSP_FAIL(MDSP_ERR_ARGUMENT, "dev_ptr is NULL");
    *dev_ptr = nullptr;
    if (bytes == 0) return MDSP_OK;
    MDSP_HIP(hipMalloc(dev_ptr, bytes));
    return MDSP_OK;
}
int mdsp_free(void* dev_ptr) {
    if (dev_ptr) MDSP_HIP(hipFree(dev_ptr));
    return MDSP_OK;
}
int mdsp_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, void* stream) {
    if (bytes == 0) return MDSP_OK;
    MDSP_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    MDSP_HIP(hipStreamSynchronize(as_stream(stream)));
    return MDSP_OK;
}
int mdsp_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream) {
    if (bytes == 0) return MDSP_OK;
    MDSP_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    MDSP_HIP(hipStreamSynchronize(as_stream(stream)));
    return MDSP_OK;
}
int mdsp_memset(void* dst_dev, int value, size_t bytes, void* stream) {
    if (bytes == 0) return MDSP_OK;
    MDSP_HIP(hipMemsetAsync(dst_dev, value, bytes, as_stream(stream)));
    return MDSP_OK;
}
int mdsp_stream_synchronize(void* stream) {
    MDSP_HIP(hipStreamSynchronize(as_stream(stream)));
    return MDSP_OK;
}

// util.jl:134
int64_t mdsp_nextfastfft(int64_t n) { return nextprod2357(n); }

// dspbase.jl:268-291
int64_t mdsp_optimal_fft_len(int64_t nb, int64_t nx) {
    if (nb < 1 || nx < 0) return -1;
    const int64_t nfull = nb + nx - 1;
    if (nfull < 1) return 1;
    const int64_t first_pow2 = ceil_log2(nb);
    const int64_t max_pow2 = ceil_log2(nfull);
    double prev = os_fft_complexity((double)(int64_t(1) << first_pow2), (double)nb);
    int64_t pow2 = first_pow2 + 1;
    while (pow2 <= max_pow2) {
        const double cur = os_fft_complexity((double)(int64_t(1) << pow2), (double)nb);
        if (cur > prev) break;
        prev = cur;
        ++pow2;
    }
    int64_t nfft = pow2 > max_pow2 ? (int64_t(1) << max_pow2) : (int64_t(1) << (pow2 - 1));
    if (nfft > nfull) nfft = nextprod2357(nfull);
    return nfft;
}

// periodograms.jl:49-50
int64_t mdsp_frame_count(int64_t len, int64_t n, int64_t noverlap) {
    if (n <= 0 || noverlap < 0 || noverlap >= n) return -1;
    return len >= n ? (len - n) / (n - noverlap) + 1 : 0;
}

// stream_filt.jl:317-322: ceil(((inputlength*L) - phi + 1) / M)
int64_t mdsp_outputlength(int64_t inputlength, int64_t L, int64_t M, int64_t initial_phi) {
    if (L <= 0 || M <= 0) return -1;
    return ceil_div_i128((__int128)inputlength * L - initial_phi + 1, M);
}

// stream_filt.jl:358-364: round((outputlength*M + phi - d) / L, r), d = M for RoundUp else 1
int64_t mdsp_inputlength(int64_t outputlength, int64_t L, int64_t M, int64_t initial_phi, int round_up) {
    if (L <= 0 || M <= 0) return -1;
    const __int128 num = (__int128)outputlength * M + initial_phi - (round_up ? M : 1);
    return round_up ? ceil_div_i128(num, L) : floor_div_i128(num, L);
}

// Filters/filt.jl:490, 504-517
int mdsp_ols_block_geometry(int64_t nb, int64_t nx, int64_t nfft, int64_t iblock, int64_t* off, int64_t* npadbefore,
                            int64_t* xstart, int64_t* n, int64_t* nout) {
    if (nb < 1 || nfft < nb || nx < 0 || iblock < 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "bad overlap-save geometry");
    const int64_t L = std::min<int64_t>(nx, nfft - (nb - 1));
    if (L <= 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "empty input has no blocks");
    const int64_t o = 1 + iblock * L;
    if (o > nx) MDSP_FAIL(MDSP_ERR_ARGUMENT, "block %lld beyond the end of x", (long long)iblock);
    const int64_t pad = std::max<int64_t>(0, nb - o);
    const int64_t xs = o - nb + pad + 1;
    if (off) *off = o;
    if (npadbefore) *npadbefore = pad;
    if (xstart) *xstart = xs;
    if (n) *n = std::min<int64_t>(nfft - pad, nx - xs + 1);
    if (nout) *nout = std::min<int64_t>(L, nx - o + 1);
    return MDSP_OK;
}

// ---- measurement helpers -----------------------------------------------------------------------------
int mdsp_event_create(void** ev) {
    if (!ev) MDSP_FAIL(MDSP_ERR_ARGUMENT, "ev is NULL");
    hipEvent_t e;
    MDSP_HIP(hipEventCreate(&e));
    *ev = (void*)e;
    return MDSP_OK;
}
int mdsp_event_destroy(void* ev) {
    if (ev) MDSP_HIP(hipEventDestroy((hipEvent_t)ev));
    return MDSP_OK;
}
int mdsp_event_record(void* ev, void* stream) {
    MDSP_HIP(hipEventRecord((hipEvent_t)ev, as_stream(stream)));
    return MDSP_OK;
}
int mdsp_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms) {
    if (!ms) MDSP_FAIL(MDSP_ERR_ARGUMENT, "ms is NULL");
    MDSP_HIP(hipEventSynchronize((hipEvent_t)ev_stop));
    MDSP_HIP(hipEventElapsedTime(ms, (hipEvent_t)ev_start, (hipEvent_t)ev_stop));
    return MDSP_OK;
}

}  // extern "C"

// float4 grid-stride copy: the on-box HBM yardstick.  MDSP_COPY_MODE selects experiments on the access pattern
// (1: four independent float4 per thread and iteration, 2: nontemporal loads/stores, 3: one contiguous chunk per workgroup,
// 4: read-only sum (no stores), 5: write-only fill); MDSP_COPY_WGS = workgroups per CU (default 8).
__global__ __launch_bounds__(256) void mdsp_copy_kernel(float4* __restrict__ dst, const float4* __restrict__ src, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void mdsp_copy4_kernel(float4* __restrict__ dst, const float4* __restrict__ src, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a;
        dst[i + stride] = b;
        dst[i + 2 * stride] = c;
        dst[i + 3 * stride] = d;
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void mdsp_copy_nt_kernel(float4* __restrict__ dst, const float4* __restrict__ src, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(src) + i);
        __builtin_nontemporal_store(v, reinterpret_cast<f4v*>(dst) + i);
    }
}

__global__ __launch_bounds__(256) void mdsp_copy_chunk_kernel(float4* __restrict__ dst, const float4* __restrict__ src, size_t n4) {
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void mdsp_read_kernel(float4* __restrict__ dst, const float4* __restrict__ src, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = src[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345678f) dst[0] = acc;   // keeps the loads alive
}

__global__ __launch_bounds__(256) void mdsp_fill_kernel(float4* __restrict__ dst, const float4* __restrict__, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const float4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) dst[i] = v;
}

// Round 6 (tools/ubench/copy_sweep.hip, profiles/r06_copy_sweep.json): the access patterns that measured fastest on this part -- 6: copy with nontemporal
// loads and sc0 | sc1 stores, four 16-byte accesses in flight per thread (5.83 TB/s against 5.75 at the default policy; the guide's 6.29 is not reached by any
// of 150 shapes x policies); 7: read-only with nontemporal loads, eight in flight (7.27 TB/s against 6.40: the one policy bit that matters).
template <int U, int LAUX, int SAUX, bool READ>
__global__ __launch_bounds__(256) void mdsp_copy_policy_kernel(float4* __restrict__ dst, const float4* __restrict__ src, size_t n4) {
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    const size_t step = (size_t)gridDim.x * 256 * U;
    u4v keep = {0, 0, 0, 0};
    for (size_t base = (size_t)blockIdx.x * 256 * U; base < n4; base += step) {   // a descriptor per trip: 32-bit offsets inside it, any buffer size
        const size_t left = n4 - base;
        const unsigned cnt = (unsigned)(left < (size_t)256 * U ? left : (size_t)256 * U);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + base), 0, (int)(cnt * 16u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(dst + base), 0, (int)(cnt * 16u), 0x00020000);
        u4v v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((u * 256u + threadIdx.x) * 16u), 0, LAUX);
        if (READ) {
#pragma unroll
            for (int u = 0; u < U; ++u) keep += v[u];
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b128(v[u], rd, (int)((u * 256u + threadIdx.x) * 16u), 0, SAUX);   // (past cnt: dropped)
        }
    }
    if (READ && keep.x == 0x12345678u && keep.y == 0x9abcdef0u) dst[0] = float4{1.f, 2.f, 3.f, 4.f};   // keeps the loads alive
}

extern "C" int mdsp_copy_bench_mode(void* dst_dev, const void* src_dev, size_t bytes, int mode, int wgs, void* stream) {
    if (bytes % 16) MDSP_FAIL(MDSP_ERR_ARGUMENT, "bytes must be a multiple of 16");
    const size_t n4 = bytes / 16;
    if (n4 == 0) return MDSP_OK;
    if (mode < 0 || mode > 7) MDSP_FAIL(MDSP_ERR_ARGUMENT, "copy mode %d out of range [0,7]", mode);
    if (wgs < 1) wgs = 8;
    const int grid = (int)std::min<size_t>((n4 + 255) / 256, (size_t)device_cu_count() * wgs);
    if (mode >= 6) {
        if (mode == 6) hipLaunchKernelGGL((mdsp_copy_policy_kernel<4, 2, 17, false>), dim3(grid), dim3(256), 0, as_stream(stream), (float4*)dst_dev, (const float4*)src_dev, n4);
        else hipLaunchKernelGGL((mdsp_copy_policy_kernel<8, 2, 0, true>), dim3(grid), dim3(256), 0, as_stream(stream), (float4*)dst_dev, (const float4*)src_dev, n4);
        MDSP_LAUNCH_CHECK();
        return MDSP_OK;
    }
    auto kern = mode == 1 ? mdsp_copy4_kernel : mode == 2 ? mdsp_copy_nt_kernel : mode == 3 ? mdsp_copy_chunk_kernel : mode == 4 ? mdsp_read_kernel
                                                                                                     : mode == 5 ? mdsp_fill_kernel : mdsp_copy_kernel;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, as_stream(stream), (float4*)dst_dev, (const float4*)src_dev, n4);
    MDSP_LAUNCH_CHECK();
    return MDSP_OK;
}

extern "C" int mdsp_copy_bench(void* dst_dev, const void* src_dev, size_t bytes, void* stream) {
    return mdsp_copy_bench_mode(dst_dev, src_dev, bytes, 0, 8, stream);
}
