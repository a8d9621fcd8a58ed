// Least-recently-used cache of device plans INSIDE the library (mdsp_*_plan_cached).
//
// DSP.jl's function-style entry points -- filt(b, x), conv(u, v), welch_pgram(s, n, noverlap), stft(...), periodogram(s) -- build their
// FFTW plans on every call; that is cheap on a CPU.  A device plan (tap spectrum, root tables, window upload, work buffers) costs
// allocations and host-to-device copies: about a millisecond, against ~45 us for the call itself at 2^20 samples (DESIGN.md section 5,
// "Per-call latency").  A host that mirrors those entry points one-to-one (the Julia twin's per-call methods) asks for its plan here:
// the key is everything the plan depends on -- device, calling thread, stream, sizes, dtype, engine, mode and the CONTENTS of the taps /
// window -- so a hit is exactly the plan a fresh create would have produced.
//
// Ownership: a cached handle is BORROWED.  It must not be passed to mdsp_*_plan_destroy.  The cache is partitioned BY CALLING THREAD: every
// thread has its own two LRU lists -- one of MDSP_PLAN_CACHE_SIZE user-visible plans ('o' overlap-save, 'w' Welch, 's' STFT), one of the same
// size for the objects the library caches for itself ('f' per-call FIR filters, 't' tap uploads of the stateful FIR entry points) -- and only
// that thread's own requests of the same class ever evict from them.  So a borrowed handle stays valid until the thread that borrowed it has
// made MDSP_PLAN_CACHE_SIZE further DISTINCT cached-plan requests, whatever other threads do, and the library's internal objects can never
// push a user's plan out.  (Round 2 kept one global list: misses of other threads, or this thread's own internal entries, could destroy a plan
// between its lookup and its use.)  mdsp_plan_cache_clear() destroys every thread's entries: call it only while no other thread is inside the
// library.  The lists of threads that have exited stay until then (at most 2 x MDSP_PLAN_CACHE_SIZE small objects per thread).
#include <functional>
#include <list>
#include <thread>
#include <unordered_map>

#include "common.h"

using namespace mdsp;

namespace {

constexpr size_t kCapacity = MDSP_PLAN_CACHE_SIZE;

struct Entry {
    std::string key;
    void* handle;
    std::function<void(void*)> destroy;
};

struct PerThread {
    std::list<Entry> user, internal;   // front = most recently used
};

struct Cache {
    std::mutex mu;                                         // guards the map and the lists (held for list surgery only, never across device work)
    std::unordered_map<std::thread::id, PerThread> by_thread;
    int64_t hits = 0, misses = 0;
};

Cache& cache() {
    static Cache* c = new Cache();   // never destroyed: no device calls from static destructors at process exit
    return *c;
}

bool is_internal(const std::string& key) { return !key.empty() && (key[0] == 'f' || key[0] == 't'); }

void put(std::string& k, const void* p, size_t n) { k.append(static_cast<const char*>(p), n); }
template <typename T> void put(std::string& k, T v) { put(k, &v, sizeof(v)); }

}  // namespace

namespace mdsp {
std::string plan_cache_key(char kind, void* stream) {
    std::string k(1, kind);
    int dev = -1;
    (void)hipGetDevice(&dev);
    put(k, dev);
    put(k, std::hash<std::thread::id>()(std::this_thread::get_id()));
    put(k, stream);
    return k;
}
}  // namespace mdsp

namespace {
std::string base_key(char kind, void* stream) { return plan_cache_key(kind, stream); }

}  // namespace

namespace mdsp {
// find or create; `make` builds a new object into *out.  Only the calling thread's list of the key's class is searched and evicted from.
int plan_cache_get(const std::string& key, void** out, const std::function<int(void**)>& make, std::function<void(void*)> destroy) {
    Cache& c = cache();
    const std::thread::id me = std::this_thread::get_id();
    const bool internal = is_internal(key);
    {
        std::lock_guard<std::mutex> lk(c.mu);
        PerThread& pt = c.by_thread[me];
        std::list<Entry>& lru = internal ? pt.internal : pt.user;
        for (auto it = lru.begin(); it != lru.end(); ++it)
            if (it->key == key) {
                lru.splice(lru.begin(), lru, it);
                ++c.hits;
                *out = it->handle;
                return MDSP_OK;
            }
    }
    void* h = nullptr;
    MDSP_TRY(make(&h));            // outside the lock: plan creation touches the device
    std::vector<Entry> evicted;
    {
        std::lock_guard<std::mutex> lk(c.mu);
        ++c.misses;
        PerThread& pt = c.by_thread[me];
        std::list<Entry>& lru = internal ? pt.internal : pt.user;
        lru.push_front(Entry{key, h, std::move(destroy)});
        while (lru.size() > kCapacity) {   // this thread's least recently used entries of this class
            evicted.push_back(std::move(lru.back()));
            lru.pop_back();
        }
    }
    for (auto& e : evicted) e.destroy(e.handle);   // hipFree inside synchronises with any launch still using the buffers
    *out = h;
    return MDSP_OK;
}
}  // namespace mdsp

namespace {
int get(const std::string& key, void** out, const std::function<int(void**)>& make, std::function<void(void*)> destroy) {
    return plan_cache_get(key, out, make, std::move(destroy));
}

size_t real_size(int dtype) { return dtype_is_double(dtype) ? 8 : 4; }

}  // namespace

extern "C" {

int mdsp_ols_plan_cached(mdsp_ols_plan* plan, const void* taps_host, int64_t nb, int64_t nfft, int64_t nx_hint, int dtype, int mode, int engine,
                         void* stream) {
    if (!plan) MDSP_FAIL(MDSP_ERR_ARGUMENT, "plan is NULL");
    *plan = nullptr;
    if (!taps_host || nb < 1) MDSP_FAIL(MDSP_ERR_ARGUMENT, "filter vector b must be non-empty");
    if (!dtype_valid(dtype)) MDSP_FAIL(MDSP_ERR_ARGUMENT, "invalid dtype %d", dtype);
    if (nfft == 0) nfft = mdsp_optimal_fft_len(nb, std::max<int64_t>(nx_hint, 1));   // the key holds the resolved length, not the hint
    std::string k = base_key('o', stream);
    put(k, nb); put(k, nfft); put(k, dtype); put(k, mode); put(k, engine == MDSP_ENGINE_AUTO ? tunables().engine : engine);
    put(k, taps_host, (size_t)nb * dtype_size(dtype));
    void* h = nullptr;
    MDSP_TRY(get(k, &h, [&](void** out) { return mdsp_ols_plan_create(reinterpret_cast<mdsp_ols_plan*>(out), taps_host, nb, nfft, nx_hint, dtype, mode, engine); },
                 [](void* p) { (void)mdsp_ols_plan_destroy(static_cast<mdsp_ols_plan>(p)); }));
    *plan = static_cast<mdsp_ols_plan>(h);
    return MDSP_OK;
}

int mdsp_welch_plan_cached(mdsp_welch_plan* plan, int64_t n, int64_t noverlap, int64_t nfft, const double* window_host, double r, int onesided,
                           int dtype, int engine, void* stream) {
    if (!plan) MDSP_FAIL(MDSP_ERR_ARGUMENT, "plan is NULL");
    *plan = nullptr;
    if (n < 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "negative size");
    std::string k = base_key('w', stream);
    put(k, n); put(k, noverlap); put(k, nfft); put(k, r); put(k, onesided); put(k, dtype); put(k, engine == MDSP_ENGINE_AUTO ? tunables().engine : engine);
    put(k, (char)(window_host != nullptr));
    if (window_host) put(k, window_host, (size_t)n * sizeof(double));
    void* h = nullptr;
    MDSP_TRY(get(k, &h, [&](void** out) { return mdsp_welch_plan_create(reinterpret_cast<mdsp_welch_plan*>(out), n, noverlap, nfft, window_host, r, onesided, dtype, engine); },
                 [](void* p) { (void)mdsp_welch_plan_destroy(static_cast<mdsp_welch_plan>(p)); }));
    *plan = static_cast<mdsp_welch_plan>(h);
    return MDSP_OK;
}

int mdsp_stft_plan_cached(mdsp_stft_plan* plan, int64_t n, int64_t noverlap, int64_t nfft, const double* window_host, double r, int onesided,
                          int psd_only, int dtype, int engine, void* stream) {
    if (!plan) MDSP_FAIL(MDSP_ERR_ARGUMENT, "plan is NULL");
    *plan = nullptr;
    if (n < 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "negative size");
    std::string k = base_key('s', stream);
    put(k, n); put(k, noverlap); put(k, nfft); put(k, r); put(k, onesided); put(k, psd_only); put(k, dtype);
    put(k, engine == MDSP_ENGINE_AUTO ? tunables().engine : engine);
    put(k, (char)(window_host != nullptr));
    if (window_host) put(k, window_host, (size_t)n * sizeof(double));
    void* h = nullptr;
    MDSP_TRY(get(k, &h, [&](void** out) { return mdsp_stft_plan_create(reinterpret_cast<mdsp_stft_plan*>(out), n, noverlap, nfft, window_host, r, onesided, psd_only, dtype, engine); },
                 [](void* p) { (void)mdsp_stft_plan_destroy(static_cast<mdsp_stft_plan>(p)); }));
    *plan = static_cast<mdsp_stft_plan>(h);
    return MDSP_OK;
}

int mdsp_plan_cache_stats(int64_t* entries, int64_t* hits, int64_t* misses) {
    Cache& c = cache();
    std::lock_guard<std::mutex> lk(c.mu);
    if (entries) {
        *entries = 0;
        for (auto& kv : c.by_thread) *entries += (int64_t)(kv.second.user.size() + kv.second.internal.size());
    }
    if (hits) *hits = c.hits;
    if (misses) *misses = c.misses;
    return MDSP_OK;
}

int mdsp_plan_cache_clear(void) {
    Cache& c = cache();
    std::list<Entry> all;
    {
        std::lock_guard<std::mutex> lk(c.mu);
        for (auto& kv : c.by_thread) {
            all.splice(all.end(), kv.second.user);
            all.splice(all.end(), kv.second.internal);
        }
        c.by_thread.clear();
    }
    for (auto& e : all) e.destroy(e.handle);
    return MDSP_OK;
}

}  // extern "C"
