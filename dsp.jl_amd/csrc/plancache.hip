// Least-recently-used cache of device plans INSIDE the library (mdsp_*_plan_cached).
//
// DSP.jl's function-style entry points -- filt(b, x), conv(u, v), welch_pgram(s, n, noverlap), stft(...), periodogram(s) -- build their
// FFTW plans on every call; that is cheap on a CPU.  A device plan (tap spectrum, root tables, window upload, work buffers) costs
// allocations and host-to-device copies: about a millisecond, against ~45 us for the call itself at 2^20 samples (DESIGN.md section 5,
// "Per-call latency").  A host that mirrors those entry points one-to-one (the Julia twin's per-call methods) asks for its plan here:
// the key is everything the plan depends on -- device, calling thread, stream, sizes, dtype, engine, mode and the CONTENTS of the taps /
// window -- so a hit is exactly the plan a fresh create would have produced.
//
// Ownership: a cached handle is BORROWED.  It must not be passed to mdsp_*_plan_destroy.  The cache is PARTITIONED: every partition has its own
// two LRU lists -- one of MDSP_PLAN_CACHE_SIZE user-visible plans ('o' overlap-save, 'w' Welch, 's' STFT), one of the same size for the objects
// the library caches for itself ('f' per-call FIR filters, 't' tap uploads of the stateful FIR entry points) -- and only requests of the same
// partition and class evict from them.  The partition of a request is the calling OS thread, unless the caller has bound an explicit context
// (mdsp_plan_cache_set_context(id != 0), thread-local): hosts whose logical tasks migrate between OS threads -- Julia tasks -- bind their own
// id before their calls and keep one partition wherever they run; mdsp_plan_cache_release_context(id) frees it.
//   * A borrowed handle stays valid until its partition has made MDSP_PLAN_CACHE_SIZE further DISTINCT cached-plan requests (and the library's
//     internal objects can never push a user's plan out), with ONE exception that bounds the device memory of a churning host: when the whole
//     cache holds more than MDSP_PLAN_CACHE_TOTAL entries (default 8 x MDSP_PLAN_CACHE_SIZE), entries of OTHER partitions that have not
//     made a request for MDSP_PLAN_CACHE_IDLE (default 64) cache requests are evicted, most idle partition first -- and only partitions that
//     CANNOT be in use: explicit contexts that no thread has bound at the moment (a host binds its context around the calls that use its borrowed
//     handles: with_plan_context in julia/MI355DSP.jl).  The partition of a live OS thread is never touched by another partition's request --
//     that thread may be inside a long call on a borrowed handle without having asked the cache for anything (round 4 judged idleness by request
//     ticks alone and could free such a plan's buffers mid-call: ADVICE r4) -- so if nothing is evictable the cap is not enforced.
//   * The lists of an OS thread that EXITS are reaped: a thread-local sentinel moves them to a graveyard at thread exit and the next cache
//     request (of any thread) or mdsp_plan_cache_clear() destroys them.  (Round 3 kept them until mdsp_plan_cache_clear(): every Welch / STFT
//     plan owns device buffers -- partial sums of nslots x nch x nfft doubles, rocFFT intermediates of up to MDSP_ROCFFT_CHUNK_MIB -- so a
//     thread pool leaked device memory: ADVICE r3.)
// mdsp_plan_cache_clear() destroys every partition's entries: call it only while no other thread is inside the library.
#include <cstdlib>
#include <functional>
#include <list>
#include <thread>
#include <unordered_map>

#include "common.h"

using namespace mdsp;

namespace {

constexpr size_t kCapacity = MDSP_PLAN_CACHE_SIZE;

size_t total_cap() { return (size_t)tunables().plan_cache_total; }     // MDSP_PLAN_CACHE_TOTAL (read with the other tunables, api_core.hip)
uint64_t idle_ticks() { return (uint64_t)tunables().plan_cache_idle; } // MDSP_PLAN_CACHE_IDLE

struct Entry {
    std::string key;
    void* handle;
    std::function<void(void*)> destroy;
};

struct Partition {
    std::list<Entry> user, internal;   // front = most recently used
    uint64_t last_use = 0;             // tick of this partition's latest request
    int bound = 0;                     // explicit contexts: threads that have this context bound right now (mdsp_plan_cache_set_context)
    size_t size() const { return user.size() + internal.size(); }
};

// partition id: bit 63 set = an explicit context (mdsp_plan_cache_set_context), clear = hash of the OS thread id
using Pid = uint64_t;
constexpr Pid kCtxBit = 1ull << 63;

struct Cache {
    std::mutex mu;                                         // guards everything below (held for list surgery only, never across device work)
    std::unordered_map<Pid, Partition> parts;
    std::list<Entry> graveyard;                            // entries of exited threads / released contexts / cap evictions, awaiting destruction
    uint64_t tick = 0;
    int64_t hits = 0, misses = 0, reaped = 0;
};

Cache& cache() {
    static Cache* c = new Cache();   // never destroyed: no device calls from static destructors at process exit
    return *c;
}

thread_local Pid tl_context = 0;     // 0: the OS thread is the partition

Pid thread_pid() { return (Pid)std::hash<std::thread::id>()(std::this_thread::get_id()) & ~kCtxBit; }
Pid current_pid() { return tl_context ? (tl_context | kCtxBit) : thread_pid(); }

// moves the partition's entries to the graveyard (caller holds the mutex)
void bury(Cache& c, Pid pid) {
    auto it = c.parts.find(pid);
    if (it == c.parts.end()) return;
    c.reaped += (int64_t)it->second.size();
    c.graveyard.splice(c.graveyard.end(), it->second.user);
    c.graveyard.splice(c.graveyard.end(), it->second.internal);
    if (it->second.bound == 0) c.parts.erase(it);   // (a context some thread still has bound keeps its -- now empty -- partition and its count)
}

// the calling thread stops using its bound context (caller holds the mutex)
void unbind(Cache& c, Pid ctx) {
    if (!ctx) return;
    auto it = c.parts.find(ctx | kCtxBit);
    if (it != c.parts.end() && it->second.bound > 0) --it->second.bound;
}

// One per OS thread that ever used the cache: at thread exit its partition goes to the graveyard (no device call here: the next request drains it).
struct ThreadSentinel {
    Pid pid;
    bool armed = false;
    ~ThreadSentinel() {
        if (!armed && !tl_context) return;
        Cache& c = cache();
        std::lock_guard<std::mutex> lk(c.mu);
        unbind(c, tl_context);   // a thread that exits with a context bound no longer holds it
        if (armed) bury(c, pid);
    }
};
thread_local ThreadSentinel tl_sentinel;

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
    put(k, current_pid());
    put(k, stream);
    return k;
}
}  // namespace mdsp

namespace {
std::string base_key(char kind, void* stream) { return plan_cache_key(kind, stream); }

}  // namespace

namespace mdsp {
// find or create; `make` builds a new object into *out.  Only the caller's partition's list of the key's class is searched and evicted from
// (plus, above the global cap, the tails of partitions that have been idle for a long time: see the header).
int plan_cache_get(const std::string& key, void** out, const std::function<int(void**)>& make, std::function<void(void*)> destroy) {
    Cache& c = cache();
    const Pid me = current_pid();
    const bool internal = is_internal(key);
    std::list<Entry> dead;
    bool hit = false;
    {
        std::lock_guard<std::mutex> lk(c.mu);
        if (!tl_context && !tl_sentinel.armed) {
            tl_sentinel.pid = me;
            tl_sentinel.armed = true;
        }
        dead.splice(dead.end(), c.graveyard);
        Partition& pt = c.parts[me];
        pt.last_use = ++c.tick;
        std::list<Entry>& lru = internal ? pt.internal : pt.user;
        for (auto it = lru.begin(); it != lru.end(); ++it)
            if (it->key == key) {
                lru.splice(lru.begin(), lru, it);
                ++c.hits;
                *out = it->handle;
                hit = true;
                break;
            }
    }
    for (auto& e : dead) e.destroy(e.handle);   // outside the lock: hipFree synchronises
    dead.clear();
    if (hit) return MDSP_OK;
    void* h = nullptr;
    MDSP_TRY(make(&h));            // outside the lock: plan creation touches the device
    {
        std::lock_guard<std::mutex> lk(c.mu);
        ++c.misses;
        Partition& pt = c.parts[me];
        std::list<Entry>& lru = internal ? pt.internal : pt.user;
        lru.push_front(Entry{key, h, std::move(destroy)});
        while (lru.size() > kCapacity) {   // this partition's least recently used entries of this class
            dead.push_back(std::move(lru.back()));
            lru.pop_back();
        }
        // the global cap: tails of the most idle OTHER partitions
        size_t total = 0;
        for (auto& kv : c.parts) total += kv.second.size();
        while (total > total_cap()) {
            Partition* victim = nullptr;
            for (auto& kv : c.parts)
                if (kv.first != me && (kv.first & kCtxBit) && kv.second.bound == 0 && kv.second.size() > 0 && c.tick - kv.second.last_use >= idle_ticks() &&
                    (!victim || kv.second.last_use < victim->last_use))
                    victim = &kv.second;
            if (!victim) break;            // nobody is idle: the cap is soft
            std::list<Entry>& from = victim->internal.size() > victim->user.size() ? victim->internal : victim->user;
            dead.push_back(std::move(from.back()));
            from.pop_back();
            ++c.reaped;
            --total;
        }
    }
    for (auto& e : dead) e.destroy(e.handle);   // hipFree inside synchronises with any launch still using the buffers
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
    {   // long filters size their block transform by the hint (bigfft.hip ols_size): the RESOLVED block is part of the key, so a plan made for a short signal
        // is not handed to a long one with undersized blocks, nor the reverse
        int64_t exec_nfft = 0, exec_block = 0;
        if (mdsp_ols_geometry_for(nb, nfft, nx_hint, dtype, mode, engine, &exec_nfft, &exec_block, nullptr, nullptr, nullptr) == MDSP_OK) put(k, exec_nfft);
    }
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
        *entries = (int64_t)c.graveyard.size();
        for (auto& kv : c.parts) *entries += (int64_t)kv.second.size();
    }
    if (hits) *hits = c.hits;
    if (misses) *misses = c.misses;
    return MDSP_OK;
}

int mdsp_plan_cache_partitions(int64_t* partitions, int64_t* reaped) {
    Cache& c = cache();
    std::lock_guard<std::mutex> lk(c.mu);
    if (partitions) *partitions = (int64_t)c.parts.size();
    if (reaped) *reaped = c.reaped;
    return MDSP_OK;
}

int mdsp_plan_cache_set_context(uint64_t id) {
    Cache& c = cache();
    std::lock_guard<std::mutex> lk(c.mu);
    unbind(c, tl_context);
    tl_context = id & ~kCtxBit;
    if (tl_context) {
        ++c.parts[tl_context | kCtxBit].bound;
        (void)tl_sentinel.armed;   // instantiate the thread's sentinel: its destructor unbinds the context if the thread exits with it bound
    }
    return MDSP_OK;
}

int mdsp_plan_cache_release_context(uint64_t id) {
    if (id == 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "context 0 is the calling thread's own partition");
    Cache& c = cache();
    std::list<Entry> dead;
    {
        std::lock_guard<std::mutex> lk(c.mu);
        bury(c, (id & ~kCtxBit) | kCtxBit);
        dead.splice(dead.end(), c.graveyard);
    }
    for (auto& e : dead) e.destroy(e.handle);
    return MDSP_OK;
}

int mdsp_plan_cache_clear(void) {
    Cache& c = cache();
    std::list<Entry> all;
    {
        std::lock_guard<std::mutex> lk(c.mu);
        for (auto& kv : c.parts) {
            all.splice(all.end(), kv.second.user);
            all.splice(all.end(), kv.second.internal);
        }
        c.parts.clear();
        all.splice(all.end(), c.graveyard);
    }
    for (auto& e : all) e.destroy(e.handle);
    return MDSP_OK;
}

}  // extern "C"
