// Host-pointer entry points (mdsp_ols_exec_host, mdsp_welch_exec_host): the reference's own call shape -- every DSP.jl entry takes
// host Arrays (Filters/filt.jl:458-476, periodograms.jl:647-744) -- as a chunked, double-buffered pipeline
//
//     lane 0:  [stage chunk 0] H2D  kernel  D2H [drain]      [stage chunk 2] H2D  kernel  D2H ...
//     lane 1:                  [stage chunk 1] H2D  kernel  D2H [drain]      [stage chunk 3] ...
//
// Two lanes, each with its own HIP stream, device buffers and page-locked staging buffers: while one lane's kernel and D2H run,
// the other lane's H2D is in flight (PCIe is full duplex) and the host thread stages / drains with a small team of memcpy threads.
// Callers whose arrays are ALREADY page-locked (mdsp_host_alloc / mdsp_host_register) pass MDSP_HOST_PINNED and skip the staging
// copies: the DMA engines then read and write the caller's memory directly.
//
// These calls are PCIe-bound (~50-60 GB/s per direction against 4-5 TB/s for the kernels): they exist so that a drop-in caller can
// hand over host arrays at all, and bench.py --host reports their rate separately from the device-resident `value`.
#include <algorithm>
#include <memory>
#include <thread>

#include "common.h"
#include "ols_plan.h"
#include "welch_plan.h"

using namespace mdsp;

namespace {

// memcpy split over a few threads (one thread moves ~10 GB/s; PCIe 5 x16 moves ~55 GB/s per direction)
void par_memcpy(void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t nt = bytes < (size_t(8) << 20) ? 1 : std::min<size_t>({size_t(8), (size_t)hw, bytes >> 22});
    if (nt <= 1) {
        memcpy(dst, src, bytes);
        return;
    }
    const size_t per = ((bytes / nt) + 4095) & ~size_t(4095);
    std::vector<std::thread> th;
    for (size_t i = 1; i < nt; ++i) {
        const size_t lo = i * per;
        if (lo >= bytes) break;
        const size_t n = std::min(per, bytes - lo);
        th.emplace_back([=] { memcpy((char*)dst + lo, (const char*)src + lo, n); });
    }
    memcpy(dst, src, std::min(per, bytes));
    for (auto& t : th) t.join();
}

struct PinBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~PinBuf() { release(); }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
    }
    int reserve(size_t n) {
        if (n <= bytes) return MDSP_OK;
        release();
        hipError_t e = hipHostMalloc(&p, n, hipHostMallocDefault);
        if (e != hipSuccess) {
            p = nullptr;
            return set_error(MDSP_ERR_NOMEM, "hipHostMalloc(%zu bytes) failed: %s", n, hipGetErrorString(e));
        }
        bytes = n;
        return MDSP_OK;
    }
};

struct Lane {
    hipStream_t s = nullptr;
    hipEvent_t done = nullptr, kdone = nullptr;   // chunk finished (after its D2H) / its kernels finished
    DevBuf din, dout;
    PinBuf hin, hout;
    // output of the chunk in flight on this lane: where it goes once `done` has fired
    void* out_dst = nullptr;
    size_t out_bytes = 0;
    bool busy = false;
    ~Lane() {
        if (done) (void)hipEventDestroy(done);
        if (kdone) (void)hipEventDestroy(kdone);
        if (s) (void)hipStreamDestroy(s);
    }
    int init() {
        if (!s) MDSP_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        if (!done) MDSP_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
        if (!kdone) MDSP_HIP(hipEventCreateWithFlags(&kdone, hipEventDisableTiming));
        return MDSP_OK;
    }
    // wait for the chunk in flight and hand its output to the caller's array
    int drain(bool pinned) {
        if (!busy) return MDSP_OK;
        MDSP_HIP(hipEventSynchronize(done));
        if (!pinned && out_bytes) par_memcpy(out_dst, hout.p, out_bytes);
        busy = false;
        out_bytes = 0;
        return MDSP_OK;
    }
};

struct Pipe {
    Lane lane[2];
    std::mutex mu;
    int init() {
        MDSP_TRY(lane[0].init());
        return lane[1].init();
    }
};

// one pipeline per device, created on first use (host entry points are synchronous; concurrent callers serialise here)
Pipe* pipe_for_device() {
    static std::mutex mu;
    static Pipe* pipes[64] = {nullptr};   // never destroyed: streams / pinned buffers must not be released from a static destructor, after the HIP runtime
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    std::lock_guard<std::mutex> lk(mu);
    if (!pipes[dev]) pipes[dev] = new Pipe();
    return pipes[dev];
}

}  // namespace

extern "C" {

int mdsp_host_alloc(void** host_ptr, size_t bytes) {
    if (!host_ptr) MDSP_FAIL(MDSP_ERR_ARGUMENT, "host_ptr is NULL");
    *host_ptr = nullptr;
    if (bytes == 0) return MDSP_OK;
    MDSP_HIP(hipHostMalloc(host_ptr, bytes, hipHostMallocDefault));
    return MDSP_OK;
}
int mdsp_host_free(void* host_ptr) {
    if (host_ptr) MDSP_HIP(hipHostFree(host_ptr));
    return MDSP_OK;
}
int mdsp_host_register(void* host_ptr, size_t bytes) {
    if (!host_ptr || bytes == 0) return MDSP_OK;
    MDSP_HIP(hipHostRegister(host_ptr, bytes, hipHostRegisterDefault));
    return MDSP_OK;
}
int mdsp_host_unregister(void* host_ptr) {
    if (host_ptr) MDSP_HIP(hipHostUnregister(host_ptr));
    return MDSP_OK;
}

// filt / conv of host arrays: columns one after the other, each in runs of whole overlap-save blocks (an even number of them: two
// real blocks share a transform).  Chunk c of a column computes blocks [g0, g1) of THE SAME block grid the device-resident call uses
// (mdsp_ols_exec_range), from the samples [g0 L - (nb-1), g1 L) -- so host and device calls return bit-identical results.
int mdsp_ols_exec_host(mdsp_ols_plan plan, const void* x_host, int64_t nx, int64_t ncols, int64_t ldx, void* y_host, int64_t nout, int64_t ldy,
                       int flags) {
    if (!plan) MDSP_FAIL(MDSP_ERR_ARGUMENT, "plan is NULL");
    if (nx < 0 || ncols < 0 || nout < 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "negative size");
    const int64_t L = plan->L, nb = plan->nb;
    const int dtype = plan->dtype;
    const bool serial_exec = plan->engine == MDSP_ENGINE_ROCFFT;   // that engine's transforms share the plan's work buffers: one at a time
    if (nout > nx + nb - 1) MDSP_FAIL(MDSP_ERR_ARGUMENT, "nout (%lld) exceeds nx+nb-1", (long long)nout);
    if (ncols > 1 && (ldx < nx || ldy < nout)) MDSP_FAIL(MDSP_ERR_ARGUMENT, "leading dimension smaller than the column length");
    if (nout == 0 || ncols == 0) return MDSP_OK;
    if (!x_host && nx > 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "x is NULL");
    if (!y_host) MDSP_FAIL(MDSP_ERR_ARGUMENT, "out is NULL");
    const bool pinned = (flags & MDSP_HOST_PINNED) != 0;
    const size_t esz = dtype_size(dtype);
    const int64_t nblocks = cdiv(nout, L);
    // blocks per chunk: ~host_chunk_mib of input, even, at least 2
    int64_t bpc = std::max<int64_t>(2, ((int64_t)tunables().host_chunk_mib << 20) / (int64_t)(L * (int64_t)esz));
    bpc &= ~int64_t(1);
    // Partitioned plans (long filters) have no block-range entry: a chunk is filtered as a signal of its own that starts nb-1 samples early
    // (zero initial state), and the nb-1 warm-up outputs are dropped -- equal to the device-resident call up to rounding, not bit for bit.
    const bool halo = plan->partitions > 1;
    if (halo) bpc = std::max<int64_t>(bpc, 4 * cdiv(nb, L));   // keep the re-filtered halo below a quarter of a chunk
    const size_t in_cap = (size_t)(bpc * L + nb - 1) * esz, out_cap = (size_t)(bpc * L + (halo ? nb - 1 : 0)) * esz;

    Pipe* pp = pipe_for_device();
    std::lock_guard<std::mutex> lk(pp->mu);
    MDSP_TRY(pp->init());
    for (Lane& ln : pp->lane) {
        MDSP_TRY(ln.din.reserve(in_cap));
        MDSP_TRY(ln.dout.reserve(out_cap));
        if (!pinned) {
            MDSP_TRY(ln.hin.reserve(in_cap));
            MDSP_TRY(ln.hout.reserve(out_cap));
        }
    }
    int rc = MDSP_OK;
    int64_t chunk = 0;
    for (int64_t col = 0; col < ncols && rc == MDSP_OK; ++col) {
        const char* xc = static_cast<const char*>(x_host) + (size_t)(col * ldx) * esz;
        char* yc = static_cast<char*>(y_host) + (size_t)(col * ldy) * esz;
        for (int64_t g0 = 0; g0 < nblocks && rc == MDSP_OK; g0 += bpc, ++chunk) {
            Lane& ln = pp->lane[chunk & 1];
            if ((rc = ln.drain(pinned)) != MDSP_OK) break;           // this lane's previous chunk (two chunks ago)
            const int64_t g1 = std::min(nblocks, g0 + bpc);
            const int64_t lo = std::max<int64_t>(0, g0 * L - (nb - 1)), hi = std::min(nx, g1 * L);
            const int64_t o0 = g0 * L, o1 = std::min(nout, g1 * L);
            const size_t inb = hi > lo ? (size_t)(hi - lo) * esz : 0, outb = (size_t)(o1 - o0) * esz;
            const void* src = xc + (size_t)lo * esz;
            if (inb) {
                if (!pinned) {
                    par_memcpy(ln.hin.p, src, inb);
                    src = ln.hin.p;
                }
                hipError_t e = hipMemcpyAsync(ln.din.p, src, inb, hipMemcpyHostToDevice, ln.s);
                if (e != hipSuccess) { rc = set_error(MDSP_ERR_DEVICE, "H2D copy failed: %s", hipGetErrorString(e)); break; }
            }
            if (serial_exec && chunk > 0) {
                hipError_t e = hipStreamWaitEvent(ln.s, pp->lane[(chunk & 1) ^ 1].kdone, 0);
                if (e != hipSuccess) { rc = set_error(MDSP_ERR_DEVICE, "stream wait failed: %s", hipGetErrorString(e)); break; }
            }
            if (halo) rc = mdsp_ols_exec(plan, ln.din.p, hi > lo ? hi - lo : 0, 1, hi > lo ? hi - lo : 0, ln.dout.p, o1 - lo, o1 - lo, ln.s);
            else rc = mdsp_ols_exec_range(plan, ln.din.p, lo, hi > lo ? hi - lo : 0, nx, ln.dout.p, g0, g1 - g0, nout, ln.s);
            if (rc != MDSP_OK) break;
            if (serial_exec) {
                hipError_t e = hipEventRecord(ln.kdone, ln.s);
                if (e != hipSuccess) { rc = set_error(MDSP_ERR_DEVICE, "event record failed: %s", hipGetErrorString(e)); break; }
            }
            void* dst = pinned ? (void*)(yc + (size_t)o0 * esz) : ln.hout.p;
            hipError_t e = hipMemcpyAsync(dst, static_cast<const char*>(ln.dout.p) + (halo ? (size_t)(o0 - lo) * esz : 0), outb, hipMemcpyDeviceToHost, ln.s);
            if (e == hipSuccess) e = hipEventRecord(ln.done, ln.s);
            if (e != hipSuccess) { rc = set_error(MDSP_ERR_DEVICE, "D2H copy failed: %s", hipGetErrorString(e)); break; }
            ln.out_dst = yc + (size_t)o0 * esz;
            ln.out_bytes = outb;
            ln.busy = true;
        }
    }
    for (Lane& ln : pp->lane) {
        const int r2 = ln.drain(pinned);
        if (rc == MDSP_OK) rc = r2;
    }
    if (rc != MDSP_OK) {   // leave no work behind on the lanes
        (void)hipStreamSynchronize(pp->lane[0].s);
        (void)hipStreamSynchronize(pp->lane[1].s);
        pp->lane[0].busy = pp->lane[1].busy = false;
    }
    return rc;
}

// welch_pgram of host arrays ((len, nch) column-major, ld lds_host): time chunks of whole frames for all channels at once; each
// chunk's frames are added to the plan's Float64 accumulators (mdsp_welch_accumulate), the PSD is formed once at the end from the
// total frame count (periodograms.jl:746-759) and copied back.
int mdsp_welch_exec_host(mdsp_welch_plan plan, const void* s_host, int64_t len, int64_t nch, int64_t lds_host, void* psd_host, int64_t ldp, int flags) {
    if (!plan) MDSP_FAIL(MDSP_ERR_ARGUMENT, "plan is NULL");
    if (len < 0 || nch < 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "negative size");
    if (nch == 0) return MDSP_OK;
    if (!psd_host) MDSP_FAIL(MDSP_ERR_ARGUMENT, "out is NULL");
    if (!s_host && len > 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "s is NULL");
    if (nch > 1 && (lds_host < len || ldp < plan->nout)) MDSP_FAIL(MDSP_ERR_DIMENSION, "leading dimension smaller than the column length");
    if (nch > 65535) MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "more than 65535 channels per call");
    const bool pinned = (flags & MDSP_HOST_PINNED) != 0;
    const size_t esz = dtype_size(plan->dtype), rsz = dtype_size(dtype_real_of(plan->dtype));
    const int64_t n = plan->n, hop = plan->n - plan->noverlap, K = mdsp_frame_count(len, n, plan->noverlap);
    // frames per chunk: ~host_chunk_mib over all channels
    const int64_t fpc = std::max<int64_t>(2, (((int64_t)tunables().host_chunk_mib << 20) / (int64_t)esz / nch - n) / hop + 1) & ~int64_t(1);
    const int64_t cl_max = (fpc - 1) * hop + n;                 // samples per channel per chunk
    const size_t in_cap = (size_t)cl_max * (size_t)nch * esz;

    Pipe* pp = pipe_for_device();
    std::lock_guard<std::mutex> lk(pp->mu);
    MDSP_TRY(pp->init());
    for (Lane& ln : pp->lane) {
        MDSP_TRY(ln.din.reserve(in_cap));
        if (!pinned) MDSP_TRY(ln.hin.reserve(in_cap));
    }
    MDSP_TRY(mdsp_welch_reset(plan));
    // The accumulators live in the plan, so the chunks' kernels must run in order on ONE stream (lane 0's); the copies of the next chunk
    // run on the other lane's stream and overlap them.
    hipStream_t ks = pp->lane[0].s;
    hipEvent_t kdone[2] = {pp->lane[0].kdone, pp->lane[1].kdone};   // kernel of the chunk that used buffer b has finished
    bool used[2] = {false, false};
    int rc = MDSP_OK;
    int64_t chunk = 0;
    for (int64_t k0 = 0; k0 < K && rc == MDSP_OK; k0 += fpc, ++chunk) {
        const int b = (int)(chunk & 1);
        Lane& ln = pp->lane[b];
        const int64_t k1 = std::min(K, k0 + fpc), cl = (k1 - k0 - 1) * hop + n;
        if (used[b]) MDSP_HIP(hipEventSynchronize(kdone[b]));   // the kernel that read this buffer two chunks ago
        const char* src = static_cast<const char*>(s_host) + (size_t)(k0 * hop) * esz;
        size_t spitch = (size_t)lds_host * esz;
        if (!pinned) {   // gather the channel slices into the staging buffer (contiguous rows of cl samples)
            for (int64_t c = 0; c < nch; ++c) par_memcpy((char*)ln.hin.p + (size_t)(c * cl) * esz, src + (size_t)c * spitch, (size_t)cl * esz);
            src = (const char*)ln.hin.p;
            spitch = (size_t)cl * esz;
        }
        hipStream_t cs = pp->lane[1].s;                         // copy stream
        hipError_t e = hipMemcpy2DAsync(ln.din.p, (size_t)cl * esz, src, spitch, (size_t)cl * esz, (size_t)nch, hipMemcpyHostToDevice, cs);
        hipEvent_t copied = nullptr;
        if (e == hipSuccess) e = hipEventCreateWithFlags(&copied, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(copied, cs);
        if (e == hipSuccess) e = hipStreamWaitEvent(ks, copied, 0);
        if (copied) (void)hipEventDestroy(copied);              // destruction is deferred until the event has completed
        if (e != hipSuccess) { rc = set_error(MDSP_ERR_DEVICE, "H2D copy failed: %s", hipGetErrorString(e)); break; }
        rc = mdsp_welch_accumulate(plan, ln.din.p, cl, nch, cl, ks);
        if (rc != MDSP_OK) break;
        e = hipEventRecord(kdone[b], ks);
        if (e != hipSuccess) { rc = set_error(MDSP_ERR_DEVICE, "event record failed: %s", hipGetErrorString(e)); break; }
        used[b] = true;
    }
    if (rc == MDSP_OK && K == 0) rc = mdsp_welch_accumulate(plan, pp->lane[0].din.p, 0, nch, 0, ks);   // no frames: zero PSD (fill!(out, 0))
    if (rc == MDSP_OK) {
        DevBuf& pd = pp->lane[0].dout;
        rc = pd.reserve((size_t)nch * (size_t)plan->nout * rsz);
        if (rc == MDSP_OK) rc = mdsp_welch_finalize(plan, 0, pd.p, plan->nout, ks);
        if (rc == MDSP_OK) {
            hipError_t e = hipMemcpy2DAsync(psd_host, (size_t)ldp * rsz, pd.p, (size_t)plan->nout * rsz, (size_t)plan->nout * rsz, (size_t)nch,
                                            hipMemcpyDeviceToHost, ks);
            if (e != hipSuccess) rc = set_error(MDSP_ERR_DEVICE, "D2H copy failed: %s", hipGetErrorString(e));
        }
    }
    (void)hipStreamSynchronize(pp->lane[1].s);
    hipError_t e = hipStreamSynchronize(ks);
    if (rc == MDSP_OK && e != hipSuccess) rc = set_error(MDSP_ERR_DEVICE, "stream synchronize failed: %s", hipGetErrorString(e));
    return rc;
}

}  // extern "C"
