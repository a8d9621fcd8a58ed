// Host-pointer entry points (mdsp_ols_exec_host, mdsp_welch_exec_host; mdsp_stft_exec_host and mdsp_fir_exec_host live next to their plans in
// spectral.hip / fir.hip): the reference's own call shape -- every DSP.jl entry takes host Arrays (Filters/filt.jl:458-476,
// periodograms.jl:647-744) -- on the chunked three-stage pipeline of hostpipe.h: one stream per PCIe direction plus one for the kernels, three
// lanes of buffers, so that chunk k's D2H, chunk k+1's kernels and chunk k+2's H2D are in flight together.
// Callers whose arrays are ALREADY page-locked (mdsp_host_alloc / mdsp_host_register) pass MDSP_HOST_PINNED and skip the staging
// copies: the DMA engines then read and write the caller's memory directly.
//
// These calls are PCIe-bound (~55 GB/s per direction against 4-5 TB/s for the kernels): they exist so that a drop-in caller can
// hand over host arrays at all, and bench.py reports their rate separately from the device-resident `value`.
#include <algorithm>
#include <memory>
#include <thread>

#include "common.h"
#include "hostpipe.h"
#include "ols_plan.h"
#include "welch_plan.h"

using namespace mdsp;

namespace mdsp {
namespace hostpipe {

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

namespace {
// rows x row bytes between pitched host arrays (dense on one side)
void par_memcpy2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t row, size_t rows) {
    if (rows == 0 || row == 0) return;
    if (dpitch == row && spitch == row) return par_memcpy(dst, src, row * rows);
    if (row >= (size_t(1) << 20) || rows < 8) {
        for (size_t r = 0; r < rows; ++r) par_memcpy((char*)dst + r * dpitch, (const char*)src + r * spitch, row);
        return;
    }
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t nt = std::min<size_t>({size_t(8), (size_t)hw, rows});
    std::vector<std::thread> th;
    for (size_t i = 1; i < nt; ++i)
        th.emplace_back([=] { for (size_t r = i; r < rows; r += nt) memcpy((char*)dst + r * dpitch, (const char*)src + r * spitch, row); });
    for (size_t r = 0; r < rows; r += nt) memcpy((char*)dst + r * dpitch, (const char*)src + r * spitch, row);
    for (auto& t : th) t.join();
}
}  // namespace

void PinBuf::release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    bytes = 0;
}
int PinBuf::reserve(size_t n) {
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

struct Pipe {
    Lane lane[NL];
    hipStream_t s_up = nullptr, s_k = nullptr, s_down = nullptr;
    std::mutex mu;
    int init() {
        if (!s_up) MDSP_HIP(hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking));
        if (!s_k) MDSP_HIP(hipStreamCreateWithFlags(&s_k, hipStreamNonBlocking));
        if (!s_down) MDSP_HIP(hipStreamCreateWithFlags(&s_down, hipStreamNonBlocking));
        for (Lane& ln : lane) {
            if (!ln.up) MDSP_HIP(hipEventCreateWithFlags(&ln.up, hipEventDisableTiming));
            if (!ln.kd) MDSP_HIP(hipEventCreateWithFlags(&ln.kd, hipEventDisableTiming));
            if (!ln.down) MDSP_HIP(hipEventCreateWithFlags(&ln.down, hipEventDisableTiming));
        }
        return MDSP_OK;
    }
    void release() {   // buffers only: streams and events are a few hundred bytes and stay for the next use
        for (Lane& ln : lane) {
            ln.din.release(); ln.dout.release(); ln.hin.release(); ln.hout.release();
        }
    }
};

namespace {
std::mutex g_mu;
Pipe* g_pipes[64] = {nullptr};   // never destroyed from a static destructor (after the HIP runtime); mdsp_shutdown releases the buffers

// one pipeline per device, created on first use
Pipe* pipe_for_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_pipes[dev]) g_pipes[dev] = new Pipe();
    return g_pipes[dev];
}
}  // namespace

void release_all() {
    std::lock_guard<std::mutex> lk(g_mu);
    for (Pipe* p : g_pipes)
        if (p) {
            std::lock_guard<std::mutex> lk2(p->mu);
            (void)hipStreamSynchronize(p->s_up);
            (void)hipStreamSynchronize(p->s_k);
            (void)hipStreamSynchronize(p->s_down);
            p->release();
        }
}

Session::Session(size_t in_cap, size_t out_cap, bool pinned) : p_(pipe_for_device()), pinned_(pinned) {
    p_->mu.lock();
    rc_ = p_->init();
    // The pipeline's streams are non-blocking: nothing orders them behind work the SAME plan / filter handle has queued on the caller's (or
    // torch's) stream -- a device-resident mdsp_fir_exec chunk whose shift-in kernel still runs while this call rewrites the history.
    // The host-array entries are synchronous calls that move tens of megabytes over PCIe: one device-wide synchronisation up front costs nothing
    // measurable and closes that window whatever stream the earlier work was queued on.
    if (rc_ == MDSP_OK && hipDeviceSynchronize() != hipSuccess) rc_ = set_error(MDSP_ERR_DEVICE, "hipDeviceSynchronize failed in front of a host-array call");
    for (Lane& ln : p_->lane) {
        ln.busy = ln.has_out = false;
        if (rc_ == MDSP_OK) rc_ = ln.din.reserve(in_cap);
        if (rc_ == MDSP_OK && out_cap) rc_ = ln.dout.reserve(out_cap);
        if (rc_ == MDSP_OK && !pinned) rc_ = ln.hin.reserve(in_cap);
        if (rc_ == MDSP_OK && !pinned && out_cap) rc_ = ln.hout.reserve(out_cap);
    }
}
Session::~Session() { p_->mu.unlock(); }
hipStream_t Session::kstream() const { return p_->s_k; }

int Session::drain(Lane& ln) {
    if (!ln.busy) return MDSP_OK;
    MDSP_HIP(hipEventSynchronize(ln.has_out ? ln.down : ln.kd));
    if (ln.has_out && !pinned_) par_memcpy2d(ln.out_dst, ln.out_pitch, ln.hout.p, ln.out_row, ln.out_row, ln.out_rows);
    ln.busy = ln.has_out = false;
    return MDSP_OK;
}

int Session::acquire(Lane** out) {
    Lane& ln = p_->lane[next_++ % NL];
    MDSP_TRY(drain(ln));
    *out = &ln;
    return MDSP_OK;
}

int Session::upload(Lane* ln, const void* src, size_t spitch, size_t row, size_t rows) {
    if (row * rows) {
        if (!pinned_) {
            par_memcpy2d(ln->hin.p, row, src, spitch, row, rows);
            src = ln->hin.p;
            spitch = row;
        }
        if (spitch == row || rows == 1) MDSP_HIP(hipMemcpyAsync(ln->din.p, src, row * rows, hipMemcpyHostToDevice, p_->s_up));
        else MDSP_HIP(hipMemcpy2DAsync(ln->din.p, row, src, spitch, row, rows, hipMemcpyHostToDevice, p_->s_up));
    }
    MDSP_HIP(hipEventRecord(ln->up, p_->s_up));
    MDSP_HIP(hipStreamWaitEvent(p_->s_k, ln->up, 0));
    ln->busy = true;   // from here on the lane has work in flight (finish() synchronises the streams on error)
    return MDSP_OK;
}

int Session::download(Lane* ln, void* dst, size_t dpitch, size_t row, size_t rows, size_t dev_off, size_t dev_pitch) {
    MDSP_HIP(hipEventRecord(ln->kd, p_->s_k));
    MDSP_HIP(hipStreamWaitEvent(p_->s_down, ln->kd, 0));
    ln->busy = true;
    if (row * rows) {
        const char* dsrc = static_cast<const char*>(ln->dout.p) + dev_off;
        void* hdst = pinned_ ? dst : ln->hout.p;
        const size_t hpitch = pinned_ ? dpitch : row;
        if ((hpitch == row && dev_pitch == row) || rows == 1) MDSP_HIP(hipMemcpyAsync(hdst, dsrc, row * rows, hipMemcpyDeviceToHost, p_->s_down));
        else MDSP_HIP(hipMemcpy2DAsync(hdst, hpitch, dsrc, dev_pitch, row, rows, hipMemcpyDeviceToHost, p_->s_down));
    }
    MDSP_HIP(hipEventRecord(ln->down, p_->s_down));
    ln->has_out = true;
    ln->out_dst = dst;
    ln->out_pitch = dpitch;
    ln->out_row = row;
    ln->out_rows = rows;
    return MDSP_OK;
}

int Session::no_download(Lane* ln) {
    MDSP_HIP(hipEventRecord(ln->kd, p_->s_k));
    ln->busy = true;
    ln->has_out = false;
    return MDSP_OK;
}

int Session::finish(int rc) {
    if (rc == MDSP_OK) {
        for (int i = 0; i < NL; ++i) {   // oldest first
            const int r2 = drain(p_->lane[(next_ + i) % NL]);
            if (rc == MDSP_OK) rc = r2;
        }
    }
    if (rc != MDSP_OK) {   // leave no work behind on the streams
        (void)hipStreamSynchronize(p_->s_up);
        (void)hipStreamSynchronize(p_->s_k);
        (void)hipStreamSynchronize(p_->s_down);
        for (Lane& ln : p_->lane) ln.busy = ln.has_out = false;
    }
    return rc;
}

}  // namespace hostpipe
}  // namespace mdsp

extern "C" {

int mdsp_host_pipeline_trim(void) {
    hostpipe::release_all();
    return MDSP_OK;
}

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
    if (nout > nx + nb - 1) MDSP_FAIL(MDSP_ERR_ARGUMENT, "nout (%lld) exceeds nx+nb-1", (long long)nout);
    if (ncols > 1 && (ldx < nx || ldy < nout)) MDSP_FAIL(MDSP_ERR_ARGUMENT, "leading dimension smaller than the column length");
    if (nout == 0 || ncols == 0) return MDSP_OK;
    if (!x_host && nx > 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "x is NULL");
    if (!y_host) MDSP_FAIL(MDSP_ERR_ARGUMENT, "out is NULL");
    const bool pinned = (flags & MDSP_HOST_PINNED) != 0;
    const size_t esz = dtype_size(dtype);
    const int64_t nblocks = cdiv(nout, L);
    // blocks per chunk: ~host_chunk_mib of input, even, at least 2
    int64_t bpc = std::max<int64_t>(4, ((int64_t)tunables().host_chunk_mib << 20) / (int64_t)(L * (int64_t)esz));
    bpc &= ~int64_t(3);   // a multiple of four blocks: the hand-allocated kernel's units (mdsp_ols_w64_asm) start at multiples of four, so a chunk and the
                          // whole column run every block through the same kernel and the results stay bit-identical
    const size_t in_cap = (size_t)(bpc * L + nb - 1) * esz, out_cap = (size_t)(bpc * L) * esz;

    hostpipe::Session ss(in_cap, out_cap, pinned);
    int rc = ss.status();
    for (int64_t col = 0; col < ncols && rc == MDSP_OK; ++col) {
        const char* xc = static_cast<const char*>(x_host) + (size_t)(col * ldx) * esz;
        char* yc = static_cast<char*>(y_host) + (size_t)(col * ldy) * esz;
        for (int64_t g0 = 0; g0 < nblocks && rc == MDSP_OK; g0 += bpc) {
            hostpipe::Lane* ln = nullptr;
            if ((rc = ss.acquire(&ln)) != MDSP_OK) break;
            const int64_t g1 = std::min(nblocks, g0 + bpc);
            const int64_t lo = std::max<int64_t>(0, g0 * L - (nb - 1)), hi = std::min(nx, g1 * L);
            const int64_t o0 = g0 * L, o1 = std::min(nout, g1 * L);
            const size_t inb = hi > lo ? (size_t)(hi - lo) * esz : 0, outb = (size_t)(o1 - o0) * esz;
            if ((rc = ss.upload(ln, xc + (size_t)lo * esz, inb, inb, 1)) != MDSP_OK) break;
            rc = mdsp_ols_exec_range(plan, ln->din.p, lo, hi > lo ? hi - lo : 0, nx, ln->dout.p, g0, g1 - g0, nout, ss.kstream());
            if (rc != MDSP_OK) break;
            rc = ss.download(ln, yc + (size_t)o0 * esz, outb, outb, 1, 0, outb);
        }
    }
    return ss.finish(rc);
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
    const size_t in_cap = (size_t)cl_max * (size_t)nch * esz, psd_bytes = (size_t)nch * (size_t)plan->nout * rsz;

    hostpipe::Session ss(in_cap, psd_bytes, pinned);
    int rc = ss.status();
    if (rc == MDSP_OK) rc = mdsp_welch_reset(plan);
    hostpipe::Lane* ln = nullptr;
    // the accumulators live in the plan: the chunks' kernels run in order on the pipeline's kernel stream while the next chunks upload
    for (int64_t k0 = 0; k0 < K && rc == MDSP_OK; k0 += fpc) {
        if ((rc = ss.acquire(&ln)) != MDSP_OK) break;
        const int64_t k1 = std::min(K, k0 + fpc), cl = (k1 - k0 - 1) * hop + n;
        // the channel slices land as contiguous rows of cl samples
        if ((rc = ss.upload(ln, static_cast<const char*>(s_host) + (size_t)(k0 * hop) * esz, (size_t)lds_host * esz, (size_t)cl * esz, (size_t)nch)) != MDSP_OK) break;
        if ((rc = mdsp_welch_accumulate(plan, ln->din.p, cl, nch, cl, ss.kstream())) != MDSP_OK) break;
        rc = ss.no_download(ln);
    }
    if (rc == MDSP_OK) rc = ss.acquire(&ln);
    if (rc == MDSP_OK) rc = ss.upload(ln, nullptr, 0, 0, 0);
    if (rc == MDSP_OK && K == 0) rc = mdsp_welch_accumulate(plan, ln->din.p, 0, nch, 0, ss.kstream());   // no frames: zero PSD (fill!(out, 0))
    if (rc == MDSP_OK) rc = mdsp_welch_finalize(plan, 0, ln->dout.p, plan->nout, ss.kstream());
    if (rc == MDSP_OK) rc = ss.download(ln, psd_host, (size_t)ldp * rsz, (size_t)plan->nout * rsz, (size_t)nch, 0, (size_t)plan->nout * rsz);
    return ss.finish(rc);
}

}  // extern "C"
