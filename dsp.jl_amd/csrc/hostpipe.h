// Chunked host <-> device pipeline shared by the host-array entry points (mdsp_ols_exec_host, mdsp_welch_exec_host, mdsp_stft_exec_host,
// mdsp_fir_exec_host): the reference's own call shape -- every DSP.jl entry takes host Arrays (Filters/filt.jl:458-476,
// periodograms.jl:647-744, :872-897, stream_filt.jl:688-775).
//
//     s_up   :  H2D(k)      H2D(k+1)      H2D(k+2)   ...            one stream per DIRECTION: chunk k's D2H runs while chunk k+1's H2D does
//     s_k    :       kernel(k)     kernel(k+1)    ...               (PCIe is full duplex; round 2 put a chunk's H2D, kernel and D2H on ONE stream
//     s_down :              D2H(k)        D2H(k+1)   ...             per lane and measured the half-duplex rate)
//
// Three lanes of device buffers (+ page-locked staging buffers for pageable arrays) rotate through the three stages; events order a
// lane's stages across the streams.  All kernels run in order on s_k, so plans whose work buffers or accumulators are shared between
// chunks (rocFFT engine, Welch sums, the polyphase filter state) need no extra care.
#pragma once

#include "common.h"

namespace mdsp {
namespace hostpipe {

constexpr int NL = 3;

struct PinBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~PinBuf() { release(); }
    void release();
    int reserve(size_t n);
};

struct Lane {
    hipEvent_t up = nullptr, kd = nullptr, down = nullptr;   // H2D finished / kernels finished / D2H finished
    DevBuf din, dout;
    PinBuf hin, hout;
    // output of the chunk in flight: rows to hand to the caller's array once `down` has fired (pageable arrays only)
    void* out_dst = nullptr;
    size_t out_pitch = 0, out_row = 0, out_rows = 0;
    bool busy = false, has_out = false;
};

struct Pipe;

// One use of the device's pipeline (RAII: holds the pipe's lock; host entry points are synchronous, concurrent callers serialise here).
class Session {
   public:
    Session(size_t in_cap, size_t out_cap, bool pinned);
    ~Session();
    int status() const { return rc_; }
    hipStream_t kstream() const;
    // next lane in rotation; waits for (and hands over the output of) the chunk that used it three chunks ago
    int acquire(Lane** ln);
    // rows x row_bytes from the caller's array (row pitch spitch) -> ln->din, dense rows; s_k will wait for it
    int upload(Lane* ln, const void* src, size_t spitch, size_t row_bytes, size_t rows);
    // after the chunk's kernels were enqueued on kstream(): (ln->dout + dev_off), rows x row_bytes with pitch dev_pitch -> dst (pitch dpitch)
    int download(Lane* ln, void* dst, size_t dpitch, size_t row_bytes, size_t rows, size_t dev_off, size_t dev_pitch);
    // a chunk without output (Welch): just mark the lane busy until its kernels have finished
    int no_download(Lane* ln);
    // drain everything; on error leave no work behind.  Returns the first error.
    int finish(int rc);

   private:
    Pipe* p_;
    bool pinned_;
    int rc_ = MDSP_OK;
    int64_t next_ = 0;
    int drain(Lane& ln);
};

void par_memcpy(void* dst, const void* src, size_t bytes);
// free every device's lanes (mdsp_shutdown)
void release_all();

}  // namespace hostpipe
}  // namespace mdsp
