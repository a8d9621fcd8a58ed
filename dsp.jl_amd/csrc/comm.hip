// Multi-GPU: one process per GPU, one RCCL communicator per process, behind the C ABI (mdsp_comm_*).
//
// The path shards by channel with NO data-path collective (Filters/filt.jl:504 column loop; stream_filt.jl:768 mapslices;
// welch_pgram / stft are per-vector).  The only exchange is the all-reduce (sum) of nout values for the cross-channel Welch
// mean -- 8 KiB for nfft = 4096: latency-bound, one ring step per xGMI link, bandwidth irrelevant -- and, when ONE
// stream is split along time over ranks, of the Float64 |X|^2 sums a Welch plan accumulates (mdsp_welch_allreduce).
//
// librccl is bound at run time (dlopen of its SONAME: inside a process that already loaded RCCL -- PyTorch's "nccl" backend IS
// RCCL -- this resolves to that copy, so there is one RCCL per process; in a plain Julia / C host it is /opt/rocm/lib/librccl.so.1).
// The library itself therefore loads on hosts without RCCL; only mdsp_comm_* needs it and says so when it is missing.
#include <dlfcn.h>

#include <rccl/rccl.h>

#include "common.h"
#include "welch_plan.h"

using namespace mdsp;

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            r.why = std::string("librccl.so.1 not found: ") + (dlerror() ? dlerror() : "?");
            return;
        }
        auto sym = [&](const char* name) {
            void* p = dlsym(r.handle, name);
            if (!p && r.why.empty()) r.why = std::string("librccl lacks ") + name;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    });
    return r;
}

int need_rccl() {
    Rccl& r = rccl();
    if (!r.why.empty()) MDSP_FAIL(MDSP_ERR_DEVICE, "RCCL unavailable: %s", r.why.c_str());
    return MDSP_OK;
}

#define MDSP_NCCL(expr)                                                                                                 \
    do {                                                                                                                \
        ncclResult_t mdsp_r_ = (expr);                                                                                  \
        if (mdsp_r_ != ncclSuccess)                                                                                     \
            return ::mdsp::set_error(MDSP_ERR_DEVICE, "%s failed: %s", #expr, rccl().GetErrorString(mdsp_r_));          \
    } while (0)

__global__ void set_scalar_kernel(double* __restrict__ p, double v) { *p = v; }

template <typename R> __global__ __launch_bounds__(256) void scale_kernel(R* __restrict__ v, int64_t n, double f) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) v[j] = (R)((double)v[j] * f);
}

}  // namespace

struct mdsp_comm_s {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1, device = 0;
    DevBuf scratch;   // a few words for reductions of scalars (frame counts)
};

extern "C" {

int mdsp_comm_unique_id(void* id128) {
    if (!id128) MDSP_FAIL(MDSP_ERR_ARGUMENT, "id buffer is NULL");
    MDSP_TRY(need_rccl());
    static_assert(sizeof(ncclUniqueId) == MDSP_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    MDSP_NCCL(rccl().GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return MDSP_OK;
}

int mdsp_comm_init_rank(mdsp_comm* comm, const void* id128, int rank, int nranks) {
    if (!comm) MDSP_FAIL(MDSP_ERR_ARGUMENT, "comm is NULL");
    *comm = nullptr;
    if (!id128) MDSP_FAIL(MDSP_ERR_ARGUMENT, "id buffer is NULL");
    if (nranks < 1 || rank < 0 || rank >= nranks) MDSP_FAIL(MDSP_ERR_ARGUMENT, "rank %d out of range [0,%d)", rank, nranks);
    MDSP_TRY(need_rccl());
    int dev = 0;
    MDSP_HIP(hipGetDevice(&dev));       // one process per GPU: the device selected by mdsp_init()
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    auto c = new mdsp_comm_s();
    c->rank = rank;
    c->nranks = nranks;
    c->device = dev;
    ncclResult_t r = rccl().CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        delete c;
        MDSP_FAIL(MDSP_ERR_DEVICE, "ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, rccl().GetErrorString(r));
    }
    *comm = c;
    return MDSP_OK;
}

int mdsp_comm_destroy(mdsp_comm comm) {
    if (!comm) return MDSP_OK;
    if (comm->comm) (void)rccl().CommDestroy(comm->comm);
    delete comm;
    return MDSP_OK;
}

int mdsp_comm_info(mdsp_comm comm, int* rank, int* nranks) {
    if (!comm) MDSP_FAIL(MDSP_ERR_ARGUMENT, "comm is NULL");
    if (rank) *rank = comm->rank;
    if (nranks) *nranks = comm->nranks;
    return MDSP_OK;
}

int mdsp_allreduce_sum(mdsp_comm comm, void* buf_dev, int64_t count, int real_dtype, void* stream) {
    if (!comm) MDSP_FAIL(MDSP_ERR_ARGUMENT, "comm is NULL");
    if (real_dtype != MDSP_F32 && real_dtype != MDSP_F64) MDSP_FAIL(MDSP_ERR_ARGUMENT, "real dtype expected");
    if (count < 0) MDSP_FAIL(MDSP_ERR_ARGUMENT, "negative count");
    if (count == 0) return MDSP_OK;
    if (!buf_dev) MDSP_FAIL(MDSP_ERR_ARGUMENT, "buffer is NULL");
    MDSP_NCCL(rccl().AllReduce(buf_dev, buf_dev, (size_t)count, real_dtype == MDSP_F32 ? ncclFloat32 : ncclFloat64, ncclSum, comm->comm,
                               as_stream(stream)));
    return MDSP_OK;
}

// Cross-channel Welch mean (the one collective of north_star): psd_dev holds this rank's per-channel PSDs
// (nch_local rows of ldp, the plan's nout bins each); mean_dev[j] = (1/nch_total) sum over ALL ranks' channels.
//   local sum on the device -> ncclAllReduce(sum) of nout values over xGMI -> scale.   comm == NULL: single rank.
int mdsp_welch_mean_allreduce(mdsp_welch_plan plan, const void* psd_dev, int64_t nch_local, int64_t ldp, int64_t nch_total, void* mean_dev,
                              mdsp_comm comm, void* stream) {
    if (!plan) MDSP_FAIL(MDSP_ERR_ARGUMENT, "plan is NULL");
    if (!mean_dev) MDSP_FAIL(MDSP_ERR_ARGUMENT, "mean is NULL");
    if (nch_local < 0 || nch_total < 1 || nch_local > nch_total) MDSP_FAIL(MDSP_ERR_ARGUMENT, "channel counts: local %lld of total %lld", (long long)nch_local, (long long)nch_total);
    if (nch_local > 0 && !psd_dev) MDSP_FAIL(MDSP_ERR_ARGUMENT, "psd is NULL");
    if (nch_local > 1 && ldp < plan->nout) MDSP_FAIL(MDSP_ERR_DIMENSION, "leading dimension smaller than the PSD length");
    const int T = dtype_real_of(plan->dtype);
    const int64_t nout = plan->nout;
    hipStream_t st = as_stream(stream);
    const bool collective = comm && comm->nranks > 1;
    if (!collective && nch_local > 0)   // one rank: sum and 1 / nch in ONE launch (the same roundings as sum-then-scale)
        return channel_sum_scaled(psd_dev, nout, nch_local, ldp, T, mean_dev, 1.0 / (double)nch_total, stream);
    if (nch_local == 0) MDSP_HIP(hipMemsetAsync(mean_dev, 0, dtype_size(T) * (size_t)nout, st));   // a rank without channels contributes zeros
    else MDSP_TRY(mdsp_channel_sum(psd_dev, nout, nch_local, ldp, T, mean_dev, stream));
    if (collective) MDSP_TRY(mdsp_allreduce_sum(comm, mean_dev, nout, T, stream));
    const dim3 g((unsigned)cdiv(nout, 256));
    if (T == MDSP_F32) hipLaunchKernelGGL(scale_kernel<float>, g, dim3(256), 0, st, (float*)mean_dev, nout, 1.0 / (double)nch_total);
    else hipLaunchKernelGGL(scale_kernel<double>, g, dim3(256), 0, st, (double*)mean_dev, nout, 1.0 / (double)nch_total);
    MDSP_LAUNCH_CHECK();
    return MDSP_OK;
}

// One stream split along TIME over ranks: every rank accumulated the frames of its slice (mdsp_welch_accumulate); the Float64 |X|^2 sums are
// added over ranks in place and the frame counts with them -- ONE RCCL group (the sums and one more double), the local count written by a
// kernel on the same stream, the total left on the device -- so that mdsp_welch_finalize(plan, 0, ...) on every rank yields the PSD of the whole
// stream (periodograms.jl:746-759: sum over ALL frames / (K fs sum w^2)).  Stream-ordered: nothing is copied from the host stack, nothing
// synchronises (round 3 did both).
int mdsp_welch_allreduce(mdsp_welch_plan plan, mdsp_comm comm, void* stream) {
    if (!plan) MDSP_FAIL(MDSP_ERR_ARGUMENT, "plan is NULL");
    if (!comm || comm->nranks <= 1) return MDSP_OK;
    if (plan->sums_global)   // the sums already are totals over ranks: reducing them again would count every rank's frames nranks times while the count is re-seeded
        MDSP_FAIL(MDSP_ERR_ARGUMENT, "these sums have been all-reduced already (mdsp_welch_reset, then accumulate this rank's frames again)");
    void* acc = nullptr;
    int64_t count = 0;
    MDSP_TRY(mdsp_welch_accumulator(plan, &acc, &count));
    MDSP_TRY(plan->kdev.reserve(sizeof(double)));
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(set_scalar_kernel, dim3(1), dim3(1), 0, st, plan->kdev.as<double>(), (double)plan->acc_frames);   // exact below 2^53 frames
    MDSP_LAUNCH_CHECK();
    MDSP_NCCL(rccl().GroupStart());
    ncclResult_t r1 = rccl().AllReduce(acc, acc, (size_t)count, ncclFloat64, ncclSum, comm->comm, st);
    ncclResult_t r2 = rccl().AllReduce(plan->kdev.p, plan->kdev.p, 1, ncclFloat64, ncclSum, comm->comm, st);
    ncclResult_t r3 = rccl().GroupEnd();
    for (ncclResult_t r : {r1, r2, r3})
        if (r != ncclSuccess) MDSP_FAIL(MDSP_ERR_DEVICE, "ncclAllReduce (Welch sums + frame count) failed: %s", rccl().GetErrorString(r));
    plan->frames_on_device = true;
    plan->sums_global = true;
    plan->count_stream = st;   // mdsp_welch_frames_accumulated reads the count behind THIS stream's work
    return MDSP_OK;
}

}  // extern "C"
