#include "rocfft_wrap.h"

namespace mdsp {

static std::mutex g_rocfft_mu;
static bool g_rocfft_up = false;

int rocfft_ensure_setup() {
    std::lock_guard<std::mutex> lk(g_rocfft_mu);
    if (!g_rocfft_up) {
        if (rocfft_setup() != rocfft_status_success) MDSP_FAIL(MDSP_ERR_DEVICE, "rocfft_setup failed");
        g_rocfft_up = true;
    }
    return MDSP_OK;
}

#define MDSP_ROCFFT(expr)                                                                              \
    do {                                                                                               \
        rocfft_status st_ = (expr);                                                                    \
        if (st_ != rocfft_status_success)                                                              \
            return set_error(MDSP_ERR_DEVICE, "%s failed with rocfft_status %d", #expr, (int)st_);     \
    } while (0)

void RocPlan::destroy() {
    if (info) (void)rocfft_execution_info_destroy(info);
    if (plan) (void)rocfft_plan_destroy(plan);
    info = nullptr;
    plan = nullptr;
    work.release();
}

int RocPlan::create(FftKind k, bool is_double, int64_t n_, int64_t batch_, bool inplace_) {
    destroy();
    MDSP_TRY(rocfft_ensure_setup());
    kind = k;
    n = n_;
    batch = batch_;
    inplace = inplace_;
    rocfft_transform_type tt;
    switch (k) {
        case FftKind::R2C: tt = rocfft_transform_type_real_forward; break;
        case FftKind::C2R: tt = rocfft_transform_type_real_inverse; break;
        case FftKind::C2C_FWD: tt = rocfft_transform_type_complex_forward; break;
        default: tt = rocfft_transform_type_complex_inverse; break;
    }
    const size_t len[1] = {(size_t)n};
    std::lock_guard<std::mutex> lk(g_rocfft_mu);  // plan creation may JIT kernels; serialise
    MDSP_ROCFFT(rocfft_plan_create(&plan, inplace ? rocfft_placement_inplace : rocfft_placement_notinplace, tt,
                                   is_double ? rocfft_precision_double : rocfft_precision_single, 1, len,
                                   (size_t)batch, nullptr));
    MDSP_ROCFFT(rocfft_execution_info_create(&info));
    size_t wb = 0;
    MDSP_ROCFFT(rocfft_plan_get_work_buffer_size(plan, &wb));
    if (wb) {
        MDSP_TRY(work.reserve(wb));
        MDSP_ROCFFT(rocfft_execution_info_set_work_buffer(info, work.p, wb));
    }
    return MDSP_OK;
}

int RocPlan::create_nd(FftKind k, bool is_double, int ndim, const int64_t* lens, bool inplace_) {
    destroy();
    if (ndim < 1 || ndim > 3) MDSP_FAIL(MDSP_ERR_UNSUPPORTED, "rocFFT plans have 1..3 dimensions");
    MDSP_TRY(rocfft_ensure_setup());
    kind = k;
    n = lens[0];
    batch = 1;
    inplace = inplace_;
    rocfft_transform_type tt;
    switch (k) {
        case FftKind::R2C: tt = rocfft_transform_type_real_forward; break;
        case FftKind::C2R: tt = rocfft_transform_type_real_inverse; break;
        case FftKind::C2C_FWD: tt = rocfft_transform_type_complex_forward; break;
        default: tt = rocfft_transform_type_complex_inverse; break;
    }
    size_t len[3] = {1, 1, 1};
    for (int d = 0; d < ndim; ++d) len[d] = (size_t)lens[d];
    std::lock_guard<std::mutex> lk(g_rocfft_mu);
    MDSP_ROCFFT(rocfft_plan_create(&plan, inplace ? rocfft_placement_inplace : rocfft_placement_notinplace, tt,
                                   is_double ? rocfft_precision_double : rocfft_precision_single, (size_t)ndim, len, 1, nullptr));
    MDSP_ROCFFT(rocfft_execution_info_create(&info));
    size_t wb = 0;
    MDSP_ROCFFT(rocfft_plan_get_work_buffer_size(plan, &wb));
    if (wb) {
        MDSP_TRY(work.reserve(wb));
        MDSP_ROCFFT(rocfft_execution_info_set_work_buffer(info, work.p, wb));
    }
    return MDSP_OK;
}

int RocPlan::exec(void* in, void* out, hipStream_t stream) {
    if (!plan) MDSP_FAIL(MDSP_ERR_DEVICE, "rocFFT plan not created");
    MDSP_ROCFFT(rocfft_execution_info_set_stream(info, (void*)stream));
    void* ib[1] = {in};
    void* ob[1] = {out};
    MDSP_ROCFFT(rocfft_execute(plan, ib, inplace ? nullptr : ob, info));
    return MDSP_OK;
}

}  // namespace mdsp
