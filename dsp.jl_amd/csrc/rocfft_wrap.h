// Thin RAII wrapper over rocFFT batched 1-D plans (the MDSP_ENGINE_ROCFFT transforms).
// Plans and their work buffers are created once and kept by the owning handle (SURVEY 8b "ownership").
#pragma once

#include <rocfft/rocfft.h>

#include "common.h"

namespace mdsp {

int rocfft_ensure_setup();

enum class FftKind { R2C, C2R, C2C_FWD, C2C_INV };

struct RocPlan {
    rocfft_plan plan = nullptr;
    rocfft_execution_info info = nullptr;
    DevBuf work;
    FftKind kind = FftKind::R2C;
    bool inplace = false;
    int64_t n = 0, batch = 0;

    RocPlan() = default;
    RocPlan(const RocPlan&) = delete;
    RocPlan& operator=(const RocPlan&) = delete;
    ~RocPlan() { destroy(); }
    void destroy();
    // Contiguous batches: real side distance n, hermitian side distance n/2+1, complex side distance n.
    int create(FftKind kind, bool is_double, int64_t n, int64_t batch, bool inplace);
    // One N-d transform (N <= 3), lens[0] fastest, default contiguous layouts (Hermitian side: lens[0]/2 + 1 along dimension 0).
    int create_nd(FftKind kind, bool is_double, int ndim, const int64_t* lens, bool inplace);
    int exec(void* in, void* out, hipStream_t stream);
};

}  // namespace mdsp
