// The overlap-save plan object (opaque behind the C ABI), shared by ols.hip (kernels) and hostpath.hip (host-pointer pipeline).
#pragma once

#include "common.h"
#include "rocfft_wrap.h"

struct mdsp_ols_plan_s {
    int dtype = MDSP_F32, mode = MDSP_OLS_FILT, engine = MDSP_ENGINE_ROCFFT;
    int64_t nb = 0, nfft = 0, L = 0;
    mdsp::DevBuf H;       // rocFFT engine: nspec (real) or nfft (complex) entries; fused: nfft entries
    mdsp::DevBuf table;   // fused: nfft forward roots
    // rocFFT engine state
    mdsp::RocPlan fwd, inv;
    mdsp::DevBuf td, fd;
    int64_t batch = 0;
    int variant = 0;  // fused kernel variant (tuning knob, MDSP_OLS_VARIANT)
};
