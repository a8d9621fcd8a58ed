// The overlap-save plan object (opaque behind the C ABI), shared by ols.hip (kernels) and hostpath.hip (host-pointer pipeline).
#pragma once

#include "common.h"
#include "rocfft_wrap.h"
#include "bigfft.h"

struct mdsp_ols_plan_s {
    int dtype = MDSP_F32, mode = MDSP_OLS_FILT, engine = MDSP_ENGINE_ROCFFT;
    int64_t nb = 0, nfft = 0, L = 0;      // nfft / L: the geometry that EXECUTES (== the reference's unless the fused engine re-blocked a long filter)
    int64_t ref_nfft = 0, ref_L = 0;     // what the caller / optimalfftfiltlength asked for (plan_info, mdsp_ols_segment: the reference's tmp1 blocks)
    int partitions = 1;                  // > 1: uniformly partitioned overlap-save (upols_fused_kernel): nfft = 2 B, `partitions` spectra in H
    mdsp::DevBuf H;       // rocFFT engine: nspec (real) or nfft (complex) entries; fused: nfft entries
    mdsp::DevBuf table;   // fused: nfft forward roots
    // rocFFT engine state
    mdsp::RocPlan fwd, inv;
    mdsp::DevBuf td, fd;
    int64_t batch = 0;
    int variant = 0;  // fused kernel variant (tuning knob, MDSP_OLS_VARIANT)
    // filters beyond the partitioned kernels: blocks of nfft = 8 .. 16 nb points on the multi-pass engine (bigfft.hip), H = nfft entries, natural order
    bool big = false;
    int big_rows = 0;                    // > 0: H is row-major for the rows form of that engine (R0 rows: bigfft.h ols_rows_r0)
    mdsp::big::EngineHolder bigeng;
};
