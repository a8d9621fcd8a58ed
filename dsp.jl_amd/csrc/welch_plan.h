// The Welch plan object (opaque behind the C ABI), shared by spectral.hip (kernels), comm.hip (all-reduce of its accumulators)
// and hostpath.hip (host-pointer pipeline).
#pragma once

#include "common.h"
#include "rocfft_wrap.h"
#include "bigfft.h"
#include "gx_plan.h"
#include "spectral_ctcols.h"
#include "spectral_ctrows.h"

struct mdsp_welch_plan_s {
    int dtype = MDSP_F32, engine = MDSP_ENGINE_ROCFFT, onesided = 1;
    int64_t n = 0, noverlap = 0, nfft = 0, nout = 0;
    double r = 1;
    bool have_win = false;
    mdsp::DevBuf win, table, partial, reduced;
    mdsp::RocPlan fwd;
    mdsp::DevBuf fr, spec;
    int64_t batch = 0;
    int variant = 0;
    // Float64 sums of |X[k]|^2 over the frames accumulated since the last reset (mdsp_welch_reset / _accumulate / _finalize):
    //   fused engine : `reduced`, acc_nslices = 1, acc_nacc = nfft bins per channel (pair-packed full spectrum for real signals)
    //   rocFFT engine: `partial`, acc_nslices = 32 deterministic slices, acc_nacc = nspec bins per channel
    bool acc_fresh = true;       // nothing accumulated yet: the next accumulate overwrites instead of adding
    int64_t acc_nch = 0, acc_frames = 0;
    // after mdsp_welch_allreduce the frame count summed over ranks lives on the DEVICE (kdev, one double, reduced together with the sums: no host
    // round trip, no stream synchronisation); mdsp_welch_finalize(plan, 0, ...) reads it there.  acc_frames stays this rank's own count.
    mdsp::DevBuf kdev;
    mdsp::DevBuf redtmp;         // group sums of the two-step slice reduction (reduce_partials, spectral.hip)
    mdsp::big::EngineHolder big;   // nfft above the single-workgroup kernels: the multi-pass engine (bigfft.hip), built at the first accumulate
    mdsp::GxPlan gx;             // 7-smooth sizes without a compile-time schedule, up to 32 x 16384 points: the run-time-schedule kernel (spectral_gx.h)
    mdsp::DevBuf winr;           // lean compile-time schedules (CtSched flag 4096): the window in the working precision, nfft values, zero tail
    bool winr_ready = false;
    mdsp::CtRowsPlan ctrows;     // ... or R0 x S in two kernels from R0 = 5 (spectral_ctrows.hip)
    mdsp::CtColsPlan ctcols;     // ... and those that are R0 x a compile-time row schedule (spectral_ctcols.hip)
    mdsp::DevBuf w64prep;        // mdsp_welch_w64_asm: Float32 window pairs + per-lane twiddles (built at the plan's first launch of that kernel)
    bool frames_on_device = false;
    bool sums_global = false;    // the sums are totals over ranks (mdsp_welch_allreduce ran): only mdsp_welch_reset clears it -- accumulating into them
                                 // and reducing again would add the earlier global totals nranks times
    hipStream_t count_stream = nullptr;   // the stream mdsp_welch_allreduce queued the count's reduction on
    int acc_nslices = 1, acc_nacc = 0, acc_mode = 0;   // welch_finalize_kernel MODE
    double* acc_ptr() const { return engine == MDSP_ENGINE_ROCFFT ? partial.as<double>() : reduced.as<double>(); }
};
