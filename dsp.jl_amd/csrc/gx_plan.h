// What a Welch / STFT plan keeps for the run-time-schedule spectral kernel (spectral_gx.h): the schedule, the split nfft = R0 x S, the root tables
// and the window in working precision.  Built at the plan's first launch of that kernel.
#pragma once

#include "common.h"
#include "gx_sched.h"

namespace mdsp {
struct GxPlan {
    bool ready = false;
    gx::Sched sc;
    int R0 = 1, nhs = 0, nhn = 0;
    size_t lds_bytes = 0;
    DevBuf tw, win;
};
}  // namespace mdsp
