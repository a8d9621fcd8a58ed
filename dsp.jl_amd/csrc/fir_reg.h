// Register-tap polyphase kernel for every signal type (fir_reg.hip, round 6): where the matrix-core tile of a shape misses the LDS (ComplexF64 at L or M >= 147:
// 16 rows x 16 L / 16 ComplexF64 outputs next to the double-buffered input exceed 160 KiB) the filter used to fall to the generic kernel -- a tap fetched
// from L2 and a sample from LDS per multiply-add, 0.06 - 0.14 of the roofline (bench row resample_441_160_c64: 31.7 ms).  Internal to the library.
#pragma once

#include "common.h"

namespace mdsp {
struct FirRegArgs {
    const void* x;       // (xlen, nch), ld ldx, the signal's storage type
    const void* hist;    // (hl, nch)
    void* y;             // (ycap, nch), ld ldy, accumulate / output type
    const void* pfbT;    // tp * L taps (Float32 or Float64), pfbT[i * L + phi]
    int64_t xlen, ldx, ldy, nout, nrounds;
    int64_t phi0m1, d0;
    int L, M, tp, hl;
    int Q;               // rounds per tile
    int NP, RL;          // phase groups (threads along the residues), round lanes; blockDim.x = NP * RL
    int span;            // staged samples per tile
};
// true: a register-tap instantiation exists for (x_dtype, double arithmetic?, taps per phase, L, M)
bool fir_reg_ok(int x_dtype, bool acc_double, int64_t tp, int64_t L, int64_t M);
int fir_reg_run(int x_dtype, bool acc_double, FirRegArgs& a, int64_t nch, hipStream_t st);
}  // namespace mdsp
