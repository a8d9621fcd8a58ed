// Compile-time schedules of the mixed-radix spectral kernels (spectral_gen.h gen_ct_kernel): the schedule type with its index arithmetic, and the two
// size tables.  Plain C++ (no HIP): tests/cpu_harness/ct_layout.cpp runs every schedule's passes on the host with THESE formulas -- radices, twiddle
// indices, group padding, the single-buffer layout -- against a long-double DFT.
#pragma once

#include <cstdint>

template <int N_, int T_, int LDSIN_, int... RS> struct CtSched {
    static constexpr int N = N_, T = T_, P = (int)sizeof...(RS);
    static constexpr bool LDSIN = (LDSIN_ & 7) != 0;
    static constexpr bool PREFETCH = (LDSIN_ & 8) != 0;      // the next unit's samples are loaded behind the first pass and ride through the other passes in registers
    static constexpr bool INPLACE = (LDSIN_ & 16) != 0;      // ONE LDS buffer: a pass reads its operands, waits for every thread's reads, writes its results in place
                                                             // (a barrier more per pass, half the LDS: twice the workgroups per CU) -- register-consumed modes only
    static constexpr int MINW_WELCH = (LDSIN_ & 32) ? 2 : 1; // waves per SIMD the Welch kernel is compiled for (register cap 256)
    static constexpr int MINW_REAL = (LDSIN_ & 7) > 1 ? (LDSIN_ & 7) : 1;   // waves per SIMD the column (STFT) kernels are compiled for (register cap): a
                                                                 // workgroup of 6 waves puts two on some SIMDs, and two such workgroups need four there   // real-signal column modes window the frame pair into LDS first (the register-fed first pass costs them a resident workgroup)
    // Welch sums at 16 - 50 points per thread (round 6, spectral_ctbig.hip): 4096 the window is loaded beside the samples (spectral_gen.h ct_pass0_lean)
    // instead of living in registers; 8192 the sums are kept in the working precision and flushed to the Float64 partials every 64 units
    static constexpr bool LEANW = (LDSIN_ & 4096) != 0, LEANA = (LDSIN_ & 8192) != 0;
    static constexpr bool INPLACE_ALL = (LDSIN_ & 65536) != 0;   // with 16: the real-signal column modes run on the one buffer as well (spectral_ctbig_cols.hip: to 16384 points)
    static constexpr bool TOUCH = (LDSIN_ & 32768) != 0;     // with 4096: the next unit's samples touched (one dword per 128-byte line) behind this unit's first pass
    static constexpr int radix(int p) {
        constexpr int r[] = {RS...};
        return r[p];
    }
    static constexpr int ns(int p) {
        int v = 1;
        for (int i = 0; i < p; ++i) v *= radix(i);
        return v;
    }
    static constexpr int nbf(int p) { return N / radix(p); }
    static constexpr int M(int p) { return (nbf(p) + T - 1) / T; }
    static constexpr int ntw(int p) { return p == 0 ? 0 : M(p) * (radix(p) - 1); }
    static constexpr int twoff(int p) {
        int v = 0;
        for (int i = 0; i < p; ++i) v += ntw(i);
        return v;
    }
    // Round 5, flag 2048 (Float64 from 4800 points): NO register-resident twiddles -- W_N^m = hi[m >> 7] lo[m & 127] from two small LDS tables (the
    // first 128 roots and every 128th), one complex product more per twiddle.  The register form needs sum_p M(p) (R_p - 1) complex values: 60 at
    // 8000 = 5 5 5 8 8 -- 240 VGPRs of doubles next to the butterflies, the window and the sums, i.e. 100 - 184 scratch operations per kernel and HBM
    // traffic 1.8 (5000) to 6.9 (8000) times the algorithmic bytes (bench.py rows welch_f64_5000 / _8000, profiles/r05_f64_twiddles.json).
    static constexpr bool TW2L = (LDSIN_ & 2048) != 0;
    static constexpr bool TWD = (LDSIN_ & 16384) != 0;      // with TW2L: a butterfly's twiddles as products of ~2 sqrt(R) table values (spectral_gen.h ct_apply_twiddles)
    static constexpr int TWS = 128, NTWHI = (N + TWS - 1) / TWS;
    static constexpr int NTW = (!TW2L && twoff(P) > 0) ? twoff(P) : 1;
    static constexpr int BINS = (N + T - 1) / T;
    // Round 4: one element of padding behind every output GROUP of a pass (group = Ns R consecutive elements = the next pass's Ns) wherever the group's
    // byte stride aliases the LDS banks (G % 4 == 0: 16-lane groups of a ds_write_b64 that straddle two groups hit the same banks -- 28 % of the LDS
    // cycles of 3 x 8 x 8 x 8, 20 % of 5 x 24 x 25).  It costs nothing at either end: the scatter is  hi (G + 1) + k + Ns q,  and the next pass reads
    // j + nbf q with nbf a multiple of G, i.e.  (j + j / G) + (nbf + nbf / G) q  -- one per-thread constant and compile-time strides.  The last
    // pass's output (registers, or the natural-order spectrum the real-column modes read back) is never padded.  Flag 512 turns it on for a schedule
    // (1024: only in the register-consumed modes): measured on all 23 sizes it is worth +12 % / +5-9 % at 3072 and 6144 (Welch / columns), within
    // +-2 % elsewhere, and costs the real-column modes 8-20 % at 1536 / 2560 / 3072 (a resident workgroup, where the larger buffer crosses an LDS step).
#ifndef MDSP_GEN_CT_PAD
#define MDSP_GEN_CT_PAD 1
#endif
    static constexpr int G(int p) { return ns(p) * radix(p); }
    static constexpr bool PAD = (LDSIN_ & 512) != 0;         // per schedule: measured per size and mode (tools/sessions/r04_s30.sh)
    static constexpr bool padded(int p) { return MDSP_GEN_CT_PAD && PAD && p >= 0 && p < P - 1 && G(p) % 4 == 0; }
    static constexpr int gin(int p) { return p > 0 && padded(p - 1) ? G(p - 1) : 0; }               // padding group of the layout pass p READS (0: none)
    static constexpr int rstride(int p) { return gin(p) ? nbf(p) + nbf(p) / gin(p) : nbf(p); }      // distance of a butterfly's operands in that layout
    static constexpr int extra() {
        int e = 0;
        for (int p = 0; p < P; ++p)
            if (padded(p) && N / G(p) > e) e = N / G(p);
        return e;
    }
    static constexpr int NP = N + extra();   // elements per LDS buffer
    static_assert(ns(P) == N && T % 64 == 0, "the radices multiply to N; whole wavefronts");
};

// the sizes with a compile-time schedule (Float32 / ComplexF32): odd radix first, the widest radix last.  -DMDSP_GEN_CT=0 keeps the run-time kernel.
#ifndef MDSP_GEN_CT
#define MDSP_GEN_CT 1
#endif
// N -> schedule (odd radix first where N has one, the widest radix last; T ~ N / 8 threads so that a thread runs 1-4 butterflies per pass)
#define MDSP_GEN_CT_SIZES(X)                                                                                                      \
    X(1000, 128, 0, 5, 5, 5, 8) X(1200, 192, 32768, 3, 5, 5, 16) X(1280, 128, 0, 5, 16, 16) X(1500, 192, 0, 3, 5, 5, 5, 4)                \
    X(1536, 192, 32768, 3, 8, 8, 8) X(1600, 128, 0, 5, 5, 8, 8) X(1920, 128, 32768, 3, 5, 8, 16) X(2000, 256, 32768, 5, 5, 5, 16)                 \
    X(2400, 256, 32768, 3, 5, 5, 4, 8) X(2500, 256, 32768, 5, 5, 5, 5, 4) X(2560, 320, 0, 5, 8, 8, 8) X(3000, 384, 4, 3, 5, 5, 5, 8)          \
    X(3072, 256, 1536, 3, 16, 8, 8) X(3200, 256, 32768, 5, 5, 8, 16) X(3840, 256, 0, 3, 5, 16, 16) X(4000, 512, 32769, 5, 5, 5, 4, 8)            \
    X(4800, 512, 32768, 3, 5, 5, 8, 8) X(5000, 512, 32768, 5, 5, 5, 5, 8) X(5120, 320, 0, 5, 16, 8, 8) X(6000, 512, 0, 3, 5, 5, 5, 16)        \
    X(6144, 512, 33280, 3, 16, 16, 8) X(6400, 448, 0, 5, 5, 16, 16) X(8000, 512, 32768, 5, 5, 5, 8, 8)
// Round 4: Float32 schedules with composite radices (fft_lds.h bfly_comp: 6 ... 25 inside one thread's registers) -- THREE passes where the list
// above runs four or five, one or two LDS round trips and barriers less per transform; fewer, fatter threads (T ~ N / 24).  Taken where the
// butterfly counts N / R fill the lanes of T threads (>= 78 % in every pass); MDSP_GEN_WIDE=0 keeps the list above.
// Flags: 16 one LDS buffer, 32 Welch kernel compiled for two waves per SIMD, 64 / 128 / 256 NOT taken for Welch / complex columns / real columns
// (measured per mode, tools/bench_wide.py, profiles/r04_wide_schedules.json; 1000, 1600 and 8000 were tried and lost in every mode).
#define MDSP_GEN_CT_WIDE_SIZES(X)                                                                                               \
    X(1200, 64, 32816, 5, 12, 20) X(1500, 64, 304, 5, 12, 25) X(1920, 128, 32816, 15, 8, 16) X(2000, 128, 33072, 5, 16, 25)                 \
    X(2400, 128, 32816, 5, 20, 24) X(2500, 128, 33072, 25, 10, 10) X(3000, 128, 48, 5, 24, 25) X(3200, 128, 33136, 25, 8, 16)               \
    X(3840, 256, 48, 15, 16, 16) X(4800, 320, 32816, 15, 16, 20) X(5000, 256, 32816, 25, 10, 20) X(6000, 256, 304, 25, 24, 10)            \
    X(6400, 256, 368, 25, 16, 16)
constexpr int gen_ct_touch(int flags, int mode) { return mode == 0 ? flags : (flags & ~32768); }   // 32768 (the next unit touched into the L2): measured for Welch sums only (r06s42)
constexpr int gen_ct_flags(int flags, int mode, bool cplx) { return gen_ct_touch(((flags & 1024) && !(mode == 0 || cplx)) ? (flags & ~512) : flags, mode); }   // 1024: padding only where the last pass is consumed from registers
constexpr bool gen_ct_wide_mode(int flags, int mode, bool cplx) { return !(flags & (mode == 0 ? 64 : cplx ? 128 : 256)); }
// (The single LDS buffer was also tried on the small-radix list above -- flag 16 on all of it, tools/sessions/r04_s28: Welch -1 ... -5 %, ComplexF32 STFT
// -1 ... -4 %: those kernels are register-, not LDS-limited in residency, and the extra barrier per pass costs.)
// Float64 / ComplexF64 (DSP.jl's default element type): up to 3000 points two buffers of N x 16 bytes as in Float32; beyond (round 4), the
// register-consumed modes (Welch sums, complex columns) run on ONE buffer (CtSched flag 16, ct_passes_inplace): 8000 x 16 bytes = 125 KiB; real-signal
// columns need the natural-order spectrum in LDS next to the pass's input, i.e. two buffers, which fit up to 5000 points.
constexpr int GEN_CT_F64_TWO_BUF = 3000, GEN_CT_F64_MAX = 8000, GEN_CT_F64_REAL_COLUMNS_TWO_BUF = 5000, GEN_CT_F64_REAL_COLUMNS_MAX = 8000;   // (real columns above 5000 points: one buffer, round 6)
// direct: Welch sums or complex columns (the last pass is consumed from registers)
