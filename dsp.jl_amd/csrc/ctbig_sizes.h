// The single-workgroup compile-time schedules of 2100 .. 16384 points (round 6): size tables shared by spectral_ctbig.hip (one workgroup per frame pair) and
// spectral_ctcols_big.hip (the same schedules as the rows of nfft = R0 x S).  CtSched flags: ct_sched.h.
#pragma once

// flags: 16 one LDS buffer, 512 group padding, 2048 table twiddles (+ MDSP_CTBIG_PREF: 8, the next unit's samples in registers through the passes)
#ifndef MDSP_CTBIG_PREF
#define MDSP_CTBIG_PREF 0   // (8: the next unit's samples in registers through the passes -- 130 - 270 spilled registers at these sizes)
#endif
#define MDSP_CTBIG_F (16 | 512 | 2048 | MDSP_CTBIG_PREF)
// ... | 4096: the LEAN form (no window registers, Float32 sums flushed every 64 units): what lets 16 points per thread at up to 1024 threads fit 128 registers
#define MDSP_CTBIG_L (16 | 512 | 2048 | 4096 | 8192)
#define MDSP_CTBIG_A (16 | 512 | 2048 | 8192)   // Float32 sums, window in registers
// ... | 16384: a butterfly's twiddles as products of ~2 sqrt(R) table values (ct_apply_twiddles): per size, where it measured faster (all but 7 of 192 sizes)
#define MDSP_CTBIG_FD (MDSP_CTBIG_F | 16384)
#define MDSP_CTBIG_LD (MDSP_CTBIG_L | 16384)
// ... | 32768: the next unit's samples touched into the L2 behind this unit's first pass (per size: +4 % median, 12 500 1.45 -> 1.57 TB/s, 16 384 1.65 -> 1.78; r06s39)
#define MDSP_CTBIG_LT (MDSP_CTBIG_L | 32768)
#define MDSP_CTBIG_FT (MDSP_CTBIG_F | 32768)
#define MDSP_CTBIG_FDT (MDSP_CTBIG_FD | 32768)
#define MDSP_CTBIG_AT (MDSP_CTBIG_A | 32768)
#define MDSP_CTBIG_ADT (MDSP_CTBIG_AD | 32768)   // (register forms: r06s42, +8 % median over the sizes 1000 .. 9600, adopted at 97 of 129)
#define MDSP_CTBIG_LDT (MDSP_CTBIG_LD | 32768)
#define MDSP_CTBIG_AD (MDSP_CTBIG_A | 16384)
#ifdef MDSP_CTBIG_LIST_H   // (schedule A/B sessions: a header with other MDSP_CTBIG_SIZES / MDSP_CTBIG_LEAN_SIZES lists)
#include MDSP_CTBIG_LIST_H
#endif
// The table of 8193 .. 16384 points: per size the fastest of up to twelve measured schedules -- three passes at 512 threads (every ordering of the factor
// triple) or four passes at 640 .. 1024 threads, lean or register form (tools/sessions/r06_ctbig_pick.py; profiles/r06_ctbig_lean.json holds every
// measurement).  Sizes with a factor of 49 or 7^3 and no triple / quadruple of supported radices (8575, 12005, 12250, 14406, ...) stay R0 x S.
#ifndef MDSP_CTBIG_LEAN_SIZES
#define MDSP_CTBIG_LEAN_SIZES(X) \
    X(8232, 768, MDSP_CTBIG_ADT, 14, 7, 7, 12) X(8505, 768, MDSP_CTBIG_ADT, 9, 7, 9, 15) X(8750, 512, MDSP_CTBIG_LD, 25, 14, 25) \
    X(8820, 512, MDSP_CTBIG_LDT, 21, 21, 20) X(8960, 768, MDSP_CTBIG_LDT, 16, 5, 7, 16) X(9216, 768, MDSP_CTBIG_ADT, 16, 4, 12, 12) \
    X(9261, 512, MDSP_CTBIG_LDT, 21, 21, 21) X(9375, 512, MDSP_CTBIG_LDT, 25, 15, 25) X(9408, 512, MDSP_CTBIG_LDT, 28, 16, 21) \
    X(9450, 512, MDSP_CTBIG_LD, 25, 21, 18) X(9604, 768, MDSP_CTBIG_ADT, 14, 7, 7, 14) X(9720, 512, MDSP_CTBIG_LDT, 18, 20, 27) \
    X(9800, 512, MDSP_CTBIG_LDT, 28, 25, 14) X(10000, 512, MDSP_CTBIG_LDT, 25, 20, 20) X(10080, 512, MDSP_CTBIG_LDT, 21, 20, 24) \
    X(10125, 512, MDSP_CTBIG_LD, 27, 25, 15) X(10206, 512, MDSP_CTBIG_LD, 18, 21, 27) X(10240, 512, MDSP_CTBIG_LD, 32, 16, 20) \
    X(10290, 768, MDSP_CTBIG_ADT, 15, 7, 7, 14) X(10368, 512, MDSP_CTBIG_LD, 27, 16, 24) X(10500, 512, MDSP_CTBIG_ADT, 20, 25, 21) \
    X(10584, 512, MDSP_CTBIG_ADT, 24, 21, 21) X(10752, 512, MDSP_CTBIG_LD, 28, 16, 24) X(10800, 512, MDSP_CTBIG_LDT, 25, 24, 18) \
    X(10935, 512, MDSP_CTBIG_LD, 27, 15, 27) X(10976, 512, MDSP_CTBIG_LDT, 14, 28, 28) X(11025, 768, MDSP_CTBIG_LDT, 7, 7, 15, 15) \
    X(11200, 512, MDSP_CTBIG_LDT, 25, 28, 16) X(11250, 512, MDSP_CTBIG_LDT, 25, 18, 25) X(11340, 512, MDSP_CTBIG_LD, 27, 28, 15) \
    X(11520, 512, MDSP_CTBIG_LDT, 20, 24, 24) X(11664, 512, MDSP_CTBIG_LDT, 27, 24, 18) X(11760, 512, MDSP_CTBIG_LDT, 28, 28, 15) \
    X(11907, 512, MDSP_CTBIG_LDT, 21, 21, 27) X(12000, 512, MDSP_CTBIG_LD, 20, 24, 25) X(12096, 512, MDSP_CTBIG_LDT, 21, 24, 24) \
    X(12150, 512, MDSP_CTBIG_LD, 18, 25, 27) X(12288, 512, MDSP_CTBIG_LDT, 32, 16, 24) X(12348, 896, MDSP_CTBIG_LDT, 9, 14, 14, 7) \
    X(12500, 512, MDSP_CTBIG_LDT, 25, 20, 25) X(12544, 512, MDSP_CTBIG_LDT, 16, 28, 28) X(12600, 512, MDSP_CTBIG_LDT, 25, 28, 18) \
    X(12800, 512, MDSP_CTBIG_LDT, 32, 16, 25) X(12960, 512, MDSP_CTBIG_LDT, 27, 30, 16) X(13122, 512, MDSP_CTBIG_LDT, 18, 27, 27) \
    X(13125, 512, MDSP_CTBIG_LD, 25, 25, 21) X(13230, 1024, MDSP_CTBIG_LDT, 9, 14, 15, 7) X(13440, 512, MDSP_CTBIG_LDT, 30, 28, 16) \
    X(13500, 512, MDSP_CTBIG_LDT, 15, 30, 30) X(13608, 512, MDSP_CTBIG_LDT, 27, 28, 18) X(13720, 1024, MDSP_CTBIG_LD, 14, 14, 14, 5) \
    X(13824, 512, MDSP_CTBIG_LDT, 32, 16, 27) X(14000, 768, MDSP_CTBIG_LDT, 14, 10, 10, 10) X(14112, 512, MDSP_CTBIG_LDT, 28, 28, 18) \
    X(14175, 1024, MDSP_CTBIG_LD, 15, 15, 9, 7) X(14336, 512, MDSP_CTBIG_LDT, 32, 16, 28) X(14400, 512, MDSP_CTBIG_LDT, 30, 30, 16) \
    X(14580, 768, MDSP_CTBIG_LDT, 15, 12, 9, 9) X(14700, 768, MDSP_CTBIG_LDT, 15, 14, 10, 7) X(15000, 768, MDSP_CTBIG_LDT, 15, 10, 10, 10) \
    X(15120, 768, MDSP_CTBIG_LD, 14, 9, 10, 12) X(15360, 512, MDSP_CTBIG_LDT, 16, 32, 30) X(15552, 768, MDSP_CTBIG_LDT, 9, 12, 12, 12) \
    X(15625, 640, MDSP_CTBIG_LD, 25, 25, 25) X(15680, 768, MDSP_CTBIG_LDT, 14, 14, 10, 8) X(15750, 640, MDSP_CTBIG_LD, 14, 15, 15, 5) \
    X(15876, 768, MDSP_CTBIG_LDT, 14, 14, 9, 9) X(16000, 1024, MDSP_CTBIG_LD, 10, 10, 16, 10) X(16128, 768, MDSP_CTBIG_LDT, 12, 12, 14, 8) \
    X(16200, 768, MDSP_CTBIG_L, 12, 15, 15, 6) X(16384, 512, MDSP_CTBIG_LDT, 32, 32, 16)
#endif
#ifndef MDSP_CTBIG_SIZES
#define MDSP_CTBIG_SIZES(X) \
    X(8400, 512, MDSP_CTBIG_FDT, 20, 20, 21) X(8640, 512, MDSP_CTBIG_FDT, 18, 20, 24) X(8748, 512, MDSP_CTBIG_FDT, 27, 18, 18) \
    X(9000, 512, MDSP_CTBIG_FDT, 18, 20, 25) X(9072, 512, MDSP_CTBIG_FDT, 28, 18, 18) X(9600, 512, MDSP_CTBIG_FDT, 20, 20, 24)
#endif

// Sizes that ALSO have a schedule in ct_sched.h's tables (every mode) but whose Welch sums measured faster on a three-pass schedule of this file with derived
// twiddles (profiles/r06_ctbig_lean.json "nextfastfft_23": 2560 1.39 -> 1.78 TB/s, 3072 1.62 -> 1.97, 5120 1.22 -> 1.46, 1280 / 1600 / 3840 / 6400 +5 .. 10 %;
// the other sixteen sizes keep their round-4 schedules)
#ifndef MDSP_CTBIG_PREF_SIZES
#define MDSP_CTBIG_PREF_SIZES(X) \
    X(1280, 192, MDSP_CTBIG_FD, 20, 8, 8) X(1600, 192, MDSP_CTBIG_FD, 16, 10, 10) X(2560, 256, MDSP_CTBIG_FDT, 16, 10, 16) X(3072, 256, MDSP_CTBIG_FDT, 16, 12, 16) \
    X(3840, 256, MDSP_CTBIG_FDT, 16, 15, 16) X(5120, 320, MDSP_CTBIG_FD, 16, 16, 20) X(6400, 448, MDSP_CTBIG_FDT, 16, 16, 25)
#endif

// ... and the 7-smooth sizes from 2100 to 8192 points that have no schedule in ct_sched.h's table (which carries every mode for 23 sizes): generated -- the
// three factors out of {4 .. 30} and the thread count out of {256 .. 512} that give every thread at most ONE butterfly per pass with the fewest idle lanes
// (93 of the 109 sizes have such a triple; the others stay on the run-time schedule).  Welch sums only.
#define MDSP_CTBIG_SMALL_SIZES(X) \
    X(2100, 256, MDSP_CTBIG_FDT, 10, 10, 21) X(2160, 256, MDSP_CTBIG_FD, 9, 10, 24) X(2187, 256, MDSP_CTBIG_FD, 9, 9, 27) X(2205, 320, MDSP_CTBIG_FDT, 7, 15, 21) \
    X(2240, 256, MDSP_CTBIG_FDT, 10, 14, 16) X(2250, 256, MDSP_CTBIG_F, 9, 10, 25) X(2268, 256, MDSP_CTBIG_FD, 9, 9, 28) X(2304, 256, MDSP_CTBIG_FDT, 9, 16, 16) \
    X(2352, 256, MDSP_CTBIG_FDT, 12, 14, 14) X(2430, 320, MDSP_CTBIG_FDT, 9, 9, 30) X(2450, 384, MDSP_CTBIG_FT, 7, 14, 25) X(2520, 256, MDSP_CTBIG_FDT, 10, 12, 21) \
    X(2592, 256, MDSP_CTBIG_FDT, 12, 12, 18) X(2625, 384, MDSP_CTBIG_FT, 7, 15, 25) X(2646, 320, MDSP_CTBIG_FDT, 9, 14, 21) X(2688, 256, MDSP_CTBIG_FDT, 12, 14, 16) \
    X(2700, 256, MDSP_CTBIG_FDT, 12, 15, 15) X(2744, 256, MDSP_CTBIG_FDT, 14, 14, 14) X(2800, 320, MDSP_CTBIG_FDT, 10, 10, 28) X(2835, 320, MDSP_CTBIG_FDT, 9, 15, 21) \
    X(2880, 256, MDSP_CTBIG_FD, 12, 12, 20) X(2916, 384, MDSP_CTBIG_FDT, 9, 12, 27) X(2940, 256, MDSP_CTBIG_FD, 14, 14, 15) X(3024, 256, MDSP_CTBIG_FDT, 12, 12, 21) \
    X(3087, 448, MDSP_CTBIG_FDT, 7, 21, 21) X(3136, 256, MDSP_CTBIG_FD, 14, 14, 16) X(3150, 256, MDSP_CTBIG_FD, 14, 15, 15) X(3240, 320, MDSP_CTBIG_FDT, 12, 15, 18) \
    X(3360, 256, MDSP_CTBIG_FDT, 14, 15, 16) X(3375, 256, MDSP_CTBIG_FD, 15, 15, 15) X(3402, 384, MDSP_CTBIG_FD, 9, 14, 27) X(3456, 320, MDSP_CTBIG_FDT, 12, 12, 24) \
    X(3500, 384, MDSP_CTBIG_FDT, 10, 14, 25) X(3528, 256, MDSP_CTBIG_FDT, 14, 14, 18) X(3584, 256, MDSP_CTBIG_FD, 14, 16, 16) X(3600, 256, MDSP_CTBIG_FD, 15, 15, 16) \
    X(3645, 448, MDSP_CTBIG_FD, 9, 15, 27) X(3750, 384, MDSP_CTBIG_FDT, 10, 15, 25) X(3780, 320, MDSP_CTBIG_FDT, 12, 15, 21) X(3888, 384, MDSP_CTBIG_FDT, 12, 12, 27) \
    X(3920, 320, MDSP_CTBIG_FDT, 14, 14, 20) X(3969, 448, MDSP_CTBIG_FDT, 9, 21, 21) X(4032, 320, MDSP_CTBIG_FDT, 14, 16, 18) X(4050, 320, MDSP_CTBIG_FDT, 15, 15, 18) \
    X(4116, 320, MDSP_CTBIG_FDT, 14, 14, 21) X(4200, 320, MDSP_CTBIG_FDT, 14, 15, 20) X(4320, 320, MDSP_CTBIG_FDT, 15, 16, 18) X(4374, 512, MDSP_CTBIG_FDT, 9, 18, 27) \
    X(4410, 320, MDSP_CTBIG_FDT, 14, 15, 21) X(4480, 320, MDSP_CTBIG_FDT, 14, 16, 20) X(4500, 320, MDSP_CTBIG_FDT, 15, 15, 20) X(4536, 384, MDSP_CTBIG_FDT, 12, 14, 27) \
    X(4608, 320, MDSP_CTBIG_FDT, 16, 16, 18) X(4704, 384, MDSP_CTBIG_FDT, 14, 14, 24) X(4725, 320, MDSP_CTBIG_FDT, 15, 15, 21) X(4860, 384, MDSP_CTBIG_FDT, 15, 18, 18) \
    X(4900, 384, MDSP_CTBIG_FDT, 14, 14, 25) X(5040, 384, MDSP_CTBIG_FDT, 14, 15, 24) X(5184, 384, MDSP_CTBIG_FT, 16, 18, 18) X(5250, 384, MDSP_CTBIG_FDT, 14, 15, 25) \
    X(5292, 384, MDSP_CTBIG_FDT, 14, 14, 27) X(5376, 384, MDSP_CTBIG_FDT, 14, 16, 24) X(5400, 384, MDSP_CTBIG_FDT, 15, 15, 24) X(5488, 448, MDSP_CTBIG_FDT, 14, 14, 28) \
    X(5600, 448, MDSP_CTBIG_FDT, 14, 16, 25) X(5625, 384, MDSP_CTBIG_FDT, 15, 15, 25) X(5670, 384, MDSP_CTBIG_FT, 15, 18, 21) X(5760, 384, MDSP_CTBIG_FDT, 15, 16, 24) \
    X(5832, 384, MDSP_CTBIG_FT, 18, 18, 18) X(5880, 448, MDSP_CTBIG_FDT, 14, 14, 30) X(6048, 384, MDSP_CTBIG_FDT, 16, 18, 21) X(6075, 448, MDSP_CTBIG_FDT, 15, 15, 27) \
    X(6174, 448, MDSP_CTBIG_FDT, 14, 21, 21) X(6272, 448, MDSP_CTBIG_FDT, 14, 16, 28) X(6300, 448, MDSP_CTBIG_FDT, 15, 15, 28) X(6480, 384, MDSP_CTBIG_FDT, 18, 18, 20) \
    X(6615, 448, MDSP_CTBIG_FDT, 15, 21, 21) X(6720, 448, MDSP_CTBIG_FDT, 15, 16, 28) X(6750, 512, MDSP_CTBIG_FDT, 15, 15, 30) X(6804, 384, MDSP_CTBIG_FDT, 18, 18, 21) \
    X(6912, 448, MDSP_CTBIG_FDT, 16, 16, 27) X(7000, 512, MDSP_CTBIG_FDT, 14, 20, 25) X(7056, 448, MDSP_CTBIG_FDT, 16, 21, 21) X(7168, 448, MDSP_CTBIG_FDT, 16, 16, 28) \
    X(7200, 448, MDSP_CTBIG_FDT, 18, 20, 20) X(7290, 512, MDSP_CTBIG_FDT, 15, 18, 27) X(7500, 512, MDSP_CTBIG_FDT, 15, 20, 25) X(7560, 448, MDSP_CTBIG_FDT, 18, 20, 21) \
    X(7680, 512, MDSP_CTBIG_FDT, 16, 16, 30) X(7776, 448, MDSP_CTBIG_FDT, 18, 18, 24) X(7938, 448, MDSP_CTBIG_FDT, 18, 21, 21) X(8064, 512, MDSP_CTBIG_FDT, 16, 18, 28) \
    X(8100, 512, MDSP_CTBIG_FDT, 18, 18, 25)

