// Internal helpers shared by the translation units of libmi355dsp.so (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_complex.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mi355dsp.h"

namespace mdsp {

// ---------------------------------------------------------------- error reporting (thread-local message)
int set_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
void clear_error();

#define MDSP_FAIL(code, ...) return ::mdsp::set_error((code), __VA_ARGS__)
#define MDSP_HIP(expr)                                                                                   \
    do {                                                                                                 \
        hipError_t mdsp_e_ = (expr);                                                                     \
        if (mdsp_e_ != hipSuccess)                                                                       \
            return ::mdsp::set_error(mdsp_e_ == hipErrorOutOfMemory ? MDSP_ERR_NOMEM : MDSP_ERR_DEVICE,  \
                                     "%s failed: %s (%s:%d)", #expr, hipGetErrorString(mdsp_e_),         \
                                     __FILE__, __LINE__);                                                \
    } while (0)
#define MDSP_TRY(expr)                 \
    do {                               \
        int mdsp_s_ = (expr);          \
        if (mdsp_s_ != MDSP_OK)        \
            return mdsp_s_;            \
    } while (0)
// kernel launch check: launch errors only (no sync)
#define MDSP_LAUNCH_CHECK() MDSP_HIP(hipGetLastError())

// ---------------------------------------------------------------- element types
inline bool dtype_valid(int dt) { return dt >= MDSP_F32 && dt <= MDSP_C64; }
inline bool dtype_is_complex(int dt) { return dt == MDSP_C32 || dt == MDSP_C64; }
inline bool dtype_is_double(int dt) { return dt == MDSP_F64 || dt == MDSP_C64; }
inline size_t dtype_size(int dt) {
    switch (dt) {
        case MDSP_F32: return 4;
        case MDSP_F64: return 8;
        case MDSP_C32: return 8;
        default: return 16;
    }
}
inline int dtype_real_of(int dt) { return dtype_is_double(dt) ? MDSP_F64 : MDSP_F32; }
inline int dtype_complex_of(int dt) { return dtype_is_double(dt) ? MDSP_C64 : MDSP_C32; }

template <typename R> struct cplx { R x, y; };
using cf32 = cplx<float>;
using cf64 = cplx<double>;

// ---------------------------------------------------------------- device buffer with RAII
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    // grow-only reservation
    int reserve(size_t n) {
        if (n <= bytes) return MDSP_OK;
        release();
        if (n == 0) return MDSP_OK;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) {
            p = nullptr;
            return set_error(MDSP_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", n, hipGetErrorString(e));
        }
        bytes = n;
        return MDSP_OK;
    }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

inline hipStream_t as_stream(void* s) { return static_cast<hipStream_t>(s); }

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// device properties (cached per device)
int device_cu_count();

// Transform slots per workgroup.  A transform of T > 64 threads synchronises its waves with s_barrier, and s_barrier spans the WORKGROUP:
// two 128-thread transforms in one 256-thread workgroup wait for each other at every exchange although they share nothing.  One transform
// per workgroup whenever barriers are real (T >= 128); several one-wave transforms (wave-level fences only) still share a workgroup.
// Measured on the headline overlap-save shape, 12 interleaved rounds on two boxes (profiles/r02l_ols_decoupled.json): 1.85 / 1.78 ms
// against 2.04 / 1.96 for the same kernel with two transforms per workgroup.  -DMDSP_COUPLED_SLOTS=1 restores the round-1 grouping.
#ifndef MDSP_COUPLED_SLOTS
#define MDSP_COUPLED_SLOTS 0
#endif
constexpr int slots_per_workgroup(int T) { return MDSP_COUPLED_SLOTS ? (T >= 256 ? 1 : 256 / T) : (T >= 128 ? 1 : 256 / T); }

// ---------------------------------------------------------------- tunables
// Optional tuning variables (DESIGN.md section 5 "Tuning knobs") are read from the environment ONCE -- by mdsp_init(), or on
// first use -- into this struct; exec / plan paths only ever look at the struct.  mdsp_reload_tunables() re-reads them
// (tools/tune.py sweeps variants inside one process).  None changes results beyond rounding.
// One line of the polyphase choice file (read only when MDSP_FIR_CHOICE_FILE names one; first line: "# mi355dsp-fir-choices v<version> gfx950"):
//     L M ntaps taps_dtype x_dtype KNOB=value[,KNOB=value ...]          e.g.   1 16 583 0 0 MDSP_FIR_MM_CH=1
// -- the polyphase knobs a box's own measurement (tools/tune_fir.py TUNE_PERSIST=1) found faster than the library's rule for that shape.  The dispatch of a
// filter with that shape runs under those values (FirChoiceScope, fir.hip); every other shape and every other kernel family is untouched.
struct FirChoice {
    int64_t L = 0, M = 0, hlen = 0;
    int taps_dtype = 0, x_dtype = 0;
    int nset = 0;
    int field[12];      // index into the table of fir.hip / api_core.hip (fir_choice_field)
    int value[12];
};
struct Tunables {
    std::vector<FirChoice> fir_choices; // the choice file, read with the environment (empty: no file)
    int engine = MDSP_ENGINE_AUTO;      // MDSP_ENGINE=fused|rocfft : engine of plans created with MDSP_ENGINE_AUTO
    int wg_per_cu = 0;                  // MDSP_WG_PER_CU           : persistent-grid workgroups per CU (0 = occupancy query / kernel default)
    int runs_per_slot = 1;              // MDSP_RUNS_PER_SLOT       : contiguous runs per transform slot
    int spec_prio = 1;                  // MDSP_SPEC_PRIO           : Welch / STFT kernels: bit 0 a unit's loads, bit 1 the STFT column stores issued at raised wave
                                        //                            priority (default 1: Welch 1.374 -> 1.356 ms; STFT loads no effect, stores +5 % time)
    int ols_prio = 1;                   // MDSP_OLS_PRIO            : overlap-save kernel: bit 0 loads, bit 1 stores issued at raised wave priority (default 1:
                                        //                            1.799 -> 1.774 ms per 2^30 samples, profiles/r03l_tune_prio.json)
    int ols_variant = 0, welch_variant = 0, stft_variant = 1;   // MDSP_{OLS,WELCH,STFT}_VARIANT : alternative kernel instantiations
    int rocfft_chunk_mib = 192;         // MDSP_ROCFFT_CHUNK_MIB    : intermediates per rocFFT-engine chunk
    int fir_lds_kib = 20;               // MDSP_FIR_LDS_KIB         : staging tile of the fast polyphase kernel
    int arb_prio = 0;                   // MDSP_ARB_PRIO            : FIRArbitrary: the prologue of a workgroup at raised wave priority
    int arb_nch = 4, arb_tile = 0;      // MDSP_ARB_NCH / _TILE     : FIRArbitrary channels per group / outputs per workgroup (0 = default)
    int arb_scan = 1;                   // MDSP_ARB_SCAN=0          : serial host recurrence only
    int64_t arb_scan_min = (int64_t)1 << 19;   // MDSP_ARB_SCAN_MIN : outputs from which the device scan is used
    int host_chunk_mib = 64;            // MDSP_HOST_CHUNK_MIB      : pinned staging chunk of the host-pointer entry points
    int fir_p = 0;                      // MDSP_FIR_P=4             : four output residues per thread in the fast polyphase kernel (default 2)
    int fir_mm = -1;                    // MDSP_FIR_MM=0            : matrix-core polyphase kernel off (default: wherever the shape fits)
    int fir_exact = 0;                  // MDSP_FIR_EXACT=1         : polyphase filters run the generic kernel only -- every output reads exactly its own tapsPerPhi-sample
                                        //                            window (stream_filt.jl:496-509), so a NaN / Inf sample leaves exactly the reference's hole
    int fir_dec = 1;                    // MDSP_FIR_DEC=0|2|3       : 0 = decimators (L = 1) on the matrix-core / register-tap kernels as up to round 4; default: the phase-per-lane
                                        //                            decimator kernel where it measured faster; 2 = its run-time M form for every M; 3 = the kernel for every M <= 64
    int fir_dec_wgs = 0;                // MDSP_FIR_DEC_WGS         : workgroups per CU of the decimator kernel's launch (Float64 / ComplexF64: the persistent form; 0 = 2, what stays resident)
    int fir_dec_nc = 1;                 // MDSP_FIR_DEC_NC=0        : the decimator kernel's run-time chunk loop also for filters of five chunks per phase (default: unrolled, taps in registers, sliding window)
    int fir_dec_ablate = 0;             // MDSP_FIR_DEC_ABLATE      : profiling only: phases of the decimator kernel switched off (fir.hip)
    int fir_mm_rows = -1;               // MDSP_FIR_MM_ROWS=0|1|2   : its tiles staged as one run / row by row / one run with padded rows (default: by cost; padded rows
                                        //                            where the rows' sample stride is bank-hostile)
    int fir_mm_nblk = 1;                // MDSP_FIR_MM_NBLK=0       : L > 192: the taps of a wave's column blocks fetched per tile (round 2) instead of all in registers
    int fir_mm_t64 = 1;                 // MDSP_FIR_MM_T64=0        : round 2's register limits of the matrix-core polyphase kernel (taps fetched per tile beyond 48 / 64 / 32 / 24
                                        //                            k-steps whatever the chunk count; default: fewer chunks per wave, taps in registers)
    int fir_mm_tight = 1;               // MDSP_FIR_MM_TIGHT=0      : no last-resort tile forms (shorter rows for L < 16, the 40-step Float32 register form): shapes that do not
                                        //                            fit the LDS otherwise go to the generic kernel, as up to round 3
    int fir_mm_rpx = 0;                 // MDSP_FIR_MM_RPX=1        : (round 5 prep, unmeasured) padded runs also for fetched taps and for register forms whose window meets two pads
    int fir_mm_tiewaves = 0;            // MDSP_FIR_MM_TIEWAVES=1   : (round 5 prep) equal tiles at equal cost: the one with more multiplying waves
    int fir_mm_prio = -1;               // MDSP_FIR_MM_PRIO=0|1     : DMA and store waves of the matrix-core kernel at normal / raised priority (default: raised where a
                                        //                            multiplying wave owns one column block)
    int fir_mm_rpad = 0;                // MDSP_FIR_MM_RPAD         : dwords of padding behind a granule of a padded run (0 = 4)
    int fir_mm_vstore = 1;              // MDSP_FIR_MM_VSTORE       : 0 = the matrix-core kernel stores output rows that are not whole vectors element by element (round 2)
    int fir_mm_pad = -1;                // MDSP_FIR_MM_PAD          : 0 = output rows of the matrix-core kernel's LDS tile without their 16 bytes of padding
    int fir_mm_ch = 0;                  // MDSP_FIR_MM_CH           : at most this many 16-row chunks per multiplying wave (0 = 4 Float32, 2 otherwise)
    int fir_mm_ng = 0;                  // MDSP_FIR_MM_NG           : at most this many 16 CH-row groups per tile (0 = default 8)
    int fir_mm_nd = 0, fir_mm_ns = 0;   // MDSP_FIR_MM_ND / _NS     : its DMA / store waves (0 = default: 2 DMA waves, 2 store waves for ratios >= 1, else 4)
    int gen_wide = 1;                   // MDSP_GEN_WIDE=0          : nextfastfft sizes: round 3's schedules of small radices instead of the three-pass composite-radix ones
    int gen_ct_f64_max = 8000;          // MDSP_GEN_CT_F64_MAX      : Float64 nextfastfft sizes above this leave the single-workgroup compile-time schedules (for the multi-pass engine)
    int ols_prefetch = 0;               // MDSP_OLS_PREFETCH=1      : overlap-save kernel with software prefetch of the next unit (default: off)
    int gx = 1;                         // MDSP_GX=0|2              : 0 = no run-time-schedule single-workgroup kernel (spectral_gx.h: sizes without a compile-time schedule go to
                                        //                            the round-2 LDS kernel / the multi-pass engine / rocFFT as up to round 5); 2 = that kernel for EVERY size it plans (A/B); 6 = no rows above 8192 points (spectral_ctcols_big.hip); 4 = never the compile-time kernels of spectral_ctcols.hip / spectral_ctbig.hip, 5 = never spectral_ctbig.hip (A/B)
    int bigfft = 1;                     // MDSP_BIGFFT=0            : transforms above the one-workgroup sizes go to the rocFFT pipeline instead of the multi-pass fused engine (bigfft.hip)
    int big_chunk_mib = 1024;           // MDSP_BIG_CHUNK_MIB       : work buffer of the multi-pass engine per launch group.  Measured 16 .. 2048 MiB (profiles/r05_bigfft_sessions.json):
                                        //                            larger is faster up to all transforms in one group (fewer launches, one accumulator update); no Infinity Cache effect seen
    int big_wgs = 0;                    // MDSP_BIG_WGS             : workgroups per CU of its pass kernels (0 = 2, what the LDS admits)
    int big_rmax = 512;                 // MDSP_BIG_RMAX            : longest sub-transform of its passes (512; 256 / 128 / 64 force more, shorter passes)
    int big_ols_log2n = 0;              // MDSP_BIG_OLS_LOG2N       : overlap-save beyond the partitioned kernels: log2 of the block transform (0: the rule of bigfft.hip ols_size)
    int big_ols_rows = 1;               // MDSP_BIG_OLS_ROWS=0      : long filters on three passes each way (natural-order spectra) instead of the rows form (column pass + row kernel)
    int big_welch_rows = 1;             // MDSP_BIG_WELCH_ROWS=0    : Welch at nfft = 32 .. 256 x 8192 (Float64: x 4096) on three passes instead of column pass + single-workgroup Welch kernel
    int big_fast = 1;                   // MDSP_BIG_FAST=0          : generic LDS phases also where the two-stage register form applies (sub-transforms of 32 .. 256 points); 2: Float32 128 = 8 x 16 (round 5's first form) instead of 16 x 8; 3: Float32 64 = 16 x 4 instead of 8 x 8 (slower)
    int big_ablate = 0;                 // MDSP_BIG_ABLATE          : profiling only (results are garbage): phases of its pass kernels switched off, see bigfft.hip
    int big_groups = 0;                 // MDSP_BIG_GROUPS          : transform groups per launch of its passes (0 = enough workgroups for four per CU)
    int plan_cache_total = 8 * MDSP_PLAN_CACHE_SIZE;   // MDSP_PLAN_CACHE_TOTAL : entries in the whole plan cache above which idle partitions are trimmed
    int plan_cache_idle = 64;           // MDSP_PLAN_CACHE_IDLE     : cache requests without one of its own after which a partition counts as idle
#ifdef MDSP_DEBUG_KNOBS
    // Profiling / bisecting switches: only in builds made with -DMDSP_DEBUG_KNOBS (build.py --tag dbg --cflags -DMDSP_DEBUG_KNOBS).
    int ablate = 0;                     // MDSP_ABLATE: 1 skip HBM loads, 2 skip transforms, 4 skip stores / accumulation (results are garbage)
    bool welch_nohalf = false, stft_noshift = false, stft_nopair = false, stft_nodirect = false, fir_generic = false,
         fir_identity_lanes = false, mt_passes = false, arb_prof = false;
#endif
};
const Tunables& tunables();
void reload_tunables();
// The calling thread sees `t` instead of the process-wide struct until the returned previous override is put back (nullptr = none): per-shape choices.
const Tunables* tunables_override(const Tunables* t);
// field of the polyphase knob MDSP_FIR_<...> inside a Tunables (nullptr: not a knob a choice file may set)
int* fir_choice_field(Tunables& t, int index);
int fir_choice_index(const char* name);
uint64_t tunables_generation();   // bumped by every reload: keys host-side memos of values derived from the tunables

// ---------------------------------------------------------------- library-wide LRU of device objects (plancache.hip)
// key = plan_cache_key(kind, stream) [device, calling thread, stream] + whatever else the object depends on, appended by the caller.
std::string plan_cache_key(char kind, void* stream);
int plan_cache_get(const std::string& key, void** out, const std::function<int(void**)>& make, std::function<void(void*)> destroy);

#ifdef MDSP_DEBUG_KNOBS
#define MDSP_DBG(field) (::mdsp::tunables().field)
#define MDSP_ABLATED(a, bit) (((a).ablate & (bit)) != 0)
#else
#define MDSP_DBG(field) 0
#define MDSP_ABLATED(a, bit) false
#endif

// spectral.hip: out[j] = (R)(sum over channels of psd[c][j]) (* scale, rounded once more) -- the local part of the cross-channel Welch mean (comm.hip)
int channel_sum_scaled(const void* psd_dev, int64_t nout, int64_t nch, int64_t ldp, int real_dtype, void* sum_dev, double scale, void* stream);
}  // namespace mdsp
