/*
 * mi355dsp.h -- C ABI of libmi355dsp.so: DSP.jl's FFT-filtering / spectral-estimation hot path on MI355X.
 *
 * DSP.jl (v0.8.5) has no FFI of its own: the hot path leaves Julia only through AbstractFFTs plans
 * (FFTW) and BLAS.dot.  This header is the boundary a thin Julia `ccall` module (julia/MI355DSP.jl) or the
 * Python/ctypes host (dsp.jl_amd/) binds instead of those calls; every entry point below names the
 * reference routine whose inner loop it replaces (paths relative to DSP.jl `src/`).
 *
 * Conventions
 *   - Plain C: pointers, int64 sizes, int status.  Nothing throws or aborts.  0 = MDSP_OK, <0 = error class
 *     (the host wrapper maps classes to the same exception types DSP.jl raises); mdsp_last_error_string()
 *     returns a thread-local message.
 *   - `*_dev` pointers are device (HBM) pointers owned by the caller; `*_host` pointers are host memory read
 *     during the call.  Handles own plans, work buffers, tables and streaming state.  A handle is
 *     single-owner; calls are ordered on the `stream` argument (a hipStream_t passed as void*; NULL = the
 *     null stream).  Distinct handles may be used from distinct host threads.
 *   - Arrays are column-major: a multi-channel signal is (len, nch) with leading dimension `ld` (elements).
 *   - Lengths, offsets and phase indices are int64 and follow the reference's 1-based state values where the
 *     reference exposes them (phi_idx, input_deficit).
 */
#ifndef MI355DSP_H
#define MI355DSP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDSP_VERSION 100 /* 0.1.0 */

/* status codes: error classes mirror the Julia exception types of the reference */
enum {
    MDSP_OK = 0,
    MDSP_ERR_ARGUMENT = -1,    /* ArgumentError      (dspbase.jl:28-33, filt.jl:474,531, periodograms.jl:396,564,876, stream_filt.jl:415,490) */
    MDSP_ERR_DOMAIN = -2,      /* DomainError        (periodograms.jl:44-45,397,565; stream_filt.jl:194,217) */
    MDSP_ERR_DIMENSION = -3,   /* DimensionMismatch  (periodograms.jl:255,735-737) */
    MDSP_ERR_ASSERTION = -4,   /* AssertionError     (stream_filt.jl:634,718,722) */
    MDSP_ERR_UNSUPPORTED = -5, /* valid in DSP.jl but outside this library's scope -> caller falls back to CPU */
    MDSP_ERR_DEVICE = -6,      /* HIP / rocFFT / RCCL runtime failure */
    MDSP_ERR_NOMEM = -7
};

/* element types (fftintype / fftouttype / fftabs2type rules, util.jl:92-104, are applied by the host) */
enum { MDSP_F32 = 0, MDSP_F64 = 1, MDSP_C32 = 2 /* ComplexF32 */, MDSP_C64 = 3 /* ComplexF64 */ };

/* transform engine */
enum {
    MDSP_ENGINE_AUTO = 0,  /* fused in-LDS FFT kernels when the size is supported, rocFFT otherwise */
    MDSP_ENGINE_FUSED = 1, /* segment/window -> FFT -> epilogue in ONE kernel, FFT held in LDS/registers */
    MDSP_ENGINE_ROCFFT = 2 /* segmenter / epilogue kernels around cached, batched rocFFT plans */
};

/* ------------------------------------------------------------------------------------------------------
 * Library / device
 * ---------------------------------------------------------------------------------------------------- */
int mdsp_version(void);
const char* mdsp_last_error_string(void);
/* Select the device for the calling thread and create the per-device context (rocFFT setup, plan cache). */
int mdsp_init(int device);
int mdsp_shutdown(void);
/* Fifteen optional environment variables (INTEGRATION.md "Environment": MDSP_ENGINE, MDSP_WG_PER_CU, MDSP_PLAN_CACHE_TOTAL / _IDLE, MDSP_ROCFFT_CHUNK_MIB,
 * MDSP_HOST_CHUNK_MIB, MDSP_BIG_CHUNK_MIB, MDSP_BIGFFT, MDSP_GX, MDSP_FIR_MM, MDSP_FIR_DEC, MDSP_FIR_EXACT, MDSP_ARB_SCAN, MDSP_ARB_SCAN_MIN,
 * MDSP_FIR_CHOICE_FILE) are read once, by mdsp_init() or on first use; exec and plan paths never call getenv.  mdsp_reload_tunables() re-reads them and is NOT
 * thread-safe against concurrent library calls (exec and plan paths read the table without a lock: tuning tools are single-threaded).
 * Everything that only steers an experiment -- kernel variants, tile shapes, wave priorities (about fifty names up to round 5) -- is a KNOB, not an environment
 * variable: mdsp_set_knob(name, value, 0) sets one, (name, 0, 1) puts its default back; used by tools/ and tests/ only.  mdsp_debug_knobs() is 1 only for a
 * library built with -DMDSP_DEBUG_KNOBS, which also reads every knob from the environment and carries the profiling switches (MDSP_ABLATE, ...). */
int mdsp_reload_tunables(void);
int mdsp_set_knob(const char* name, int value, int unset);
int mdsp_debug_knobs(void);
int mdsp_device_count(int* count);

/* Device-memory helpers for hosts without their own GPU array type (ctypes tests, the Julia wrapper). */
int mdsp_malloc(void** dev_ptr, size_t bytes);
int mdsp_free(void* dev_ptr);
int mdsp_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, void* stream);
int mdsp_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream);
int mdsp_memset(void* dst_dev, int value, size_t bytes, void* stream);
int mdsp_stream_synchronize(void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Pure index arithmetic (host; bit-exact with the reference)
 * ---------------------------------------------------------------------------------------------------- */
/* util.jl:134  nextfastfft(n) = nextprod((2,3,5,7), n) */
int64_t mdsp_nextfastfft(int64_t n);
/* dspbase.jl:268-291  optimalfftfiltlength(nb, nx) */
int64_t mdsp_optimal_fft_len(int64_t nb, int64_t nx);
/* periodograms.jl:49-50  number of frames of ArraySplit (trailing partial frame dropped) */
int64_t mdsp_frame_count(int64_t len, int64_t n, int64_t noverlap);
/* stream_filt.jl:317-322  outputlength(inputlength, L//M, initial_phi) */
int64_t mdsp_outputlength(int64_t inputlength, int64_t L, int64_t M, int64_t initial_phi);
/* stream_filt.jl:358-364  inputlength(outputlength, L//M, initial_phi, RoundDown|RoundUp) */
int64_t mdsp_inputlength(int64_t outputlength, int64_t L, int64_t M, int64_t initial_phi, int round_up);
/* Overlap-save block geometry of _fftfilt! (Filters/filt.jl:490,504-517) for block `iblock` (0-based):
 * off (1-based first output), npadbefore, xstart (1-based), n (samples copied), nout (samples saved). */
int mdsp_ols_block_geometry(int64_t nb, int64_t nx, int64_t nfft, int64_t iblock, int64_t* off,
                            int64_t* npadbefore, int64_t* xstart, int64_t* n, int64_t* nout);

/* ------------------------------------------------------------------------------------------------------
 * Overlap-save FIR filtering / convolution
 *   replaces the block loop of _fftfilt! (Filters/filt.jl:504-518: zero+copy, rfft, .* filterft, brfft, copy)
 *   and of unsafe_conv_kern_os! (dspbase.jl:546-606) incl. its padded edge blocks (:371-486), N = 1.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct mdsp_ols_plan_s* mdsp_ols_plan;
enum {
    MDSP_OLS_FILT = 0, /* y has the length of x; taps pre-scaled by 1/nfft            (filt.jl:499) */
    MDSP_OLS_CONV = 1  /* y has length nx+nb-1; filter SPECTRUM scaled by 1/nfft (dspbase.jl:516) */
};
/* taps_host: nb elements of `dtype` (real taps for MDSP_F32/F64, complex for MDSP_C32/C64).
 * nfft = 0 selects mdsp_optimal_fft_len(nb, nx_hint). */
int mdsp_ols_plan_create(mdsp_ols_plan* plan, const void* taps_host, int64_t nb, int64_t nfft, int64_t nx_hint,
                         int dtype, int mode, int engine);
int mdsp_ols_plan_destroy(mdsp_ols_plan plan);
int mdsp_ols_plan_info(mdsp_ols_plan plan, int64_t* nfft, int64_t* block_len /* L */, int* engine_used);
/* What actually executes.  mdsp_ols_plan_info reports the REFERENCE's geometry (the nfft optimalfftfiltlength / the caller chose, dspbase.jl:268-291);
 * the fused engine runs it as it is up to nfft = 8192 (4096 in Float64).  Longer filters -- ~1100 taps and more, where the reference asks for
 * nfft = 16384 ... 2^20 -- are re-blocked: one block of the largest in-LDS transform while the filter covers at most half of it, else a
 * uniformly partitioned filter (2..4 partitions of exec_nfft/2 taps, spectra of the last blocks kept in registers).  Beyond four partitions
 * (more than 16384 Float32 / 8192 Float64 taps, any of the four dtypes) a block no longer fits a workgroup and is transformed in passes over HBM
 * (partitions = 1): exec_nfft = R0 x S with S the longest single-workgroup transform (8192 / 4096) -- a column pass of R0-point
 * transforms, ONE kernel per row for transform, filter spectrum, inverse transform and twiddle, a column pass back (R0 = 64 up to S x 16 taps, 256 up
 * to S x 64) -- and beyond that three passes each way on natural-order spectra (2^20 points and more).  Same outputs within rounding, one plan for
 * every filter length with 2 nb <= 2^26. */
int mdsp_ols_plan_geometry(mdsp_ols_plan plan, int64_t* exec_nfft, int64_t* exec_block_len, int* partitions);
/* The same answer without a plan and without a device (pure host arithmetic: what mdsp_ols_plan_create WOULD execute for these arguments, so a host
 * can size a time-axis split or a test can check the rule on a machine without a GPU).  engine_used: MDSP_ENGINE_*; rows: > 0 when the blocks run in
 * the rows form of the multi-pass engine (R0 rows of 8192 / 4096 points), else 0.  Any output pointer may be NULL. */
int mdsp_ols_geometry_for(int64_t nb, int64_t nfft, int64_t nx_hint, int dtype, int mode, int engine, int64_t* exec_nfft, int64_t* exec_block_len,
                          int* partitions, int* engine_used, int* rows);
/* x_dev: (nx, ncols) ld ldx;  y_dev: (nout, ncols) ld ldy.  nout = nx (filt), nx+nb-1 (conv), or any
 * 0 <= nout <= nx+nb-1.  x and y must not alias (Filters/filt.jl:438-439). */
int mdsp_ols_exec(mdsp_ols_plan plan, const void* x_dev, int64_t nx, int64_t ncols, int64_t ldx, void* y_dev,
                  int64_t nout, int64_t ldy, void* stream);
/* Blocks [first_block, first_block + nblocks_range) of the SAME block grid mdsp_ols_exec uses for one column of nx samples /
 * nout outputs, from a slice of the signal: xs_dev holds x[xs_first .. xs_first + xs_len) and must cover the samples those blocks
 * read, [first_block L - (nb-1), (first_block + nblocks_range) L) clipped to [0, nx); ys_dev[0..] receives the outputs from
 * first_block L on (L = exec_block_len of mdsp_ols_plan_geometry).  first_block must be even for real dtypes on single-block plans
 * (bit-identical to the whole-column call); partitioned plans (long filters) take any first_block and agree with the whole-column call
 * within rounding.  Building block of mdsp_ols_exec_host and of a time-axis split
 * of one stream over GPUs (no collective: overlap-save blocks are independent, Filters/filt.jl:504-518). */
int mdsp_ols_exec_range(mdsp_ols_plan plan, const void* xs_dev, int64_t xs_first, int64_t xs_len, int64_t nx, void* ys_dev,
                        int64_t first_block, int64_t nblocks_range, int64_t nout, void* stream);
/* Segmenter only (K1): materialise blocks [first_block, first_block+nblocks) of column 0 as (nfft, nblocks)
 * -- the exact contents of `tmp1` before the forward transform (filt.jl:509-510).  For parity tests. */
int mdsp_ols_segment(mdsp_ols_plan plan, const void* x_dev, int64_t nx, int64_t first_block, int64_t nblocks,
                     void* seg_dev, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Framing (ArraySplit), Welch, STFT / spectrogram / periodogram
 *   replaces ArraySplit getindex (periodograms.jl:57-69), mul!(outbuf, plan, sig) (:754,:888),
 *   fft2pow! (:142-172) and fft2oneortwosided! (:234-244).
 * ---------------------------------------------------------------------------------------------------- */
/* K4 only: frames [first, first+count) of one channel as (nfft, count) of the signal's fftintype, windowed
 * in Float64 and rounded once, zero tail -- bit-identical to the reference's buffer contents. */
int mdsp_frames(const void* s_dev, int64_t len, int dtype, int64_t n, int64_t noverlap, int64_t nfft,
                const double* window_host /* n doubles or NULL */, int64_t first, int64_t count, void* frames_dev,
                void* stream);

typedef struct mdsp_welch_plan_s* mdsp_welch_plan;
/* dtype: element type of the signal (fftintype already applied).  window_host: n Float64 or NULL.
 * r = fs * sum(abs2, window) (or fs * n without window), as WelchConfig computes it (periodograms.jl:567-568).
 * onesided requires a real dtype (ArgumentError otherwise, :564). */
int mdsp_welch_plan_create(mdsp_welch_plan* plan, int64_t n, int64_t noverlap, int64_t nfft,
                           const double* window_host, double r, int onesided, int dtype, int engine);
int mdsp_welch_plan_destroy(mdsp_welch_plan plan);
int mdsp_welch_plan_info(mdsp_welch_plan plan, int64_t* nout, int* engine_used);
/* s_dev: (len, nch) ld lds.  psd_dev: (nout, nch) ld ldp of fftabs2type(dtype); nout = nfft/2+1 or nfft.
 * Each channel gets its own PSD (welch_pgram_helper!, periodograms.jl:746-759). */
int mdsp_welch_exec(mdsp_welch_plan plan, const void* s_dev, int64_t len, int64_t nch, int64_t lds,
                    void* psd_dev, int64_t ldp, void* stream);
/* Streaming form of the same computation -- mdsp_welch_exec IS reset + accumulate + finalize.  The plan owns Float64 sums of
 * |X[k]|^2 over every frame accumulated since the last reset; a long stream may be handed over slice by slice (each slice: whole
 * frames, i.e. consecutive slices overlap by n - hop samples), by several ranks (mdsp_welch_allreduce), or from host memory
 * (mdsp_welch_exec_host).  finalize forms psd = fold(sums) / (K fs sum(w^2)) with K = frames_total (0: the frames accumulated here),
 * periodograms.jl:751-757.  All slices of one accumulation must have the same channel count. */
int mdsp_welch_reset(mdsp_welch_plan plan);
int mdsp_welch_accumulate(mdsp_welch_plan plan, const void* s_dev, int64_t len, int64_t nch, int64_t lds, void* stream);
int mdsp_welch_frames_accumulated(mdsp_welch_plan plan, int64_t* frames_per_channel);
int mdsp_welch_finalize(mdsp_welch_plan plan, int64_t frames_total, void* psd_dev, int64_t ldp, void* stream);
/* The accumulator itself (Float64, `count` values, layout private to the engine): what a caller-side collective would sum over ranks. */
int mdsp_welch_accumulator(mdsp_welch_plan plan, void** acc_dev, int64_t* count);
/* Local part of the cross-channel mean: sum over nch rows (ld ldp) of nout values, in `real_dtype`, into sum_dev (nout). */
int mdsp_channel_sum(const void* psd_dev, int64_t nout, int64_t nch, int64_t ldp, int real_dtype, void* sum_dev,
                     void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Multi-GPU: one process per GPU, channels sharded over ranks, RCCL over xGMI for the ONE collective of the path.
 *   The reference loops over columns / channels serially (Filters/filt.jl:504, stream_filt.jl:768 `mapslices`); here rank r owns
 *   a contiguous block of channels and runs the single-GPU entry points on them -- no data-path collective -- and only the
 *   cross-channel Welch mean needs an exchange: an all-reduce (sum) of nout values.
 *   Bootstrap: rank 0 calls mdsp_comm_unique_id and ships the 128 bytes to the other ranks by whatever the host has (MPI.jl bcast,
 *   a shared file, torch.distributed's store); every rank then calls mdsp_comm_init_rank (collective) after mdsp_init(device).
 *   librccl is bound at run time; without it these entries return MDSP_ERR_DEVICE and everything else keeps working.
 * ---------------------------------------------------------------------------------------------------- */
#define MDSP_COMM_ID_BYTES 128
typedef struct mdsp_comm_s* mdsp_comm;
int mdsp_comm_unique_id(void* id128 /* MDSP_COMM_ID_BYTES, host */);
int mdsp_comm_init_rank(mdsp_comm* comm, const void* id128, int rank, int nranks);
int mdsp_comm_destroy(mdsp_comm comm);
int mdsp_comm_info(mdsp_comm comm, int* rank, int* nranks);
/* in-place sum over ranks of `count` Float32 / Float64 values (ncclAllReduce on `stream`) */
int mdsp_allreduce_sum(mdsp_comm comm, void* buf_dev, int64_t count, int real_dtype, void* stream);
/* Cross-channel Welch mean: psd_dev = this rank's per-channel PSDs (nch_local rows of ld ldp, plan's nout bins, fftabs2type);
 * mean_dev (nout) = (1 / nch_total) * sum over the channels of ALL ranks.  comm == NULL or a 1-rank communicator: local mean. */
int mdsp_welch_mean_allreduce(mdsp_welch_plan plan, const void* psd_dev, int64_t nch_local, int64_t ldp, int64_t nch_total,
                              void* mean_dev, mdsp_comm comm, void* stream);
/* ONE stream split along time over ranks: sums the plans' Float64 accumulators and frame counts over ranks in place; a following
 * mdsp_welch_finalize(plan, 0, ...) gives every rank the PSD of the whole stream.  Stream-ordered (one RCCL group of the sums and the
 * frame count; the total count stays on the device and finalize reads it there); mdsp_welch_frames_accumulated afterwards reads it back. */
int mdsp_welch_allreduce(mdsp_welch_plan plan, mdsp_comm comm, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Host-array entry points: DSP.jl's own call shape (host Arrays in, host Arrays out; Filters/filt.jl:458-476,
 * periodograms.jl:647-744, :872-897, stream_filt.jl:627-637, :688-775) as a chunked H2D || kernel || D2H pipeline: one internal stream per
 * PCIe direction plus one for the kernels, three lanes of device (and, for pageable arrays, page-locked staging) buffers, so that chunk k's
 * download, chunk k+1's kernels and chunk k+2's upload are in flight together (full duplex).
 * Synchronous (results are in the host arrays on return); PCIe-bound -- see DESIGN.md section 5 for the measured rates.
 *   flags: MDSP_HOST_PINNED = the caller's arrays are already page-locked (mdsp_host_alloc / mdsp_host_register): no staging copies.
 *   Chunk size: MDSP_HOST_CHUNK_MIB (default 64).  Overlap-save results are bit-identical to the device-resident calls for single-block
 *   plans (same block grid); PARTITIONED (long-filter) plans run block ranges from slices of the signal (mdsp_ols_exec_range: samples under
 *   the zero taps in front of a slice read as zero), so they equal the whole-column call within rounding, and a NaN / Inf earlier in the
 *   stream does not propagate into later slices the way it does through the device-resident delay line.  Welch: equal up to Float64
 *   summation order.
 *   Ordering and resources: every host-array call starts with a device-wide synchronisation (the pipeline's streams are non-blocking; this
 *   orders them behind whatever the same handle has queued on other streams); one pipeline per device, guarded by a mutex for the whole
 *   call -- host-array calls of several threads on one device run one after the other; its three lanes of device (and page-locked staging)
 *   buffers only grow and stay until mdsp_host_pipeline_trim() or mdsp_shutdown().  mdsp_fir_exec_host advances the filter state chunk by
 *   chunk: after an error return the filter's state is unspecified -- mdsp_fir_reset (or set_state) before using it again.
 * ---------------------------------------------------------------------------------------------------- */
#define MDSP_HOST_PINNED 1
/* frees the host pipelines' device and page-locked buffers of every device (they are re-grown on the next host-array call) */
int mdsp_host_pipeline_trim(void);
int mdsp_host_alloc(void** host_ptr, size_t bytes);      /* hipHostMalloc */
int mdsp_host_free(void* host_ptr);
int mdsp_host_register(void* host_ptr, size_t bytes);    /* hipHostRegister: page-lock an existing allocation (e.g. a Julia Array) */
int mdsp_host_unregister(void* host_ptr);
int mdsp_ols_exec_host(mdsp_ols_plan plan, const void* x_host, int64_t nx, int64_t ncols, int64_t ldx, void* y_host, int64_t nout,
                       int64_t ldy, int flags);
int mdsp_welch_exec_host(mdsp_welch_plan plan, const void* s_host, int64_t len, int64_t nch, int64_t lds, void* psd_host,
                         int64_t ldp, int flags);
/* stft / spectrogram / periodogram of host arrays (mdsp_stft_exec with host pointers, periodograms.jl:872-897): channel by channel in runs of
 * whole frames, the chunk sized by its OUTPUT (2-8x the input); bit-identical to the device-resident call.  Declared after mdsp_stft_plan below. */

typedef struct mdsp_stft_plan_s* mdsp_stft_plan;
/* psd_only = 0: raw STFT columns (fftouttype), unnormalised (periodograms.jl:892);
 * psd_only = 1: spectrogram columns fft2pow!(…, r, onesided, offset) (:890) in fftabs2type.
 * periodogram(s) (periodograms.jl:393-417) is the single-frame case n = length(s), noverlap = 0, psd_only = 1. */
int mdsp_stft_plan_create(mdsp_stft_plan* plan, int64_t n, int64_t noverlap, int64_t nfft,
                          const double* window_host, double r, int onesided, int psd_only, int dtype, int engine);
int mdsp_stft_plan_destroy(mdsp_stft_plan plan);
int mdsp_stft_plan_info(mdsp_stft_plan plan, int64_t* nout, int* engine_used);
/* s_dev: (len, nch) ld lds.  out_dev: nch matrices (nout, k) column-major with column stride ldo (>= nout)
 * and channel stride chs (elements of the output type). */
int mdsp_stft_exec(mdsp_stft_plan plan, const void* s_dev, int64_t len, int64_t nch, int64_t lds, void* out_dev,
                   int64_t ldo, int64_t chs, void* stream);
/* the same with host arrays (see "Host-array entry points" above); flags: MDSP_HOST_PINNED */
int mdsp_stft_exec_host(mdsp_stft_plan plan, const void* s_host, int64_t len, int64_t nch, int64_t lds, void* out_host,
                        int64_t ldo, int64_t chs, int flags);

/* ------------------------------------------------------------------------------------------------------
 * Multitaper spectral estimation (src/multitaper.jl)
 *   plan    = MTConfig (:5-135): frames of n samples, nfft, `ntapers` tapers (n x ntapers, column-major Float64,
 *             e.g. dpss(n, nw, ntapers)), inverse normalisations r[taper] (fs ./ taper_weights, :127-131)
 *   psd     = mt_pgram! (:225-245) for every frame of arraysplit(signal, n, noverlap): mt_spectrogram! (:312-330);
 *             out: (nout, K) per channel, like mdsp_stft_exec with psd_only
 *   spectra = mt_fft_tapered_multichannel! (:596-600): x_mt[f, taper, channel] of ONE n-sample frame per channel
 *             (real input, onesided), optionally demeaned per channel (:566-570)
 *   cross   = cs_inner! (:602-616) incl. the DC / Nyquist 1/sqrt(2) (:577-580): out (nch, nch, nfi) complex
 *   coherence_from_cs! (:704-723): (nch, nch, nf) complex -> real
 * ---------------------------------------------------------------------------------------------------- */
typedef struct mdsp_mt_plan_s* mdsp_mt_plan;
int mdsp_mt_plan_create(mdsp_mt_plan* plan, int64_t n, int64_t nfft, const double* tapers_host, int64_t ntapers,
                        const double* r_host, int onesided, int dtype, int engine);
int mdsp_mt_plan_destroy(mdsp_mt_plan plan);
int mdsp_mt_plan_info(mdsp_mt_plan plan, int64_t* nout, int64_t* ntapers, int* engine_used);
int mdsp_mt_psd_exec(mdsp_mt_plan plan, const void* s_dev, int64_t len, int64_t noverlap, int64_t nch, int64_t lds,
                     void* out_dev, int64_t ldo, int64_t chs, void* stream);
int mdsp_mt_spectra_exec(mdsp_mt_plan plan, const void* s_dev, int64_t nch, int64_t lds, int demean, void* xmt_dev,
                         void* stream);
int mdsp_mt_cross_spectra(mdsp_mt_plan plan, const void* xmt_dev, int64_t nch, const int64_t* freq_inds_host /*0-based*/,
                          int64_t nfi, void* out_dev, void* stream);
int mdsp_coherence_from_cs(const void* cs_dev, int64_t nch, int64_t nf, int real_dtype, void* out_dev, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Stateful polyphase FIR (FIRFilter{FIRStandard|FIRInterpolator|FIRDecimator|FIRRational})
 *   replaces the while loop of filt!(buffer, ::FIRFilter, x) (stream_filt.jl:409-558) and its
 *   unsafe_dot / BLAS.dot inner products (util.jl:225-283) and shiftin! (util.jl:299-314).
 * ---------------------------------------------------------------------------------------------------- */
typedef struct mdsp_fir_s* mdsp_fir;
/* taps_host: hlen taps of taps_dtype (MDSP_F32 | MDSP_F64); ratio L/M is reduced to lowest terms.
 * x_dtype: element type of the input; output type = promote(taps, x).  nch channels share the scalar state
 * (phi_idx, input_deficit) and keep separate histories, as resample(...; dims) does per slice (:768-774). */
int mdsp_fir_create(mdsp_fir* f, const void* taps_host, int64_t hlen, int64_t L, int64_t M, int taps_dtype,
                    int x_dtype, int64_t nch);
int mdsp_fir_destroy(mdsp_fir f);
/* exact != 0: this filter runs the generic kernel only, in which every output reads exactly its own tapsPerPhi-sample window
 * (stream_filt.jl:496-509) -- a NaN / Inf sample then leaves exactly the reference's hole.  The default kernels multiply a tile's common window
 * by explicit zero taps, so their hole can be wider by up to 15 outputs plus the outputs of 64 more input positions (the decimator kernel, L = 1,
 * is exact either way).  Same results otherwise, to rounding.  (Round 4 had this as the process-wide environment variable MDSP_FIR_EXACT only.) */
int mdsp_fir_set_exact(mdsp_fir f, int exact);
int mdsp_fir_reset(mdsp_fir f);                       /* reset!     stream_filt.jl:247-276 */
int mdsp_fir_setphase(mdsp_fir f, double phi);        /* setphase!  :216-229 */
int mdsp_fir_timedelay(mdsp_fir f, double* tau);      /* timedelay  :400-403 */
int mdsp_fir_outputlength(mdsp_fir f, int64_t inputlength, int64_t* outlen);            /* :324-338 */
int mdsp_fir_inputlength(mdsp_fir f, int64_t outputlength, int round_up, int64_t* inlen); /* :366-383 */
int mdsp_fir_info(mdsp_fir f, int* kind /*0 std,1 interp,2 decim,3 rational*/, int64_t* L, int64_t* M,
                  int64_t* taps_per_phase, int64_t* history_len, int* out_dtype);
/* Which kernel mdsp_fir_exec would run for a chunk of `xlen` samples in the filter's current state: 0 generic polyphase kernel
 * (any dtype), 1 register-tap kernel (Float32, <= 64 taps per phase), 2 matrix-core kernel (rows of 16 outputs on
 * v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64; Float32 results bit-identical to 0 / 1), 3 decimator kernel (L = 1, M <= 64, any length: a lane
 * per input phase; sums phases first, so it agrees with 0 to rounding, and reads exactly the reference's windows).  Diagnostics / tests. */
int mdsp_fir_kernel_path(mdsp_fir f, int64_t xlen, int* path);
/* Geometry the matrix-core kernel would use for a filter of hlen taps at ratio L // M (pure host arithmetic, no device):
 * out12 = {fits, rounds per row RB, outputs per row Lr = RB L, samples per row Mr = RB M, column blocks NB, row groups NG, k-steps of four taps (in registers up to 64, Float64 32; beyond that fetched per tile),
 * 16-row chunks per wave CH, parts per sample CS, DMA waves, store waves, LDS bytes}.  Diagnostics / tests. */
int mdsp_fir_mm_geometry(int64_t L, int64_t M, int64_t hlen, int taps_dtype, int x_dtype, int64_t* out12);
/* state is exactly the reference's: 1-based phi_idx and input_deficit, history (history_len, nch) of x_dtype */
int mdsp_fir_get_state(mdsp_fir f, int64_t* phi_idx, int64_t* input_deficit, void* history_host);
int mdsp_fir_set_state(mdsp_fir f, int64_t phi_idx, int64_t input_deficit, const void* history_host);
/* x_dev: (xlen, nch) ld ldx;  y_dev: (ycap, nch) ld ldy, ycap >= outputlength (ArgumentError otherwise, :490).
 * *nwritten = samples written per channel (the return value of filt!). */
int mdsp_fir_exec(mdsp_fir f, const void* x_dev, int64_t xlen, int64_t ldx, void* y_dev, int64_t ycap,
                  int64_t ldy, int64_t* nwritten, void* stream);
/* filt(::FIRFilter, x) / resample of host arrays (stream_filt.jl:627-637, :688-775): the stream passes through the filter in time chunks of
 * all channels, the filter state carried from chunk to chunk exactly as in the reference's streaming use -- the result and the final state
 * are those of ONE mdsp_fir_exec over the whole stream, bit for bit.  ycap >= mdsp_fir_outputlength(f, xlen).  flags: MDSP_HOST_PINNED */
int mdsp_fir_exec_host(mdsp_fir f, const void* x_host, int64_t xlen, int64_t ldx, void* y_host, int64_t ycap,
                       int64_t ldy, int64_t* nwritten, int flags);

/* ------------------------------------------------------------------------------------------------------
 * Arbitrary-rate resampler (FIRFilter{FIRArbitrary}: rate::AbstractFloat, Nphi phases, linear interpolation
 * between neighbouring phases through the derivative bank)
 *   replaces FIRArbitrary / FIRFilter(h, rate, Nphi) (stream_filt.jl:92-156), update! (:567-577) and
 *   filt!(buffer, ::FIRFilter{FIRArbitrary}, x) (:579-625).  The Float64 phase-accumulator recurrence is evaluated
 *   with the reference's own IEEE operations (host, once per call, shared by all channels), so the number of
 *   samples written and the state after every chunk are bit-exact; the dot products run on the device.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct mdsp_firarb_s* mdsp_firarb;
int mdsp_firarb_create(mdsp_firarb* f, const void* taps_host, int64_t hlen, double rate, int64_t nphi, int taps_dtype,
                       int x_dtype, int64_t nch);
int mdsp_firarb_destroy(mdsp_firarb f);
int mdsp_firarb_reset(mdsp_firarb f);                     /* reset!      stream_filt.jl:260-276 */
int mdsp_firarb_setphase(mdsp_firarb f, double phi);      /* setphase!   :231-239 */
int mdsp_firarb_timedelay(mdsp_firarb f, double* tau);    /* timedelay   :400-401 */
int mdsp_firarb_outputlength(mdsp_firarb f, int64_t inputlength, int64_t* outlen);              /* :340-342 */
int mdsp_firarb_inputlength(mdsp_firarb f, int64_t outputlength, int round_up, int64_t* inlen); /* :385-389 */
int mdsp_firarb_info(mdsp_firarb f, int64_t* nphi, int64_t* taps_per_phase, int64_t* history_len, int* out_dtype,
                     double* delta);
/* the reference's state: phiAccumulator, alpha = frac(phiAccumulator), phiIdx = 1 + trunc(phiAccumulator) (1-based),
 * inputDeficit, xIdx (1-based), history (history_len, nch) of x_dtype */
int mdsp_firarb_get_state(mdsp_firarb f, double* phi_acc, double* alpha, int64_t* phi_idx, int64_t* input_deficit,
                          int64_t* x_idx, void* history_host);
int mdsp_firarb_set_state(mdsp_firarb f, double phi_acc, int64_t input_deficit, const void* history_host);
/* y_dev: (ycap, nch); ycap must hold every sample the loop writes (outputlength(...) + 1 always suffices, which is
 * what allocate_output reserves, :639-655).  *nwritten = samplesWritten. */
int mdsp_firarb_exec(mdsp_firarb f, const void* x_dev, int64_t xlen, int64_t ldx, void* y_dev, int64_t ycap,
                     int64_t ldy, int64_t* nwritten, void* stream);
/* Pure host function (no device): the index trajectory of the loop above.  Writes the (xIdx, phiAccumulator) of
 * outputs 0, block, 2*block, ... into anchors_* (may be NULL), the number of outputs and the final state. */
int mdsp_arb_trajectory(double phi_acc, int64_t input_deficit, double rate, int64_t nphi, int64_t xlen, int64_t block,
                        int64_t* anchors_x, double* anchors_acc, int64_t anchors_cap, int64_t* nout,
                        double* phi_acc_end, int64_t* input_deficit_end);
/* For streams of >= 2^19 outputs mdsp_firarb_exec evaluates that same recurrence in parallel on the device, bit for bit
 * (integer image of the IEEE operations + a multi-level scan of the rounding, dsp.jl_amd/csrc/arb_scan.h), after a
 * serial pilot of `pilot` outputs; it falls back to the serial loop when its self-check cannot certify the result.
 * mdsp_arb_trajectory_scan is the host emulation of that device code (anchors every 16 outputs; *used = 0: the scan
 * does not apply to these arguments); mdsp_firarb_scan_stats counts how each trajectory of a filter was evaluated. */
int mdsp_arb_trajectory_scan(double phi_acc, int64_t input_deficit, double rate, int64_t nphi, int64_t xlen, int64_t pilot,
                             int64_t* anchors_x, double* anchors_acc, int64_t anchors_cap, int64_t* nout,
                             double* phi_acc_end, int64_t* input_deficit_end, int* used, int* passes);
int mdsp_firarb_scan_stats(mdsp_firarb f, int64_t* scanned, int64_t* serial);
/* test helper (host): updates at which the device replay's branch-free form of update! differs from the reference form */
int mdsp_arb_replay_check(double phi_acc, double rate, int64_t nphi, int64_t nsteps, int64_t* mismatches);

/* ------------------------------------------------------------------------------------------------------
 * N-dimensional convolution: conv(u, v) / conv!(out, u, v) for arrays (dspbase.jl:709-792).
 *   u_dev / v_dev: column-major arrays (first dimension fastest) of sizes su[ndim] / sv[ndim], both of `dtype`;
 *   out_dev: column-major, sizes su[d] + sv[d] - 1.
 *   mdsp_convnd_fft    replaces _conv_kern_fft! (dspbase.jl:611-644): per-dimension zero-padding to
 *                      nextfastfft(outsize), one N-d transform per operand (real transforms for real dtypes),
 *                      product, inverse, crop -- and stands in for unsafe_conv_kern_os! (:490-609), which evaluates
 *                      the same sums block-wise.  At most three dimensions with extent > 1 (rocFFT).
 *   mdsp_convnd_direct replaces _conv_td! (dspbase.jl:646-660): the convolution sum, any ndim <= 8.
 * ---------------------------------------------------------------------------------------------------- */
int mdsp_convnd_fft(const void* u_dev, const int64_t* su, const void* v_dev, const int64_t* sv, int ndim, int dtype,
                    void* out_dev, void* stream);
int mdsp_convnd_direct(const void* u_dev, const int64_t* su, const void* v_dev, const int64_t* sv, int ndim, int dtype,
                       void* out_dev, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Time-domain FIR (filt(b, a::Number, x) and the nb <= 66 branch of filt(b, x)):
 *   replaces _filt_fir! (dspbase.jl:95-105,118-141).  Zero initial state, per column.
 *   taps_host: nb REAL taps in the real precision of `dtype` (float for MDSP_F32/C32, double for F64/C64);
 *   x_dev / y_dev: (nx, ncols) of `dtype`.
 * ---------------------------------------------------------------------------------------------------- */
int mdsp_tdfir_exec(const void* taps_host, int64_t nb, int dtype, const void* x_dev, int64_t nx, int64_t ncols,
                    int64_t ldx, void* y_dev, int64_t ldy, void* stream);

/* Stateful time-domain FIR: filt!(out, f::DF2TFilter{<:PolynomialRatio}, x) with FIR coefficients (Filters/filt.jl:153-181)
 *   advancing the TDF-II state exactly as _filt_fir!(out, b, x, si, col) does (dspbase.jl:95-105).
 *   si_dev: (nb-1, ncols) of `dtype`, read as the initial state and overwritten with the final one. */
int mdsp_tdfir_state_exec(const void* taps_host, int64_t nb, int dtype, const void* x_dev, int64_t nx, int64_t ncols,
                          int64_t ldx, void* y_dev, int64_t ldy, void* si_dev, void* stream);
/* extrapolate_signal! (Filters/filt.jl:243-257), the odd-symmetric extension filtfilt applies per column:
 *   out (n + 2 pad, ncols) = [2 x[1] .- x[pad+1:-1:2]; x; 2 x[end] .- x[end-1:-1:end-pad]] */
int mdsp_extrapolate(const void* x_dev, int64_t n, int64_t ncols, int64_t ldx, int dtype, int64_t pad, void* out_dev,
                     int64_t ldo, void* stream);

/* hilbert(x) (src/util.jl:31-87): analytic signal of every real column, out (n, ncols) complex of the same precision */
int mdsp_hilbert(const void* x_dev, int64_t n, int64_t ncols, int64_t ldx, int real_dtype, void* out_dev, int64_t ldo,
                 void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Plan cache inside the library: the per-call fast path for hosts that mirror DSP.jl's function-style entry points one-to-one
 * (filt(b, x), conv(u, v), welch_pgram(s, n, noverlap), stft / spectrogram / periodogram build their FFTW plans on every call;
 * a device plan costs ~1 ms, the call ~45 us).  Same arguments as the matching *_plan_create plus the stream the plan will run on;
 * the key is (device, partition, stream, every argument, CONTENTS of taps / window).  The returned handle is BORROWED: never destroy it.
 * Partitions: the calling OS thread by default; a host whose logical tasks migrate between OS threads (Julia tasks) binds its own id with
 * mdsp_plan_cache_set_context(id != 0) -- thread-local, 0 restores the per-thread default -- and frees that partition with
 * mdsp_plan_cache_release_context(id).  Only requests of the same partition evict from its two lists (user plans / the library's own cached
 * objects, MDSP_PLAN_CACHE_SIZE each), so a handle stays valid until ITS PARTITION has made MDSP_PLAN_CACHE_SIZE further distinct cached requests,
 * with one exception that bounds device memory: above MDSP_PLAN_CACHE_TOTAL entries in the whole cache (environment, default 8 x
 * MDSP_PLAN_CACHE_SIZE) entries of explicit contexts that NO thread has bound at the moment and that have made no request for MDSP_PLAN_CACHE_IDLE
 * (default 64) cache requests are evicted -- a host therefore keeps its context bound for as long as it uses handles borrowed under it; the
 * partition of a live OS thread is never evicted by others (it may be inside a long call on a borrowed handle).
 * The entries of an OS thread that exits are reaped automatically (destroyed at the next cache request of any thread).
 * mdsp_plan_cache_clear() destroys every partition's entries and must not race with other threads' library calls.
 * mdsp_plan_cache_partitions: live partitions and the number of entries reaped so far (exited threads, released contexts, cap evictions).
 * ---------------------------------------------------------------------------------------------------- */
#define MDSP_PLAN_CACHE_SIZE 16
int mdsp_ols_plan_cached(mdsp_ols_plan* plan, const void* taps_host, int64_t nb, int64_t nfft, int64_t nx_hint, int dtype, int mode,
                         int engine, void* stream);
int mdsp_welch_plan_cached(mdsp_welch_plan* plan, int64_t n, int64_t noverlap, int64_t nfft, const double* window_host, double r,
                           int onesided, int dtype, int engine, void* stream);
int mdsp_stft_plan_cached(mdsp_stft_plan* plan, int64_t n, int64_t noverlap, int64_t nfft, const double* window_host, double r,
                          int onesided, int psd_only, int dtype, int engine, void* stream);
int mdsp_plan_cache_stats(int64_t* entries, int64_t* hits, int64_t* misses);
int mdsp_plan_cache_partitions(int64_t* partitions, int64_t* reaped);
int mdsp_plan_cache_set_context(uint64_t id);
int mdsp_plan_cache_release_context(uint64_t id);
int mdsp_plan_cache_clear(void);

/* ------------------------------------------------------------------------------------------------------
 * Measurement helpers (used by bench.py; not part of the drop-in surface)
 * ---------------------------------------------------------------------------------------------------- */
/* Times `reps` back-to-back launches of the last exec recorded on a plan with HIP events on `stream`. */
int mdsp_event_create(void** ev);
int mdsp_event_destroy(void* ev);
int mdsp_event_record(void* ev, void* stream);
int mdsp_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms);
/* float4 device copy kernel: the on-box achievable-HBM yardstick (bytes read + written = 2*bytes). */
int mdsp_copy_bench(void* dst_dev, const void* src_dev, size_t bytes, void* stream);
/* The same yardstick with an explicit access pattern: mode 0 float4 copy (grid-stride), 1 four float loads/stores, 2 nontemporal float4,
 * 3 one contiguous chunk per workgroup, 4 read-only (float4 loads summed), 5 write-only; wgs = workgroups per CU (< 1: 8). */
int mdsp_copy_bench_mode(void* dst_dev, const void* src_dev, size_t bytes, int mode, int wgs, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355DSP_H */
