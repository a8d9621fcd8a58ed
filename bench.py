#!/usr/bin/env python3
"""bench.py -- benchmark of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config filtwelch|stft|resample]
        N > 1: one rank per GPU.  Either the caller launches the ranks (python -m torch.distributed.run --nproc-per-node N ... bench.py
        --gpus N: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment), or -- when no launcher environment is present --
        this script launches them ITSELF (re-exec under torch.distributed.run on 127.0.0.1).  Fewer than N visible devices: exit code 2 and
        no JSON line (never an `n_gpus: 1` line for --gpus 8).

--config filtwelch (default; the BASELINE.json metric "Gsamples/s filt+welch, 1 Gsample Float32 stream"):
    one step = one pass of the hot path over one 2^30-sample Float32 stream per GPU, already resident in HBM:
        y = fftfilt(b, x)          256-tap overlap-save FIR             (BASELINE config 2, Filters/filt.jl:479-521)
        P = welch_pgram(x)         nfft = 4096, hanning, 50 % overlap   (BASELINE config 3, periodograms.jl:746-759)
        mean over channels         one RCCL all-reduce of 2049 floats   (N > 1: one channel (stream) per GPU)
    value = (2^30 samples x N) / step time, i.e. input samples that went through BOTH filt and Welch per second.
--config stft      BASELINE config 4: stft nfft = 1024, hop = 256, ComplexF32; 8 channels x 2^26 samples per GPU (64 channels on 8 GPUs);
                   channels are independent: NO collective (periodograms.jl:872-897 is per-vector).
--config resample  BASELINE config 5: FIRFilter 160//147, 5120 taps (32 per phase), Float32; 4 channels x 2^28 samples per GPU
                   (32 channels on 8 GPUs) + one RCCL all-reduce for the cross-channel average of the last output block.

The collective goes through the library's own communicator (mdsp_comm_* / mdsp_welch_mean_allreduce: RCCL behind the C ABI, the
entry points a Julia host binds); torch.distributed ("nccl" = RCCL) carries the bootstrap id, the barrier and the max-over-ranks
of the timing, and is the fallback transport should the library communicator fail to initialise (reported in `config.collective`).

The JSON line also carries
    roofline      the dominant kernel: algorithmic bytes / its mean launch duration, measured live with HIP events on
                  the launch stream; `traffic` = HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes -- measured
                  live by re-running this script under rocprofv3 (N = 1, rocprofv3 on PATH), else the committed profiles/pmc_latest.json;
                  `traffic_source` says which.
    kernels       the other kernel of the step, the float4 copy / read yardsticks, and (N = 1) the remaining SURVEY section-8 rows at their
                  single-GPU shares: stft, spectrogram, resample, firarb -- each with ms, algorithmic GB/s and frac of 8 TB/s.
    host_path     (N = 1) the host-pointer entry points (mdsp_{ols,welch,stft,fir}_exec_host: three-stage H2D || kernel || D2H pipeline): PCIe-inclusive
                  rates on a bounded sample, reported SEPARATELY -- never part of `value`.
    cpu_baseline  the CPU oracle (numpy/scipy restatement of DSP.jl's algorithm; kind "port") timed on a bounded sample: 1 thread
                  (`value`), all cores (`multi`), and the line-faithful per-block loop (`faithful`).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured float4-copy rate
METRIC = "Gsamples/s filt+welch, 1 Gsample Float32 stream; achieved HBM GB/s vs roofline"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node (default: WORLD_SIZE when launched by torchrun, else 1)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=["filtwelch", "stft", "resample"], default="filtwelch")
    ap.add_argument("--log2n", type=int, default=0, help="samples per channel = 2^log2n (default: 30 / 26 / 28 by config, BASELINE sizes)")
    ap.add_argument("--engine", choices=["auto", "fused", "rocfft"], default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-log2n", type=int, default=29, help="CPU baseline sample = 2^k samples")
    ap.add_argument("--no-rows", action="store_true", help="skip the other section-8 rows (kernels.stft / spectrogram / resample / firarb)")
    ap.add_argument("--no-host", action="store_true", help="skip the host-pointer (PCIe-inclusive) measurement")
    ap.add_argument("--host-log2n", type=int, default=28)
    ap.add_argument("--no-live-pmc", action="store_true", help="do not re-run under rocprofv3 for roofline.traffic")
    ap.add_argument("--describe-rows", action="store_true", help="print what every row of `kernels` runs (ROW_INFO) and exit")
    ap.add_argument("--dry", action="store_true", help="CPU dry mode (tests): gloo, no device work; exercises sharding, the collective and the JSON schema")
    return ap.parse_args()


def lowpass_taps(n):
    import dsp_jl_amd as d
    return d.design.lowpass_firwindow(0.25, d.hamming(n), fs=1.0).astype("float32")


def resample_taps():
    import numpy as np
    from fractions import Fraction
    import dsp_jl_amd as d
    h = d.resample_filter(Fraction(160, 147))
    return (np.concatenate([h, np.zeros(5120 - len(h))]) if len(h) < 5120 else h[:5120]).astype(np.float32)


# ------------------------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(log2n: int):
    """Oracle timed on the host, same workload, bounded sample: 1 thread, all cores, and the line-faithful block loop."""
    import numpy as np
    import scipy.fft as sfft
    from oracle import filt as ofilt, periodograms as opg, windows as ow
    try:
        import threadpoolctl
        ctl = threadpoolctl.threadpool_limits(1)
    except Exception:
        ctl = None
    ncores = os.cpu_count() or 1
    b = np.asarray(lowpass_taps(256))
    nb, nfft = 256, 2048
    L = nfft - nb + 1
    H = sfft.rfft(np.concatenate([b / np.float32(nfft), np.zeros(nfft - nb, np.float32)]))

    def run(n, workers):
        seg = min(n, 1 << 24)          # the stream is walked in 2^24-sample segments to bound host memory (~1.5 GB)
        rng = np.random.default_rng(1776)
        t_filt = t_welch = 0.0
        with sfft.set_workers(workers):
            for s0 in range(0, n, seg):
                x = rng.standard_normal(seg, dtype=np.float32)
                # vectorised form of the oracle's block loop (same algorithm: rfft -> *H -> irfft per 2048-point block), batched
                t0 = time.perf_counter()
                xp = np.concatenate([np.zeros(nb - 1, np.float32), x, np.zeros(nfft, np.float32)])
                nblk = -(-seg // L)
                y = np.empty(nblk * L, np.float32)
                CH = 2048
                for k0 in range(0, nblk, CH):
                    k1 = min(nblk, k0 + CH)
                    idx = (np.arange(k0, k1) * L)[:, None] + np.arange(nfft)[None, :]
                    blk = sfft.irfft(sfft.rfft(xp[idx], axis=1) * H, nfft, axis=1) * np.float32(nfft)
                    y[k0 * L:k1 * L] = blk[:, nb - 1:].reshape(-1)
                t_filt += time.perf_counter() - t0
                if s0 == 0:  # spot-check the batched form against the line-faithful oracle on a prefix
                    ref = ofilt._fftfilt(b, x[:20000], nfft)
                    assert np.allclose(y[:20000], ref, rtol=1e-4, atol=1e-5)
                t0 = time.perf_counter()
                opg.welch_pgram(x, 4096, 2048, window=ow.hanning)
                t_welch += time.perf_counter() - t0
                del xp, y
        return t_filt, t_welch

    n = 1 << log2n
    tf1, tw1 = run(n, 1)
    tfm, twm = run(n, ncores)
    # the line-faithful oracle (one Python iteration per 2048-point block, exactly Filters/filt.jl:504-518) on a smaller sample
    nf = 1 << min(log2n, 24)
    xf = np.random.default_rng(1776).standard_normal(nf, dtype=np.float32)
    t0 = time.perf_counter()
    ofilt._fftfilt(b, xf, nfft)
    t_faith_filt = time.perf_counter() - t0
    t0 = time.perf_counter()
    opg.welch_pgram(xf, 4096, 2048, window=ow.hanning, sequential=True)
    t_faith_welch = time.perf_counter() - t0
    if ctl is not None and hasattr(ctl, "restore_original_limits"):
        ctl.restore_original_limits()
    return {"value": round(n / (tf1 + tw1) / 1e9, 5), "unit": "Gsamples/s", "cores": 1, "kind": "port",
            "sample": f"2^{log2n} Float32 samples: filt {tf1:.2f}s + welch {tw1:.2f}s; numpy/scipy(pocketfft) restatement of DSP.jl, not DSP.jl/FFTW; host has {ncores} cores",
            "multi": {"value": round(n / (tfm + twm) / 1e9, 5), "cores": ncores, "sample": f"same, scipy.fft workers={ncores}: {tfm:.2f}s + {twm:.2f}s"},
            "faithful": {"value": round(nf / (t_faith_filt + t_faith_welch) / 1e9, 5), "cores": 1,
                         "sample": f"2^{min(log2n, 24)} samples, line-faithful block / frame loops: {t_faith_filt:.2f}s + {t_faith_welch:.2f}s"}}


def cpu_baseline_config(config: str, log2n: int):
    """CPU baseline of the --config stft / resample lines: the oracle (numpy restatement of DSP.jl, one thread) on a bounded single-channel
    sample of the same workload; for the resampler also SciPy's compiled polyphase routine as a second opinion."""
    import numpy as np
    from fractions import Fraction
    try:
        import threadpoolctl
        ctl = threadpoolctl.threadpool_limits(1)
    except Exception:
        ctl = None
    ncores = os.cpu_count() or 1
    n = 1 << log2n
    rng = np.random.default_rng(1776)
    if config == "resample":
        from oracle import stream_filt as osf
        h = resample_taps()
        x = rng.standard_normal(n, dtype=np.float32)
        t0 = time.perf_counter()
        y = osf.FIRFilter(h, Fraction(160, 147)).filt(x)
        t = time.perf_counter() - t0
        res = {"value": round(n / t / 1e9, 6), "unit": "Gsamples/s", "cores": 1, "kind": "port",
               "sample": f"one channel of 2^{log2n} Float32 samples, 160//147, 5120 taps: oracle FIRFilter.filt (numpy restatement of stream_filt.jl:476-515, "
                         f"not DSP.jl/BLAS) {t:.2f}s -> {len(y)} outputs; host has {ncores} cores"}
        try:
            import scipy.signal as ss
            t0 = time.perf_counter()
            ss.upfirdn(h, x, 160, 147)
            t2 = time.perf_counter() - t0
            res["compiled"] = {"value": round(n / t2 / 1e9, 6), "cores": 1, "sample": f"same sample through scipy.signal.upfirdn (compiled polyphase loop, different edge handling): {t2:.2f}s"}
        except Exception as e:  # pragma: no cover
            res["compiled"] = {"value": None, "sample": f"scipy.signal.upfirdn failed: {e}"}
    else:
        from oracle import periodograms as opg, windows as ow
        s = (rng.standard_normal(n, dtype=np.float32) + 1j * rng.standard_normal(n, dtype=np.float32)).astype(np.complex64)
        t0 = time.perf_counter()
        S = opg.stft(s, 1024, 768, window=ow.hanning, onesided=False)
        t = time.perf_counter() - t0
        res = {"value": round(n / t / 1e9, 6), "unit": "Gsamples/s", "cores": 1, "kind": "port",
               "sample": f"one channel of 2^{log2n} ComplexF32 samples, nfft 1024, hop 256: oracle stft (numpy / pocketfft restatement of periodograms.jl:872-897, "
                         f"not DSP.jl/FFTW) {t:.2f}s -> {S.shape[1]} columns; host has {ncores} cores"}
    if ctl is not None and hasattr(ctl, "restore_original_limits"):
        ctl.restore_original_limits()
    return res


# ------------------------------------------------------------------------------------------------------------------ helpers
class Timer:
    """HIP events on the launch stream (torch.cuda.Event would only see torch's current stream; these are recorded on the
    stream handed to the library)."""

    def __init__(self, lib, _lib, stream):
        self.lib, self._lib, self.stream = lib, _lib, stream

    def ev(self):
        e = C.c_void_p()
        self._lib.check(self.lib.mdsp_event_create(C.byref(e)))
        return e

    def rec(self, e):
        self._lib.check(self.lib.mdsp_event_record(e, self.stream))

    def ms(self, a, b):
        v = C.c_float()
        self._lib.check(self.lib.mdsp_event_elapsed_ms(a, b, C.byref(v)))
        return v.value

    def time(self, fn, reps=9):
        """Median and best of `reps` timed launches; `self.last` keeps min / max / reps of the same sample for the row.  Every timed launch directly
        follows an untimed one on the same stream -- steady state, as in the headline's K back-to-back steps: a launch behind an idle gap (the
        synchronisation between repetitions) pays the clock ramp of the idle part, 5-10 % of a 2 ms kernel."""
        import torch
        fn()
        torch.cuda.synchronize()
        a, b = self.ev(), self.ev()
        ts = []
        for _ in range(reps):
            fn(); self.rec(a); fn(); self.rec(b)
            torch.cuda.synchronize()
            ts.append(self.ms(a, b))
        ts.sort()
        self.last = {"min_ms": round(ts[0], 4), "max_ms": round(ts[-1], 4), "reps": reps}
        return ts[len(ts) // 2], ts[0]


class Marks:
    """Row markers for the profiler: before each measured row, one mdsp_fill_kernel launch with (100 + row index) workgroups on the launch stream
    (tools/prof_summary.py rows() cuts the dispatch list at them).  ~2 us each, always outside the timed regions."""
    ROWS = ("step", "yardsticks", "stft", "spectrogram", "resample", "firarb", "resample_f64", "resample_c32", "interp2_f32", "decim2_f32",
            "welch_3000", "welch_1536", "filt_5120", "decim8_f32", "resample_147_160_f32", "resample_160_441_f64",
            # round 5
            "decim16_f32", "resample_441_160_c64", "welch_default", "welch_default_2p24", "spectrogram_default", "filt_32768", "filt_f64", "welch_f64",
            "welch_f64_5000", "welch_f64_8000", "mt_pgram", "hilbert", "conv2d", "filtfilt", "welch_2p19",
            "welch_8192", "welch_12500", "welch_16384", "welch_65536", "welch_125000", "welch_200000", "spectrogram_8400")

    def __init__(self, lib, _lib, stream):
        import torch
        self.lib, self._lib, self.stream = lib, _lib, stream
        self.buf = torch.empty(16 * 256 * (100 + len(self.ROWS)) // 4, dtype=torch.float32, device="cuda")

    def __call__(self, row):
        nwg = 100 + self.ROWS.index(row)
        self._lib.check(self.lib.mdsp_copy_bench_mode(self.buf.data_ptr(), self.buf.data_ptr(), 16 * 256 * nwg, 5, 8, self.stream))


def roof(kernel, ms, alg_bytes, traffic=None, extra=None):
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    r = {"bound": "hbm", "kernel": kernel, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
         "traffic": traffic, "ms_per_launch": round(ms, 4), "algorithmic_bytes_per_launch": int(alg_bytes)}
    if extra:
        r.update(extra)
    return r


def crow(tm, med, alg_bytes, **extra):
    """One compact row of `kernels`: median ms of the timer's last sample, algorithmic bytes / time as a fraction of the 8 TB/s HBM peak, the sample's fastest ms.
    (What each row runs is in ROW_INFO below and in DESIGN.md section 5: the line stays small enough for the driver's record to hold every row.)"""
    gbs = alg_bytes / (med * 1e-3) / 1e9
    last = getattr(tm, "last", {})
    return {"ms": round(med, 4), "frac": round(gbs / HBM_PEAK_GBS, 4), "min": last.get("min_ms"), **extra}   # GB/s = frac x 8000; min: the fastest of the sample


# row -> (what runs, shape, algorithmic bytes per input sample); printed by `python bench.py --describe-rows`, not in the JSON line
ROW_INFO = {
    "stft": ("stft_fused_kernel", "config 4 share: 8 ch x 2^26 ComplexF32, nfft 1024, hop 256", "40"),
    "spectrogram": ("stft_fused_kernel (PSD form)", "same signal", "24"),
    "resample": ("polyphase_mfma_kernel", "config 5 share: 4 ch x 2^28 Float32, 160//147, 5120 taps", "8.354"),
    "firarb": ("arbitrary_fir_kernel", "row f1: 4 ch x 2^28 Float32, rate 160/147 as Float64, warm trajectory", "8.354"),
    "resample_f64": ("polyphase kernel", "4 ch x 2^26 Float64, 160//147, 5120 taps", "16.7"),
    "resample_c32": ("polyphase kernel", "4 ch x 2^26 ComplexF32, 160//147, 5120 taps", "16.7"),
    "interp2_f32": ("polyphase kernel", "4 ch x 2^26 Float32, 2//1, resample_filter taps", "12"),
    "decim2_f32": ("polyphase kernel", "4 ch x 2^26 Float32, 1//2", "6"),
    "decim8_f32": ("decimator kernel (round 5) / matrix cores before", "4 ch x 2^26 Float32, 1//8", "4.5"),
    "decim16_f32": ("decimator kernel (round 5)", "4 ch x 2^26 Float32, 1//16", "4.25"),
    "resample_147_160_f32": ("polyphase kernel", "4 ch x 2^26 Float32, 147//160", "7.675"),
    "resample_160_441_f64": ("polyphase kernel", "4 ch x 2^26 Float64, 160//441", "10.9"),
    "resample_441_160_c64": ("polyphase kernel", "4 ch x 2^26 ComplexF64, 441//160", "60.1"),
    "welch_3000": ("gen_ct_kernel<3000>", "2^27 Float32, n = nfft = 3000, 50 % overlap", "4"),
    "welch_1536": ("gen_ct_kernel<1536>", "2^27 Float32, n = nfft = 1536", "4"),
    "welch_default": ("multi-pass engine (bigfft.hip)", "welch_pgram(s) with DEFAULT arguments, 2^27 Float32: n = nfft = 2^24, 15 frames", "4"),
    "welch_default_2p24": ("multi-pass engine", "welch_pgram(s), 2^24 Float32: n = nfft = 2^21", "4"),
    "welch_8192": ("welch_half_kernel<8192>", "2^27 Float32, n = nfft = 8192, 50 % overlap (the largest register-resident power of two: the yardstick of the sizes above it)", "4"),
    "welch_12500": ("gen_ct_kernel<12500 = 25 20 25>, lean form (one workgroup, one LDS buffer: csrc/spectral_ctbig.hip)", "2^27 Float32, n = nfft = 12500 = nextfastfft(10^5 >> 3), 50 % overlap", "4"),
    "welch_16384": ("gen_ct_kernel<16384 = 32 32 16>, lean form (one workgroup of 512 threads, 134 KiB of LDS: csrc/spectral_ctbig.hip)", "2^27 Float32, n = nfft = 16384, 50 % overlap", "4"),
    "welch_65536": ("gen_ct_cols_kernel<16384>, 4 x 16384 (column step fused into the loads: csrc/spectral_ctcols_big.hip; the multi-pass engine's 256 x 256 measured 0.44 TB/s)", "2^27 Float32, n = nfft = 65536, 50 % overlap", "4"),
    "welch_125000": ("rows_col_kernel<8> + gen_ct_kernel<15625 = 25 25 25> over the rows: 8 x 15625 in two kernels (csrc/spectral_ctrows.hip; the multi-pass engine's 250 x 500 measured 0.25 TB/s)", "2^27 Float32, n = nfft = 125000 = nextfastfft(10^6 >> 3), 50 % overlap", "4"),
    "welch_200000": ("rows_col_kernel<16> + gen_ct_kernel<12500> over the rows: 16 x 12500 in two kernels (csrc/spectral_ctrows.hip; the multi-pass engine measured 0.15 TB/s)", "2^27 Float32, n = nfft = 200000 = nextfastfft(1.6 10^6 >> 3), 50 % overlap", "4"),
    "welch_2p19": ("rows_col_kernel<32> + gen_ct_kernel<16384> over the rows: 32 x 16384 in two kernels (round 5: the multi-pass engine's rows form, 64 x 8192, 0.56 TB/s)", "2^27 Float32, n = nfft = 2^19, 50 % overlap: 511 frames", "4"),
    "spectrogram_8400": ("gen_ct_kernel<8400 = 20 20 21>, column mode (one workgroup, two LDS buffers: csrc/spectral_ctbig_cols.hip; round 5: rocFFT pipeline 0.29 TB/s)", "2^27 Float32 -> 4201 x 31955 Float32, n = nfft = 8400, 50 % overlap", "4 in + 4 per bin out"),
    "spectrogram_default": ("multi-pass engine + untangle", "spectrogram(s) with DEFAULT arguments, 2^27 Float32 -> (2^23 + 1) x 15 Float32", "4 in + 4 per bin out"),
    "filt_5120": ("upols2_fused_kernel", "filt, 5120 taps, 2^28 Float32", "8"),
    "filt_32768": ("d.filt(b, x) through the host mirror: ONE plan, rows form of the multi-pass engine (column pass, row kernel, column pass back)", "32768 taps, 2^27 Float32", "8"),
    "filt_f64": ("ols_fused_kernel<double>", "the headline filt in Float64: 256 taps, nfft 2048, 2^29 samples", "16"),
    "welch_f64": ("welch_half_kernel<double>", "the headline welch_pgram in Float64: nfft 4096, 2^29 samples", "8"),
    "welch_f64_5000": ("gen_ct_kernel<double, 5000>", "2^27 Float64, n = nfft = 5000", "8"),
    "welch_f64_8000": ("gen_ct_kernel<double, 8000>", "2^27 Float64, n = nfft = 8000", "8"),
    "mt_pgram": ("d.mt_pgram(x; nw = 4, ntapers = 7) through the host mirror (one multi-pass transform per taper)", "2^22 Float32 (the DPSS tapers are a host eigenproblem: 2^26 would take minutes to set up)", "4"),
    "hilbert": ("mdsp_hilbert (rocFFT forward, spectrum kernel, rocFFT inverse, scale)", "2^27 Float32 -> ComplexF32", "12"),
    "conv2d": ("d.conv(A, B): mdsp_convnd_fft", "4096 x 4096 Float32 with a 33 x 33 kernel", "4 in + 4 out"),
    "filtfilt": ("d.filtfilt(b, x) through the host mirror", "256 taps, 2^26 Float32", "8"),
}


def spread(tm):
    """min / max / reps of the timer's last sample (the row's median is ms_per_launch)."""
    return dict(getattr(tm, "last", {}))


def git_head():
    """HEAD of the checkout; on the GPU box (a snapshot without .git) the hash the builder wrote into .build_commit before sending it."""
    try:
        h = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip()
        if h:
            return h
    except Exception:
        pass
    try:
        return open(os.path.join(ROOT, ".build_commit")).read().strip()[:12] or None
    except Exception:
        return None


def committed_traffic():
    p = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        d = json.load(open(p))
        return d, f"profiles/pmc_latest.json (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py @ {d.get('commit', 'round-1 HEAD a2de93b')}; NOT measured in this run)"
    except Exception:
        return {}, None


def live_traffic(args):
    """Re-run this script (2 steps, no extras) under rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, kernel-trace only, as
    MI355X_MICROARCH.md prescribes) and return HBM bytes per launch of the fused kernels.  FETCH_SIZE is doubled (gfx950 correction)."""
    import shutil
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import prof_summary
    tmp = tempfile.mkdtemp(prefix="mdsp_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    out = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(tmp, ctr)
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "-d", d, "-o", "bench", "--", sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2",
               "--warmup", "1", "--no-cpu-baseline", "--no-host", "--no-live-pmc", "--config", args.config, "--engine", args.engine] + (["--no-rows"] if args.no_rows else [])
        try:
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                p.wait(timeout=240)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, 9)           # our own child process group only
                return None, f"rocprofv3 --pmc {ctr} pass timed out"
        except Exception as e:
            return None, f"rocprofv3 failed to start: {e}"
        dbs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
        if not dbs:
            return None, f"rocprofv3 --pmc {ctr} produced no database (rc {p.returncode})"
        out[ctr] = dbs[0]
    try:
        res = prof_summary.traffic_rows(out["FETCH_SIZE"], out["WRITE_SIZE"])     # per bench row (marker-separated), dominant kernel of the row
    except Exception as e:
        return None, f"cannot read the PMC databases: {e}"
    shutil.rmtree(tmp, ignore_errors=True)
    # bytes = 2 FETCH_SIZE 1024 + WRITE_SIZE 1024 (gfx950 FETCH correction, MI355X_MICROARCH.md); each row's dispatches sit between marker launches
    return res, "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (2 steps), bytes = 2*FETCH*1024 + WRITE*1024"


# ------------------------------------------------------------------------------------------------------------------ rows / host path
def measure_rows(tm, lib, _lib, d, stream, mark=lambda row: None):
    """The remaining SURVEY section-8 rows (device-resident, HIP events on the launch stream, median of 9 back-to-back pairs): what each row runs
    is in ROW_INFO; every row comes back as crow()'s compact dict."""
    import numpy as np
    import torch
    from fractions import Fraction
    from dsp_jl_amd.periodograms import _StftPlan, compute_window
    from dsp_jl_amd.dspbase import OlsPlan
    rows = {}
    g = torch.Generator(device="cuda"); g.manual_seed(1776)

    def guarded(name, fn):
        """a row that fails must not take the others with it"""
        try:
            fn()
        except Exception as e:  # pragma: no cover
            rows[name] = {"error": str(e)[:120]}
        torch.cuda.empty_cache()

    # ---- config 4 share: stft / spectrogram
    nch, n = 8, 1 << 26
    s = torch.view_as_complex(torch.randn((nch, n, 2), generator=g, device="cuda", dtype=torch.float32) * math.sqrt(0.5))
    win, norm2 = compute_window(d.hanning, 1024)
    K = d.frame_count(n, 1024, 768)
    for name, psd, outdt, bps in (("stft", 0, torch.complex64, 40.0), ("spectrogram", 1, torch.float32, 24.0)):
        plan = _StftPlan(1024, 768, 1024, win, 1.0 * norm2, False, psd, np.complex64, d.ENGINE_FUSED)
        out = torch.empty((nch, K, 1024), dtype=outdt, device="cuda")
        mark(name)
        med, _ = tm.time(lambda: _lib.check(lib.mdsp_stft_exec(plan._h, s.data_ptr(), n, nch, n, out.data_ptr(), 1024, K * 1024, stream)))
        rows[name] = crow(tm, med, bps * n * nch)
        del out, plan
    del s
    torch.cuda.empty_cache()
    # ---- config 5 share: resample 160//147, and FIRArbitrary at the same shape
    nch, n = 4, 1 << 28
    h = resample_taps()
    x = torch.randn((nch, n), generator=g, device="cuda", dtype=torch.float32)
    fh = C.c_void_p()
    _lib.check(lib.mdsp_fir_create(C.byref(fh), h.ctypes.data_as(C.c_void_p), len(h), 160, 147, _lib.F32, _lib.F32, nch))
    ol = C.c_int64(); _lib.check(lib.mdsp_fir_outputlength(fh, n, C.byref(ol)))
    y = torch.empty((nch, ol.value + 1), dtype=torch.float32, device="cuda")
    nw = C.c_int64()

    def fir():
        _lib.check(lib.mdsp_fir_reset(fh))
        _lib.check(lib.mdsp_fir_exec(fh, x.data_ptr(), n, n, y.data_ptr(), ol.value, ol.value + 1, C.byref(nw), stream))

    mark("resample")
    med, _ = tm.time(fir)
    rows["resample"] = crow(tm, med, (4 + 4 * 160 / 147) * n * nch)
    _lib.check(lib.mdsp_fir_destroy(fh))
    rate = 160 / 147
    ha = d.resample_filter(rate, 32).astype(np.float32)
    fa = C.c_void_p()
    _lib.check(lib.mdsp_firarb_create(C.byref(fa), ha.ctypes.data_as(C.c_void_p), len(ha), rate, 32, _lib.F32, _lib.F32, nch))
    ola = C.c_int64(); _lib.check(lib.mdsp_firarb_outputlength(fa, n, C.byref(ola)))
    ya = y if ola.value + 1 <= y.shape[1] else torch.empty((nch, ola.value + 1), dtype=torch.float32, device="cuda")
    ld = ya.shape[1]

    def arb():
        _lib.check(lib.mdsp_firarb_reset(fa))
        _lib.check(lib.mdsp_firarb_exec(fa, x.data_ptr(), n, n, ya.data_ptr(), ola.value + 1, ld, C.byref(nw), stream))

    mark("firarb")
    med, _ = tm.time(arb)
    # the kernel's own roofs (it is nowhere near HBM's): 24 bytes of LDS reads per (output, tap) against 256 B / clock / CU, and four v_pk_fma_f32 against the
    # measured packed-FMA issue rate (profiles/r02t_valu_rate.txt), both at 2.4 GHz -- DESIGN.md section 4.7
    tpp = -(-len(ha) // 32)
    tfl = 2.0 * 2.0 * tpp * ola.value * nch / (med * 1e-3) / 1e12
    lds_tbs = float(tpp) * ola.value * -(-nch // 4) * (8 + 4 * min(nch, 4)) / (med * 1e-3) / 1e12
    rows["firarb"] = crow(tm, med, (4 + 4 * rate) * n * nch, lds_frac=round(lds_tbs / (256.0 * 256 * 2.4e9 / 1e12), 4),
                          valu_frac=round(tfl / (2.0 * 29.1 * 1024 * 2.4e9 / 1e12), 4))
    _lib.check(lib.mdsp_firarb_destroy(fa))
    del x, y, ya
    torch.cuda.empty_cache()
    # ---- the same resampler on the other signal types and ratios: 4 channels x 2^26 samples, DSP.jl's own resample_filter design per ratio
    n2 = 1 << 26
    for name, tdt, hdt, lt, lx, esz, (L, M) in (("resample_f64", torch.float64, np.float64, _lib.F64, _lib.F64, 8, (160, 147)),
                                                ("resample_c32", torch.complex64, np.float32, _lib.F32, _lib.C32, 8, (160, 147)),
                                                ("interp2_f32", torch.float32, np.float32, _lib.F32, _lib.F32, 4, (2, 1)),
                                                ("decim2_f32", torch.float32, np.float32, _lib.F32, _lib.F32, 4, (1, 2)),
                                                ("decim8_f32", torch.float32, np.float32, _lib.F32, _lib.F32, 4, (1, 8)),
                                                ("decim16_f32", torch.float32, np.float32, _lib.F32, _lib.F32, 4, (1, 16)),
                                                ("resample_147_160_f32", torch.float32, np.float32, _lib.F32, _lib.F32, 4, (147, 160)),
                                                ("resample_160_441_f64", torch.float64, np.float64, _lib.F64, _lib.F64, 8, (160, 441)),
                                                ("resample_441_160_c64", torch.complex128, np.float64, _lib.F64, _lib.C64, 16, (441, 160))):
        def one():
            hh = resample_taps().astype(hdt) if (L, M) == (160, 147) else np.asarray(d.resample_filter(Fraction(L, M)), dtype=hdt)
            xx = torch.randn((nch, n2), generator=g, device="cuda", dtype=tdt)
            f2 = C.c_void_p()
            _lib.check(lib.mdsp_fir_create(C.byref(f2), hh.ctypes.data_as(C.c_void_p), len(hh), L, M, lt, lx, nch))
            o2 = C.c_int64(); _lib.check(lib.mdsp_fir_outputlength(f2, n2, C.byref(o2)))
            yy = torch.empty((nch, o2.value + 1), dtype=tdt, device="cuda")
            pth = C.c_int(-1); _lib.check(lib.mdsp_fir_kernel_path(f2, n2, C.byref(pth)))

            def fir2():
                _lib.check(lib.mdsp_fir_reset(f2))
                _lib.check(lib.mdsp_fir_exec(f2, xx.data_ptr(), n2, n2, yy.data_ptr(), o2.value, o2.value + 1, C.byref(nw), stream))

            mark(name)
            med, _ = tm.time(fir2)
            rows[name] = crow(tm, med, esz * (1 + L / M) * n2 * nch, path=pth.value)   # 0 generic, 1 register taps, 2 matrix cores, 3 decimator kernel
            _lib.check(lib.mdsp_fir_destroy(f2))
        guarded(name, one)
    # ---- Welch at the reference's DEFAULT transform sizes (nfft = nextfastfft(n), util.jl:134, periodograms.jl:560): the compile-time mixed-radix
    #      schedules at 3000 / 1536, and (round 5) the DEFAULT call itself -- n = length >> 3 -- on the multi-pass engine
    n3 = 1 << 27
    xr = torch.randn(n3, generator=g, device="cuda", dtype=torch.float32)
    for nfft in (3000, 1536):
        cfg = d.WelchConfig(n3, np.float32, n=nfft, noverlap=nfft // 2, nfft=nfft, window=d.hanning, engine=d.ENGINE_FUSED)
        psd = torch.empty(cfg.nout, dtype=torch.float32, device="cuda")
        mark(f"welch_{nfft}")
        med, _ = tm.time(lambda: _lib.check(lib.mdsp_welch_exec(cfg._h, xr.data_ptr(), n3, 1, n3, psd.data_ptr(), cfg.nout, stream)))
        rows[f"welch_{nfft}"] = crow(tm, med, 4.0 * n3)
        del cfg, psd

    def welch_default(name, length):
        cfg = d.WelchConfig(length, np.float32, window=d.hanning)          # n = length >> 3, noverlap = n >> 1, nfft = nextfastfft(n)
        psd = torch.empty(cfg.nout, dtype=torch.float32, device="cuda")
        mark(name)
        med, _ = tm.time(lambda: _lib.check(lib.mdsp_welch_exec(cfg._h, xr.data_ptr(), length, 1, length, psd.data_ptr(), cfg.nout, stream)))
        rows[name] = crow(tm, med, 4.0 * length, engine=cfg.engine, nfft=cfg.nfft)

    guarded("welch_default", lambda: welch_default("welch_default", n3))
    guarded("welch_default_2p24", lambda: welch_default("welch_default_2p24", 1 << 24))

    def welch_2p19():
        cfg = d.WelchConfig(n3, np.float32, n=1 << 19, noverlap=1 << 18, nfft=1 << 19, window=d.hanning)
        psd = torch.empty(cfg.nout, dtype=torch.float32, device="cuda")
        mark("welch_2p19")
        med, _ = tm.time(lambda: _lib.check(lib.mdsp_welch_exec(cfg._h, xr.data_ptr(), n3, 1, n3, psd.data_ptr(), cfg.nout, stream)))
        rows["welch_2p19"] = crow(tm, med, 4.0 * n3, engine=cfg.engine)

    guarded("welch_2p19", welch_2p19)

    # (round 6) the sizes either side of the old cliff at 8192 points: nextfastfft sizes of 10^5- and 10^6-sample default calls, 2^14, 2^16
    def welch_n(nfft):
        cfg = d.WelchConfig(n3, np.float32, n=nfft, noverlap=nfft // 2, nfft=nfft, window=d.hanning)   # AUTO
        psd = torch.empty(cfg.nout, dtype=torch.float32, device="cuda")
        mark(f"welch_{nfft}")
        med, _ = tm.time(lambda: _lib.check(lib.mdsp_welch_exec(cfg._h, xr.data_ptr(), n3, 1, n3, psd.data_ptr(), cfg.nout, stream)))
        rows[f"welch_{nfft}"] = crow(tm, med, 4.0 * n3, engine=cfg.engine)

    for nfft_ in (8192, 12500, 16384, 65536, 125000, 200000):
        guarded(f"welch_{nfft_}", lambda nfft_=nfft_: welch_n(nfft_))

    def spectrogram_default():
        nn = n3 >> 3
        w_, norm2_ = compute_window(d.hanning, nn)
        plan = _StftPlan(nn, nn >> 1, nn, w_, norm2_, True, 1, np.float32, d.ENGINE_AUTO)
        KK = d.frame_count(n3, nn, nn >> 1)
        out = torch.empty((KK, plan.nout), dtype=torch.float32, device="cuda")
        mark("spectrogram_default")
        med, _ = tm.time(lambda: _lib.check(lib.mdsp_stft_exec(plan._h, xr.data_ptr(), n3, 1, n3, out.data_ptr(), plan.nout, KK * plan.nout, stream)))
        rows["spectrogram_default"] = crow(tm, med, 4.0 * n3 + 4.0 * KK * plan.nout, engine=plan.engine)

    guarded("spectrogram_default", spectrogram_default)

    # (round 6) spectrogram of a real signal at a nextfastfft size past 8192 points: n = nfft = 8400 = nextfastfft(67000 >> 3), 50 % overlap
    def spectrogram_8400():
        nn = 8400
        w_, norm2_ = compute_window(d.hanning, nn)
        plan = _StftPlan(nn, nn >> 1, nn, w_, norm2_, True, 1, np.float32, d.ENGINE_AUTO)
        KK = d.frame_count(n3, nn, nn >> 1)
        out = torch.empty((KK, plan.nout), dtype=torch.float32, device="cuda")
        mark("spectrogram_8400")
        med, _ = tm.time(lambda: _lib.check(lib.mdsp_stft_exec(plan._h, xr.data_ptr(), n3, 1, n3, out.data_ptr(), plan.nout, KK * plan.nout, stream)))
        rows["spectrogram_8400"] = crow(tm, med, 4.0 * n3 + 4.0 * KK * plan.nout, engine=plan.engine)

    guarded("spectrogram_8400", spectrogram_8400)

    def hilbert():
        out = torch.empty(n3, dtype=torch.complex64, device="cuda")
        mark("hilbert")
        med, _ = tm.time(lambda: _lib.check(lib.mdsp_hilbert(xr.data_ptr(), n3, 1, n3, _lib.F32, out.data_ptr(), n3, stream)))
        rows["hilbert"] = crow(tm, med, 12.0 * n3)

    guarded("hilbert", hilbert)

    def filt_long():
        taps = (np.random.default_rng(32768).standard_normal(32768) / math.sqrt(32768)).astype(np.float32)
        d.filt(taps, xr)                       # builds and caches the segment plans
        mark("filt_32768")
        med, _ = tm.time(lambda: d.filt(taps, xr))
        rows["filt_32768"] = crow(tm, med, 8.0 * n3)

    guarded("filt_32768", filt_long)

    def filtfilt():
        xs = xr[: 1 << 26]
        b = np.asarray(lowpass_taps(256))
        d.filtfilt(b, xs)
        mark("filtfilt")
        med, _ = tm.time(lambda: d.filtfilt(b, xs))
        rows["filtfilt"] = crow(tm, med, 8.0 * (1 << 26))

    guarded("filtfilt", filtfilt)

    def mt():
        xs = xr[: 1 << 22].contiguous()
        d.mt_pgram(xs, nw=4, ntapers=7)        # DPSS tapers (host eigenproblem) and the per-taper plans are built here, outside the timing
        mark("mt_pgram")
        med, _ = tm.time(lambda: d.mt_pgram(xs, nw=4, ntapers=7))
        rows["mt_pgram"] = crow(tm, med, 4.0 * (1 << 22), tapers=7)

    guarded("mt_pgram", mt)
    del xr
    torch.cuda.empty_cache()

    def conv2d():
        A = torch.randn((4096, 4096), generator=g, device="cuda", dtype=torch.float32)
        B = np.random.default_rng(33).standard_normal((33, 33)).astype(np.float32)
        d.conv(A, B)
        mark("conv2d")
        med, _ = tm.time(lambda: d.conv(A, B))
        rows["conv2d"] = crow(tm, med, 4.0 * 4096 * 4096 + 4.0 * (4096 + 32) ** 2)

    guarded("conv2d", conv2d)
    # ---- filt with a LONG filter: 5120 taps (partitioned overlap-save), 2^28 Float32
    n4 = 1 << 28
    xl = torch.randn(n4, generator=g, device="cuda", dtype=torch.float32)
    yl = torch.empty_like(xl)
    tl = (np.random.default_rng(5120).standard_normal(5120) / math.sqrt(5120)).astype(np.float32)
    pl = OlsPlan(tl, d.optimalfftfiltlength(5120, n4), n4, 0, d.ENGINE_FUSED)
    mark("filt_5120")
    med, _ = tm.time(lambda: _lib.check(lib.mdsp_ols_exec(pl._h, xl.data_ptr(), n4, 1, n4, yl.data_ptr(), n4, n4, stream)))
    rows["filt_5120"] = crow(tm, med, 8.0 * n4)
    del pl, xl, yl
    torch.cuda.empty_cache()

    # ---- Float64 (DSP.jl's default eltype: util.jl:92-104): the headline pair at 2^29 samples, Welch at the 5000- and 8000-point nextfastfft sizes
    def f64_rows():
        n5 = 1 << 29
        xd = torch.randn(n5, generator=g, device="cuda", dtype=torch.float64)
        yd = torch.empty_like(xd)
        pl64 = OlsPlan(np.asarray(lowpass_taps(256), dtype=np.float64), 2048, n5, _lib.OLS_FILT, d.ENGINE_AUTO)
        mark("filt_f64")
        med, _ = tm.time(lambda: _lib.check(lib.mdsp_ols_exec(pl64._h, xd.data_ptr(), n5, 1, n5, yd.data_ptr(), n5, n5, stream)))
        rows["filt_f64"] = crow(tm, med, 16.0 * n5)
        del yd, pl64
        cfg = d.WelchConfig(n5, np.float64, n=4096, noverlap=2048, window=d.hanning)
        psd = torch.empty(cfg.nout, dtype=torch.float64, device="cuda")
        mark("welch_f64")
        med, _ = tm.time(lambda: _lib.check(lib.mdsp_welch_exec(cfg._h, xd.data_ptr(), n5, 1, n5, psd.data_ptr(), cfg.nout, stream)))
        rows["welch_f64"] = crow(tm, med, 8.0 * n5)
        n6 = 1 << 27
        for nfft in (5000, 8000):
            cfg = d.WelchConfig(n6, np.float64, n=nfft, noverlap=nfft // 2, nfft=nfft, window=d.hanning)
            psd = torch.empty(cfg.nout, dtype=torch.float64, device="cuda")
            mark(f"welch_f64_{nfft}")
            med, _ = tm.time(lambda: _lib.check(lib.mdsp_welch_exec(cfg._h, xd.data_ptr(), n6, 1, n6, psd.data_ptr(), cfg.nout, stream)))
            rows[f"welch_f64_{nfft}"] = crow(tm, med, 8.0 * n6, engine=cfg.engine)

    guarded("f64", f64_rows)
    return rows




def measure_host_path(lib, _lib, d, log2n):
    """mdsp_ols_exec_host / mdsp_welch_exec_host on a bounded sample: pageable numpy arrays (staged) and page-locked arrays."""
    import numpy as np
    from dsp_jl_amd.dspbase import OlsPlan
    n = 1 << log2n
    rng = np.random.default_rng(1776)
    x = rng.standard_normal(n, dtype=np.float32)
    plan = OlsPlan(np.asarray(lowpass_taps(256)), 2048, n, _lib.OLS_FILT, d.ENGINE_AUTO)
    cfg = d.WelchConfig(n, np.float32, n=4096, noverlap=2048, window=d.hanning)
    res = {"sample": f"2^{log2n} Float32 samples; wall clock of the synchronous host-pointer calls, PCIe included; never part of `value`", "unit": "Gsamples/s"}

    def wall(fn, reps=3):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        return min(ts)

    y = np.empty_like(x)
    psd = np.empty(cfg.nout, np.float32)
    xp, yp, pp = x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), psd.ctypes.data_as(C.c_void_p)
    t = wall(lambda: _lib.check(lib.mdsp_ols_exec_host(plan._h, xp, n, 1, n, yp, n, n, 0)))
    res["filt_pageable"] = {"Gsamples_per_s": round(n / t / 1e9, 3), "GBps_pcie_both_ways": round(8.0 * n / t / 1e9, 1)}
    t = wall(lambda: _lib.check(lib.mdsp_welch_exec_host(cfg._h, xp, n, 1, n, pp, cfg.nout, 0)))
    res["welch_pageable"] = {"Gsamples_per_s": round(n / t / 1e9, 3), "GBps_pcie_h2d": round(4.0 * n / t / 1e9, 1)}
    pin_in, pin_out = C.c_void_p(), C.c_void_p()
    _lib.check(lib.mdsp_host_alloc(C.byref(pin_in), n * 4)); _lib.check(lib.mdsp_host_alloc(C.byref(pin_out), n * 4))
    try:
        C.memmove(pin_in, xp, n * 4)
        t = wall(lambda: _lib.check(lib.mdsp_ols_exec_host(plan._h, pin_in, n, 1, n, pin_out, n, n, _lib.HOST_PINNED)))
        res["filt_pinned"] = {"Gsamples_per_s": round(n / t / 1e9, 3), "GBps_pcie_both_ways": round(8.0 * n / t / 1e9, 1)}
        t = wall(lambda: _lib.check(lib.mdsp_welch_exec_host(cfg._h, pin_in, n, 1, n, pp, cfg.nout, _lib.HOST_PINNED)))
        res["welch_pinned"] = {"Gsamples_per_s": round(n / t / 1e9, 3), "GBps_pcie_h2d": round(4.0 * n / t / 1e9, 1)}
    finally:
        lib.mdsp_host_free(pin_in); lib.mdsp_host_free(pin_out)
    del x, y
    # the other two hot entries from host memory (round 3): config 4's stft (output 4x the input: the D2H direction is the bound) and
    # config 5's resampler, page-locked arrays, bounded single-channel samples
    try:
        from fractions import Fraction
        from dsp_jl_amd.periodograms import _StftPlan, compute_window
        ns = 1 << (log2n - 3)
        win, norm2 = compute_window(d.hanning, 1024)
        sp = _StftPlan(1024, 768, 1024, win, 1.0 * norm2, False, 0, np.complex64, d.ENGINE_AUTO)
        K = d.frame_count(ns, 1024, 768)
        pi, po = C.c_void_p(), C.c_void_p()
        _lib.check(lib.mdsp_host_alloc(C.byref(pi), ns * 8)); _lib.check(lib.mdsp_host_alloc(C.byref(po), K * 1024 * 8))
        try:
            sig = (rng.standard_normal(ns, dtype=np.float32) + 1j * rng.standard_normal(ns, dtype=np.float32)).astype(np.complex64)
            C.memmove(pi, sig.ctypes.data_as(C.c_void_p), ns * 8)
            t = wall(lambda: _lib.check(lib.mdsp_stft_exec_host(sp._h, pi, ns, 1, ns, po, 1024, K * 1024, _lib.HOST_PINNED)))
            res["stft_pinned"] = {"Gsamples_per_s": round(ns / t / 1e9, 3), "GBps_pcie_d2h": round(8.0 * K * 1024 / t / 1e9, 1), "log2n": log2n - 3}
        finally:
            lib.mdsp_host_free(pi); lib.mdsp_host_free(po)
        nr = 1 << (log2n - 1)
        h = resample_taps()
        fh = C.c_void_p()
        _lib.check(lib.mdsp_fir_create(C.byref(fh), h.ctypes.data_as(C.c_void_p), len(h), 160, 147, _lib.F32, _lib.F32, 1))
        ol = C.c_int64(); _lib.check(lib.mdsp_fir_outputlength(fh, nr, C.byref(ol)))
        pi, po = C.c_void_p(), C.c_void_p()
        _lib.check(lib.mdsp_host_alloc(C.byref(pi), nr * 4)); _lib.check(lib.mdsp_host_alloc(C.byref(po), (ol.value + 1) * 4))
        try:
            xr = rng.standard_normal(nr, dtype=np.float32)
            C.memmove(pi, xr.ctypes.data_as(C.c_void_p), nr * 4)
            nw = C.c_int64()

            def fir_host():
                _lib.check(lib.mdsp_fir_reset(fh))
                _lib.check(lib.mdsp_fir_exec_host(fh, pi, nr, nr, po, ol.value, ol.value + 1, C.byref(nw), _lib.HOST_PINNED))

            t = wall(fir_host)
            res["resample_pinned"] = {"Gsamples_per_s": round(nr / t / 1e9, 3), "GBps_pcie_both_ways": round(4.0 * (nr + ol.value) / t / 1e9, 1), "log2n": log2n - 1}
        finally:
            lib.mdsp_host_free(pi); lib.mdsp_host_free(po)
            lib.mdsp_fir_destroy(fh)
    except Exception as e:  # pragma: no cover
        res["rows_error"] = str(e)
    return res


# ------------------------------------------------------------------------------------------------------------------ main
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """--gpus N > 1 without a launcher environment: start the N ranks ourselves (one process per GPU) under torch.distributed.run, rendezvous on
    127.0.0.1, same command line; the children see WORLD_SIZE and take the normal path.  Returns the launcher's exit code."""
    if not args.dry:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {have} HIP device(s) visible; refusing to print a line for fewer GPUs than asked", file=sys.stderr, flush=True)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.describe_rows:
        for k, (what, shape, bps) in ROW_INFO.items():
            print(f"{k:24s} {what}; {shape}; {bps} B per input sample")
        return
    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    world = int(os.environ.get("WORLD_SIZE", "1")) if launched else 1
    if args.gpus is None:
        args.gpus = world
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        sys.exit(2)
    if not launched and args.gpus > 1:
        sys.exit(self_launch(args))
    if launched and world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} contradicts the launcher's WORLD_SIZE={world}", file=sys.stderr, flush=True)
        sys.exit(2)
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")) if launched else 0
    local = int(os.environ.get("LOCAL_RANK", "0")) if launched else 0
    if args.dry:
        return dry_main(args, world, rank)
    if world > 1 and torch.cuda.device_count() < world:
        print(f"bench.py: {world} ranks but only {torch.cuda.device_count()} HIP device(s) visible", file=sys.stderr, flush=True)
        sys.exit(2)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    import numpy as np
    import dsp_jl_amd as d
    from dsp_jl_amd import _lib
    from dsp_jl_amd.dspbase import OlsPlan
    lib = _lib.lib()
    _lib.check(lib.mdsp_init(local if world > 1 else 0))
    eng = {"auto": d.ENGINE_AUTO, "fused": d.ENGINE_FUSED, "rocfft": d.ENGINE_ROCFFT}[args.engine]
    dev = torch.device("cuda", torch.cuda.current_device())
    stream = torch.cuda.current_stream().cuda_stream
    tm = Timer(lib, _lib, stream)

    # the library's RCCL communicator (C ABI); torch.distributed is the fallback transport
    comm, collective = None, "none (1 GPU)"
    if world > 1:
        # 1. a LOCAL probe on every rank (binds librccl, ncclGetUniqueId needs no communicator), 2. agree, 3. only then the collective init:
        #    a rank that cannot load RCCL must not leave the others waiting inside ncclCommInitRank
        try:
            my_id, err = d.Comm.unique_id(), ""
        except Exception as e:  # pragma: no cover - needs a broken RCCL
            my_id, err = None, str(e)
        flags = torch.tensor([1.0 if my_id is not None else 0.0], device=dev)
        dist.all_reduce(flags, op=dist.ReduceOp.MIN)
        if float(flags.item()) >= 1.0:
            box = [my_id if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            try:
                comm = d.Comm(box[0], rank, world)
                collective = "mdsp_comm (RCCL ncclAllReduce behind the C ABI)"
            except Exception as e:  # pragma: no cover - needs > 1 GPU
                comm, collective = None, f"torch.distributed nccl (mdsp_comm init failed: {e})"
            flags = torch.tensor([1.0 if comm is not None else 0.0], device=dev)
            dist.all_reduce(flags, op=dist.ReduceOp.MIN)          # all ranks must agree on the transport
            if float(flags.item()) < 1.0 and comm is not None:
                comm.close(); comm = None
                collective = "torch.distributed nccl (mdsp_comm unavailable on some rank)"
        else:
            collective = f"torch.distributed nccl (RCCL probe failed on some rank{': ' + err if err else ''})"

    g = torch.Generator(device=dev)
    g.manual_seed(1776 + rank)                          # seed 1776 = test/runtests.jl:20; independent data per GPU

    if args.config == "filtwelch":
        log2n = args.log2n or 30
        n = 1 << log2n
        x = torch.randn(n, generator=g, device=dev, dtype=torch.float32)
        t = torch.arange(n, device=dev, dtype=torch.float32)
        x += 0.5 * torch.sin((2 * np.pi * 0.1234) * t)      # BASELINE.md config 3 line (phase accuracy is irrelevant here)
        del t
        y = torch.empty(n, dtype=torch.float32, device=dev)
        plan = OlsPlan(np.asarray(lowpass_taps(256)), 2048, n, _lib.OLS_FILT, eng)
        cfg = d.WelchConfig(n, np.float32, n=4096, noverlap=2048, window=d.hanning, engine=eng)
        psd = torch.empty((1, cfg.nout), dtype=torch.float32, device=dev)
        mean = torch.empty(cfg.nout, dtype=torch.float32, device=dev)
        units_per_rank = n

        def step(evt):
            a, b_, c = evt
            tm.rec(a)
            _lib.check(lib.mdsp_ols_exec(plan._h, x.data_ptr(), n, 1, n, y.data_ptr(), n, n, stream))
            tm.rec(b_)
            _lib.check(lib.mdsp_welch_exec(cfg._h, x.data_ptr(), n, 1, n, psd.data_ptr(), cfg.nout, stream))
            tm.rec(c)
            if comm is not None or world == 1:   # local channel sum -> RCCL all-reduce of 2049 floats over xGMI -> 1/nch: ONE C-ABI call
                _lib.check(lib.mdsp_welch_mean_allreduce(cfg._h, psd.data_ptr(), 1, cfg.nout, world, mean.data_ptr(), comm._h if comm else None, stream))
            else:
                _lib.check(lib.mdsp_channel_sum(psd.data_ptr(), cfg.nout, 1, cfg.nout, _lib.F32, mean.data_ptr(), stream))
                dist.all_reduce(mean, op=dist.ReduceOp.SUM)
                mean.mul_(1.0 / world)
        coll = lambda: (psd.sum(0, dtype=torch.float64), mean, float(world))     # (this rank's contribution, the collective's result, divisor)
        names = ("filt", "welch")
        alg = (8.0 * n, 4.0 * n)
        kern = ("ols_fused_kernel (overlap-save filt, 8 B/sample)", "mdsp_welch_w64c_asm + reduce + finalize (4 B/sample)")
        workload = f"filt(256 taps, overlap-save nfft 2048) + welch_pgram(nfft 4096, hanning, 50 %) per 2^{log2n}-sample Float32 stream; one stream per GPU; RCCL all-reduce of the PSD for N>1"
        metric, dtype, engine_used = METRIC, "f32", {1: "fused", 2: "rocfft"}[plan.engine]
    elif args.config == "stft":
        from dsp_jl_amd.periodograms import _StftPlan, compute_window
        log2n = args.log2n or 26
        nch, n = 8, 1 << log2n
        s = torch.view_as_complex(torch.randn((nch, n, 2), generator=g, device=dev, dtype=torch.float32) * math.sqrt(0.5))
        win, norm2 = compute_window(d.hanning, 1024)
        K = d.frame_count(n, 1024, 768)
        plan = _StftPlan(1024, 768, 1024, win, 1.0 * norm2, False, 0, np.complex64, eng)
        out = torch.empty((nch, K, 1024), dtype=torch.complex64, device=dev)
        units_per_rank = nch * n

        def step(evt):
            a, b_, c = evt
            tm.rec(a)
            _lib.check(lib.mdsp_stft_exec(plan._h, s.data_ptr(), n, nch, n, out.data_ptr(), 1024, K * 1024, stream))
            tm.rec(b_)
            tm.rec(c)                            # channels are independent: no collective on this path
        coll = None
        names = ("stft", "none")
        alg = (40.0 * n * nch, 0.0)
        kern = ("stft_fused_kernel (config 4: 8 + 8*1024/256 = 40 B/sample)", "-")
        workload = (f"stft(nfft=1024, hop=256, hanning, two-sided) of {nch} channels x 2^{log2n} ComplexF32 samples per GPU -> {nch} x (1024 x {K}) ComplexF32; "
                    f"channel c of {nch}*N on rank c div {nch}; no collective")
        metric, dtype, engine_used = "Gsamples/s stft nfft=1024 hop=256 ComplexF32, 8 channels x 64 Msample per GPU (BASELINE config 4)", "c64(f32 pairs)", "fused" if eng != d.ENGINE_ROCFFT else "rocfft"
    else:
        log2n = args.log2n or 28
        nch, n = 4, 1 << log2n
        h = resample_taps()
        x = torch.randn((nch, n), generator=g, device=dev, dtype=torch.float32)
        fh = C.c_void_p()
        _lib.check(lib.mdsp_fir_create(C.byref(fh), h.ctypes.data_as(C.c_void_p), len(h), 160, 147, _lib.F32, _lib.F32, nch))
        ol = C.c_int64(); _lib.check(lib.mdsp_fir_outputlength(fh, n, C.byref(ol)))
        y = torch.empty((nch, ol.value), dtype=torch.float32, device=dev)
        nw = C.c_int64()
        TAIL = 1024
        avg = torch.empty(TAIL, dtype=torch.float32, device=dev)
        units_per_rank = nch * n

        def step(evt):
            a, b_, c = evt
            tm.rec(a)
            _lib.check(lib.mdsp_fir_reset(fh))
            _lib.check(lib.mdsp_fir_exec(fh, x.data_ptr(), n, n, y.data_ptr(), ol.value, ol.value, C.byref(nw), stream))
            tm.rec(b_)
            # "RCCL avg" (BASELINE config 5): cross-channel average of the last output block -- local sum over this rank's channels,
            # one all-reduce of 1024 floats, 1/nch_total
            tail = y[:, ol.value - TAIL:]
            _lib.check(lib.mdsp_channel_sum(tail.data_ptr(), TAIL, nch, ol.value, _lib.F32, avg.data_ptr(), stream))
            if comm is not None:
                _lib.check(lib.mdsp_allreduce_sum(comm._h, avg.data_ptr(), TAIL, _lib.F32, stream))
            elif world > 1:
                dist.all_reduce(avg, op=dist.ReduceOp.SUM)
            avg.mul_(1.0 / (nch * world))
            tm.rec(c)
        coll = lambda: (y[:, ol.value - TAIL:].sum(0, dtype=torch.float64), avg, float(nch * world))
        names = ("resample", "channel average + all-reduce")
        alg = ((4 + 4 * 160 / 147) * n * nch, 0.0)
        kern = ("polyphase_mfma_kernel (config 5: 4 + 4*160/147 = 8.354 B/input sample)", "-")
        workload = (f"resample 160//147 (FIRFilter, 5120 taps = 32 per phase) of {nch} channels x 2^{log2n} Float32 samples per GPU -> {ol.value} outputs per channel; "
                    "RCCL all-reduce (1024 floats) for the cross-channel average of the last output block")
        metric, dtype, engine_used = "Gsamples/s FIRFilter polyphase resample 160//147, 32 taps/phase, Float32, 4 channels x 256 Msample per GPU (BASELINE config 5)", "f32", "hip"

    mark = Marks(lib, _lib, stream)
    mark("step")
    evs = [(tm.ev(), tm.ev(), tm.ev()) for _ in range(args.steps)]        # HIP events on the launch stream, created up front
    scratch = (tm.ev(), tm.ev(), tm.ev())
    for _ in range(args.warmup):
        step(scratch)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):                                  # the timed region: exactly K steps
        step(evs[k])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t_a = sum(tm.ms(a, b_) for a, b_, _ in evs) / max(1, args.steps)
    t_b = sum(tm.ms(b_, c) for _, b_, c in evs) / max(1, args.steps)
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())

    # self-check of the step's collective (outside the timed region): every rank's contribution is gathered over a DIFFERENT transport
    # (torch.distributed all_gather), summed in Float64 on each rank, and compared with what the step's own all-reduce left behind
    collective_check, collective_failed = "n/a (no collective on this path)", False
    if coll is not None:
        contrib, result, div = coll()
        if world > 1:
            parts = [torch.empty_like(contrib) for _ in range(world)]
            dist.all_gather(parts, contrib)
            want = torch.stack(parts).sum(0) / div
        else:
            want = contrib / div
        err = float((result.to(torch.float64) - want).norm() / want.norm().clamp_min(1e-300))
        bad = torch.tensor([0.0 if err < 1e-5 else 1.0], device=dev)
        if world > 1:
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        collective_failed = float(bad.item()) != 0.0
        collective_check = (f"FAILED (rank 0 relative error {err:.3e})" if collective_failed else
                            "ok" if world > 1 else "ok (1 rank: local channel sum only, nothing crossed a link)")

    if rank == 0:
        # ONE compact line: the driver keeps a bounded tail of it, so nothing is said twice -- one traffic_source, no per-row descriptions (ROW_INFO /
        # --describe-rows have them), stage times as flat numbers in `config` and every row's fraction of the roof once more as flat numbers in `roofline`
        out = {
            "metric": metric,
            "value": round(units_per_rank * world / dt * args.steps / 1e9, 3), "unit": "Gsamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": workload, "name": args.config, "samples_per_gpu": units_per_rank, "engine": engine_used, "collective": collective,
                       "collective_check": collective_check, f"stage_ms_{names[0]}": round(t_a, 4), f"stage_ms_{names[1].split()[0]}": round(t_b, 4)},
        }
        traffic, traffic_source = committed_traffic()
        if world == 1 and not args.no_live_pmc:
            try:
                live, src = live_traffic(args)
            except Exception as e:  # pragma: no cover
                live, src = None, f"live PMC pass failed: {e}"
            if live:
                traffic, traffic_source = live, src
            else:
                traffic_source = f"{traffic_source}; live pass unavailable ({src})"
        tkey = {"filtwelch": ("ols_fused_bytes_per_launch", "welch_fused_bytes_per_launch"), "stft": ("stft_bytes_per_launch", None),
                "resample": ("resample_bytes_per_launch", None)}[args.config]
        main_roof = roof(kern[0], t_a, alg[0], traffic.get(tkey[0]) if traffic else None)
        kernels = {}
        if args.config == "filtwelch":
            other = roof(kern[1], t_b, alg[1], traffic.get(tkey[1]) if traffic else None)
            # the Welch kernel's other roof: 5 N log2 N flop per 4096-point transform of 4096 new samples against the packed-FP32 add/multiply
            # rate (butterflies are adds, not FMAs): 256 CUs x 4 SIMDs x 16 lanes x 2 x 2.4 GHz
            tfl = 5.0 * 4096 * 12 * (units_per_rank / 4096) / (t_b * 1e-3) / 1e12
            other["valu_frac"] = round(tfl / 78.6, 4)
            if t_b > t_a:            # `roofline` is the dominant (longer) kernel of the step
                main_roof, other = other, main_roof
            kernels["other"] = {k: other[k] for k in ("kernel", "achieved", "frac", "traffic", "ms_per_launch", "algorithmic_bytes_per_launch") if k in other}
            if "valu_frac" in other:
                kernels["other"]["valu_frac"] = other["valu_frac"]
            main_roof["other_kernel_ms"] = round(min(t_a, t_b), 4)
            main_roof["other_kernel_frac"] = kernels["other"]["frac"]
            # on-box yardsticks: float4 copy (2 x 4 GiB moved), read-only stream, and the best 1:1 copy this box does (nontemporal float4, four workgroups per CU)
            nb = units_per_rank * 4
            mark("yardsticks")
            med, _ = tm.time(lambda: _lib.check(lib.mdsp_copy_bench(y.data_ptr(), x.data_ptr(), nb, stream)), reps=3)
            kernels["copy_GBps"] = round(2 * nb / med / 1e6, 1)
            med, _ = tm.time(lambda: _lib.check(lib.mdsp_copy_bench_mode(y.data_ptr(), x.data_ptr(), nb, 4, 8, stream)), reps=3)
            kernels["read_GBps"] = round(nb / med / 1e6, 1)
            med, _ = tm.time(lambda: _lib.check(lib.mdsp_copy_bench_mode(y.data_ptr(), x.data_ptr(), nb, 2, 4, stream)), reps=3)
            kernels["copy_nt_4wg_GBps"] = round(2 * nb / med / 1e6, 1)
            # round 6: the fastest of 150 copy shapes x cache policies of tools/ubench/copy_sweep.hip (nontemporal loads, sc0 | sc1 stores, four
            # 16-byte accesses in flight, two workgroups per CU) and the read-only stream with nontemporal loads -- the yardsticks the step kernels are priced against
            med, _ = tm.time(lambda: _lib.check(lib.mdsp_copy_bench_mode(y.data_ptr(), x.data_ptr(), nb, 6, 2, stream)), reps=3)
            kernels["copy_ntload_sc_store_GBps"] = round(2 * nb / med / 1e6, 1)
            kernels["copy_best_GBps"] = max(kernels["copy_GBps"], kernels["copy_nt_4wg_GBps"], kernels["copy_ntload_sc_store_GBps"])   # the best 1:1 copy of THIS box on THESE buffers
            med, _ = tm.time(lambda: _lib.check(lib.mdsp_copy_bench_mode(y.data_ptr(), x.data_ptr(), nb, 7, 4, stream)), reps=3)
            kernels["read_nt_GBps"] = round(nb / med / 1e6, 1)
        out["roofline"] = main_roof
        out["traffic_source"] = traffic_source
        if world == 1 and not args.no_rows and args.config == "filtwelch":
            del x, y
            torch.cuda.empty_cache()
            try:
                rows = measure_rows(tm, lib, _lib, d, stream, mark)
                for key, r in rows.items():
                    tb = traffic.get(f"{key}_bytes_per_launch") if traffic else None
                    if tb and "frac" in r:     # HBM bytes per launch of the row's dominant kernel / its algorithmic bytes
                        r["traffic_x"] = round(tb / (r["frac"] * HBM_PEAK_GBS * 1e9 * r["ms"] * 1e-3), 3)
                    if "frac" in r:
                        main_roof[f"f_{key}"] = r["frac"]
                kernels.update(rows)
            except Exception as e:  # pragma: no cover
                kernels["rows_error"] = str(e)
        out["kernels"] = kernels
        if world == 1 and not args.no_host and args.config == "filtwelch":
            try:
                out["host_path"] = measure_host_path(lib, _lib, d, args.host_log2n)
            except Exception as e:  # pragma: no cover
                out["host_path"] = {"error": str(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_log2n) if args.config == "filtwelch" else cpu_baseline_config(args.config, min(args.cpu_log2n, 23))
            except Exception as e:  # pragma: no cover
                out["cpu_baseline"] = {"value": None, "unit": "Gsamples/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
        out["commit"] = git_head()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()                                   # rank 0 may still be printing / measuring its yardsticks: leave together
        if comm is not None:
            comm.close()
        dist.destroy_process_group()
    if collective_failed:
        sys.exit(3)


def dry_main(args, world, rank):
    """CPU dry mode: the launcher contract (env, gloo rendezvous on 127.0.0.1, barrier + max-over-ranks timing, rank-0 JSON line), the
    channel sharding and the collective call pattern of each --config with world_size ranks and NO device work.  tests/test_bench_dry.py
    runs it with 2 ranks; it measures nothing (value = null)."""
    import torch
    import torch.distributed as dist
    import dsp_jl_amd as d
    if world > 1:
        dist.init_process_group("gloo")
    per_gpu = {"filtwelch": 1, "stft": 8, "resample": 4}[args.config]
    nch_total = per_gpu * world
    mine = d.channel_shard(nch_total, rank, world)
    assert len(mine) == per_gpu                                     # weak scaling: fixed channels per GPU
    nred = {"filtwelch": 2049, "stft": 0, "resample": 1024}[args.config]
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if nred:
            v = torch.full((nred,), float(len(mine)), dtype=torch.float32)   # stands for the local channel sum
            if world > 1:
                dist.all_reduce(v, op=dist.ReduceOp.SUM)
            v.mul_(1.0 / nch_total)
            assert abs(float(v[0]) - 1.0) < 1e-6                    # every channel counted exactly once
    if world > 1:
        dist.barrier()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    # the same self-check as the device path: rank-dependent contributions, all-reduce against an all_gather + Float64 sum
    check = "n/a (no collective on this path)"
    if nred:
        contrib = (torch.arange(nred, dtype=torch.float64) + 1.0) * (rank + 1) * len(mine)
        red = contrib.to(torch.float32)
        want = contrib
        if world > 1:
            dist.all_reduce(red, op=dist.ReduceOp.SUM)
            parts = [torch.empty_like(contrib) for _ in range(world)]
            dist.all_gather(parts, contrib)
            want = torch.stack(parts).sum(0)
        err = float((red.to(torch.float64) / nch_total - want / nch_total).norm() / (want / nch_total).norm())
        check = "ok" if err < 1e-5 else f"FAILED ({err:.3e})"
    # every rank's channel block, gathered: rank 0 reports them (the first 8-GPU run must see rank r own channels [r per_gpu, (r + 1) per_gpu))
    blocks = [[mine.start, mine.stop]]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, [mine.start, mine.stop])
        blocks = gathered
    if rank == 0:
        print(json.dumps({"metric": METRIC if args.config == "filtwelch" else f"dry:{args.config}", "value": None, "unit": "Gsamples/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(float(tt.item()) / max(1, args.steps) * 1e3, 4),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "none (dry mode)",
                          "config": {"workload": f"dry run of --config {args.config}", "name": args.config, "channels_total": nch_total,
                                     "channels_per_gpu": per_gpu, "channel_blocks": blocks, "collective": "gloo (dry)", "collective_check": check,
                                     "allreduce_floats": nred}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
