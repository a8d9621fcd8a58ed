"""ctypes binding of libmi355dsp.so (the C ABI declared in include/mi355dsp.h).

The library is the product: there is NO CPU fallback.  If the shared object is missing it is built with hipcc
(cross-compiles without a GPU); if no HIP device is visible every compute entry point raises DeviceError.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
# MDSP_LIB_TAG selects an experimental build made by `python dsp.jl_amd/build.py --tag T --cflags "..."` (tools/io_policy_sweep.sh)
LIB_PATH = os.path.join(HERE, "libmi355dsp" + ("_" + os.environ["MDSP_LIB_TAG"] if os.environ.get("MDSP_LIB_TAG") else "") + ".so")

# ---- status codes / dtype / engine enums (mirror include/mi355dsp.h) -----------------------------------
OK, ERR_ARGUMENT, ERR_DOMAIN, ERR_DIMENSION, ERR_ASSERTION, ERR_UNSUPPORTED, ERR_DEVICE, ERR_NOMEM = 0, -1, -2, -3, -4, -5, -6, -7
F32, F64, C32, C64 = 0, 1, 2, 3
ENGINE_AUTO, ENGINE_FUSED, ENGINE_ROCFFT = 0, 1, 2
OLS_FILT, OLS_CONV = 0, 1
HOST_PINNED = 1
COMM_ID_BYTES = 128


class ArgumentError(ValueError):
    """Julia ``ArgumentError``."""


class DomainError(ValueError):
    """Julia ``DomainError``."""


class DimensionMismatch(ValueError):
    """Julia ``DimensionMismatch``."""


class UnsupportedError(NotImplementedError):
    """Valid DSP.jl call that this library does not accelerate (caller should use DSP.jl's CPU path)."""


class DeviceError(RuntimeError):
    """HIP / rocFFT failure, or no MI355X visible."""


_EXC = {ERR_ARGUMENT: ArgumentError, ERR_DOMAIN: DomainError, ERR_DIMENSION: DimensionMismatch,
        ERR_ASSERTION: AssertionError, ERR_UNSUPPORTED: UnsupportedError, ERR_DEVICE: DeviceError, ERR_NOMEM: MemoryError}

i64, vp, ci, cd = C.c_int64, C.c_void_p, C.c_int, C.c_double
pi64, pint, pdbl, pvp = C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_void_p)

# name -> (restype, argtypes); every symbol include/mi355dsp.h declares
PROTOTYPES = {
    "mdsp_version": (ci, []),
    "mdsp_last_error_string": (C.c_char_p, []),
    "mdsp_init": (ci, [ci]),
    "mdsp_shutdown": (ci, []),
    "mdsp_reload_tunables": (ci, []),
    "mdsp_set_knob": (ci, [C.c_char_p, ci, ci]),
    "mdsp_debug_knobs": (ci, []),
    "mdsp_device_count": (ci, [pint]),
    "mdsp_malloc": (ci, [pvp, C.c_size_t]),
    "mdsp_free": (ci, [vp]),
    "mdsp_memcpy_h2d": (ci, [vp, vp, C.c_size_t, vp]),
    "mdsp_memcpy_d2h": (ci, [vp, vp, C.c_size_t, vp]),
    "mdsp_memset": (ci, [vp, ci, C.c_size_t, vp]),
    "mdsp_stream_synchronize": (ci, [vp]),
    "mdsp_nextfastfft": (i64, [i64]),
    "mdsp_optimal_fft_len": (i64, [i64, i64]),
    "mdsp_frame_count": (i64, [i64, i64, i64]),
    "mdsp_outputlength": (i64, [i64, i64, i64, i64]),
    "mdsp_inputlength": (i64, [i64, i64, i64, i64, ci]),
    "mdsp_ols_block_geometry": (ci, [i64, i64, i64, i64, pi64, pi64, pi64, pi64, pi64]),
    "mdsp_ols_plan_create": (ci, [pvp, vp, i64, i64, i64, ci, ci, ci]),
    "mdsp_ols_plan_destroy": (ci, [vp]),
    "mdsp_ols_plan_info": (ci, [vp, pi64, pi64, pint]),
    "mdsp_ols_plan_geometry": (ci, [vp, pi64, pi64, pint]),
    "mdsp_ols_geometry_for": (ci, [i64, i64, i64, ci, ci, ci, pi64, pi64, pint, pint, pint]),
    "mdsp_ols_exec": (ci, [vp, vp, i64, i64, i64, vp, i64, i64, vp]),
    "mdsp_ols_exec_range": (ci, [vp, vp, i64, i64, i64, vp, i64, i64, i64, vp]),
    "mdsp_ols_exec_host": (ci, [vp, vp, i64, i64, i64, vp, i64, i64, ci]),
    "mdsp_ols_segment": (ci, [vp, vp, i64, i64, i64, vp, vp]),
    "mdsp_frames": (ci, [vp, i64, ci, i64, i64, i64, pdbl, i64, i64, vp, vp]),
    "mdsp_welch_plan_create": (ci, [pvp, i64, i64, i64, pdbl, cd, ci, ci, ci]),
    "mdsp_welch_plan_destroy": (ci, [vp]),
    "mdsp_welch_plan_info": (ci, [vp, pi64, pint]),
    "mdsp_welch_exec": (ci, [vp, vp, i64, i64, i64, vp, i64, vp]),
    "mdsp_welch_reset": (ci, [vp]),
    "mdsp_welch_accumulate": (ci, [vp, vp, i64, i64, i64, vp]),
    "mdsp_welch_frames_accumulated": (ci, [vp, pi64]),
    "mdsp_welch_finalize": (ci, [vp, i64, vp, i64, vp]),
    "mdsp_welch_accumulator": (ci, [vp, pvp, pi64]),
    "mdsp_welch_exec_host": (ci, [vp, vp, i64, i64, i64, vp, i64, ci]),
    "mdsp_comm_unique_id": (ci, [vp]),
    "mdsp_comm_init_rank": (ci, [pvp, vp, ci, ci]),
    "mdsp_comm_destroy": (ci, [vp]),
    "mdsp_comm_info": (ci, [vp, pint, pint]),
    "mdsp_allreduce_sum": (ci, [vp, vp, i64, ci, vp]),
    "mdsp_welch_mean_allreduce": (ci, [vp, vp, i64, i64, i64, vp, vp, vp]),
    "mdsp_welch_allreduce": (ci, [vp, vp, vp]),
    "mdsp_host_alloc": (ci, [pvp, C.c_size_t]),
    "mdsp_host_free": (ci, [vp]),
    "mdsp_host_register": (ci, [vp, C.c_size_t]),
    "mdsp_host_unregister": (ci, [vp]),
    "mdsp_channel_sum": (ci, [vp, i64, i64, i64, ci, vp, vp]),
    "mdsp_stft_plan_create": (ci, [pvp, i64, i64, i64, pdbl, cd, ci, ci, ci, ci]),
    "mdsp_stft_plan_destroy": (ci, [vp]),
    "mdsp_stft_plan_info": (ci, [vp, pi64, pint]),
    "mdsp_stft_exec": (ci, [vp, vp, i64, i64, i64, vp, i64, i64, vp]),
    "mdsp_stft_exec_host": (ci, [vp, vp, i64, i64, i64, vp, i64, i64, ci]),
    "mdsp_hilbert": (ci, [vp, i64, i64, i64, ci, vp, i64, vp]),
    "mdsp_tdfir_state_exec": (ci, [vp, i64, ci, vp, i64, i64, i64, vp, i64, vp, vp]),
    "mdsp_extrapolate": (ci, [vp, i64, i64, i64, ci, i64, vp, i64, vp]),
    "mdsp_mt_plan_create": (ci, [pvp, i64, i64, vp, i64, vp, ci, ci, ci]),
    "mdsp_mt_plan_destroy": (ci, [vp]),
    "mdsp_mt_plan_info": (ci, [vp, pi64, pi64, pint]),
    "mdsp_mt_psd_exec": (ci, [vp, vp, i64, i64, i64, i64, vp, i64, i64, vp]),
    "mdsp_mt_spectra_exec": (ci, [vp, vp, i64, i64, ci, vp, vp]),
    "mdsp_mt_cross_spectra": (ci, [vp, vp, i64, vp, i64, vp, vp]),
    "mdsp_coherence_from_cs": (ci, [vp, i64, i64, ci, vp, vp]),
    "mdsp_fir_create": (ci, [pvp, vp, i64, i64, i64, ci, ci, i64]),
    "mdsp_fir_destroy": (ci, [vp]),
    "mdsp_fir_set_exact": (ci, [vp, ci]),
    "mdsp_fir_reset": (ci, [vp]),
    "mdsp_fir_setphase": (ci, [vp, cd]),
    "mdsp_fir_timedelay": (ci, [vp, pdbl]),
    "mdsp_fir_outputlength": (ci, [vp, i64, pi64]),
    "mdsp_fir_inputlength": (ci, [vp, i64, ci, pi64]),
    "mdsp_fir_info": (ci, [vp, pint, pi64, pi64, pi64, pi64, pint]),
    "mdsp_fir_kernel_path": (ci, [vp, i64, pint]),
    "mdsp_fir_mm_geometry": (ci, [i64, i64, i64, ci, ci, pi64]),
    "mdsp_fir_get_state": (ci, [vp, pi64, pi64, vp]),
    "mdsp_fir_set_state": (ci, [vp, i64, i64, vp]),
    "mdsp_fir_exec": (ci, [vp, vp, i64, i64, vp, i64, i64, pi64, vp]),
    "mdsp_fir_exec_host": (ci, [vp, vp, i64, i64, vp, i64, i64, pi64, ci]),
    "mdsp_firarb_create": (ci, [pvp, vp, i64, cd, i64, ci, ci, i64]),
    "mdsp_firarb_destroy": (ci, [vp]),
    "mdsp_firarb_reset": (ci, [vp]),
    "mdsp_firarb_setphase": (ci, [vp, cd]),
    "mdsp_firarb_timedelay": (ci, [vp, pdbl]),
    "mdsp_firarb_outputlength": (ci, [vp, i64, pi64]),
    "mdsp_firarb_inputlength": (ci, [vp, i64, ci, pi64]),
    "mdsp_firarb_info": (ci, [vp, pi64, pi64, pi64, pint, pdbl]),
    "mdsp_firarb_get_state": (ci, [vp, pdbl, pdbl, pi64, pi64, pi64, vp]),
    "mdsp_firarb_set_state": (ci, [vp, cd, i64, vp]),
    "mdsp_firarb_exec": (ci, [vp, vp, i64, i64, vp, i64, i64, pi64, vp]),
    "mdsp_arb_trajectory": (ci, [cd, i64, cd, i64, i64, i64, pi64, pdbl, i64, pi64, pdbl, pi64]),
    "mdsp_arb_trajectory_scan": (ci, [cd, i64, cd, i64, i64, i64, pi64, pdbl, i64, pi64, pdbl, pi64, pint, pint]),
    "mdsp_firarb_scan_stats": (ci, [vp, pi64, pi64]),
    "mdsp_arb_replay_check": (ci, [cd, cd, i64, i64, pi64]),
    "mdsp_convnd_fft": (ci, [vp, pi64, vp, pi64, ci, ci, vp, vp]),
    "mdsp_convnd_direct": (ci, [vp, pi64, vp, pi64, ci, ci, vp, vp]),
    "mdsp_tdfir_exec": (ci, [vp, i64, ci, vp, i64, i64, i64, vp, i64, vp]),
    "mdsp_ols_plan_cached": (ci, [pvp, vp, i64, i64, i64, ci, ci, ci, vp]),
    "mdsp_welch_plan_cached": (ci, [pvp, i64, i64, i64, pdbl, cd, ci, ci, ci, vp]),
    "mdsp_stft_plan_cached": (ci, [pvp, i64, i64, i64, pdbl, cd, ci, ci, ci, ci, vp]),
    "mdsp_plan_cache_stats": (ci, [pi64, pi64, pi64]),
    "mdsp_plan_cache_clear": (ci, []),
    "mdsp_host_pipeline_trim": (ci, []),
    "mdsp_plan_cache_partitions": (ci, [pi64, pi64]),
    "mdsp_plan_cache_set_context": (ci, [C.c_uint64]),
    "mdsp_plan_cache_release_context": (ci, [C.c_uint64]),
    "mdsp_event_create": (ci, [pvp]),
    "mdsp_event_destroy": (ci, [vp]),
    "mdsp_event_record": (ci, [vp, vp]),
    "mdsp_event_elapsed_ms": (ci, [vp, vp, C.POINTER(C.c_float)]),
    "mdsp_copy_bench": (ci, [vp, vp, C.c_size_t, vp]),
    "mdsp_copy_bench_mode": (ci, [vp, vp, C.c_size_t, ci, ci, vp]),
}

_lock = threading.Lock()
_lib = None


def build(force: bool = False, verbose: bool = False) -> str:
    import importlib.util
    spec = importlib.util.spec_from_file_location("_mdsp_build", os.path.join(HERE, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build(force=force, verbose=verbose)


def lib() -> C.CDLL:
    """Load (building first if needed) the shared library and attach prototypes."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                build()
            try:
                # torch (if the caller uses it for device arrays) must own the HIP runtime it was built with:
                # importing it first makes libmi355dsp resolve libamdhip64/librocfft to the already-loaded copies.
                import torch  # noqa: F401
            except Exception:  # pragma: no cover - torch is optional for the C ABI itself
                pass
            handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
            for name, (res, args) in PROTOTYPES.items():
                fn = getattr(handle, name)      # AttributeError here = the .so does not export a declared symbol
                fn.restype = res
                fn.argtypes = args
            _lib = handle
    return _lib


# the environment variables of a product build (include/mi355dsp.h); every other MDSP_* name the tools and tests set is a knob (mdsp_set_knob)
ENV_VARIABLES = ("MDSP_ENGINE", "MDSP_WG_PER_CU", "MDSP_PLAN_CACHE_TOTAL", "MDSP_PLAN_CACHE_IDLE", "MDSP_ROCFFT_CHUNK_MIB", "MDSP_HOST_CHUNK_MIB", "MDSP_BIG_CHUNK_MIB",
                 "MDSP_BIGFFT", "MDSP_GX", "MDSP_FIR_MM", "MDSP_FIR_DEC", "MDSP_FIR_EXACT", "MDSP_ARB_SCAN", "MDSP_ARB_SCAN_MIN", "MDSP_FIR_CHOICE_FILE")
_DEBUG_ENV = ("MDSP_ABLATE", "MDSP_WELCH_NOHALF", "MDSP_STFT_NOSHIFT", "MDSP_STFT_NOPAIR", "MDSP_STFT_NODIRECT", "MDSP_FIR_GENERIC", "MDSP_FIR_IDENTITY_LANES",
              "MDSP_MT_PASSES", "MDSP_ARB_PROF")   # profiling switches of -DMDSP_DEBUG_KNOBS builds: environment only


def set_tunable(name: str, value) -> None:
    """Set (or, with None, put back the default of) a tuning variable and make the library take it: one of the fifteen environment variables goes into the
    environment (libmi355dsp reads it once -- mdsp_init / first use -- and again on mdsp_reload_tunables, never on exec or plan paths); anything else is a
    knob and goes through mdsp_set_knob (round 6: experiments are not steered through the environment of a product build).  Tuning tools and tests only."""
    if name in ENV_VARIABLES or name in _DEBUG_ENV:
        if value is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = str(value)
        check(lib().mdsp_reload_tunables())
        return
    check(lib().mdsp_set_knob(name.encode(), 0 if value is None else int(value), 1 if value is None else 0))


def check(status: int) -> None:
    if status == OK:
        return
    msg = lib().mdsp_last_error_string()
    msg = msg.decode("utf-8", "replace") if msg else f"libmi355dsp status {status}"
    raise _EXC.get(status, RuntimeError)(msg)


def device_count() -> int:
    n = C.c_int(0)
    check(lib().mdsp_device_count(C.byref(n)))
    return n.value


def require_device() -> None:
    """The product path fails loudly without a GPU (no CPU fallback)."""
    if device_count() < 1:
        raise DeviceError("no HIP device visible: dsp.jl_amd runs on MI355X only (there is no CPU fallback; "
                          "use DSP.jl itself on the CPU)")
