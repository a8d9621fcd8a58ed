"""Least-recently-used cache of device plans for the per-call API.

DSP.jl's `filt(b, x)`, `welch_pgram(s, n, noverlap)`, `stft(...)` build their FFTW plans on every call; that is cheap on a CPU.
A device plan (tap spectrum, root tables, window, work buffers) costs allocations and host-to-device copies -- about a millisecond,
two orders of magnitude more than the kernel needs for a 2^20-sample signal -- so the per-call entry points keep the most recent
plans alive, keyed by everything the plan depends on.  Explicit plan objects (`WelchConfig`, `FIRFilter`, ...) are never cached here:
their lifetime is the caller's."""
from __future__ import annotations

import threading
from collections import OrderedDict


class PlanCache:
    def __init__(self, maxsize: int = 8):
        self._d: OrderedDict = OrderedDict()
        self._lock = threading.Lock()
        self.maxsize = maxsize
        self.hits = self.misses = 0

    def get(self, key, make):
        with self._lock:
            if key in self._d:
                self._d.move_to_end(key)
                self.hits += 1
                return self._d[key]
        plan = make()                       # outside the lock: plan creation touches the device
        with self._lock:
            self.misses += 1
            self._d[key] = plan
            self._d.move_to_end(key)
            while len(self._d) > self.maxsize:
                self._d.popitem(last=False)   # the evicted plan is destroyed by its __del__ (hipFree synchronises)
        return plan

    def clear(self):
        with self._lock:
            self._d.clear()


def array_key(a):
    """Hashable identity of a host array's contents."""
    import numpy as np
    a = np.ascontiguousarray(a.cpu() if hasattr(a, "cpu") else a)
    return (a.dtype.str, a.shape, a.tobytes())


def window_key(window):
    """Windows are functions (hashable as they are), None, or vectors (keyed by content)."""
    if window is None or callable(window):
        return window
    return array_key(window.cpu() if hasattr(window, "cpu") else window)   # device tensors are keyed by their host copy


def ctx_key():
    """Plans own work buffers in ONE device's HBM: one cached plan per (device, thread, stream) so that concurrent callers never share
    one and a plan is never reused after torch.cuda.set_device() moved the caller to another GPU (the default stream is 0 on every device)."""
    from . import _dev
    dev = _dev.device()                # DeviceError without a GPU, before anything else touches the runtime
    return (dev.index, threading.get_ident(), _dev.stream_ptr())


plans = PlanCache()
