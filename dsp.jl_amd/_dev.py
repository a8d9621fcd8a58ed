"""Device-array plumbing for the Python host: torch tensors hold HBM buffers and streams (nothing else of torch is used).

Julia arrays are column-major and DSP.jl filters along the first axis; the Python mirror takes arrays of shape
``(n,)`` or ``(n, channels)`` with the same meaning.  On the device every channel is one contiguous column, i.e. a
C-contiguous ``(channels, n)`` tensor -- exactly a Julia ``(n, channels)`` array in memory.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib

_NP2MD = {np.dtype(np.float32): _lib.F32, np.dtype(np.float64): _lib.F64,
          np.dtype(np.complex64): _lib.C32, np.dtype(np.complex128): _lib.C64}
_TORCH2NP = {torch.float32: np.dtype(np.float32), torch.float64: np.dtype(np.float64),
             torch.complex64: np.dtype(np.complex64), torch.complex128: np.dtype(np.complex128),
             torch.int32: np.dtype(np.int32), torch.int64: np.dtype(np.int64), torch.int16: np.dtype(np.int16),
             torch.int8: np.dtype(np.int8), torch.uint8: np.dtype(np.uint8), torch.bool: np.dtype(np.bool_),
             torch.float16: np.dtype(np.float16)}
_NP2TORCH = {v: k for k, v in _TORCH2NP.items()}


def np_dtype_of(x) -> np.dtype:
    if isinstance(x, torch.Tensor):
        return _TORCH2NP[x.dtype]
    return np.asarray(x).dtype


def md_dtype(dt) -> int:
    return _NP2MD[np.dtype(dt)]


def torch_dtype(dt) -> torch.dtype:
    return _NP2TORCH[np.dtype(dt)]


def is_device_array(x) -> bool:
    return isinstance(x, torch.Tensor) and x.is_cuda


def device() -> torch.device:
    _lib.require_device()
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


HOST_PIPELINE_MIN_BYTES = 32 << 20   # numpy inputs at least this large take the library's host-pointer pipeline (pinned, chunked, overlapped)


def host_columns(x, dtype):
    """``x`` as HOST columns for the host-pointer entry points: a C-contiguous (ncols, n) numpy view of ``dtype`` WITHOUT copying, or
    None when ``x`` is not a large numpy array whose columns are already contiguous (1-d, or 2-d in Fortran order like a Julia matrix)."""
    if not isinstance(x, np.ndarray) or x.dtype != np.dtype(dtype) or x.nbytes < HOST_PIPELINE_MIN_BYTES:
        return None
    if x.ndim == 1 and x.flags.c_contiguous:
        return x.reshape(1, -1)
    if x.ndim == 2 and x.flags.f_contiguous:
        return x.T
    return None


def to_columns(x, dtype) -> tuple[torch.Tensor, tuple]:
    """Return a C-contiguous (ncols, n) device tensor of ``dtype`` holding the columns of ``x`` (shape (n, cols...)),
    plus the original shape."""
    dev = device()
    td = torch_dtype(dtype)
    if isinstance(x, torch.Tensor):
        t = x.to(device=dev)
    else:
        a = np.asarray(x)
        if a.dtype == np.dtype(object):
            raise TypeError("non-numeric array")
        t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    shape = tuple(t.shape)
    if t.dim() == 0:
        raise _lib.ArgumentError("expected at least a vector")
    n = shape[0]
    ncols = 1
    for k in shape[1:]:
        ncols *= int(k)
    t = t.reshape(n, ncols).t()           # (ncols, n) view; Julia's trailing dims are flattened like CartesianIndices
    if t.dtype != td:
        t = t.to(td)
    return t.contiguous(), shape


def as_device(x, dtype) -> torch.Tensor:
    """``x`` as a device tensor of ``dtype`` with its own shape (no column flattening)."""
    dev = device()
    td = torch_dtype(dtype)
    if isinstance(x, torch.Tensor):
        t = x.to(device=dev)
    else:
        a = np.asarray(x)
        if a.dtype == np.dtype(object):
            raise TypeError("non-numeric array")
        t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t if t.dtype == td else t.to(td)


def from_columns(t: torch.Tensor, shape: tuple, like):
    """(ncols, n_out) device tensor -> array shaped like the input container (numpy in -> numpy out)."""
    n_out = t.shape[1]
    out = t.t().reshape((n_out,) + tuple(shape[1:]))
    if isinstance(like, torch.Tensor):
        return out
    return out.cpu().numpy()


def empty_columns(ncols: int, n: int, dtype) -> torch.Tensor:
    return torch.empty((ncols, n), dtype=torch_dtype(dtype), device=device())


def ptr(t: torch.Tensor) -> int:
    return t.data_ptr()


def tensor_view(ptr_: int, count: int, dtype, dev=None) -> torch.Tensor:
    """A 1-d tensor VIEW of ``count`` elements of ``dtype`` at device address ``ptr_`` (memory owned by the library)."""
    dt = np.dtype(dtype)

    class _Iface:
        __cuda_array_interface__ = {"shape": (int(count),), "typestr": dt.str, "data": (int(ptr_), False), "version": 3, "strides": None}

    return torch.as_tensor(_Iface(), device=dev if dev is not None else device())
