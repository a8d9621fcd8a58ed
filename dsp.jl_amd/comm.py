"""``Comm``: the library's RCCL communicator (``mdsp_comm_*`` in include/mi355dsp.h), one per process / GPU.

The collective of the path lives behind the C ABI so that ANY host (the Julia twin, a C program, this Python mirror) gets the
multi-GPU part from the same entry points.  Bootstrap needs 128 bytes shipped from rank 0 to every rank once; here that rides
on whatever ``torch.distributed`` group the launcher (``torch.distributed.run``) already set up -- the data path itself never
goes through torch.
"""
from __future__ import annotations

import ctypes as C

from . import _dev, _lib


class Comm:
    def __init__(self, unique_id: bytes, rank: int, nranks: int):
        if len(unique_id) != _lib.COMM_ID_BYTES:
            raise _lib.ArgumentError(f"unique id must be {_lib.COMM_ID_BYTES} bytes")
        _lib.require_device()
        self._h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), _lib.COMM_ID_BYTES)
        _lib.check(_lib.lib().mdsp_comm_init_rank(C.byref(self._h), buf, int(rank), int(nranks)))
        self.rank, self.nranks = int(rank), int(nranks)

    @staticmethod
    def unique_id() -> bytes:
        """``ncclGetUniqueId`` through the C ABI (rank 0 calls this and ships the bytes to the other ranks)."""
        _lib.require_device()
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        _lib.check(_lib.lib().mdsp_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def from_torch_distributed(cls, group=None) -> "Comm":
        """One ``Comm`` per rank of an initialised ``torch.distributed`` group (any backend: only the 128-byte id travels on it)."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(box[0], rank, world)

    @classmethod
    def single(cls) -> "Comm":
        return cls(cls.unique_id(), 0, 1)

    def allreduce_sum(self, t):
        """In-place sum over ranks of a Float32 / Float64 device tensor (``ncclAllReduce`` on the current stream)."""
        if not (_dev.is_device_array(t) and t.is_contiguous()):
            raise _lib.ArgumentError("a contiguous device tensor is required")
        _lib.check(_lib.lib().mdsp_allreduce_sum(self._h, _dev.ptr(t), t.numel(), _dev.md_dtype(_dev.np_dtype_of(t)), _dev.stream_ptr()))
        return t

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().mdsp_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
