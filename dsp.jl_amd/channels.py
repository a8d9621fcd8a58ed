"""Multi-channel / multi-GPU plumbing: channels shard across ranks (one process per GPU), every rank runs the
single-GPU path on its own channels, and the only collective is the all-reduce (sum) of an ``nout``-float PSD for
the cross-channel Welch mean.  ``torch.distributed`` with backend "nccl" IS RCCL on ROCm.
"""
from __future__ import annotations

import torch

from . import _dev, _lib
from .periodograms import WelchConfig, _welch_exec
from . import util


def channel_shard(nch_total: int, rank: int, world: int) -> range:
    """Contiguous block partition of channels over ranks (channel c -> rank c // ceil(nch/world))."""
    per = -(-nch_total // world)
    lo = min(rank * per, nch_total)
    return range(lo, min(lo + per, nch_total))


def welch_channel_mean(cols: torch.Tensor, config: WelchConfig, nch_total: int | None = None, group=None) -> torch.Tensor:
    """Mean over ALL channels (across ranks) of the per-channel Welch PSDs.

    ``cols``: this rank's channels as a C-contiguous (nch_local, len) device tensor.  Local sum on the device
    (``mdsp_channel_sum``), one all-reduce(sum) of ``nout`` values over xGMI, then the 1/nch scale.
    """
    psd = _welch_exec(cols, config)                                   # (nch_local, nout)
    T = util.fftabs2type(config.intype)
    tot = torch.empty(config.nout, dtype=_dev.torch_dtype(T), device=cols.device)
    _lib.check(_lib.lib().mdsp_channel_sum(_dev.ptr(psd), config.nout, psd.shape[0], config.nout, _dev.md_dtype(T), _dev.ptr(tot),
                                           _dev.stream_ptr()))
    n_all = psd.shape[0] if nch_total is None else nch_total
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size(group) > 1:
        torch.distributed.all_reduce(tot, op=torch.distributed.ReduceOp.SUM, group=group)
    return tot / n_all
