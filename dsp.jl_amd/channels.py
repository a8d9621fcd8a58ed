"""Multi-channel / multi-GPU plumbing: channels shard across ranks (one process per GPU), every rank runs the
single-GPU path on its own channels, and the only collective is the all-reduce (sum) of an ``nout``-float PSD for
the cross-channel Welch mean.  ``torch.distributed`` with backend "nccl" IS RCCL on ROCm.

A single long stream can also be split along TIME ("next"): rank r takes a contiguous range of Welch frames (its
samples are a slice with an ``n - hop`` overlap into the next rank's) or, for filtering, a slice preceded by ``nb - 1`` samples
of halo; the PSD needs the same one all-reduce, weighted by the frame counts, and filtering needs none.
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


def _allreduce_sum(t: torch.Tensor, comm=None, group=None) -> torch.Tensor:
    """The one transport of the path: in-place sum over ranks.  ``comm`` (dsp_jl_amd.Comm: RCCL through the C ABI,
    ``mdsp_allreduce_sum``) when given; otherwise an initialised ``torch.distributed`` group (backend "nccl" IS RCCL on ROCm;
    "gloo" in the CPU tests); single process: nothing to do."""
    if comm is not None:
        if comm.nranks > 1:
            comm.allreduce_sum(t)
    elif torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size(group) > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM, group=group)
    return t


def _local_channel_sum(psd: torch.Tensor, nout: int, T) -> torch.Tensor:
    """Sum over this rank's channels on the device (``mdsp_channel_sum``); a rank without channels contributes zeros."""
    tot = torch.zeros(nout, dtype=_dev.torch_dtype(T), device=psd.device)
    if psd.shape[0] > 0:
        _lib.check(_lib.lib().mdsp_channel_sum(_dev.ptr(psd), nout, psd.shape[0], nout, _dev.md_dtype(T), _dev.ptr(tot), _dev.stream_ptr()))
    return tot


def welch_channel_mean(cols: torch.Tensor, config: WelchConfig, nch_total: int | None = None, group=None, comm=None) -> torch.Tensor:
    """Mean over ALL channels (across ranks) of the per-channel Welch PSDs.

    ``cols``: this rank's channels as a C-contiguous (nch_local, len) device tensor (may have 0 rows).  With ``comm`` (a
    ``dsp_jl_amd.Comm``) the whole reduction is ONE C-ABI call, ``mdsp_welch_mean_allreduce``: local sum on the device, one
    ``ncclAllReduce(sum)`` of ``nout`` values over xGMI, the 1/nch scale.  Without it the same three steps run here with
    ``torch.distributed`` as the transport.
    """
    nloc = int(cols.shape[0])
    psd = _welch_exec(cols, config) if nloc > 0 else cols.new_zeros((0, config.nout))                   # (nch_local, nout)
    T = util.fftabs2type(config.intype)
    n_all = nloc if nch_total is None else int(nch_total)
    if comm is not None:
        mean = torch.empty(config.nout, dtype=_dev.torch_dtype(T), device=cols.device)
        _lib.check(_lib.lib().mdsp_welch_mean_allreduce(config._h, _dev.ptr(psd) if nloc else None, nloc, config.nout, n_all, _dev.ptr(mean),
                                                        comm._h, _dev.stream_ptr()))
        return mean
    tot = _local_channel_sum(psd, config.nout, T)
    _allreduce_sum(tot, None, group)
    return tot / n_all


# ---------------------------------------------------------------------------------------------------------------------
# one stream over several GPUs: split along time
# ---------------------------------------------------------------------------------------------------------------------
def frame_shard(nframes: int, rank: int, world: int) -> range:
    """Contiguous block partition of the frames 0..nframes-1 of ONE stream over ranks."""
    per = -(-nframes // world)
    lo = min(rank * per, nframes)
    return range(lo, min(lo + per, nframes))


def frame_span(frames: range, n: int, noverlap: int) -> tuple[int, int]:
    """Half-open 0-based sample range [lo, hi) that the frames cover (frame k = samples k*hop .. k*hop + n, periodograms.jl:57-69)."""
    if len(frames) == 0:
        return 0, 0
    hop = n - noverlap
    return frames.start * hop, (frames.stop - 1) * hop + n


def welch_time_split(x_slice, nframes_total: int, n: int, noverlap: int, group=None, comm=None, **kw) -> torch.Tensor:
    """Welch PSD of one stream whose frames are split over ranks.  ``x_slice``: this rank's samples (``frame_span`` of its
    ``frame_shard``; may be empty).  Every rank adds the reference's frames of its range to a Welch plan's Float64 |X|^2 sums
    (``mdsp_welch_accumulate``), the sums are added over ranks by one all-reduce, and every rank forms the PSD with the TOTAL frame
    count (``mdsp_welch_finalize``) -- the same frames and the same normalisation as ``welch_pgram`` of the whole stream
    (periodograms.jl:746-759)."""
    length = int(x_slice.shape[0])
    sdt = _dev.np_dtype_of(x_slice)
    cfg = WelchConfig(max(length, n), sdt, n=n, noverlap=noverlap, **kw)
    cols, _ = _dev.to_columns(x_slice, cfg.intype)                # (1, length); a rank without frames passes an empty slice
    cfg.reset()
    cfg.accumulate(cols)
    acc = cfg.accumulator()
    _allreduce_sum(acc, comm, group)
    return cfg.finalize(int(nframes_total))[0]


def filt_time_split_span(lo: int, hi: int, nb: int) -> tuple[int, int]:
    """Samples a rank needs to produce outputs [lo, hi) of ``filt(b, x)`` with ``nb`` taps: its own range preceded by ``nb - 1``
    samples of halo (fewer at the start of the stream); the first ``lo - span_lo`` outputs of filtering that slice are dropped."""
    return max(0, lo - (nb - 1)), hi


def filt_time_split(b, x_slice, drop: int):
    """``filt(b, x)`` restricted to this rank's output range: filter the slice of ``filt_time_split_span`` (zero initial state)
    and drop the ``drop`` outputs computed from the halo.  No collective: the overlap-save blocks of different ranks are
    independent (Filters/filt.jl:479-521)."""
    from . import filt as _filt
    y = _filt(b, x_slice)
    return y[drop:]
