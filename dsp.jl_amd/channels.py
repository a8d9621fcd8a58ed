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


def welch_time_split(x_slice, nframes_total: int, n: int, noverlap: int, group=None, **kw) -> torch.Tensor:
    """Welch PSD of one stream whose frames are split over ranks.  ``x_slice``: this rank's samples (``frame_span`` of its
    ``frame_shard``; may be empty).  Every rank evaluates the reference's frames of its range, the per-rank means are
    recombined with their frame counts by one all-reduce(sum) of ``nout`` values: sum_r K_r P_r / K -- the same frames and the
    same normalisation as ``welch_pgram`` of the whole stream (periodograms.jl:746-759)."""
    length = int(x_slice.shape[0])
    sdt = _dev.np_dtype_of(x_slice)
    k_local = 0 if length < n else (length - n) // (n - noverlap) + 1
    T = util.fftabs2type(util.fftintype(sdt))
    if k_local > 0:
        cfg = WelchConfig(length, sdt, n=n, noverlap=noverlap, **kw)
        cols, _ = _dev.to_columns(x_slice, cfg.intype)
        tot = _welch_exec(cols, cfg)[0] * float(k_local)
    else:
        nfft = int(kw.get("nfft", util.nextfastfft(n)))
        onesided = kw.get("onesided", sdt.kind != "c")
        tot = torch.zeros(nfft // 2 + 1 if onesided else nfft, dtype=_dev.torch_dtype(T), device=_dev.device())
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size(group) > 1:
        torch.distributed.all_reduce(tot, op=torch.distributed.ReduceOp.SUM, group=group)
    return tot / float(nframes_total)


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
