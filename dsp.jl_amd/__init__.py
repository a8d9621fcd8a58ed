"""dsp.jl_amd -- DSP.jl's FFT-filtering / spectral-estimation hot path on AMD MI355X.

Host-side mirror (Python, because Julia is absent from the build image; the Julia ``ccall`` twin is
``julia/MI355DSP.jl``) of the DSP.jl API for this path.  All arithmetic happens in ``libmi355dsp.so``
(hand-written HIP kernels for gfx950 + rocFFT) through the C ABI of ``include/mi355dsp.h``; there is no CPU
fallback.  Mutating Julia methods ``f!`` are spelled ``f_``.
"""
from . import _lib
from ._lib import ArgumentError, DeviceError, DimensionMismatch, DomainError, UnsupportedError, build  # noqa: F401
from ._lib import ENGINE_AUTO, ENGINE_FUSED, ENGINE_ROCFFT  # noqa: F401
from .util import nextfastfft, fftintype, fftouttype, fftabs2type  # noqa: F401
from . import windows, design  # noqa: F401
from .windows import (hanning, hann, hamming, rect, bartlett, cosine, blackman, kaiser, dpss, dpsseig, tukey, lanczos, triang,  # noqa: F401
                      gaussian, bartlett_hann, blackmanharris, nuttall, flattop)
from .design import resample_filter, kaiserord  # noqa: F401
from .dspbase import conv, conv_, conv_separable, xcorr, hilbert, optimalfftfiltlength, os_fft_complexity, SMALL_FILT_CUTOFF  # noqa: F401
from .dspbase import filt as _filt_ba, filt_ as _filt_ba_
from .filters import (FIRFilter, fftfilt, fftfilt_, tdfilt, tdfilt_, resample, inputlength, outputlength,  # noqa: F401
                      filt as _filt_bx, filt_ as _filt_bx_, filt_stateless, DF2TFilter, filtfilt)
from .multitaper import (MTConfig, MTSpectrogramConfig, MTCrossSpectraConfig, MTCoherenceConfig, CrossPowerSpectra, Coherence,  # noqa: F401
                         coherence, dpss_config, mt_pgram, mt_pgram_, mt_spectrogram, mt_spectrogram_, mt_cross_power_spectra,
                         mt_cross_power_spectra_, mt_coherence, mt_coherence_)
from .periodograms import (Periodogram, Spectrogram, WelchConfig, arraysplit, fftshift, periodogram, welch_pgram, welch_pgram_,  # noqa: F401
                           spectrogram, stft, power, freq, time, frame_count)
from .comm import Comm  # noqa: F401
from .channels import (channel_shard, welch_channel_mean, frame_shard, frame_span, welch_time_split,  # noqa: F401
                       filt_time_split_span, filt_time_split)


def filt(*args, **kw):
    """DSP.jl's ``filt`` methods on this path:

    ``filt(b, a, x)``          time-domain FIR, scalar ``a``            (dspbase.jl:14)
    ``filt(b, x)``             FIR, FFT overlap-save when length(b) > 66 (Filters/filt.jl:445, :525)
    ``filt(f::FIRFilter, x)``  stateful polyphase filter                 (stream_filt.jl:627)
    ``filt(f::DF2TFilter, x)`` stateful FIR filter (TDF-II state)        (Filters/filt.jl:153)
    ``filt(h, x, ratio)``      stateless polyphase filter                (stream_filt.jl:663)
    ``filt(h, x, rate::float, Nphi=32)``  stateless arbitrary-rate resampler (stream_filt.jl:669)
    """
    if len(args) == 2 and isinstance(args[0], (FIRFilter, DF2TFilter)):
        return args[0].filt(args[1])
    if len(args) == 2:
        return _filt_bx(*args, **kw)
    if len(args) in (3, 4):
        from fractions import Fraction
        import numpy as _np
        third = args[2]
        scalar_rate = isinstance(third, (int, Fraction, float, _np.floating, _np.integer)) and not isinstance(third, bool)
        vector_x = hasattr(args[1], "shape") and len(getattr(args[1], "shape")) >= 1
        if scalar_rate and vector_x:                      # filt(h, x, ratio) / filt(h, x, rate::AbstractFloat, Nphi=32)
            if len(args) == 4 and not isinstance(third, (float, _np.floating)):
                raise TypeError("filt(h, x, ratio, Nphi): Nphi is only defined for a floating-point rate")
            return filt_stateless(*args)
        if len(args) == 3:
            return _filt_ba(*args)
    raise TypeError("filt: no method matching the given arguments")


def filt_(out, *args):
    """``filt!(out, b, x)`` / ``filt!(out, b, a, x)``."""
    if len(args) == 2:
        return _filt_bx_(out, *args)
    if len(args) == 3:
        return _filt_ba_(out, *args)
    raise TypeError("filt!: no method matching the given arguments")
