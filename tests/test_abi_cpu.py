"""CPU-only checks of the C-ABI boundary: the shared library loads, exports every declared symbol, fails loudly
without a GPU, and its pure index arithmetic is bit-exact with the oracle (= the reference's formulas)."""
import ctypes as C
import os
import re
from fractions import Fraction

import numpy as np
import pytest

import dsp_jl_amd as d
from dsp_jl_amd import _lib
from oracle import dspbase as odsp, filt as ofilt, periodograms as opg, stream_filt as osf, util as outil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_header_symbol():
    hdr = open(os.path.join(ROOT, "include", "mi355dsp.h")).read()
    declared = set(re.findall(r"\b(mdsp_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"mdsp_ols_plan", "mdsp_welch_plan", "mdsp_stft_plan", "mdsp_fir"}
    assert len(declared) > 40
    lib = _lib.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/mi355dsp.h but not exported"
        assert name in _lib.PROTOTYPES, f"{name} has no ctypes prototype"
    assert lib.mdsp_version() == 100


def test_no_cpu_fallback():
    if _lib.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(d.DeviceError):
        d.filt(np.ones(3), 1.0, np.ones(8))
    with pytest.raises(d.DeviceError):
        d.welch_pgram(np.ones(64), 8, 4, window=None)
    with pytest.raises(d.DeviceError):
        _lib.check(_lib.lib().mdsp_init(0))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "dsp.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f"{f} imports the oracle"


def test_nextfastfft_matches_oracle():
    lib = _lib.lib()
    for n in list(range(0, 2000)) + [2 ** 20 - 1, 2 ** 20 + 1, 10 ** 6, 123456789]:
        assert lib.mdsp_nextfastfft(n) == outil.nextfastfft(n), n
    assert d.nextfastfft((64, 65, 127)) == (64, 70, 128)            # test/util.jl:59


def test_optimal_fft_len_matches_oracle():
    lib = _lib.lib()
    rng = np.random.default_rng(1)
    cases = [(1, 3), (256, 2 ** 30), (127, 10 ** 6), (12, 128), (128, 128), (13, 32), (4, 25), (1, 1), (7, 1)]
    cases += [(int(rng.integers(1, 3000)), int(rng.integers(1, 10 ** 7))) for _ in range(2000)]
    for nb, nx in cases:
        assert lib.mdsp_optimal_fft_len(nb, nx) == odsp.optimalfftfiltlength(nb, nx), (nb, nx)


def test_frame_count_and_lengths_match_oracle():
    lib = _lib.lib()
    rng = np.random.default_rng(2)
    for _ in range(3000):
        n = int(rng.integers(1, 5000)); nov = int(rng.integers(0, n)); ln = int(rng.integers(0, 10 ** 6))
        assert lib.mdsp_frame_count(ln, n, nov) == opg.frame_count(ln, n, nov)
    assert lib.mdsp_frame_count(2 ** 30, 4096, 2048) == 524287
    for _ in range(3000):
        L = int(rng.integers(1, 200)); M = int(rng.integers(1, 200)); phi = int(rng.integers(1, L + 1))
        r = Fraction(L, M)
        il = int(rng.integers(0, 10 ** 7)); ol = int(rng.integers(0, 10 ** 7))
        # the oracle (like the reference) works on the ratio's numerator/denominator as given by Rational
        assert lib.mdsp_outputlength(il, L, M, phi) == -(-(il * L - phi + 1) // M)
        assert lib.mdsp_outputlength(il, r.numerator, r.denominator, min(phi, r.numerator)) == osf.outputlength_ratio(il, r, min(phi, r.numerator))
        for up in (0, 1):
            assert lib.mdsp_inputlength(ol, r.numerator, r.denominator, min(phi, r.numerator), up) == \
                osf.inputlength_ratio(ol, r, min(phi, r.numerator), bool(up))


def test_ols_block_geometry_matches_reference_table():
    lib = _lib.lib()
    i64 = C.c_int64
    for nb, nx, nfft in ((256, 10 ** 4, 2048), (127, 5000, 1024), (12, 128, 32), (13, 140, 32), (4, 25, 16), (200, 300, 256), (3, 2, 4)):
        L, rows = ofilt.fftfilt_block_table(nb, nx, nfft)
        for ib, row in enumerate(rows):
            out = [i64() for _ in range(5)]
            _lib.check(lib.mdsp_ols_block_geometry(nb, nx, nfft, ib, *[C.byref(o) for o in out]))
            assert tuple(o.value for o in out) == row, (nb, nx, nfft, ib)
        with pytest.raises(d.ArgumentError):
            _lib.check(lib.mdsp_ols_block_geometry(nb, nx, nfft, len(rows), None, None, None, None, None))


def test_host_windows_and_design_match_oracle(golden):
    from oracle import windows as ow, design as od
    for n in (1, 2, 8, 127, 128, 4096):
        for name in ("hanning", "hamming", "bartlett", "rect"):
            assert np.array_equal(getattr(d.windows, name)(n), getattr(ow, name)(n)), (name, n)
    assert np.max(np.abs(d.hanning(128) - golden["hanning128"])) < 5e-16
    assert np.array_equal(d.kaiser(33, 2.5), ow.kaiser(33, 2.5))
    for r in (Fraction(160, 147), Fraction(1, 2), Fraction(3, 2), 2):
        assert np.allclose(d.resample_filter(r), od.resample_filter(r), rtol=1e-14, atol=0)
    assert d.kaiserord(0.1, 60) == od.kaiserord(0.1, 60)


def test_type_rules():
    for dt in (np.int32, np.int64, np.float32, np.float64, np.complex64, np.complex128, np.float16):
        assert d.fftintype(dt) == outil.fftintype(dt)
        assert d.fftouttype(dt) == outil.fftouttype(dt)
        assert d.fftabs2type(dt) == outil.fftabs2type(dt)


def test_firfilter_host_state_matches_oracle():
    """setphase!/inputlength/outputlength/timedelay need no device: compare the host mirror with the oracle."""
    rng = np.random.default_rng(5)
    for _ in range(500):
        ratio = Fraction(int(rng.integers(1, 11)), int(rng.integers(1, 11)))
        hlen = int(rng.integers(1, 101))
        a, b = d.FIRFilter(np.zeros(hlen), ratio), osf.FIRFilter(np.zeros(hlen), ratio)
        assert a.kernel.lower().endswith(b.kind)
        if ratio != 1:
            ph = 10 * rng.random()
            a.setphase(ph); b.setphase(ph)
        assert (a.phi_idx, a.input_deficit, a.historyLen) == (b.phi_idx, b.input_deficit, b.history_len)
        assert a.timedelay() == b.timedelay()
        yl = int(rng.integers(1, 101))
        assert a.inputlength(yl) == b.inputlength(yl) and a.inputlength(yl, True) == b.inputlength(yl, True)
        assert a.outputlength(yl) == b.outputlength(yl)
        # test/resample.jl:160-161 bracketing invariants on the host mirror
        assert a.outputlength(a.inputlength(yl)) <= yl < a.outputlength(a.inputlength(yl) + 1)
        assert a.outputlength(a.inputlength(yl, True) - 1) < yl <= a.outputlength(a.inputlength(yl, True))


def test_argument_errors_raised_before_device_work():
    with pytest.raises(d.ArgumentError):
        d.filt(np.array([]), 1.0, np.ones(4))                         # dspbase.jl:28
    with pytest.raises(d.ArgumentError):
        d.filt(np.ones(2), 0.0, np.ones(4))                           # dspbase.jl:30
    with pytest.raises(d.ArgumentError):
        d.stft(np.ones(64, dtype=np.complex64), 8, 4, onesided=True)  # periodograms.jl:876
    with pytest.raises(d.DomainError):
        d.stft(np.ones(64), 8, 8)                                     # periodograms.jl:44
    with pytest.raises(d.DomainError):
        d.stft(np.ones(64), 8, 2, nfft=4)                             # periodograms.jl:45
    with pytest.raises(d.DimensionMismatch):
        d.stft(np.ones(64), 8, 2, window=np.ones(7))                  # periodograms.jl:255
    with pytest.raises(d.DomainError):
        d.periodogram(np.ones(64), nfft=32)                           # periodograms.jl:397
    with pytest.raises(d.ArgumentError):
        d.conv(np.ones(300), np.ones(300), "quantum")                 # dspbase.jl:754
    with pytest.raises(d.UnsupportedError):
        d.FIRFilter(np.ones(8), 1.5)
    with pytest.raises(d.DomainError):
        d.FIRFilter(np.ones(8), 3).setphase(-1.0)                     # stream_filt.jl:224
