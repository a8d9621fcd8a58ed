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


def test_overlap_save_plan_choice_without_a_device():
    """mdsp_ols_geometry_for: what mdsp_ols_plan_create would execute, as pure host arithmetic.  The reference's nfft runs as it is up to the largest
    in-LDS transform (8192 Float32 / 4096 Float64), longer filters are re-blocked (one block, then 2 .. 4 partitions), and beyond 16384 / 8192 taps the
    blocks go through the multi-pass engine: 64 rows of the longest single-workgroup transform while the filter is under a quarter of the block, then
    256 rows, then three passes each way on 2^20 points and more; less when one block holds the whole signal."""
    lib = _lib.lib()

    def geo(nb, nx, dt, nfft=0, mode=_lib.OLS_FILT, engine=d.ENGINE_AUTO):
        en, el, ep, eg, er = C.c_int64(), C.c_int64(), C.c_int(), C.c_int(), C.c_int()
        _lib.check(lib.mdsp_ols_geometry_for(nb, nfft, nx, dt, mode, engine, C.byref(en), C.byref(el), C.byref(ep), C.byref(eg), C.byref(er)))
        return en.value, el.value, ep.value, eg.value, er.value

    F, R = d.ENGINE_FUSED, d.ENGINE_ROCFFT
    big = 1 << 28
    assert geo(256, big, _lib.F32) == (2048, 2048 - 255, 1, F, 0)                       # the headline shape: the reference's own block
    assert geo(5120, big, _lib.F32)[:4] == (4096, 2048, 3, F)                           # partitioned: 3 x 2048 taps
    assert geo(16384, big, _lib.F32)[:4] == (8192, 4096, 4, F)
    assert geo(16385, big, _lib.F32) == (1 << 19, (1 << 19) - 16384, 1, F, 64)          # rows form, 64 x 8192
    assert geo(131072, big, _lib.F32) == (1 << 19, (1 << 19) - 131071, 1, F, 64)
    assert geo(131073, big, _lib.F32) == (1 << 21, (1 << 21) - 131072, 1, F, 256)       # 256 x 8192
    assert geo(1 << 19, big, _lib.F32)[4] == 256
    assert geo((1 << 19) + 1, 1 << 26, _lib.F32, nfft=1 << 24) == (1 << 23, (1 << 23) - (1 << 19), 1, F, 0)   # three passes each way: 8 x the filter
    assert geo(8193, big, _lib.F64) == (1 << 18, (1 << 18) - 8192, 1, F, 64)            # Float64: 64 x 4096
    assert geo(65537, big, _lib.C64) == (1 << 20, (1 << 20) - 65536, 1, F, 256)
    assert geo(32768, 80_000, _lib.F32)[0] == 1 << 17 and geo(32768, 80_000, _lib.F32)[4] == 0        # one block of 2^17 points holds the whole signal
    assert geo(32768, 80_000, _lib.F32, mode=_lib.OLS_CONV)[0] == 1 << 18                         # conv: nx + nb - 1 outputs do not fit that block
    assert geo(32768, big, _lib.F32, engine=R)[3] == R and geo(32768, big, _lib.F32, engine=R)[4] == 0
    assert geo(300, 10_000, _lib.F32, nfft=3000)[3] == R                                # a 7-smooth block the caller asked for: rocFFT engine
    with pytest.raises(d.ArgumentError):
        geo(0, 100, _lib.F32)
    with pytest.raises(d.ArgumentError):
        geo(100, 1000, _lib.F32, nfft=64)


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
    with pytest.raises(d.DomainError):
        d.FIRFilter(np.ones(8), -1.5)                                 # stream_filt.jl:151
    with pytest.raises(d.DomainError):
        d.FIRFilter(np.ones(64), 1.5).setphase(-0.5)                  # stream_filt.jl:232
    with pytest.raises(d.DomainError):
        d.FIRFilter(np.ones(8), 3).setphase(-1.0)                     # stream_filt.jl:224


# ---- FIRArbitrary: host-side trajectory and length / phase arithmetic (no device needed) --------------------------
def _c_trajectory(acc, deficit, rate, nphi, xlen, block=64):
    import ctypes as C
    lib = _lib.lib()
    cap = int(xlen * rate / block) + 16
    ax = (C.c_int64 * cap)()
    aa = (C.c_double * cap)()
    nout, dend, aend = C.c_int64(), C.c_int64(), C.c_double()
    _lib.check(lib.mdsp_arb_trajectory(acc, deficit, rate, nphi, xlen, block, ax, aa, cap, C.byref(nout), C.byref(aend), C.byref(dend)))
    na = -(-nout.value // block)
    return np.array(ax[:na]), np.array(aa[:na]), nout.value, aend.value, dend.value


@pytest.mark.parametrize("rate,nphi,acc0,def0,xlen", [
    (1.1, 32, 0.0, 1, 5000), (0.9802414928649835, 32, 7.25, 3, 1822), (3.141592653589793, 32, 0.0, 1, 1000),
    (1 / 55.55, 32, 11.0, 5, 35546), (0.012, 32, 0.0, 1, 1000), (2.5, 7, 3.5, 2, 777), (1e-3, 32, 0.0, 40, 50000),
    (0.5, 32, 0.0, 1, 10), (1.7, 32, 0.0, 12, 5)])
def test_arbitrary_trajectory_bit_exact_with_oracle(rate, nphi, acc0, def0, xlen):
    """update! (stream_filt.jl:567-577) replayed by the library vs the oracle's literal loop: anchors, count, final state."""
    from oracle import stream_filt as osf
    xs, accs, acc_end, def_end = osf.arb_trajectory(acc0, def0, nphi / rate, nphi, xlen)
    ax, aa, nout, aend, dend = _c_trajectory(acc0, def0, rate, nphi, xlen)
    assert nout == len(xs)
    assert aend == acc_end and dend == def_end           # bit-exact Float64 / integer state
    assert np.array_equal(ax, xs[::64]) and np.array_equal(aa, accs[::64])


def _c_trajectory_scan(acc, deficit, rate, nphi, xlen, pilot):
    import ctypes as C
    lib = _lib.lib()
    cap = int(xlen * rate / 16) + 64
    ax = np.zeros(cap, np.int64)
    aa = np.zeros(cap, np.float64)
    nout, dend, aend, used, passes = C.c_int64(), C.c_int64(), C.c_double(), C.c_int(), C.c_int()
    _lib.check(lib.mdsp_arb_trajectory_scan(acc, deficit, rate, nphi, xlen, pilot, ax.ctypes.data_as(C.POINTER(C.c_int64)),
                                            aa.ctypes.data_as(C.POINTER(C.c_double)), cap, C.byref(nout), C.byref(aend), C.byref(dend),
                                            C.byref(used), C.byref(passes)))
    na = -(-nout.value // 16)
    return bool(used.value), passes.value, ax[:na], aa[:na], nout.value, aend.value, dend.value


def test_arbitrary_parallel_scan_is_bit_exact():
    """The device's parallel evaluation of update! (integer image of the IEEE operations + scan of the roundings,
    csrc/arb_scan.h), run through its host emulation, against the serial loop: every anchor, the output count and the final
    state, over up/down-sampling rates, non-power-of-two Nphi, off-grid initial phases, exactly representable rates
    (positions on the wrap point) and a three-level scan."""
    rng = np.random.default_rng(5)
    rates = [160 / 147, 147 / 160, 0.3721, 1.5, 2.7, 1.0000001, 0.99, 3.3, 0.2, 1 / 3, 2 / 3, 6.5, 0.13, 5.99, 1.0, 2.0, 0.5, 4 / 3, np.pi / 3]
    rates += list(np.exp(rng.uniform(np.log(0.13), np.log(6.9), 12)))
    scanned = 0
    cases = []
    for rate in rates:
        for nphi in (32, 48, 7):
            cases.append((rate, nphi, int(rng.integers(60000, 120000) / max(rate, 1)), int(rng.choice([64, 1024, 4096]))))
    cases.append((1.5, 32, 6_000_000, 65536))           # 9e6 outputs: 281k blocks, three scan levels
    for rate, nphi, xlen, pilot in cases:
        acc = float(rng.uniform(0, nphi)) if rng.random() < 0.7 else 0.0
        deficit = int(rng.integers(1, 5))
        ax, aa, nout, aend, dend = _c_trajectory(acc, deficit, rate, nphi, xlen, block=16)
        used, passes, sx, sa, snout, saend, sdend = _c_trajectory_scan(acc, deficit, rate, nphi, xlen, pilot)
        if not used:
            continue
        scanned += 1
        assert (snout, saend, sdend) == (nout, aend, dend), (rate, nphi, xlen)
        assert np.array_equal(sx, ax) and np.array_equal(sa, aa), (rate, nphi, xlen)
    assert scanned >= len(cases) - 3                     # the scan certifies itself on (almost) every case


def test_argument_errors_of_the_array_paths_come_before_device_work():
    """conv for arrays / separable conv / xcorr: the reference's argument errors (dspbase.jl:754, :872-880; test/dsp.jl:172, :347-356)
    are raised on the host, before anything needs a device."""
    with pytest.raises(d.ArgumentError):
        d.conv(np.ones((2, 2)), np.ones((2, 2)), algorithm="quantum")
    with pytest.raises(d.ArgumentError):
        d.conv(np.ones((2, 2)), np.ones(2), np.ones((2, 2)))             # conv(u, v', A) takes two vectors and a matrix
    with pytest.raises(TypeError):
        d.xcorr(np.ones((2, 2)), np.ones((2, 2)))
    with pytest.raises(d.ArgumentError):
        d.xcorr(np.ones(2), np.ones(2), padmode="bug")
    with pytest.raises(d.DimensionMismatch):
        d.xcorr(np.ones(1), np.ones(2), scaling="biased")
    if _lib.device_count() == 0:
        with pytest.raises(d.DeviceError):
            d.conv(np.ones((2, 2)), np.ones((2, 2)))
        with pytest.raises(d.DeviceError):
            d.hilbert(np.ones(8))
        with pytest.raises(d.DeviceError):
            d.resample(np.ones(64), 1.5)


def test_arbitrary_device_replay_update_is_exact():
    """The branch-free update the device replays between anchors (two candidate quotients, Sterbenz-exact remainder) against
    the reference-form update (stream_filt.jl:567-577), step by step: up- and down-sampling, large skips, odd Nphi."""
    import ctypes as C
    lib = _lib.lib()
    rng = np.random.default_rng(9)
    rates = [160 / 147, 147 / 160, 0.3721, 1.5, 2.7, 6.5, 0.13, 0.012, 1 / 55.55, 0.002, 1.0, 2.0, 0.5, 1e-4, 31.7, 1 - 2 ** -40, 1 + 2 ** -40]
    rates += list(np.exp(rng.uniform(np.log(1e-3), np.log(30), 40)))
    for rate in rates:
        for nphi in (32, 7, 48, 1):
            bad = C.c_int64(-1)
            _lib.check(lib.mdsp_arb_replay_check(float(rng.uniform(0, nphi)), float(rate), nphi, 200000, C.byref(bad)))
            assert bad.value == 0, (rate, nphi)


def test_arbitrary_reference_length_regressions():
    """test/resample.jl:96-101 (issue #317): output lengths of resample() at awkward arbitrary rates."""
    from oracle import design as od

    def resample_len(n, rate):
        h = od.resample_filter(float(rate), 32)
        f = d.FIRFilter(h, float(rate))
        f.setphase(f.timedelay())
        out_len = int(np.ceil(n * rate))
        npad = f.inputlength(out_len, round_up=True)
        _, _, nout, _, _ = _c_trajectory(f.phi_accumulator, f.input_deficit, rate, 32, npad)
        assert nout >= out_len                             # checked_resample_output! (stream_filt.jl:722)
        return out_len

    assert resample_len(35546, 1 / 55.55) == 640
    assert resample_len(1822, 0.9802414928649835) == 1786
    assert resample_len(16_367_000 * 2, 10_000_000 / 16_367_000) == 20_000_000
    assert resample_len(1000, 0.012) == 12


def test_arbitrary_host_state_matches_oracle():
    from oracle import stream_filt as osf
    rng = np.random.default_rng(3)
    h = rng.standard_normal(640)
    for rate, nphi in ((1.37, 32), (0.61, 32), (2.2, 20)):
        o = osf.FIRFilter(h, rate, nphi)
        p = d.FIRFilter(h, rate, nphi)
        assert p.kernel == "FIRArbitrary" and p.tapsPerphi == o.taps_per_phi and p.historyLen == o.history_len
        assert p.timedelay() == o.timedelay() and p.delta == o.delta
        for phi in (0.0, 0.3, 9.984375, 17.5):
            o.reset(); p.reset()
            o.setphase(phi); p.setphase(phi)
            assert (p.phi_accumulator, p.phi_idx, p.alpha, p.input_deficit) == (o.phi_acc, o.phi_idx, o.alpha, o.input_deficit)
            for n in (0, 1, 17, 1000, 123457):
                assert p.outputlength(n) == o.outputlength(n)
                assert p.inputlength(n) == o.inputlength(n) and p.inputlength(n, round_up=True) == o.inputlength(n, roundup=True)


def test_window_generators_match_reference_goldens(golden):
    """test/windows.jl:45-128: every window the host can generate against the reference's MATLAB / regression vectors."""
    from conftest import isapprox
    assert np.array_equal(d.rect(128), np.ones(128))
    pairs = [(d.hanning(128), "hanning128"), (d.hann(128), "hanning128"), (d.hamming(128), "hamming128"), (d.triang(128), "triang128"),
             (d.bartlett(128), "bartlett128"), (d.bartlett_hann(128), "bartlett_hann128"), (d.blackman(128), "blackman128"),
             (d.blackmanharris(128, 3), "blackmanharris_3term_128"), (d.blackmanharris(128), "blackmanharris_4term_128"),
             (d.nuttall(128, 3), "nuttall_3term_128"), (d.nuttall(128), "nuttall_4term_128"), (d.kaiser(128, 0.4 / np.pi), "kaiser128_0p4"),
             (d.flattop(128), "flattop"), (d.gaussian(128, 0.2), "gaussian128_0p2"), (d.tukey(128, 0.4), "tukey128_0p4"),
             (d.lanczos(128), "lanczos128"), (d.cosine(128), "cosine128")]
    for got, key in pairs:
        assert isapprox(got, golden[key]), key
    assert isapprox(d.triang(5), d.bartlett(7)[1:6])
    assert d.blackman(128).min() == 0.0
    assert np.array_equal(d.tukey(128, 0), d.rect(128))
    with pytest.raises(d.ArgumentError):
        d.blackmanharris(128, 2)
    with pytest.raises(d.ArgumentError):
        d.nuttall(128, 2)
    with pytest.raises(d.DomainError):
        d.tukey(128, 1.5)


def test_plan_cache_lru_semantics():
    """dsp.jl_amd/_plancache.py: most-recently-used retention, eviction order, content keys of arrays and windows."""
    from dsp_jl_amd import _plancache as pc
    c = pc.PlanCache(maxsize=3)
    made = []

    def make(tag):
        made.append(tag)
        return object()

    a = c.get("a", lambda: make("a"))
    assert c.get("a", lambda: make("a2")) is a and made == ["a"] and (c.hits, c.misses) == (1, 1)
    c.get("b", lambda: make("b")); c.get("c", lambda: make("c"))
    c.get("a", lambda: make("a3"))                       # refresh a: b is now the oldest
    c.get("d", lambda: make("d"))                        # evicts b
    assert made == ["a", "b", "c", "d"]
    c.get("b", lambda: make("b2"))                       # rebuilt, evicts c
    assert made[-1] == "b2" and c.get("a", lambda: make("a4")) is a
    c.clear()
    assert c.get("a", lambda: make("a5")) is not a
    x = np.arange(6, dtype=np.float32)
    assert pc.array_key(x) == pc.array_key(x.copy()) and pc.array_key(x) != pc.array_key(x.astype(np.float64))
    assert pc.array_key(x) != pc.array_key(x.reshape(2, 3)) and pc.array_key(x) != pc.array_key(x[::-1])
    assert pc.window_key(None) is None and pc.window_key(d.hanning) is d.hanning and pc.window_key(x) == pc.array_key(x)
    # memoised designs return fresh arrays with identical contents
    h1, h2 = d.resample_filter(1.2345), d.resample_filter(1.2345)
    assert h1 is not h2 and np.array_equal(h1, h2)
    h1[0] = 123.0
    assert d.resample_filter(1.2345)[0] != 123.0
    t1 = d.dpss(64, 4); t1[0, 0] = 9.0
    assert d.dpss(64, 4)[0, 0] != 9.0


def test_time_split_helpers_cover_every_frame_and_sample_once():
    """frame_shard / frame_span / filt_time_split_span (SURVEY 8e "next"): the shards partition the frames, neighbouring spans overlap
    by n - hop samples, a span holds exactly its frames, and filter halos are nb - 1 samples (clipped at the stream start)."""
    rng = np.random.default_rng(12)
    for _ in range(300):
        n = int(rng.integers(2, 400)); nov = int(rng.integers(0, n)); hop = n - nov
        length = int(rng.integers(n, 20000)); world = int(rng.integers(1, 9))
        K = opg.frame_count(length, n, nov)
        shards = [d.frame_shard(K, r, world) for r in range(world)]
        assert [k for sh in shards for k in sh] == list(range(K))
        prev_hi = None
        for sh in shards:
            lo, hi = d.frame_span(sh, n, nov)
            if len(sh) == 0:
                assert (lo, hi) == (0, 0)
                continue
            assert hi <= length and opg.frame_count(hi - lo, n, nov) == len(sh) and lo == sh.start * hop
            if prev_hi is not None:
                assert prev_hi - lo == n - hop
            prev_hi = hi
        nb = int(rng.integers(1, 300)); lo = int(rng.integers(0, length)); hi = int(rng.integers(lo, length + 1))
        slo, shi = d.filt_time_split_span(lo, hi, nb)
        assert shi == hi and slo == max(0, lo - (nb - 1))


def test_arbitrary_scan_declines_what_it_cannot_model():
    """Rates outside the integer model (more than three rounding bits, more than eight input samples per output) and streams that end
    inside the pilot are reported as not handled -- the product then runs the serial loop."""
    for rate, xlen in ((9.5, 50000), (0.05, 2_000_000), (1.1, 100)):
        used, passes, *_ = _c_trajectory_scan(0.3, 1, rate, 32, xlen, 1024)
        assert not used
    used, passes, sx, sa, snout, saend, sdend = _c_trajectory_scan(0.3, 1, 6.9, 32, 30000, 1024)
    ax, aa, nout, aend, dend = _c_trajectory(0.3, 1, 6.9, 32, 30000, block=16)
    assert used and (snout, saend, sdend) == (nout, aend, dend) and np.array_equal(sx, ax) and np.array_equal(sa, aa)


def test_environment_is_read_in_one_place_and_debug_knobs_are_compiled_out():
    """exec / plan paths never call getenv (tunables are read once, in api_core.hip), and the profiling
    switches (MDSP_ABLATE, ...) do not exist in the product build."""
    import glob
    import re
    from dsp_jl_amd import _lib
    csrc = os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), "csrc")
    offenders = []
    for path in glob.glob(os.path.join(csrc, "*")):
        if os.path.isdir(path) or os.path.basename(path) == "api_core.hip":
            continue
        if re.search(r"\bgetenv\s*\(", open(path, errors="replace").read()):
            offenders.append(os.path.basename(path))
    assert not offenders, offenders
    lib = _lib.lib()
    assert lib.mdsp_debug_knobs() == 0
    os.environ["MDSP_ABLATE"] = "7"
    try:
        assert lib.mdsp_reload_tunables() == 0 and lib.mdsp_debug_knobs() == 0
    finally:
        del os.environ["MDSP_ABLATE"]
        lib.mdsp_reload_tunables()


def test_promote_type_follows_julia_not_numpy():
    """filt(b::Float32, 1, x::Float32) must stay Float32 (numpy's result_type says float64)."""
    from dsp_jl_amd import util
    P = util.promote_type
    f4, f8, c8, c16, i8, i4, u8, u4, b1 = (np.dtype(t) for t in (np.float32, np.float64, np.complex64, np.complex128, np.int64, np.int32, np.uint64, np.uint32, np.bool_))
    assert P(f4, i8, f4) == f4 and P(f4, i8) == f4 and P(i8, f8) == f8 and P(f4, f8) == f8
    assert P(c8, i8) == c8 and P(c8, f8) == c16 and P(f4, c8) == c8 and P(c16, f4) == c16
    assert P(i8, i4) == i8 and P(i4, u4) == u4 and P(i8, u4) == i8 and P(i8, u8) == u8 and P(b1, i4) == i4 and P(b1, b1) == b1
    assert P(np.float16, i8) == np.dtype(np.float16) and P(np.float16, f4) == f4


def _julia_ccalls(src):
    """(name, return type, [argument types]) of every `ccall((:mdsp_x, lib), Ret, (T1, T2, ...), args...)` in the Julia source."""
    out = []
    for m in re.finditer(r"ccall\(\(:(mdsp_[a-z0-9_]+), lib\),\s*([A-Za-z0-9]+),\s*\(", src):
        i, depth, cur, types = m.end(), 1, "", []
        while depth:                                   # walk the type tuple, respecting Ptr{...} / Ref{Ptr{...}} nesting
            c = src[i]
            if c in "({":
                depth += 1
            elif c in ")}":
                depth -= 1
            if depth == 0:
                break
            if c == "," and depth == 1:
                types.append(cur.strip()); cur = ""
            else:
                cur += c
            i += 1
        if cur.strip():
            types.append(cur.strip())
        out.append((m.group(1), m.group(2), types))
    return out


def _jl_kind(t):
    """Julia ccall type -> ABI class."""
    scal = {"Cint": "i32", "Int64": "i64", "UInt64": "u64", "Cdouble": "f64", "Float64": "f64", "Cfloat": "f32", "Csize_t": "u64", "Cstring": "cstr"}
    if t in scal:
        return scal[t]
    m = re.fullmatch(r"(?:Ptr|Ref)\{(.*)\}", t)
    assert m, f"unrecognised Julia ccall type {t!r}"
    inner = m.group(1)
    if inner == "Cvoid":
        return "ptr:void"
    if re.fullmatch(r"(?:Ptr|Ref)\{Cvoid\}", inner):
        return "ptr:pvoid"
    return "ptr:" + {"Int64": "i64", "Cint": "i32", "Cdouble": "f64", "Float64": "f64", "Cfloat": "f32", "UInt8": "u8"}[inner]


def _ct_kind(t):
    """ctypes prototype entry -> ABI class."""
    import ctypes as C
    table = {C.c_int: "i32", C.c_int64: "i64", C.c_uint64: "u64", C.c_double: "f64", C.c_float: "f32", C.c_size_t: "u64", C.c_void_p: "ptr:void", C.c_char_p: "cstr",
             C.POINTER(C.c_void_p): "ptr:pvoid", C.POINTER(C.c_int64): "ptr:i64", C.POINTER(C.c_int): "ptr:i32", C.POINTER(C.c_double): "ptr:f64",
             C.POINTER(C.c_float): "ptr:f32"}
    return table[t]


def test_julia_twin_binds_exported_symbols_with_matching_argument_types():
    """julia/MI355DSP.jl cannot be run here (no Julia in the image).  What CAN be checked without running it: every
    `ccall((:sym, lib), Ret, (types...), args...)` names an exported symbol, declares as many arguments as the ctypes prototype of the same
    entry point, and every argument (and the return value) has the same ABI class -- Cint vs Int64 vs Cdouble vs Csize_t, and for pointers
    what they point to (an Int64 / Cint mix-up would pass an arity check and corrupt the call).  VERDICT r2 item 7: at least 95 of the
    library's symbols are bound."""
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "julia", "MI355DSP.jl")).read()
    calls = _julia_ccalls(src)
    assert len(calls) >= 100
    lib = _lib.lib()
    for name, ret, types in calls:
        assert name in _lib.PROTOTYPES and hasattr(lib, name), name
        cret, cargs = _lib.PROTOTYPES[name]
        assert len(types) == len(cargs), (name, types, len(cargs))
        assert _jl_kind(ret) == _ct_kind(cret), (name, "return", ret)
        for k, (jt, ct) in enumerate(zip(types, cargs)):
            jk, ck = _jl_kind(jt), _ct_kind(ct)
            # a typed Julia pointer may stand where the C side takes void* (e.g. a Vector{UInt8} id); the reverse is a bug
            assert jk == ck or (ck == "ptr:void" and jk.startswith("ptr:") and jk != "ptr:pvoid"), (name, k, jt, ck)
    bound = {c[0] for c in calls}
    assert len(bound) >= 95, (len(bound), sorted(set(_lib.PROTOTYPES) - bound))
    # the drop-in surface the reference exports on this path (Filters/Filters.jl:24-72, periodograms.jl:7-14, multitaper.jl)
    for fn in ("filt!", "fftfilt!", "tdfilt!", "conv!", "welch_pgram!", "arraysplit", "hilbert", "xcorr", "DF2TFilter", "filtfilt", "resample_filter",
               "mt_pgram!", "mt_spectrogram!", "mt_cross_power_spectra!", "mt_coherence!", "struct Periodogram", "struct Spectrogram", "Base.time(p::Spectrogram)"):
        assert fn in src, fn


def test_matrix_core_tile_choice_round3():
    # Round 3: decimating shapes whose windows leave room for a small tile only no longer take the largest tile whatever its wave
    # count (two or one multiplying waves per CU at 1//8, 1//16; five at 1//3, 3//8: idle or doubly loaded SIMDs) but the cheapest by the cost
    # line of fir_mm_geo; interpolators with one column block take four row groups (three workgroups per CU).  MDSP_FIR_MM_NG=8 is round 2's rule.
    import ctypes as C
    from dsp_jl_amd import _lib
    lib = _lib.lib()
    out = (C.c_int64 * 12)()

    def geo(L, M, hlen, tdt=_lib.F32, xdt=_lib.F32):
        _lib.check(lib.mdsp_fir_mm_geometry(L, M, hlen, tdt, xdt, out))
        ok, RB, Lr, Mr, NB, NG, T, CH, CS, nd, ns, lds = list(out)
        return dict(ok=ok, NB=NB, NG=NG, CH=CH, nd=nd, ns=ns, lds=lds, rows=16 * CH * NG)

    try:
        for L, M, hlen in ((1, 8, 293), (1, 16, 583), (1, 3, 111), (3, 8, 295), (1, 4, 147)):
            _lib.set_tunable("MDSP_FIR_MM_NG", 8)
            old = geo(L, M, hlen)
            _lib.set_tunable("MDSP_FIR_MM_NG", None)
            new = geo(L, M, hlen)
            assert new["ok"] and new["lds"] <= 160 * 1024
            assert new["NG"] % 4 == 0 or new["NG"] > old["NG"], (L, M, old, new)      # every SIMD gets a multiplying wave (or at least more of them do)
            assert new["rows"] * 2 >= old["rows"], (L, M, old, new)                   # ... without giving up more than half of the tile
        assert geo(1, 8, 293) == dict(ok=1, NB=1, NG=8, CH=1, nd=2, ns=4, lds=105472, rows=128)   # 11 outputs per row: 94 k-steps, taps in registers
        assert geo(1, 2, 75)["NG"] == 6 and geo(1, 2, 75)["CH"] == 4                # measured: the large tile wins here (0.45 against 0.53 ms)
        for L, M, hlen in ((2, 1, 75), (3, 2, 111), (4, 1, 149), (5, 3, 185)):
            g = geo(L, M, hlen)
            assert g["NG"] == 4 and g["nd"] == 2 and g["ns"] == 2 and 3 * g["lds"] <= 160 * 1024, (L, M, g)   # 4 + 2 + 2 waves, three workgroups per CU
        assert geo(160, 147, 5120)["NG"] == 1 and geo(160, 147, 5120)["CH"] == 4    # ten column blocks: unchanged
    finally:
        _lib.set_tunable("MDSP_FIR_MM_NG", None)


def test_matrix_core_last_resort_tiles():
    # Round 4: shapes whose tile misses the LDS by a kilobyte or two no longer fall to the generic kernel (0.03 - 0.07 of the roofline there):
    # decimators take rows of fewer rounds, Float32 windows of 33 - 40 k-steps the 40-step register form with single-chunk waves.  Shapes that
    # fitted before keep their geometry (the forms are tried only after everything else failed); MDSP_FIR_MM_TIGHT=0 is the old behaviour.
    import ctypes as C
    from dsp_jl_amd import _lib
    lib = _lib.lib()
    out = (C.c_int64 * 12)()

    def geo(L, M, hlen, tdt, xdt):
        _lib.check(lib.mdsp_fir_mm_geometry(L, M, hlen, tdt, xdt, out))
        return list(out)

    shapes = [(1, 16, 583, _lib.F64, _lib.C64), (160, 441, 16001, _lib.F32, _lib.C32), (1, 32, 1100, _lib.F64, _lib.F64), (1, 32, 1100, _lib.F32, _lib.C32)]
    keep = [(160, 147, 5120, _lib.F32, _lib.F32), (1, 16, 583, _lib.F32, _lib.F32), (160, 441, 16001, _lib.F64, _lib.F64), (1, 8, 293, _lib.F64, _lib.C64), (147, 160, 5881, _lib.F32, _lib.C32)]
    try:
        _lib.set_tunable("MDSP_FIR_MM_TIGHT", 0)
        assert all(geo(*sh)[0] == 0 for sh in shapes)
        before = [geo(*sh) for sh in keep]
        _lib.set_tunable("MDSP_FIR_MM_TIGHT", None)
        assert [geo(*sh) for sh in keep] == before
        g = geo(1, 16, 583, _lib.F64, _lib.C64)
        assert g[:5] == [1, 13, 13, 208, 1] and g[7] == 1 and g[-1] <= 160 * 1024          # 13 rounds per row instead of 15
        g = geo(160, 441, 16001, _lib.F32, _lib.C32)
        assert g[:5] == [1, 1, 160, 441, 10] and g[6] == 40 and g[7] == 1 and g[-1] == 160 * 1024   # 40 k-steps of taps in registers, 16 rows per tile: exactly the LDS
        for sh in shapes:
            g = geo(*sh)
            assert g[0] == 1 and g[-1] <= 160 * 1024 and (sh[0] >= 16 or 1 <= g[1] * sh[0] <= 16), (sh, g)
        assert geo(147, 160, 5881, _lib.F64, _lib.C64)[0] == 0                               # two output buffers of 41 KiB beside 2 x 45 KiB of samples: still the generic kernel
    finally:
        _lib.set_tunable("MDSP_FIR_MM_TIGHT", None)


def test_polyphase_choice_file_overrides_one_shape_only(tmp_path):
    """The per-box choice file (common.h FirChoice; written by tools/tune_fir.py TUNE_PERSIST=1): a line names a shape -- reduced L, M, taps, dtypes -- and the
    polyphase knobs this box measured faster for it; filters of that shape are dispatched under those values, every other shape under the library's rule.
    Checked on the pure-host geometry entry (no device): malformed lines and unknown knobs are skipped."""
    lib = _lib.lib()

    def geo(L, M, n, td=_lib.F32, xd=_lib.F32):
        out = (C.c_int64 * 12)()
        _lib.check(lib.mdsp_fir_mm_geometry(L, M, n, td, xd, out))
        return list(out)

    base_a, base_b, f64_before = geo(3, 2, 96), geo(2, 1, 75), geo(3, 2, 96, _lib.F64, _lib.F64)
    assert base_a[0] == 1 and base_a[5] > 2                      # ok, NG of the library's rule
    f = tmp_path / "fir_choice.txt"
    key = f"# mi355dsp-fir-choices v{lib.mdsp_version()} gfx950\n"
    f.write_text(key + "# a comment\n3 2 96 0 0 MDSP_FIR_MM_NG=2,MDSP_NOT_A_KNOB=7\nthis line is not a shape\n6 4 96 0 0 MDSP_FIR_MM_NG=1\n")
    try:
        _lib.set_tunable("MDSP_FIR_CHOICE_FILE", str(f))
        got = geo(3, 2, 96)
        assert got[0] == 1 and got[5] == 2 and got[7] == base_a[7]       # NG capped for this shape (the unreduced 6//4 line is never looked up: ratios are reduced)
        assert geo(6, 4, 96) == got                                      # ... which is the same filter
        assert geo(3, 2, 97)[5] == base_a[5] and geo(2, 1, 75) == base_b                     # other shapes: untouched
        assert geo(3, 2, 96, _lib.F64, _lib.F64) == f64_before                               # another dtype is another shape
        # a file without the key line of THIS library version / architecture is ignored as a whole (round 6)
        stale = tmp_path / "stale.txt"
        stale.write_text("# mi355dsp-fir-choices v1 gfx942\n3 2 96 0 0 MDSP_FIR_MM_NG=2\n")
        _lib.set_tunable("MDSP_FIR_CHOICE_FILE", str(stale))
        assert geo(3, 2, 96) == base_a
    finally:
        _lib.set_tunable("MDSP_FIR_CHOICE_FILE", str(tmp_path / "absent.txt"))
    assert geo(3, 2, 96) == base_a
    _lib.set_tunable("MDSP_FIR_CHOICE_FILE", None)
    assert geo(3, 2, 96) == base_a   # no variable, no file: nothing under ~/.cache is looked at (opt-in since round 6)


def test_matrix_core_polyphase_geometry_is_consistent():
    # mdsp_fir_mm_geometry is the host arithmetic that sizes the matrix-core polyphase kernel (rows of RB rounds, blocks of 16 outputs,
    # k-steps, LDS buffers, wave roles): pure integer code, checked here without a device over random ratios and tap counts.
    import ctypes as C
    from math import gcd
    from dsp_jl_amd import _lib
    lib = _lib.lib()
    out = (C.c_int64 * 12)()

    def geo(L, M, hlen, tdt, xdt):
        _lib.check(lib.mdsp_fir_mm_geometry(L, M, hlen, tdt, xdt, out))
        return list(out)

    assert geo(160, 147, 5120, _lib.F32, _lib.F32) == [1, 1, 160, 147, 10, 1, 12, 4, 1, 2, 2, 161792]      # BASELINE config 5
    assert geo(320, 294, 5120, _lib.F32, _lib.F32)[:5] == [1, 1, 160, 147, 10]                             # ratios are reduced first
    g = geo(2, 1, 75, _lib.F32, _lib.F32)
    assert g[0] == 1 and g[1] == 7 and g[2] == 14 and g[3] == 7                                             # a row is 7 rounds: odd sample stride
    assert geo(250, 249, 4000, _lib.F32, _lib.F32)[:5] == [1, 1, 250, 249, 16]                              # L > 192: two column blocks per wave
    assert geo(2000, 1999, 40000, _lib.F32, _lib.F32)[0] == 0                                               # L > 1024
    assert geo(160, 147, 5120, _lib.F64, _lib.F32)[0] == 0                                                  # Float64 taps on a Float32 signal
    assert geo(160, 147, 5120, _lib.F64, _lib.C64)[-1] == 160 * 1024                                        # ComplexF64: 16 rows per wave, unpadded output rows: exactly the LDS
    assert geo(147, 160, 5881, _lib.F64, _lib.C64)[0] == 0                                                  # tile does not fit the LDS
    assert geo(147, 160, 5881, _lib.F32, _lib.F32)[:7] == [1, 1, 147, 160, 10, 1, 16]                       # 48 kHz -> 44.1 kHz (padded rows: M = 160 is a multiple of 32)
    assert geo(160, 441, 16001, _lib.F32, _lib.F32)[7] == 2                                                 # 44.1 kHz -> 16 kHz
    assert geo(1, 2, 48 * 4 + 1, _lib.F32, _lib.F32)[6:8] == [64, 2]                                        # more than 192 window positions: two chunks per wave, 64 k-steps of taps in registers (round 3)
    assert geo(1, 2, 100 * 4 + 1, _lib.F32, _lib.F32)[6] == 112                                             # more than 384: taps fetched per tile
    assert geo(1, 2, 5000, _lib.F32, _lib.F32)[0] == 0                                                      # more than 4096 window positions
    rng = np.random.default_rng(42)
    fits = 0
    for _ in range(3000):
        L, M = int(rng.integers(1, 1100)), int(rng.integers(1, 700))
        hlen = int(rng.integers(1, 9000))
        tdt = [_lib.F32, _lib.F64][int(rng.integers(0, 2))]
        xdt = [_lib.F32, _lib.F64, _lib.C32, _lib.C64][int(rng.integers(0, 4))]
        ok, RB, Lr, Mr, NB, NG, T, CH, CS, nd, ns, lds = geo(L, M, hlen, tdt, xdt)
        g0 = gcd(L, M); Lq, Mq = L // g0, M // g0
        dbl = tdt == _lib.F64 or xdt in (_lib.F64, _lib.C64)
        if not ok:
            continue
        fits += 1
        tp = -(-hlen // Lq)
        assert dbl == (xdt in (_lib.F64, _lib.C64))                       # compute type == signal type
        assert Lq <= 1024 and Lr == RB * Lq and Mr == RB * Mq and (RB == 1 if Lq >= 16 else Lr <= 16)
        NBW = NB if NB <= 12 else -(-NB // -(-NB // 12))                 # multiplying waves per row group (several column blocks each when L > 192)
        assert NB == -(-Lr // 16) and 1 <= NG <= 8 and NBW * NG + nd + ns <= 16 and nd >= 1 and ns >= 1
        assert CS == (2 if xdt in (_lib.C32, _lib.C64) else 1) and CH in ((4, 2, 1) if (not dbl and CS == 1) else (2, 1))
        dmax = ((Lq - 1) + (min(Lr, 16) - 1) * Mq) // Lq
        assert 4 * T >= tp + dmax and T <= 1024 and (T <= (48 if dbl else 96) or T % 8 == 0)   # every tap of every column of a block has a k-step
        esz = 8 if dbl else 4
        rows = 16 * CH * NG
        dw = (esz // 4) * CS
        pitch = -(-((Mr + 4 * T + 4) * dw) // 256) * 256 + 4               # rows staged one by one (sample strides that are multiples of 8) ...
        lin = -(-((rows * Mr + Mr + 4 * T + 4) * dw) // 256) * 256           # ... or the tile as one run
        def prun(S):   # ... or one run with 4 dwords of padding behind every 256-dword granule (round 3)
            return -(-((-(-((rows * Mr + Mr + 4 * T + 4) * dw) // 256) + 1) * 260) // 256) * 256
        def total(padded, rowwise):
            ibuf = -(-(rows * pitch) // 256) * 256 if rowwise == 1 else prun(rowwise // 2) if rowwise >= 2 else lin
            return 2 * 4 * ibuf + 2 * rows * (Lr * CS if NB == 1 else 16 * NB * CS + (16 // esz if padded else 0)) * esz
        need = next((v for v in [total(pd, rw) for pd in (True, False) for rw in (0, 1, 2, 4, 8)] if v == lds), None)
        assert need is not None and lds <= 160 * 1024
    assert fits > 300


def test_experiment_knobs_are_not_environment_variables(tmp_path):
    """Round 6 (VERDICT r5 item 8 iv): a product build reads fifteen documented environment variables; what steers an experiment is a knob set through
    mdsp_set_knob -- its name in the environment does nothing (unless the library was built with -DMDSP_DEBUG_KNOBS)."""
    import os
    lib = _lib.lib()

    def geo(L, M, n):
        out = (C.c_int64 * 12)()
        _lib.check(lib.mdsp_fir_mm_geometry(L, M, n, _lib.F32, _lib.F32, out))
        return list(out)

    base = geo(3, 2, 96)
    assert base[5] > 2
    os.environ["MDSP_FIR_MM_NG"] = "2"
    try:
        _lib.check(lib.mdsp_reload_tunables())
        if not lib.mdsp_debug_knobs():
            assert geo(3, 2, 96) == base                  # the environment does not reach a knob
    finally:
        del os.environ["MDSP_FIR_MM_NG"]
        _lib.check(lib.mdsp_reload_tunables())
    _lib.set_tunable("MDSP_FIR_MM_NG", 2)                 # ... mdsp_set_knob does
    try:
        assert geo(3, 2, 96)[5] == 2
        _lib.check(lib.mdsp_reload_tunables())            # and a reload keeps what was set
        assert geo(3, 2, 96)[5] == 2
    finally:
        _lib.set_tunable("MDSP_FIR_MM_NG", None)
    assert geo(3, 2, 96) == base
    with pytest.raises(Exception):
        _lib.check(lib.mdsp_set_knob(b"MDSP_NOT_A_KNOB", 1, 0))
    # the header names exactly the variables the host module lists
    hdr = open(os.path.join(ROOT, "include", "mi355dsp.h")).read()
    block = hdr[hdr.index("Fifteen optional environment variables"):hdr.index("int mdsp_reload_tunables")]
    named = set()
    for m in re.finditer(r"MDSP_[A-Z0-9_]+(?: / _[A-Z0-9_]+)*", block):
        parts = m.group(0).split(" / ")
        named.add(parts[0])
        for suffix in parts[1:]:
            named.add(parts[0][:parts[0].rindex("_")] + suffix)
    named -= {"MDSP_ABLATE", "MDSP_DEBUG_KNOBS"}
    assert named == set(_lib.ENV_VARIABLES), named ^ set(_lib.ENV_VARIABLES)
    assert len(_lib.ENV_VARIABLES) == 15
