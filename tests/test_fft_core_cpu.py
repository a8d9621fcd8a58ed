"""Host emulation of the workgroup FFTs (tests/cpu_harness/fft_emul.cpp): the SAME pass code the kernels run (dsp.jl_amd/csrc/fft_lds.h),
executed thread by thread with g++ and checked against a long-double DFT -- register-resident Stockham passes in every geometry the
kernels instantiate (lane-permuted and wave-private exchange variants included) and the mixed-radix LDS passes for 7-smooth sizes."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fft_passes_on_the_host(tmp_path):
    exe = str(tmp_path / "fft_emul")
    r = subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cpu_harness", "fft_emul.cpp"), "-o", exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-3000:]
    assert "wave-private last exchange" in r.stdout and "gen N= 3000" in r.stdout and "windowed first pass" in r.stdout


def test_compile_time_schedules_on_the_host(tmp_path):
    """tests/cpu_harness/ct_layout.cpp: every entry of the two schedule tables (csrc/ct_sched.h: 23 small-radix + 13 composite-radix schedules) run on the
    host with the schedule type's own index arithmetic -- butterflies per pass, twiddle indices, group padding, buffer size, two buffers and one -- and
    the butterflies of fft_lds.h, against a Float64 DFT."""
    exe = str(tmp_path / "ct_layout")
    r = subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cpu_harness", "ct_layout.cpp"), "-o", exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-3000:]
    assert "36 schedules" in r.stdout and "N  3072 T 256 flags 1536 passes 4 NP  3136" in r.stdout and "FAIL" not in r.stdout


def test_single_workgroup_tables_on_the_host(tmp_path):
    """The same harness over the single-workgroup tables of round 6 (csrc/ctbig_sizes.h: 177 schedules from 1280 to 16384 points, three or four passes at 192 .. 1024
    threads on one LDS buffer): radices multiply to N, the padded buffer and the twiddle tables fit 160 KiB (static_assert), and the passes -- run with the schedule
    type's index arithmetic on one buffer -- give the DFT (every 61st bin against a Float64 sum above 4096 points)."""
    exe = str(tmp_path / "ct_layout_big")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-DCT_BIG=61", os.path.join(ROOT, "tests", "cpu_harness", "ct_layout.cpp"), "-o", exe], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "FAIL" not in r.stdout, r.stdout[-3000:]
    assert "177 schedules: OK" in r.stdout and "N 16384 T 512" in r.stdout, r.stdout[-500:]


def test_multipass_transform_on_the_host(tmp_path):
    """tests/cpu_harness/bigfft_emul.cpp: the tile / sub-pass / store phases of the multi-pass engine (csrc/bigfft_pass.h) and its planner
    (csrc/bigfft_plan.h: factorisation, radix schedules, two-level twiddle tables) run thread by thread on the host against a Float64 DFT --
    two, three and four passes, powers of two, the 2^a 5^b sizes of the default arguments, odd sizes with partial tiles, both precisions, every two-stage
    geometry (256 = 16 x 16, 128 = 16 x 8 and 8 x 16, 64 = 8 x 8 and 16 x 4, 32 = 4 x 8).  And the ROWS FORM of the long-filter convolution (bigfft.hip
    run_ols_rows): column pass with the inter-pass twiddle, per row transform x spectrum row H'[k1 S + k2] = H[k1 + R0 k2] x inverse transform x inverse twiddle,
    column pass back on the conjugate with tables of ones -- against the circular convolution, for 64 / 128 / 256 rows."""
    exe = str(tmp_path / "bigfft_emul")
    r = subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cpu_harness", "bigfft_emul.cpp"), "-o", exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-3000:]
    assert "N=   125000 f32 P=2" in r.stdout and "P=4" in r.stdout and "FAIL" not in r.stdout
    assert r.stdout.count("rows form") == 4 and "16x8 column pass" in r.stdout


def test_runtime_schedule_kernel_on_the_host(tmp_path):
    """tests/cpu_harness/gx_emul.cpp: the run-time-schedule spectral kernel's planner (csrc/gx_sched.h, with the device's planning parameters) and pass code
    (csrc/gx_pass.h) run thread by thread on the host -- one buffer of exactly the planned size, read | barrier | butterflies + write per pass, range and
    collision checks on every LDS index, every radix 2 .. 16, two-level twiddle tables -- and the fused column step (nfft = R0 x S, decimation in
    frequency with the kernel's tables) against a Float64 DFT, both precisions."""
    exe = str(tmp_path / "gx_emul")
    r = subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "cpu_harness", "gx_emul.cpp"), "-o", exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    for prec in ("f32", "f64"):
        r = subprocess.run([exe, prec], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-3000:]
        assert "columns nfft  16384 = " in r.stdout and "too large" not in r.stdout and "no schedule" not in r.stdout
