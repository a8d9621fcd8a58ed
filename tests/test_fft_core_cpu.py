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
