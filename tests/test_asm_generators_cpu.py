"""The hand-allocated Welch kernel (csrc/welch_w64c_asm.s) is GENERATED: tools/gen_welch_asm_c.py (on the machinery of tools/gen_welch_asm.py, the first form's generator) emits the dataflow on
virtual registers, list-schedule it, assign the 256 VGPRs, insert the wait counts and hazard pads -- and carry a lane-level emulator of the ~15 opcodes they
use.  Without a GPU this checks (i) the emulated instruction lists against numpy (|FFT(w (a + i b))|^2 accumulated over three consecutive units), together
with the independent wait-count replay; (ii) that the committed .s file IS what the generator emits; (iii) that it assembles for gfx950.  (Round 5 removed
the two hand-allocated kernels that lost their A/Bs -- the first Welch form, which re-read the shared half-frame, and the overlap-save kernel -- and the
latter's generator.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.mark.parametrize("gen", ["gen_welch_asm.py", "gen_welch_asm_c.py"])
def test_generated_kernel_emulates_correctly(gen):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", gen), "--check"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 problems" in r.stdout and "relerr" in r.stdout, r.stdout


@pytest.mark.parametrize("gen,mod,out", [("gen_welch_asm_c", "kernel_text", "welch_w64c_asm.s")])
def test_committed_assembly_is_the_generators_output_and_assembles(tmp_path, gen, mod, out):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        m = __import__(gen)
    finally:
        sys.path.pop(0)
    text, nbody = getattr(m, mod)()
    committed = open(os.path.join(ROOT, "dsp.jl_amd", "csrc", out)).read()
    assert text == committed, f"csrc/{out} is stale: run python tools/{gen}.py"
    assert nbody > 1500
    if not os.path.exists(os.path.join(LLVM, "clang")):
        pytest.skip("no ROCm assembler here")
    obj = str(tmp_path / "k.o")
    r = subprocess.run([os.path.join(LLVM, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", os.path.join(ROOT, "dsp.jl_amd", "csrc", out), "-o", obj],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", obj], capture_output=True, text=True, timeout=300).stdout
    assert d.count("v_pk_") > 1000 and "v_permlane32_swap" in d and "scratch_" not in d
