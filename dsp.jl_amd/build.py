"""Build libmi355dsp.so in-tree with hipcc for gfx950 (no GPU needed: hipcc cross-compiles).

    python dsp.jl_amd/build.py [--force] [--verbose]

One object per translation unit (parallel), then a shared link against rocFFT.  The .so lands next to this
file so it travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libmi355dsp.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")

SOURCES = ["api_core.hip", "rocfft_wrap.hip", "ols.hip", "spectral.hip", "fir.hip", "convnd.hip", "comm.hip", "hostpath.hip", "plancache.hip"]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
          "-Wno-implicit-fallthrough", "-ffp-contract=on", f"-I{ROCM}/include"]
LDFLAGS = ["-shared", "-fPIC", "--offload-arch=gfx950", f"-L{ROCM}/lib", "-lrocfft", "-ldl", "-lpthread", f"-Wl,-rpath,{ROCM}/lib"]   # librccl is bound at run time (comm.hip)


def _deps_hash(src: str) -> str:
    h = hashlib.sha256()
    h.update(" ".join(CFLAGS).encode())
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".h", ".hpp")) or name == src:
            with open(os.path.join(CSRC, name), "rb") as f:
                h.update(name.encode())
                h.update(f.read())
    with open(os.path.join(HERE, "..", "include", "mi355dsp.h"), "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def _compile(src: str, force: bool, verbose: bool) -> tuple[str, bool]:
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    stamp = obj + ".hash"
    want = _deps_hash(src)
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == want:
        return obj, False
    cmd = [HIPCC, *CFLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr)
    with open(stamp, "w") as f:
        f.write(want)
    return obj, True


def build(force: bool = False, verbose: bool = False, tag: str = "", cflags: str = "") -> str:
    """tag / cflags: an experimental variant (extra -D flags) built next to the product library as libmi355dsp_<tag>.so."""
    global OBJ, LIB, CFLAGS
    if tag:
        OBJ = os.path.join(HERE, "csrc", "_obj_" + tag)
        LIB = os.path.join(HERE, f"libmi355dsp_{tag}.so")
        CFLAGS = CFLAGS + cflags.split()
    if not os.path.exists(HIPCC):
        raise RuntimeError(f"hipcc not found at {HIPCC}")
    os.makedirs(OBJ, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        res = list(ex.map(lambda s: _compile(s, force, verbose), SOURCES))
    objs = [o for o, _ in res]
    if any(changed for _, changed in res) or not os.path.exists(LIB) or force:
        cmd = [HIPCC, *objs, *LDFLAGS, "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    def _opt(name):
        return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else ""
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv, tag=_opt("--tag"), cflags=_opt("--cflags")))
