import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """The reference's golden vectors, packed by tests/golden/make_golden.py."""
    return np.load(os.path.join(ROOT, "tests", "golden", "dsp_golden.npz"))


def isapprox(a, b, rtol=None, atol=0.0):
    """Julia's ``isapprox`` on arrays: norm(a-b) <= max(atol, rtol*max(norm(a), norm(b))), rtol=sqrt(eps)."""
    a = np.asarray(a)
    b = np.asarray(b)
    if a.shape != b.shape:
        return False
    if rtol is None:
        rtol = np.sqrt(np.finfo(np.result_type(a.dtype, b.dtype, np.float32)).eps) if atol == 0 else 0.0
    a64 = a.astype(np.complex128 if (a.dtype.kind == "c" or b.dtype.kind == "c") else np.float64).ravel()
    b64 = b.astype(a64.dtype).ravel()
    return np.linalg.norm(a64 - b64) <= max(atol, rtol * max(np.linalg.norm(a64), np.linalg.norm(b64)))


def relerr(a, ref):
    """Norm-wise relative error ||a-ref|| / ||ref|| in double precision."""
    a = np.asarray(a)
    ref = np.asarray(ref)
    ct = np.complex128 if (a.dtype.kind == "c" or ref.dtype.kind == "c") else np.float64
    d = np.linalg.norm(a.astype(ct).ravel() - ref.astype(ct).ravel())
    n = np.linalg.norm(ref.astype(ct).ravel())
    return d / n if n > 0 else d


@pytest.fixture(scope="session")
def approx():
    return isapprox


EPS32 = 2.0 ** -24          # unit roundoff of Float32


def ulps_of_max(a, ref, axis=None):
    """Element-wise error in units of (Float32 unit roundoff x the largest reference magnitude along `axis`): max |a - ref| / (2^-24 max|ref|).
    north_star asks for a STATED ulp tolerance; the parity tests assert this figure against c * log2(nfft) (FFT paths) or c * taps (dot products)
    next to the norm-wise bound -- a handful of badly wrong small bins cannot hide in a norm."""
    a = np.asarray(a)
    ref = np.asarray(ref)
    ct = np.complex128 if (a.dtype.kind == "c" or ref.dtype.kind == "c") else np.float64
    err = np.abs(a.astype(ct) - ref.astype(ct))
    scale = np.max(np.abs(ref.astype(ct)), axis=axis, keepdims=axis is not None)
    return float(np.max(err / (EPS32 * scale)))
