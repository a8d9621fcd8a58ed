import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, ctypes as C
from fractions import Fraction
import dsp_jl_amd as d
from dsp_jl_amd import _lib
from oracle import stream_filt as osf
lib=_lib.lib(); _lib.check(lib.mdsp_init(0))
rng=np.random.default_rng(7)
for (L,M) in ((441,160),(160,441),(147,160)):
    for dt in (np.complex128, np.complex64, np.float64):
        x=(rng.standard_normal(60000)+(1j*rng.standard_normal(60000) if np.dtype(dt).kind=='c' else 0)).astype(dt)
        y=np.asarray(d.resample(x, Fraction(L,M)))
        ref=osf.resample(x.astype(np.complex128 if np.dtype(dt).kind=='c' else np.float64), Fraction(L,M))
        e=np.linalg.norm(y-ref)/np.linalg.norm(ref)
        h=np.asarray(d.resample_filter(Fraction(L,M)), dtype=np.float32 if dt==np.complex64 else np.float64)
        fh, path = C.c_void_p(), C.c_int(-1)
        xd={np.complex128:_lib.C64,np.complex64:_lib.C32,np.float64:_lib.F64}[dt]
        _lib.check(lib.mdsp_fir_create(C.byref(fh), h.ctypes.data_as(C.c_void_p), len(h), L, M, _lib.F32 if dt==np.complex64 else _lib.F64, xd, 1))
        _lib.check(lib.mdsp_fir_kernel_path(fh, 60000, C.byref(path)))
        print(L,M,np.dtype(dt).name,'relerr',e,'path',path.value, len(y), len(ref))
