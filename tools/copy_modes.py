import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, dsp_jl_amd as d
from dsp_jl_amd import _lib
lib=_lib.lib(); _lib.check(lib.mdsp_init(0))
st=torch.cuda.current_stream().cuda_stream
def ev():
    e=C.c_void_p(); _lib.check(lib.mdsp_event_create(C.byref(e))); return e
e0,e1=ev(),ev()
for log2n in (28, 30):
    n=1<<log2n
    x=torch.randn(n, device="cuda"); y=torch.empty_like(x)
    for mode in (0,1,2,3,4,5):
        for wgs in (4,8,16,32):
            f=lambda: _lib.check(lib.mdsp_copy_bench_mode(y.data_ptr(), x.data_ptr(), n*4, mode, wgs, st))
            f(); torch.cuda.synchronize(); ts=[]
            for _ in range(5):
                _lib.check(lib.mdsp_event_record(e0, st)); f(); _lib.check(lib.mdsp_event_record(e1, st))
                ms=C.c_float(); _lib.check(lib.mdsp_event_elapsed_ms(e0,e1,C.byref(ms))); ts.append(ms.value)
            b = (2 if mode<4 else 1)*4.0*n
            print(f"2^{log2n} mode {mode} wgs {wgs}: {min(ts):.4f} ms {b/min(ts)/1e6:.0f} GB/s")
    # torch's own copy
    ts=[]
    for _ in range(5):
        _lib.check(lib.mdsp_event_record(e0, st)); y.copy_(x); _lib.check(lib.mdsp_event_record(e1, st))
        ms=C.c_float(); _lib.check(lib.mdsp_event_elapsed_ms(e0,e1,C.byref(ms))); ts.append(ms.value)
    print(f"2^{log2n} torch copy_: {min(ts):.4f} ms {8.0*n/min(ts)/1e6:.0f} GB/s")
    del x,y
