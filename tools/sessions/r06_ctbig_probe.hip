// Compile probe of one gen_ct_kernel instantiation (tools/sessions/r06_ctbig_pick.py: hipcc -DTRY_SIZES(X)=X(N,T,flags,radices...) -Idsp.jl_amd/csrc --cuda-device-only -c).
#include <algorithm>
#include "common.h"
#include "devio.h"
#include "fft_lds.h"
#include "hostfft.h"
#include "spectral_ctcols.h"
using namespace mdsp;
using mdsp::fft::cx;
namespace {
#include "spectral_gen.h"
#define F (16 | 512 | 2048)
#define FL (16 | 512 | 2048 | 4096 | 8192)
#define FA (16 | 512 | 2048 | 8192)
#define FLD (16 | 512 | 2048 | 4096 | 8192 | 16384)
#define FD (16 | 512 | 2048 | 16384)
#define X(N, T, FL, ...) template __global__ void gen_ct_kernel<float, false, 0, CtSched<N, T, FL, __VA_ARGS__>>(GenArgs);
TRY_SIZES(X)
}
