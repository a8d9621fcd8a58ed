// Micro-benchmark: write-only HBM bandwidth of a persistent grid by store width and chunking -- the overlap-save kernel stores 4 bytes
// per lane (256 B per wave instruction), 14 KiB contiguous per workgroup unit.  Does the width matter?
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/store_width.so tools/ubench/store_width.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int W>   // dwords per lane per store
__global__ __launch_bounds__(128) void k(float* y, long long n, int unit) {   // unit: floats per workgroup chunk
    const long long nunits = n / unit;
    typedef float vt __attribute__((ext_vector_type(W)));
    for (long long u = blockIdx.x; u < nunits; u += gridDim.x) {
        float* p = y + u * unit;
        for (int i = threadIdx.x * W; i < unit; i += 128 * W) {
            vt v;
            for (int j = 0; j < W; ++j) v[j] = (float)(i + j);
            *reinterpret_cast<vt*>(p + i) = v;
        }
    }
}
template <int W> void run(float* y, long long n, int unit, int wgs) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<W>, dim3(256 * wgs), dim3(128), 0, 0, y, n, unit);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<W>, dim3(256 * wgs), dim3(128), 0, 0, y, n, unit);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("store %2d B/lane, unit %6d floats, %d WG/CU : %7.3f ms  %7.1f GB/s\n", 4 * W, unit, wgs, ms, n * 4.0 / ms / 1e6);
}
int main() {
    const long long n = 1ll << 30;
    float* y; hipMalloc(&y, n * 4);
    for (int wgs : {3, 6, 12}) {
        run<1>(y, n, 3584, wgs);
        run<2>(y, n, 3584, wgs);
        run<4>(y, n, 3584, wgs);
        run<4>(y, n, 16384, wgs);
    }
    return 0;
}
