// Device copy / read / fill rates over launch shapes and cache policies (the guide's 6.29 TB/s float4 copy against the 4.8 - 5.7 the
// library's own yardsticks reach).   hipcc --offload-arch=gfx950 -O3 tools/ubench/copy_sweep.hip -o /tmp/copy_sweep && /tmp/copy_sweep [log2 bytes = 31]
// Variants: workgroups per CU (persistent grid) 1 .. 16; 16-byte accesses in flight per thread 1 .. 16; interleaved (grid-stride) or one contiguous 2 MiB-
// aligned range per workgroup; cache-policy bits of the buffer instructions on loads and stores (aux: 1 = sc0, 2 = nt, 16 = sc1).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000); }

// MODE 0 copy, 1 read (sum kept alive), 2 fill
template <int U, int LAUX, int SAUX, bool CHUNK, int MODE>
__global__ __launch_bounds__(256) void k(void* dst, const void* src, unsigned n16, unsigned chunk16) {
    const __amdgpu_buffer_rsrc_t rs = rsrc(src, n16 * 16u), rd = rsrc(dst, n16 * 16u);
    unsigned lo, hi, step;
    if (CHUNK) { lo = blockIdx.x * chunk16; hi = min(n16, lo + chunk16); step = 256u * U; }
    else { lo = blockIdx.x * 256u * U; hi = n16; step = gridDim.x * 256u * U; }
    u4 keep = {0, 0, 0, 0};
    for (unsigned base = lo; base < hi; base += step) {
        u4 v[U];
        if (MODE != 2) {
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((base + u * 256u + threadIdx.x) * 16u), 0, LAUX);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = u4{base, (unsigned)u, threadIdx.x, 1u};
        }
        if (MODE != 1) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (base + u * 256u + threadIdx.x < hi) __builtin_amdgcn_raw_buffer_store_b128(v[u], rd, (int)((base + u * 256u + threadIdx.x) * 16u), 0, SAUX);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) keep += v[u];
        }
    }
    if (MODE == 1 && keep.x == 0x12345678u) __builtin_amdgcn_raw_buffer_store_b128(keep, rd, 0, 0, 0);
}

struct Res { double gbps; const char* what; int wgs, U, laux, saux, chunk; };
static std::vector<Res> all;

template <int U, int LAUX, int SAUX, bool CHUNK, int MODE> void run(void* d, void* s, size_t bytes, int cus, const char* what) {
    const unsigned n16 = (unsigned)(bytes / 16);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wgs : {1, 2, 4, 8, 16}) {
        const unsigned grid = (unsigned)(cus * wgs);
        unsigned chunk16 = (n16 + grid - 1) / grid;
        chunk16 = (chunk16 + 131071u) / 131072u * 131072u;   // 2 MiB
        if (CHUNK && (size_t)chunk16 * (grid - 1) >= n16) continue;   // more workgroups than chunks
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL((k<U, LAUX, SAUX, CHUNK, MODE>), dim3(grid), dim3(256), 0, 0, d, s, n16, chunk16);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) best = std::min(best, ms);
        }
        all.push_back({(MODE == 0 ? 2.0 : 1.0) * bytes / best / 1e6, what, wgs, U, LAUX, SAUX, (int)CHUNK});
    }
}
template <int LAUX, int SAUX> void policies(void* d, void* s, size_t bytes, int cus) {
    run<4, LAUX, SAUX, false, 0>(d, s, bytes, cus, "copy");
    run<8, LAUX, SAUX, false, 0>(d, s, bytes, cus, "copy");
    run<8, LAUX, SAUX, true, 0>(d, s, bytes, cus, "copy");
}
int main(int argc, char** argv) {
    const size_t bytes = (size_t)1 << (argc > 1 ? atoi(argv[1]) : 31);
    void *d, *s;
    CHECK(hipMalloc(&d, bytes)); CHECK(hipMalloc(&s, bytes));
    CHECK(hipMemset(s, 1, bytes)); CHECK(hipMemset(d, 0, bytes));
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    // shapes at the default policy
    run<1, 0, 0, false, 0>(d, s, bytes, cus, "copy"); run<2, 0, 0, false, 0>(d, s, bytes, cus, "copy"); run<4, 0, 0, false, 0>(d, s, bytes, cus, "copy");
    run<8, 0, 0, false, 0>(d, s, bytes, cus, "copy"); run<16, 0, 0, false, 0>(d, s, bytes, cus, "copy");
    run<4, 0, 0, true, 0>(d, s, bytes, cus, "copy"); run<8, 0, 0, true, 0>(d, s, bytes, cus, "copy"); run<16, 0, 0, true, 0>(d, s, bytes, cus, "copy");
    // policies: loads x stores over {default, sc0, nt, sc1, sc0|sc1, nt|sc1 ...}
    policies<0, 2>(d, s, bytes, cus); policies<2, 0>(d, s, bytes, cus); policies<2, 2>(d, s, bytes, cus); policies<0, 16>(d, s, bytes, cus); policies<0, 17>(d, s, bytes, cus);
    policies<0, 18>(d, s, bytes, cus); policies<2, 18>(d, s, bytes, cus); policies<16, 0>(d, s, bytes, cus); policies<1, 0>(d, s, bytes, cus); policies<17, 17>(d, s, bytes, cus);
    policies<0, 1>(d, s, bytes, cus); policies<2, 17>(d, s, bytes, cus); policies<18, 18>(d, s, bytes, cus); policies<3, 3>(d, s, bytes, cus); policies<19, 19>(d, s, bytes, cus);
    // one direction only
    run<8, 0, 0, false, 1>(d, s, bytes, cus, "read"); run<8, 2, 0, false, 1>(d, s, bytes, cus, "read"); run<16, 0, 0, true, 1>(d, s, bytes, cus, "read");
    run<8, 0, 0, false, 2>(d, s, bytes, cus, "fill"); run<8, 0, 2, false, 2>(d, s, bytes, cus, "fill"); run<8, 0, 18, true, 2>(d, s, bytes, cus, "fill");
    std::sort(all.begin(), all.end(), [](const Res& a, const Res& b) { return a.gbps > b.gbps; });
    printf("{\"bytes\": %zu, \"cus\": %d, \"variants\": [\n", bytes, cus);
    for (size_t i = 0; i < all.size(); ++i)
        printf(" {\"what\": \"%s\", \"GBps\": %.0f, \"wgs_per_cu\": %d, \"in_flight_16B\": %d, \"load_aux\": %d, \"store_aux\": %d, \"chunked\": %d}%s\n", all[i].what, all[i].gbps, all[i].wgs,
               all[i].U, all[i].laux, all[i].saux, all[i].chunk, i + 1 < all.size() ? "," : "");
    printf("]}\n");
    return 0;
}
