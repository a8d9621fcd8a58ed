// Micro-benchmark: issue rate of the Float32 vector FMA forms the kernels use (cycles per wave instruction, one SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_rate.bin tools/ubench/valu_rate.hip && tools/ubench/valu_rate.bin
// Every variant runs NI independent accumulator chains of length LEN in one wave per SIMD (and again with 2, 4 waves per SIMD);
// clock64() around the loop, minimum over waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int LEN = 512;

template <int MODE, int NI> __global__ void k(float* out, long long* cyc, float s0, float s1) {
    f2 acc[NI];
    for (int i = 0; i < NI; ++i) acc[i] = f2{(float)threadIdx.x * 1e-3f + i, (float)i};
    f2 x = {s0 + threadIdx.x * 1e-6f, s1};
    float sx = __builtin_amdgcn_readfirstlane(__float_as_int(s0)) ? s0 : s1;   // uniform
    f2 sv = {sx, sx * 0.5f};
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < LEN; ++it) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (MODE == 0) {          // 2 x v_fma_f32
                asm volatile("v_fma_f32 %0, %2, %3, %0\n\tv_fma_f32 %1, %2, %4, %1" : "+v"(acc[i].x), "+v"(acc[i].y) : "v"(x.x), "v"(x.y), "v"(x.x));
            } else if (MODE == 1) {   // v_pk_fma_f32, VGPR operands
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(x));
            } else if (MODE == 2) {   // v_pk_fma_f32, sample broadcast by op_sel
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(x), "v"(x));
            } else if (MODE == 3) {   // v_pk_fma_f32 with an SGPR-pair operand and broadcast
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(x), "s"(sv));
            } else if (MODE == 4) {   // v_pk_add_f32
                asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(x));
            } else if (MODE == 5) {   // 2 x v_add_f32
                asm volatile("v_add_f32 %0, %2, %0\n\tv_add_f32 %1, %3, %1" : "+v"(acc[i].x), "+v"(acc[i].y) : "v"(x.x), "v"(x.y));
            } else if (MODE == 6) {   // v_pk_mul_f32
                asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(x));
            } else if (MODE == 7) {   // v_fmac_f32 x2 (VOP2)
                asm volatile("v_fmac_f32 %0, %2, %3\n\tv_fmac_f32 %1, %2, %4" : "+v"(acc[i].x), "+v"(acc[i].y) : "v"(x.x), "v"(x.y), "v"(x.x));
            }
        }
    }
    const long long t1 = clock64();
    float r = 0;
    for (int i = 0; i < NI; ++i) r += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE, int NI> void run(const char* name, int per_wave_instr) {
    float* out; long long* cyc;
    hipMalloc(&out, 4 << 20); hipMalloc(&cyc, 1 << 16);
    for (int waves : {4, 8, 16}) {   // waves per workgroup = 1, 2, 4 per SIMD; one workgroup
        hipLaunchKernelGGL((k<MODE, NI>), dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0001f, 0.9999f);
        hipDeviceSynchronize();
        std::vector<long long> h(waves);
        hipMemcpy(h.data(), cyc, waves * 8, hipMemcpyDeviceToHost);
        long long mx = 0; for (auto v : h) mx = v > mx ? v : mx;
        const double per = (double)mx / ((double)LEN * NI * per_wave_instr * (waves / 4));
        printf("%-44s NI=%d waves/SIMD=%d : %6.2f cycles per wave-instruction per SIMD  (%5.1f FMA-or-op lanes/clk/SIMD)\n", name, NI, waves / 4, per,
               64.0 * (per_wave_instr == 2 ? 1 : 2) / per);
    }
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0, 8>("2 x v_fma_f32 (VOP3)", 2);
    run<7, 8>("2 x v_fmac_f32 (VOP2)", 2);
    run<1, 8>("v_pk_fma_f32 vgpr", 1);
    run<2, 8>("v_pk_fma_f32 vgpr, op_sel broadcast", 1);
    run<3, 8>("v_pk_fma_f32 sgpr pair, op_sel broadcast", 1);
    run<4, 8>("v_pk_add_f32", 1);
    run<5, 8>("2 x v_add_f32", 2);
    run<6, 8>("v_pk_mul_f32", 1);
    run<1, 2>("v_pk_fma_f32 vgpr (2 chains: latency)", 1);
    run<0, 2>("2 x v_fma_f32 (2 chains: latency)", 2);
    return 0;
}
