// Micro-benchmark: can ONE wave per SIMD hide vector work under its own MFMA stream on gfx950?
// Each wave runs `iters` trips of: NM independent v_mfma_f32_32x32x16_bf16 (4 accumulators round-robin) with NV independent
// v_fma_f32 (or transcendental / packed-convert) instructions placed after each MFMA.  One block of 256 threads per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int KIND, int NACC = 4>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, float seedv) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x16{0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)(seedv + threadIdx.x), b[i] = (__bf16)(seedv * 2 + i);
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seedv + i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % NACC], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                if (KIND == 0) v[j & 7] = __builtin_fmaf(v[j & 7], 1.0001f, 0.5f);
                else if (KIND == 1) v[j & 7] = __builtin_amdgcn_exp2f(v[j & 7]);
                else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(v[j & 7]) : "v"(v[j & 7]), "v"(v[(j + 1) & 7]));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, int KIND, int NACC = 4>
void run(const char* name, float* out) {
    const int iters = 2000, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, KIND, NACC>), dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, KIND, NACC>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)iters * 8;
    printf("acc=%d %-10s NV=%d  %.3f ms  %.1f ns per MFMA (+%d vector ops)\n", NACC, name, NV, ms, ms * 1e6 / mfma, NV);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 256 * 4);
    run<0, 0>("fma", out);
    run<2, 0>("fma", out);
    run<4, 0>("fma", out);
    run<6, 0>("fma", out);
    run<8, 0>("fma", out);
    run<12, 0>("fma", out);
    run<16, 0>("fma", out);
    run<2, 1>("exp2", out);
    run<4, 1>("exp2", out);
    run<0, 0, 1>("chain1", out);
    run<0, 0, 2>("chain2", out);
    run<4, 0, 1>("chain1", out);
    run<4, 2>("cvt_pk", out);
    run<8, 2>("cvt_pk", out);
    return 0;
}
