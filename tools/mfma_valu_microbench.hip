// microbenchmark: does VALU work issue in the shadow of fp32 / fp16 MFMAs on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
template <int MODE, int NV, int VK = 0>   // MODE 0: f32 16x16x4, 1: f16 16x16x16, 2: no MFMA; NV = independent VALU ops per MFMA slot; VK 0: v_fma_f32, 1: v_exp_f32 (transcendental)
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    f4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f4){0, 0, 0, 0};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
    float a = seed + threadIdx.x, b = seed * 2 + threadIdx.x;
    h4 ah = {(_Float16)a, (_Float16)b, (_Float16)a, (_Float16)b}, bh = ah;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u & 3], 0, 0, 0);
            else if (MODE == 1) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, acc[u & 3], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) v[(u + j) & 7] = VK ? __builtin_amdgcn_exp2f(v[(u + j) & 7] * 0.5f) : __builtin_fmaf(v[(u + j) & 7], 1.0001f, 0.5f);
            if (MODE != 2) __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            if (MODE != 2) __builtin_amdgcn_sched_group_barrier(0x2, VK ? 2 * NV : NV, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE, int NV, int VK = 0> float run(float* d, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, NV, VK><<<blocks, 256>>>(d, 100, 1.0f);
    hipEventRecord(e0);
    k<MODE, NV, VK><<<blocks, 256>>>(d, 20000, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* d; hipMalloc(&d, 256 * 1024 * 256 * 4);
    for (int wpb = 1; wpb <= 3; wpb += 2) {       // blocks per CU (4 waves each = 1 wave/SIMD per block)
        int blocks = 256 * wpb;
        printf("waves/SIMD %d\n", wpb);
        printf(" f32 mfma only        %.3f ms\n", run<0, 0>(d, blocks));
        printf(" f32 mfma + 2 valu    %.3f ms\n", run<0, 2>(d, blocks));
        printf(" f32 mfma + 4 valu    %.3f ms\n", run<0, 4>(d, blocks));
        printf(" f32 mfma + 8 valu    %.3f ms\n", run<0, 8>(d, blocks));
        printf(" f32 mfma + 2 (mul, exp2) %.3f ms\n", run<0, 2, 1>(d, blocks));
        printf(" f32 mfma + 4 (mul, exp2) %.3f ms\n", run<0, 4, 1>(d, blocks));
        printf(" no mfma: 2 fma / slot    %.3f ms\n", run<2, 2, 0>(d, blocks));
        printf(" no mfma: 2 (mul, exp2)   %.3f ms\n", run<2, 2, 1>(d, blocks));
        printf(" no mfma: 4 (mul, exp2)   %.3f ms\n", run<2, 4, 1>(d, blocks));
        printf(" f16 mfma only        %.3f ms\n", run<1, 0>(d, blocks));
        printf(" f16 mfma + 1 valu    %.3f ms\n", run<1, 1>(d, blocks));
        printf(" f16 mfma + 2 valu    %.3f ms\n", run<1, 2>(d, blocks));
        printf(" f16 mfma + 4 valu    %.3f ms\n", run<1, 4>(d, blocks));
    }
    // valu only reference: 8 fma per "slot"
    return 0;
}
