// LDS cost of the reads k_knn_pairs is made of, on gfx950: wave-uniform (broadcast) ds_read_b128 / b64 / b32 against per-lane-distinct
// ds_read_b128, 4 / 8 / 16 waves per CU (one workgroup per CU), cycles of CU time per wave-instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_bcast_probe.hip -o variants/lds_bcast_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define ITERS 2000
template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int stride) {
    extern __shared__ float4 lds[];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = make_float4(i, i + 1, i + 2, i + 3);
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // KIND 0: wave-uniform address; 1: per-lane distinct (conflict-free); 2: uniform b64; 3: uniform b32
    float4 acc = make_float4(0, 0, 0, 0);
    int base = (wv * 64) & 4095;
    const long long t0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = (base + u * stride + (KIND == 1 ? lane : 0)) & 4095;
            if (KIND <= 1) { const float4 v = lds[idx]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
            else if (KIND == 2) { const float2 v = reinterpret_cast<const float2*>(lds)[idx * 2]; acc.x += v.x; acc.y += v.y; }
            else { const float v = reinterpret_cast<const float*>(lds)[idx * 4]; acc.x += v; }
        }
        base = (base + 8 * stride) & 4095;
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND> void run(const char* name, float* out, long long* cyc) {
    for (int threads : {256, 512, 1024}) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 65536, 0, out, cyc, 2);      // warm
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 65536, 0, out, cyc, 2);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        const double waves = threads / 64.0, reads = ITERS * 8.0;
        // wall clock (clock64 of one wave is biased: the oldest wave keeps its slots): ns of CU time per wave-instruction
        printf("%-34s %2d waves/CU: wall %8.1f us = %6.2f ns of CU time per wave-instruction (%5.1f ns per read per wave)\n", name, threads / 64,
               ms * 1e3, ms * 1e6 / (reads * waves), ms * 1e6 / reads);
    }
}
int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&cyc, 8);
    run<0>("ds_read_b128 wave-uniform", out, cyc);
    run<1>("ds_read_b128 per-lane distinct", out, cyc);
    run<2>("ds_read_b64 wave-uniform", out, cyc);
    run<3>("ds_read_b32 wave-uniform", out, cyc);
    return 0;
}
