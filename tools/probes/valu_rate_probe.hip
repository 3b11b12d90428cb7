// Issue cost of the vector instructions k_knn_pairs is made of, on gfx950: cycles per wave-instruction at 1 and 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rate_probe.hip -o /tmp/valu_rate_probe && /tmp/valu_rate_probe
// Each kernel runs ITERS x 32 independent-enough instructions of one kind (8 accumulator chains) and reports
// (clock64 end - start) / instructions for wave 0 of workgroup 0; the 4-waves-per-SIMD figure divides by 4 to give SIMD cycles per instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define ITERS 4000
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, float seed) {
    v2f a[8]; double d[8]; float f[8];
    for (int i = 0; i < 8; ++i) { a[i] = (v2f){seed + i + threadIdx.x, seed * 2 + i}; d[i] = (double)(seed + i) + threadIdx.x; f[i] = seed + i + threadIdx.x; }
    const v2f m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    const double dm = 3.0 + seed;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (KIND == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(m.x), "v"(c.x));
                REP8(X)
#undef X
            } else if (KIND == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                REP8(X)
#undef X
            } else if (KIND == 2) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                REP8(X)
#undef X
            } else if (KIND == 3) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                REP8(X)
#undef X
            } else if (KIND == 4) {
#define X(i) asm volatile("v_min_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dm));
                REP8(X)
#undef X
            } else if (KIND == 5) {
#define X(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dm));
                REP8(X)
#undef X
            } else if (KIND == 6) {
#define X(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(f[i]) : "v"(m.x));
                REP8(X)
#undef X
            } else if (KIND == 7) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(m.x), "v"(c.x));
                REP8(X)
#undef X
            } else if (KIND == 8) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dm));
                REP8(X)
#undef X
            } else if (KIND == 9) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(f[i]) : "v"(m.x));
                REP8(X)
#undef X
            } else if (KIND == 10) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
                REP8(X)
#undef X
            } else if (KIND == 11) {
#define X(i) asm volatile("v_cmp_le_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(m.x) : "vcc");
                REP8(X)
#undef X
            }
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y + (float)d[i] + f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND> void run(const char* name, float* out, long long* cyc, int per_issue) {
    for (int threads : {256, 1024}) {               // 1 and 4 waves per SIMD (one workgroup per CU: 304 >= CUs workgroups would queue; 256 = one each)
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);      // warm
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0f);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
        const double n = (double)ITERS * 32 * per_issue;
        // wall clock: n instructions per wave, threads / 256 waves per SIMD -> ns of SIMD time per instruction
        printf("%-28s %d waves/SIMD: %6.2f clock64 ticks per instruction per wave (%5.2f per SIMD); wall %8.1f us = %5.2f ns of SIMD time per instruction\n",
               name, threads / 256, c / n, c / n / (threads / 256), ms * 1e3, ms * 1e6 / (n * (threads / 256)));
    }
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    run<0>("v_fma_f32", out, cyc, 1);
    run<1>("v_pk_fma_f32", out, cyc, 1);
    run<2>("v_pk_add_f32", out, cyc, 1);
    run<3>("v_pk_mul_f32", out, cyc, 1);
    run<4>("v_min_f64", out, cyc, 1);
    run<5>("v_max_f64", out, cyc, 1);
    run<6>("v_min_f32", out, cyc, 1);
    run<7>("v_max3_f32", out, cyc, 1);
    run<8>("v_add_f64", out, cyc, 1);
    run<9>("v_mul_lo_u32", out, cyc, 1);
    run<10>("v_exp_f32", out, cyc, 1);
    run<11>("v_cmp_le_f32 + v_cndmask", out, cyc, 2);
    return 0;
}
