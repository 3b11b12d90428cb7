// Probe of the round-6 replay investigation (profiles/r6_replay_mismatch.md): packed-fp32 VALU instructions checked against their scalar
// forms while the frame pipeline (MFMA kernels of other streams) runs beside them.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/pk_probe.hip -o variants/libpkprobe.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float v2f __attribute__((ext_vector_type(2)));

// counts[kind * 4 + quarter]: mismatching lanes per instruction form and 16-lane quarter of the wave; counts[32 + kind] = checks per kind
__global__ __launch_bounds__(64) void k_pk_probe(unsigned long long* counts, int iters, float seed) {
    const int lane = threadIdx.x & 63, q = lane >> 4;
    float a0 = seed + 0.001f * lane, a1 = 1.5f - 0.002f * lane, b0 = 0.75f + 0.0003f * lane, b1 = -0.25f + 0.0007f * lane, c0 = 0.1f * lane, c1 = 3.0f - 0.01f * lane;
    unsigned long long bad[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        // fresh operands every iteration (cheap LCG-ish update keeps them finite)
        a0 = a0 * 0.999f + 0.0011f; a1 = a1 * 1.001f - 0.0013f; b0 = b0 * 0.998f + 0.0007f; b1 = b1 * 1.002f + 0.0003f; c0 = c0 * 0.5f + 0.25f; c1 = c1 * 0.5f - 0.125f;
        v2f A = {a0, a1}, B = {b0, b1}, Cc = {c0, c1}, r;
        float e0, e1;
        // kind 0: v_pk_mul_f32 plain
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(A), "v"(B));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e0) : "v"(a0), "v"(b0)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e1) : "v"(a1), "v"(b1));
        bad[0] += (__float_as_uint(r.x) != __float_as_uint(e0)) | (__float_as_uint(r.y) != __float_as_uint(e1));
        // kind 1: v_pk_add_f32 plain
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(A), "v"(B));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(e0) : "v"(a0), "v"(b0)); asm volatile("v_add_f32 %0, %1, %2" : "=v"(e1) : "v"(a1), "v"(b1));
        bad[1] += (__float_as_uint(r.x) != __float_as_uint(e0)) | (__float_as_uint(r.y) != __float_as_uint(e1));
        // kind 2: v_pk_fma_f32 plain
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(A), "v"(B), "v"(Cc));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e0) : "v"(a0), "v"(b0), "v"(c0)); asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e1) : "v"(a1), "v"(b1), "v"(c1));
        bad[2] += (__float_as_uint(r.x) != __float_as_uint(e0)) | (__float_as_uint(r.y) != __float_as_uint(e1));
        // kind 3: v_pk_mul_f32 op_sel_hi:[0,1]  (both halves of the result take the LOW half of src0: {a0*b0, a0*b1})
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(A), "v"(B));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e0) : "v"(a0), "v"(b0)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e1) : "v"(a0), "v"(b1));
        bad[3] += (__float_as_uint(r.x) != __float_as_uint(e0)) | (__float_as_uint(r.y) != __float_as_uint(e1));
        // kind 4: v_pk_fma_f32 op_sel:[1,0,0] (low result uses the HIGH half of src0; op_sel_hi default 1: {a1*b0+c0, a1*b1+c1})
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(r) : "v"(A), "v"(B), "v"(Cc));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e0) : "v"(a1), "v"(b0), "v"(c0)); asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e1) : "v"(a1), "v"(b1), "v"(c1));
        bad[4] += (__float_as_uint(r.x) != __float_as_uint(e0)) | (__float_as_uint(r.y) != __float_as_uint(e1));
        // kind 5: a DEPENDENT chain of packed ops back to back (mul -> fma -> add), as the compiler schedules them
        {
            v2f t;
            asm volatile("v_pk_mul_f32 %0, %1, %2\n\tv_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_add_f32 %0, %0, %1" : "=&v"(t) : "v"(A), "v"(B), "v"(Cc));
            float s0, s1;
            asm volatile("v_mul_f32 %0, %1, %2\n\tv_fma_f32 %0, %0, %2, %3\n\tv_add_f32 %0, %0, %1" : "=&v"(s0) : "v"(a0), "v"(b0), "v"(c0));
            asm volatile("v_mul_f32 %0, %1, %2\n\tv_fma_f32 %0, %0, %2, %3\n\tv_add_f32 %0, %0, %1" : "=&v"(s1) : "v"(a1), "v"(b1), "v"(c1));
            bad[5] += (__float_as_uint(t.x) != __float_as_uint(s0)) | (__float_as_uint(t.y) != __float_as_uint(s1));
        }
        // kind 6: a packed op consuming the result of a scalar-form VALU op issued right before it (and vice versa)
        {
            float m; v2f t; float u;
            asm volatile("v_mul_f32 %0, %2, %3\n\tv_pk_mul_f32 %1, %4, %5" : "=&v"(m), "=&v"(t) : "v"(a0), "v"(b0), "v"(A), "v"(B));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(u) : "v"(t.x), "v"(m));
            float e = a0 * b0, f;
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(f) : "v"(e), "v"(e));
            bad[6] += (__float_as_uint(u) != __float_as_uint(f));
        }
        // kind 7: v_pk_mov_b32 with op_sel (the compiler's register shuffles beside packed math)
        {
            v2f t;
            asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(t) : "v"(A), "v"(B));
            bad[7] += (__float_as_uint(t.x) != __float_as_uint(a1)) | (__float_as_uint(t.y) != __float_as_uint(b0));
        }
    }
    for (int k = 0; k < 8; ++k)
        if (bad[k]) atomicAdd(&counts[k * 4 + q], bad[k]);
    if (lane == 0) for (int k = 0; k < 8; ++k) atomicAdd(&counts[32 + k], (unsigned long long)iters * 64ull);
}

extern "C" int pk_probe_launch(void* stream, int blocks, int iters, unsigned long long* counts, float seed) {
    hipLaunchKernelGGL(k_pk_probe, dim3(blocks), dim3(64), 0, (hipStream_t)stream, counts, iters, seed);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
