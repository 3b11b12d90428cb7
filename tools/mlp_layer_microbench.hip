// Microbenchmark for VERDICT r3 item 4: ONE 64 -> 64 hidden layer of the part colour MLP (bias, 64 x 64 weights from LDS, log2-domain
// Softplus) on a 16-pair column block per wave, exactly the register layout of k_mlp.hip:st_rgb2 —
//   F32   : 64 x v_mfma_f32_16x16x4_f32 (the product's form)
//   BF16x3: activations and weights split into three bf16 terms (hi + mid + lo = the fp32 value exactly), the six leading products
//           (hi hi, hi mid, mid hi, hi lo, lo hi, mid mid) on 48 x v_mfma_f32_16x16x32_bf16 with fp32 accumulation
// chained `iters` times (the output of a layer is the input of the next, so nothing can be hoisted), 3 workgroups of 4 waves per CU
// as k_part_rgb_all runs.  Prints ms for both and the largest difference of the outputs after ONE layer.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mlp_layer_microbench.hip -o /tmp/mlpbench && /tmp/mlpbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define LOG2E 1.4426950408889634f

__device__ __forceinline__ f32x4 softplus4_log2(f32x4 v) {      // u = log2(1 + exp2(min(z, 126))): {min, exp2, add, log2} per value
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(fminf(v[k], 126.0f)));
    return r;
}
__device__ __forceinline__ int hid_col(int s, int g) { return 16 * (s >> 2) + 4 * g + (s & 3); }

// LDS images.  F32: [k-step s][lane][m-tile] floats (one ds_read_b128 feeds four MFMAs).  BF16: [split][mo][kb][lane] x 8 bf16.
#define LDS_F32 (16 * 64 * 4)
__global__ __launch_bounds__(256, 3) void k_f32(const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ in,
                                                float* __restrict__ out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_F32 + 64];
    for (int t = threadIdx.x; t < 16 * 4 * 64; t += 256) {
        const int ln = t & 63, mt = (t >> 6) & 3, s = t >> 8, g = ln >> 4, i = ln & 15;
        lds[(s * 64 + ln) * 4 + mt] = W[(16 * mt + i) * 64 + hid_col(s, g)];
    }
    if (threadIdx.x < 64) lds[LDS_F32 + threadIdx.x] = bias[threadIdx.x] * LOG2E;
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane >> 4, col = lane & 15;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    f32x4 h[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[mt][r] = in[(tile * 16 + col) * 64 + 16 * mt + 4 * g + r];
    for (int it = 0; it < iters; ++it) {
        f32x4 h2[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) h2[mt][r] = lds[LDS_F32 + 16 * mt + 4 * g + r];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float4 a = *reinterpret_cast<const float4*>(lds + (s * 64 + lane) * 4);
            const float b = h[s >> 2][s & 3];
            h2[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b, h2[0], 0, 0, 0);
            h2[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b, h2[1], 0, 0, 0);
            h2[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b, h2[2], 0, 0, 0);
            h2[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b, h2[3], 0, 0, 0);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) h[mt] = softplus4_log2(h2[mt]);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(tile * 16 + col) * 64 + 16 * mt + 4 * g + r] = h[mt][r];
}

__device__ __forceinline__ void split3(float x, __bf16& hi, __bf16& mid, __bf16& lo) {
    hi = (__bf16)x;
    const float r1 = x - (float)hi;
    mid = (__bf16)r1;
    lo = (__bf16)(r1 - (float)mid);
}

__global__ __launch_bounds__(256, 3) void k_bf16x3(const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ in,
                                                   float* __restrict__ out, int iters) {
    __shared__ __attribute__((aligned(16))) bf16x8 wl[3 * 4 * 2 * 64];      // 24 KB
    __shared__ float bl[64];
    for (int t = threadIdx.x; t < 4 * 2 * 64; t += 256) {
        const int ln = t & 63, kb = (t >> 6) & 1, mo = t >> 7, g = ln >> 4, i = ln & 15;
        bf16x8 vh, vm, vl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float w = W[(16 * mo + i) * 64 + 16 * (2 * kb + (j >> 2)) + 4 * g + (j & 3)];
            __bf16 a, b, c;
            split3(w, a, b, c);
            vh[j] = a; vm[j] = b; vl[j] = c;
        }
        wl[(0 * 8 + mo * 2 + kb) * 64 + ln] = vh;
        wl[(1 * 8 + mo * 2 + kb) * 64 + ln] = vm;
        wl[(2 * 8 + mo * 2 + kb) * 64 + ln] = vl;
    }
    if (threadIdx.x < 64) bl[threadIdx.x] = bias[threadIdx.x] * LOG2E;
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane >> 4, col = lane & 15;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    f32x4 h[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[mt][r] = in[(tile * 16 + col) * 64 + 16 * mt + 4 * g + r];
    for (int it = 0; it < iters; ++it) {
        bf16x8 bh[2], bm[2], bo[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                __bf16 a, b, c;
                split3(h[2 * kb + (j >> 2)][j & 3], a, b, c);
                bh[kb][j] = a; bm[kb][j] = b; bo[kb][j] = c;
            }
        f32x4 h2[4];
#pragma unroll
        for (int mo = 0; mo < 4; ++mo) {
#pragma unroll
            for (int r = 0; r < 4; ++r) h2[mo][r] = bl[16 * mo + 4 * g + r];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const bf16x8 ah = wl[(0 * 8 + mo * 2 + kb) * 64 + lane], am = wl[(1 * 8 + mo * 2 + kb) * 64 + lane], al = wl[(2 * 8 + mo * 2 + kb) * 64 + lane];
                // smallest terms first
                h2[mo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm[kb], h2[mo], 0, 0, 0);
                h2[mo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bo[kb], h2[mo], 0, 0, 0);
                h2[mo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[kb], h2[mo], 0, 0, 0);
                h2[mo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm[kb], h2[mo], 0, 0, 0);
                h2[mo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh[kb], h2[mo], 0, 0, 0);
                h2[mo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[kb], h2[mo], 0, 0, 0);
            }
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) h[mt] = softplus4_log2(h2[mt]);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(tile * 16 + col) * 64 + 16 * mt + 4 * g + r] = h[mt][r];
}

int main() {
    const int blocks = 256 * 3, tiles = blocks * 4, n = tiles * 16;
    std::vector<float> W(64 * 64), b(64), x((size_t)n * 64);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.0f - 1.0f; };
    for (auto& v : W) v = rnd() * 0.125f * LOG2E * 0.6931f;      // (ln2 * log2e folded as in the product: hidden-to-hidden weights are unscaled)
    for (auto& v : b) v = rnd() * 0.1f;
    for (auto& v : x) v = fabsf(rnd()) * 2.0f;                   // Softplus outputs (log2 domain) are positive
    float *dW, *db, *dx, *o1, *o2;
    hipMalloc(&dW, W.size() * 4); hipMalloc(&db, 256); hipMalloc(&dx, x.size() * 4); hipMalloc(&o1, x.size() * 4); hipMalloc(&o2, x.size() * 4);
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    // one layer: the difference of the two forms
    k_f32<<<blocks, 256>>>(dW, db, dx, o1, 1);
    k_bf16x3<<<blocks, 256>>>(dW, db, dx, o2, 1);
    std::vector<float> r1(x.size()), r2(x.size());
    hipMemcpy(r1.data(), o1, x.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), o2, x.size() * 4, hipMemcpyDeviceToHost);
    // float64 reference of the layer for both
    double e1 = 0, e2 = 0, d12 = 0;
    for (int p = 0; p < 4096; ++p)
        for (int o = 0; o < 64; ++o) {
            double acc = (double)b[o] * (double)LOG2E;
            for (int k = 0; k < 64; ++k) acc += (double)W[o * 64 + k] * (double)x[(size_t)p * 64 + k];
            const double ref = log2(1.0 + exp2(acc));
            e1 = fmax(e1, fabs(r1[(size_t)p * 64 + o] - ref)); e2 = fmax(e2, fabs(r2[(size_t)p * 64 + o] - ref));
            d12 = fmax(d12, fabs((double)r1[(size_t)p * 64 + o] - (double)r2[(size_t)p * 64 + o]));
        }
    printf("one layer, max |out - float64|: f32 mfma %.3e   bf16x3 %.3e   |f32 - bf16x3| %.3e\n", e1, e2, d12);
    hipEvent_t a, c; hipEventCreate(&a); hipEventCreate(&c);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        float ms1, ms2;
        hipEventRecord(a); k_f32<<<blocks, 256>>>(dW, db, dx, o1, iters); hipEventRecord(c); hipEventSynchronize(c); hipEventElapsedTime(&ms1, a, c);
        hipEventRecord(a); k_bf16x3<<<blocks, 256>>>(dW, db, dx, o2, iters); hipEventRecord(c); hipEventSynchronize(c); hipEventElapsedTime(&ms2, a, c);
        const double lay = (double)tiles * iters;
        printf("%d layers x %d tiles of 16 pairs: f32 mfma %.3f ms (%.1f ns / tile-layer / CU-slot), bf16x3 %.3f ms  -> ratio %.3f\n", iters, tiles, ms1,
               ms1 * 1e6 / lay * 768, ms2, ms2 / ms1);
    }
    return 0;
}
