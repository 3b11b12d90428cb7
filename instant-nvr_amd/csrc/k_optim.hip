// Row f1: the optimiser step of the training loop — torch.optim.Adam over every trainable tensor
// (lib/train/optimizer.py:13-31: one parameter group per tensor, eps = cfg.train.eps = 1e-15).  torch's foreach
// implementation cannot batch across parameter groups: ~8 element-wise launches per tensor x 186 tensors = 1.5 k
// launches and 7 ms per step for the 286 M parameters of inb_377.  This is ONE launch: a flat list of 64 k-element
// chunks over all tensors (chunk -> tensor table built once by the host), single pass over param / grad / exp_avg /
// exp_avg_sq = 28 B per parameter, the HBM floor of a dense Adam step.
// Arithmetic of torch/optim/adam.py (_single_tensor_adam, amsgrad=False, maximize=False):
//   g += wd * p ; m = lerp(m, g, 1-b1) ; v = b2*v + (1-b2) g*g ; p -= (lr / bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
#include "common.h"

static_assert(sizeof(float*) == 8, "64-bit");
#define ADAM_BLOCK 256
#define ADAM_CHUNK 16384            // elements per workgroup
#ifndef ADAM_NT
#define ADAM_NT 1
#endif
#ifndef ADAM_UNROLL
#define ADAM_UNROLL 4
#endif

struct AdamTensor {                 // device-resident table, one entry per tensor that has a gradient this step (InvrAdamTensor)
    float* p; const float* g; float* m; float* v;
    int64_t n;
    float lr, wd, bc1, bc2_sqrt;    // bias corrections 1-b1^step and sqrt(1-b2^step) of THIS tensor's step count
    const float* active;            // optional device flag: 0 -> the tensor is skipped this step (torch skips tensors without gradient)
    int32_t grad_shift;             // > 0: g is a ROW-SCALAR gradient — element i takes g[i >> grad_shift] (the gradient of a
                                    // sum-over-features table is one scalar per row of 2^shift features, k_encode.hip)
    int32_t step;                   // device-side step count (invr_adam_advance): host tables need no per-step upload
};

// One thread per tensor: step += 1 and the bias corrections of the new step, in double like the host does
// (1 - beta**step, sqrt(1 - beta**step)).  Lets a training loop replay the optimiser step without a host-built table.
__global__ void k_adam_advance(AdamTensor* tensors, int n, double b1, double b2) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (tensors[t].active && tensors[t].active[0] == 0.0f) return;
    const int s = tensors[t].step + 1;
    tensors[t].step = s;
    tensors[t].bc1 = (float)(1.0 - pow(b1, (double)s));
    tensors[t].bc2_sqrt = (float)sqrt(1.0 - pow(b2, (double)s));
}

__global__ __launch_bounds__(ADAM_BLOCK) void k_adam(const AdamTensor* __restrict__ tensors, const int32_t* __restrict__ chunk_tensor,
                                                     const int32_t* __restrict__ chunk_index, float b1, float b2, float w1, float w2, float eps) {
    const AdamTensor t = tensors[chunk_tensor[blockIdx.x]];
    if (t.active && t.active[0] == 0.0f) return;
    const int64_t base = (int64_t)chunk_index[blockIdx.x] * ADAM_CHUNK;
    const int64_t end = min(base + ADAM_CHUNK, t.n);
    const float step_size = t.lr / t.bc1;                         // (w1 = 1 - beta1, w2 = 1 - beta2: formed in double by the host)
    auto upd = [&](float& p, float g, float& m, float& v) {
        if (t.wd != 0.0f) g = fmaf(t.wd, p, g);
        m = m + w1 * (g - m);                                     // torch.lerp, weight < 0.5
        v = fmaf(w2 * g, g, b2 * v);                               // mul_(b2).addcmul_(g, g, value=1-b2)
        const float denom = sqrtf(v) / t.bc2_sqrt + eps;
        p = p - step_size * (m / denom);                           // addcdiv_(m, denom, value=-step_size)
    };
    // 28 B per parameter stream through this kernel once per step and nothing of it is read again before the next step has streamed
    // another 6.9 GB through the caches: nontemporal loads / stores (no L2 / Infinity-Cache allocation), and ADAM_UNROLL float4 per
    // array and thread in flight before the first use (the loop form issued one float4 per array, waited, computed, stored: too
    // few bytes in flight per CU for the HBM latency).  ADAM_NT=0 / ADAM_UNROLL=1: the round-4 form (A/B builds).
    auto ld4 = [](const float* q) -> float4 {
#if ADAM_NT
        typedef float v4 __attribute__((ext_vector_type(4)));
        const v4 r = __builtin_nontemporal_load(reinterpret_cast<const v4*>(q));
        return make_float4(r.x, r.y, r.z, r.w);
#else
        return *reinterpret_cast<const float4*>(q);
#endif
    };
    auto st4 = [](float* q, const float4& x) {
#if ADAM_NT
        typedef float v4 __attribute__((ext_vector_type(4)));
        v4 r = {x.x, x.y, x.z, x.w};
        __builtin_nontemporal_store(r, reinterpret_cast<v4*>(q));
#else
        *reinterpret_cast<float4*>(q) = x;
#endif
    };
    const int sh = t.grad_shift;
    const bool vec = ((((uintptr_t)t.p | (uintptr_t)t.m | (uintptr_t)t.v | (sh > 0 ? (uintptr_t)0 : (uintptr_t)t.g)) & 15) == 0) && (sh == 0 || sh >= 2);
    if (vec) {
        constexpr int U = ADAM_UNROLL;
        int64_t i = base + (int64_t)threadIdx.x * 4;
        for (; i + (int64_t)(U - 1) * ADAM_BLOCK * 4 + 3 < end; i += (int64_t)U * ADAM_BLOCK * 4) {
            float4 p[U], m[U], v[U], g[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t j = i + (int64_t)u * ADAM_BLOCK * 4;
                p[u] = ld4(t.p + j); m[u] = ld4(t.m + j); v[u] = ld4(t.v + j);
                if (sh > 0) { const float gs = t.g[j >> sh]; g[u] = make_float4(gs, gs, gs, gs); }      // row scalar: one load per 2^shift elements
                else g[u] = ld4(t.g + j);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t j = i + (int64_t)u * ADAM_BLOCK * 4;
                upd(p[u].x, g[u].x, m[u].x, v[u].x); upd(p[u].y, g[u].y, m[u].y, v[u].y); upd(p[u].z, g[u].z, m[u].z, v[u].z); upd(p[u].w, g[u].w, m[u].w, v[u].w);
                st4(t.p + j, p[u]); st4(t.m + j, m[u]); st4(t.v + j, v[u]);
            }
        }
        for (; i + 3 < end; i += ADAM_BLOCK * 4) {
            float4 p = ld4(t.p + i), m = ld4(t.m + i), v = ld4(t.v + i), g;
            if (sh > 0) { const float gs = t.g[i >> sh]; g = make_float4(gs, gs, gs, gs); }
            else g = ld4(t.g + i);
            upd(p.x, g.x, m.x, v.x); upd(p.y, g.y, m.y, v.y); upd(p.z, g.z, m.z, v.z); upd(p.w, g.w, m.w, v.w);
            st4(t.p + i, p); st4(t.m + i, m); st4(t.v + i, v);
        }
        for (int64_t k = base + ((end - base) & ~(int64_t)3) + threadIdx.x; k < end; k += ADAM_BLOCK) upd(t.p[k], t.g[k >> sh], t.m[k], t.v[k]);
    } else {
        for (int64_t k = base + threadIdx.x; k < end; k += ADAM_BLOCK) upd(t.p[k], t.g[k >> sh], t.m[k], t.v[k]);
    }
}

static_assert(sizeof(AdamTensor) == sizeof(InvrAdamTensor), "AdamTensor mirrors InvrAdamTensor");

int launch_adam_advance(void* tensors, int n, double b1, double b2, hipStream_t st) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_adam_advance, dim3((unsigned)cdiv(n, 64)), dim3(64), 0, st, reinterpret_cast<AdamTensor*>(tensors), n, b1, b2);
    INVR_LAUNCH_CHECK();
    return 0;
}

int launch_adam(const void* tensors, const int32_t* chunk_tensor, const int32_t* chunk_index, int64_t n_chunks, double b1, double b2,
                float eps, hipStream_t st) {
    if (n_chunks == 0) return 0;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)n_chunks), dim3(ADAM_BLOCK), 0, st, reinterpret_cast<const AdamTensor*>(tensors), chunk_tensor,
                       chunk_index, (float)b1, (float)b2, (float)(1.0 - b1), (float)(1.0 - b2), eps);      // 1 - beta in double, as torch/optim/adam.py
    INVR_LAUNCH_CHECK();
    return 0;
}
