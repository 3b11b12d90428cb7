// K7 + K8: max-occupancy part merge and alpha compositing along the ray.
// Replaces TPoseHuman.forward's merge (inb_part_network_multiassign.py:229-256, cfg.aggr == ""),
// the scatter into full-size raw/occ (:156-159) and volume_rendering / render_weights
// (lib/utils/net_utils.py:12-44 as called at inb_renderer.py:72, i.e. epsilon = 0, no background).
//
// One wave per ray, lane = sample (64 samples per pass): transmittance is an exclusive product
// scan done with 6 wave shuffles, rgb/acc are wave reductions.  The merge is done on the fly from
// the per-(slot,part) results, so the (N,4) raw tensor is only written when the caller wants it.
#include "pipeline.h"

#define CMP_BLOCK 256
#ifndef COMP_NT
#define COMP_NT 1
#endif
typedef float cmp_v4 __attribute__((ext_vector_type(4)));

// Wave-wide inclusive product scan and sum on the DPP network (row_shr 1 / 2 / 4 / 8 inside the 16-lane rows, then row_bcast:15 and
// row_bcast:31 across rows — the sequence LLVM's own wave scans use on gfx9): six full-rate VALU instructions per scan.  The
// __shfl_up form compiles to ds_bpermute_b32 — an LDS round trip per step, ~20 dependent ones per 128-sample ray — and the kernel
// was bound by exactly that latency at its occupancy.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float identity, float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(x), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_incl_prod(float x, int) {
    x *= dpp_f<0x111, 0xF>(1.0f, x);          // row_shr:1
    x *= dpp_f<0x112, 0xF>(1.0f, x);          // row_shr:2
    x *= dpp_f<0x114, 0xF>(1.0f, x);          // row_shr:4
    x *= dpp_f<0x118, 0xF>(1.0f, x);          // row_shr:8
    x *= dpp_f<0x142, 0xA>(1.0f, x);          // row_bcast:15 -> rows 1 and 3
    x *= dpp_f<0x143, 0xC>(1.0f, x);          // row_bcast:31 -> rows 2 and 3
    return x;
}
__device__ __forceinline__ float wave_sum(float x) {           // total in every lane that reads lane 63's value: returned broadcast
    x += dpp_f<0x111, 0xF>(0.0f, x);
    x += dpp_f<0x112, 0xF>(0.0f, x);
    x += dpp_f<0x114, 0xF>(0.0f, x);
    x += dpp_f<0x118, 0xF>(0.0f, x);
    x += dpp_f<0x142, 0xA>(0.0f, x);
    x += dpp_f<0x143, 0xC>(0.0f, x);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

struct DenseRaw {      // raw (R,S,4) given
    const float4* raw;
    __device__ __forceinline__ float4 get(int64_t i) const { return raw[i]; }
};
struct MergedRaw {     // merge per-part results of the survivor slot of sample i
    const unsigned long long* mask;      // survivor bit of sample i: bit i&63 of word i>>6
    const int32_t* word_off;             // rank of the first survivor of every word (ray-major order)
    const int32_t* byte_off;             // windowed order (Workspace::ord_rows > 0): rank of the first survivor of every mask byte; else NULL
    const uint8_t* wsel;                 // the merge's choice per survivor (k_winner_lists)
    const float4* rgbw;                  // [rgb, occ] of the winning listed pair at the survivor's slot; far constants at [lcap + p]
    int64_t const_slot;
    __device__ __forceinline__ float4 get(int64_t i) const {
        const unsigned long long m = mask[i >> 6];
        const int bit = (int)(i & 63);
        int slot = -1;
        if ((m >> bit) & 1ull) {
            if (byte_off) slot = byte_off[i >> 3] + __popc((unsigned)(m >> (bit & ~7)) & ((1u << (bit & 7)) - 1u));
            else slot = word_off[i >> 6] + __popcll(m & ((1ull << bit) - 1ull));
        }
        if (slot >= const_slot) slot = -1;                   // survivor beyond max_active (reported in stats[6])
        float4 best = make_float4(0.f, 0.f, 0.f, 0.f);
        if (slot >= 0) {
            // argmax over the 5 parts with zeros for unflagged parts, first maximum wins (:253-255): decided by k_winner_lists
            const unsigned sel = wsel[slot];
            if (sel < (unsigned)INVR_NUM_PARTS) best = rgbw[slot];
            else if (sel < 16u) best = rgbw[const_slot + 1 + (sel - 8u)];      // far pair: per-part constant
        }
        return best;
    }
};

template <class Src>
__global__ __launch_bounds__(CMP_BLOCK) void k_composite(Src src, int64_t R, int S, float eps, float* __restrict__ weights,
                                                         float* __restrict__ rgb_map, float* __restrict__ acc_map,
                                                         float4* __restrict__ raw_out, float* __restrict__ occ_out) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * (CMP_BLOCK / 64) + (threadIdx.x >> 6);
    if (ray >= R) return;
    float T_run = 1.0f, ar = 0.f, ag = 0.f, ab = 0.f, aw = 0.f;
    for (int s0 = 0; s0 < S; s0 += 64) {
        const int s = s0 + lane;
        const bool live = s < S;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) {
            v = src.get(ray * S + s);
#if COMP_NT
            // raw is the frame's largest output (16 B per ray-sample, 93 % zeros) and nothing on the device reads it again: stored
            // nontemporal so that its 0.5 GB per frame do not displace the row-sum tables and pair arrays of the frames in flight
            // from the L2 / Infinity Cache
            if (raw_out) __builtin_nontemporal_store((cmp_v4){v.x, v.y, v.z, v.w}, reinterpret_cast<cmp_v4*>(raw_out) + (ray * S + s));
#else
            if (raw_out) raw_out[ray * S + s] = v;
#endif
            if (occ_out) occ_out[ray * S + s] = v.w;
        }
        const float alpha = v.w;
        const float incl = wave_incl_prod(live ? 1.0f - alpha + eps : 1.0f, lane);      // cumprod(1 - alpha + epsilon); eps = 0 (1 with cfg.random_bg)
        const float excl = dpp_f<0x138, 0xF>(1.0f, incl);                // wave_shr:1 (lane 0 keeps the identity)
        const float wgt = alpha * (T_run * excl);                        // render_weights (:12-15)
        if (live && weights) weights[ray * S + s] = wgt;
        ar = fmaf(wgt, v.x, ar); ag = fmaf(wgt, v.y, ag); ab = fmaf(wgt, v.z, ab); aw += wgt;
        T_run *= __int_as_float(__builtin_amdgcn_readlane(__float_as_int(incl), 63));
    }
    ar = wave_sum(ar); ag = wave_sum(ag); ab = wave_sum(ab); aw = wave_sum(aw);
    if (lane == 0) {
        rgb_map[ray * 3] = ar; rgb_map[ray * 3 + 1] = ag; rgb_map[ray * 3 + 2] = ab;
        acc_map[ray] = aw;
    }
}

// (one workgroup per four rays: a fixed grid of 4096 workgroups walking the rays measured the same 135 us — the kernel is bound by
// its 16 B / sample of output, not by dispatch)
static unsigned composite_grid(int64_t n_rays) { return (unsigned)cdiv(n_rays, CMP_BLOCK / 64); }

int launch_composite(const float* raw, int64_t n_rays, int S, float eps, float* weights, float* rgb_map, float* acc_map, hipStream_t st) {
    if (n_rays == 0) return 0;
    DenseRaw src{reinterpret_cast<const float4*>(raw)};
    hipLaunchKernelGGL(k_composite<DenseRaw>, dim3(composite_grid(n_rays)), dim3(CMP_BLOCK), 0, st,
                       src, n_rays, S, eps, weights, rgb_map, acc_map, (float4*)nullptr, (float*)nullptr);
    INVR_LAUNCH_CHECK();
    return 0;
}

int launch_merge_composite(const RenderArgs& a, const Workspace& w, float* rgb_map, float* acc_map, float* raw,
                           float* occ, float* weights, hipStream_t st) {
    if (a.R == 0) return 0;
    MergedRaw src{w.mask, w.word_off, w.ord_rows > 0 ? w.byte_off : nullptr, w.wsel, w.rgbw, w.cap};
    hipLaunchKernelGGL(k_composite<MergedRaw>, dim3(composite_grid(a.R)), dim3(CMP_BLOCK), 0, st,
                       src, a.R, a.S, a.scene.comp_eps, weights, rgb_map, acc_map, reinterpret_cast<float4*>(raw), occ);
    INVR_LAUNCH_CHECK();
    return 0;
}

// ---- distortion regulariser (inb_renderer.py:96-103), O(S) per ray with running prefix sums -------
// sum_ij w_i w_j |m_i - m_j| = 2 * sum_i w_i (m_i * W_<i - M_<i) for ascending m (z is ascending);
// evaluated in the reference's O(S^2) form when S <= 64 is not needed: both agree to fp32 rounding.
__global__ __launch_bounds__(CMP_BLOCK) void k_distortion(const float* __restrict__ weights, const float* __restrict__ z,
                                                          int64_t R, int S, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * (CMP_BLOCK / 64) + (threadIdx.x >> 6);
    if (ray >= R) return;
    const float* wr = weights + ray * S;
    const float* zr = z + ray * S;
    // direct double loop split over lanes (S is 64..128 in practice): lane i accumulates row i
    float acc = 0.0f;
    for (int i = lane; i < S; i += 64) {
        const float mi = (zr[i] + zr[min(i + 1, S - 1)]) / 2.0f;
        const float wi = wr[i];
        float row = 0.0f;
        for (int j = 0; j < S; ++j) {
            const float mj = (zr[j] + zr[min(j + 1, S - 1)]) / 2.0f;
            row += (wi * wr[j]) * fabsf(mi - mj);
        }
        acc += row;
    }
    acc = wave_sum(acc);
    if (lane == 0) out[ray] = acc;
}

int launch_distortion(const float* weights, const float* z, int64_t n_rays, int S, float* out, hipStream_t st) {
    if (n_rays == 0) return 0;
    hipLaunchKernelGGL(k_distortion, dim3((unsigned)cdiv(n_rays, CMP_BLOCK / 64)), dim3(CMP_BLOCK), 0, st, weights, z, n_rays, S, out);
    INVR_LAUNCH_CHECK();
    return 0;
}

// ---- backward of the compositing (net_utils.py:12-44 under autograd) ------------------------------
// (written for epsilon = 0; with the epsilon of render_weights every factor 1-a_j below is 1-a_j+eps.)
// w_k = a_k T_k, T_k = prod_{j<k}(1-a_j).  With G_k = dL/dw_k (from rgb_map, acc_map and any direct
// weight gradient):
//     dL/da_i = T_i (G_i - Q_i),   Q_i = sum_{k>i} G_k a_k prod_{i<j<k}(1-a_j)
// (the product EXCLUDES factor i, so nothing is divided by 1-a_i: alpha == 1 — reachable, 1-exp(-softplus(h))
// rounds to 1 for h >~ 17 — gives the finite gradient torch's zero-aware cumprod backward gives, and alpha close to 1
// has no cancellation).  Q obeys the backward recurrence Q_i = G_{i+1} a_{i+1} + (1-a_{i+1}) Q_{i+1}, Q_{S-1} = 0:
// a composition of affine maps f_k(x) = b_k + m_k x, which is associative -> one wave suffix scan over (m, b) pairs per
// 64-sample pass, passes walked back to front with a carried Q.  dL/drgb_i = w_i * g_rgb.
struct Affine { float m, b; };
__device__ __forceinline__ Affine wave_suffix_compose(Affine f, int lane) {     // F_l = f_l o f_{l+1} o ... o f_63
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float m2 = __shfl_down(f.m, d), b2 = __shfl_down(f.b, d);
        if (lane + d < 64) { f.b = fmaf(f.m, b2, f.b); f.m *= m2; }
    }
    return f;
}

__global__ __launch_bounds__(CMP_BLOCK) void k_composite_bwd(const float4* __restrict__ raw, const float* __restrict__ g_rgb,
                                                             const float* __restrict__ g_acc, const float* __restrict__ g_w,
                                                             int64_t R, int S, float eps, float4* __restrict__ g_raw) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * (CMP_BLOCK / 64) + (threadIdx.x >> 6);
    if (ray >= R) return;
    const float gr = g_rgb[ray * 3], gg = g_rgb[ray * 3 + 1], gb = g_rgb[ray * 3 + 2];
    const float ga = g_acc ? g_acc[ray] : 0.0f;
    const int npass = (S + 63) / 64;
    float carry = 0.0f;     // Q of the last sample of the current pass
    for (int ps = npass - 1; ps >= 0; --ps) {
        // transmittance at the start of this pass: product over the earlier passes (recomputed per pass — S is a
        // few passes at most, and no per-thread array bounds the sample count)
        float T0 = 1.0f;
        for (int q = 0; q < ps; ++q) {
            const float aq = raw[ray * S + q * 64 + lane].w;
            T0 *= __shfl(wave_incl_prod(1.0f - aq + eps, lane), 63);
        }
        const int s = ps * 64 + lane;
        const bool live = s < S;
        float4 v = live ? raw[ray * S + s] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float a = v.w;
        const float incl = wave_incl_prod(live ? 1.0f - a + eps : 1.0f, lane);
        float excl = __shfl_up(incl, 1);
        if (lane == 0) excl = 1.0f;
        const float T = T0 * excl;
        const float wgt = a * T;
        const float G = live ? (gr * v.x + gg * v.y + gb * v.z + ga + (g_w ? g_w[ray * S + s] : 0.0f)) : 0.0f;
        const Affine F = wave_suffix_compose(Affine{live ? 1.0f - a + eps : 1.0f, G * a}, lane);       // dead lanes: identity map
        float Mn = __shfl_down(F.m, 1), Bn = __shfl_down(F.b, 1);
        if (lane == 63) { Mn = 1.0f; Bn = 0.0f; }
        const float Q = fmaf(Mn, carry, Bn);
        if (live) {
            float4 o;
            o.x = wgt * gr; o.y = wgt * gg; o.z = wgt * gb;
            o.w = T * (G - Q);
            g_raw[ray * S + s] = o;
        }
        carry = fmaf(__shfl(F.m, 0), carry, __shfl(F.b, 0));
    }
}

int launch_composite_bwd(const float* raw, const float* g_rgb, const float* g_acc, const float* g_w, int64_t n_rays, int S, float eps,
                         float* g_raw, hipStream_t st) {
    if (n_rays == 0) return 0;
    hipLaunchKernelGGL(k_composite_bwd, dim3((unsigned)cdiv(n_rays, CMP_BLOCK / 64)), dim3(CMP_BLOCK), 0, st,
                       reinterpret_cast<const float4*>(raw), g_rgb, g_acc, g_w, n_rays, S, eps, reinterpret_cast<float4*>(g_raw));
    INVR_LAUNCH_CHECK();
    return 0;
}
