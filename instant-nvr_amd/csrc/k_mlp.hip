// K6: the two tiny Softplus MLPs of one body part on the fp32 matrix cores.
// Replaces part_base_network.Network.forward after the encoder (part_base_network.py:44-63):
//   h   = occMLP(emb19)            19 -> 64 -> 17          (Softplus between linears)
//   occ = 1 - exp(-softplus(h[0])) ; feat = h[1:17]
//   rgb = sigmoid(rgbMLP([emb19, dirPE27, feat16, latent8]))   70 -> 64 (-> 64) -> 3
//
// fp32 parity (1e-4 per pixel) rules out bf16/fp16 MFMA inputs, so the layers run on
// v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains, MI355X_MICROARCH "Matrix cores").
// Orientation: D^T = W . X^T — weights are the A operand (M = 16 output features per tile),
// 16 (point,part) pairs are the N columns.  With that orientation the accumulator registers of
// one layer ARE the B operands of the next (lane group g = lane>>4 holds features 16*mt+4g+r of
// pair lane&15 in register r), provided the next layer's weights are stored in the matching
// K order — so activations never leave registers between layers.  Weights are re-ordered into
// that per-(k-step, m-tile, lane) order while they are staged into LDS (46 KB per part).
// The 1-wide / 3-wide heads (occupancy logit, rgb out) are 16-term VALU dot products plus two
// cross-lane adds instead of wasting 15/16 of an MFMA tile.
#include <stdlib.h>
#include "pipeline.h"

#include "mlp_common.h"

// compile-time ablations for profiling experiments (tools/mlp_ablate.sh): 1 no activations, 2 no MFMA, 3 no LDS weight reads
#ifndef MLP_DBG
#define MLP_DBG 0
#endif
#if MLP_DBG == 2
#define mfma4(a, b, c) ((c) + (f32x4){(a) * (b), 0.f, 0.f, 0.f})
#endif
#if MLP_DBG == 1
#define softplus4_log2(v) (v)
#endif

// One 16-pair column block of a wave: inputs, activations and outputs stay in registers from the embedding to [rgb, occ].
struct MlpCol {
    float eb[EMB_STEPS];        // k-slots 4s+g of the (padded) 20-wide embedding of pair `col`
    float dv[3];                // canonical view direction
    f32x4 h[4];                 // hidden activations (log2 domain), feature 16 mt + 4 g + r
    f32x4 feat;                 // occ features 1..16 (true scale)
    float occ;
    float k5[11];               // rgb layer-1 k-slots 5..15: sin/cos (6), d (1), feat (4)
};

// The stages of the two MLPs for ONE column block.  mlp_part runs two column blocks per wave SKEWED by one stage, so that in
// every scheduling region the matrix-core instructions of one block sit beside the vector instructions (activation, view-
// direction encoding, heads) of the other: a SIMD issues VALU in the shadow of a 32-cycle fp32 MFMA only if independent VALU
// work is available at that point of the (in-order) wave.
__device__ __forceinline__ void st_occ1(const float* lds, int lane, int g, MlpCol& c) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) c.h[mt] = bias4(lds + O_B_OCC1, mt, g);
#pragma unroll
    for (int s = 0; s < EMB_STEPS; ++s) {
        const float4 a = *reinterpret_cast<const float4*>(lds + O_W_OCC1 + (s * 64 + lane) * 4);
        c.h[0] = mfma4(a.x, c.eb[s], c.h[0]); c.h[1] = mfma4(a.y, c.eb[s], c.h[1]);
        c.h[2] = mfma4(a.z, c.eb[s], c.h[2]); c.h[3] = mfma4(a.w, c.eb[s], c.h[3]);
    }
}
__device__ __forceinline__ void st_act(MlpCol& c) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) c.h[mt] = softplus4_log2(c.h[mt]);
}
__device__ __forceinline__ void st_occ2(const float* lds, int lane, int g, MlpCol& c) {       // features 1..16 on MFMA, logit 0 on VALU
    c.feat = bias4(lds + O_B_OCC2, 0, g);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 a = *reinterpret_cast<const float4*>(lds + O_W_OCC2 + (q * 64 + lane) * 4);
        c.feat = mfma4(a.x, c.h[q][0], c.feat); c.feat = mfma4(a.y, c.h[q][1], c.feat);
        c.feat = mfma4(a.z, c.h[q][2], c.feat); c.feat = mfma4(a.w, c.h[q][3], c.feat);
    }
    const float lg = head_dot_s(c.h, lds + O_V_OCC, g) + lds[O_V_OCC + 64];
    c.occ = one_minus_exp_neg(softplus_f(lg));                    // 1 - exp(-softplus(h0))  (:52)
}
__device__ __forceinline__ void st_rgb_in(MlpCol& c, int g, float fmul) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float sn, cs;
        sincos_hw(c.dv[k] * fmul, &sn, &cs);
        c.k5[2 * k] = sn;
        c.k5[2 * k + 1] = cs;
    }
    c.k5[6] = g == 0 ? c.dv[0] : (g == 1 ? c.dv[1] : (g == 2 ? c.dv[2] : 0.0f));
#pragma unroll
    for (int r = 0; r < 4; ++r) c.k5[7 + r] = c.feat[r];
}
__device__ __forceinline__ void st_rgb1(const float* lds, int lane, int g, MlpCol& c) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) c.h[mt] = bias4(lds + O_B_RGB1, mt, g);
#pragma unroll
    for (int s = 0; s < RGB1F_STEPS; ++s) {
        const float4 a = *reinterpret_cast<const float4*>(lds + O_W_RGB1 + (s * 64 + lane) * 4);
        const float b = s < EMB_STEPS ? c.eb[s < EMB_STEPS ? s : 0] : c.k5[s >= EMB_STEPS ? s - EMB_STEPS : 0];
        c.h[0] = mfma4(a.x, b, c.h[0]); c.h[1] = mfma4(a.y, b, c.h[1]);
        c.h[2] = mfma4(a.z, b, c.h[2]); c.h[3] = mfma4(a.w, b, c.h[3]);
    }
}
__device__ __forceinline__ void st_rgb2(const float* lds, int lane, int g, MlpCol& c) {
    f32x4 h2[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) h2[mt] = bias4(lds + O_B_RGB2, mt, g);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float4 a = *reinterpret_cast<const float4*>(lds + O_W_RGB2 + (s * 64 + lane) * 4);
        const float b = c.h[s >> 2][s & 3];
        h2[0] = mfma4(a.x, b, h2[0]); h2[1] = mfma4(a.y, b, h2[1]);
        h2[2] = mfma4(a.z, b, h2[2]); h2[3] = mfma4(a.w, b, h2[3]);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) c.h[mt] = h2[mt];
}
__device__ __forceinline__ float4 st_head(const float* lds, int g, const MlpCol& c) {          // rgb head 64 -> 3, sigmoid
    float o[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = sigmoid_f(head_dot_s(c.h, lds + O_V_OUT + k * 64, g) + lds[O_V_OUT + 3 * 64 + k]);
    return make_float4(o[0], o[1], o[2], c.occ);
}
#define MLP_FENCE() __builtin_amdgcn_sched_barrier(0)
// Interleave directive for one scheduling region (between two MLP_FENCEs): N_MFMA groups of {1 matrix instruction, V vector
// instructions} — the vector work of the other column block is issued in the shadows of this block's MFMAs (<= 5 single-issue
// VALU fit beside a 32-cycle fp32 MFMA; MI355X_MICROARCH.md).  LDS reads of the weights float freely.
template <int N_MFMA, int V>
__device__ __forceinline__ void mlp_interleave() {
#pragma unroll
    for (int i = 0; i < N_MFMA; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, V, 0);      // VALU
    }
}

template <int NRGB>
__device__ __forceinline__ void mlp_part(float* lds, const PartMlpDev& pm, const float* __restrict__ emb,
                                         const float* __restrict__ ds, int64_t stride, const int32_t* __restrict__ l_slot,
                                         int cnt, int64_t cap, float4* __restrict__ raws, int part,
                                         float4* __restrict__ raw_direct, const int vblock, const int nblocks) {
    if ((int64_t)vblock * (MLP_BLOCK / 64) * MLP_CB * 16 >= cnt) return;
    __syncthreads();                                   // previous part's weights no longer in use
    stage_weights<NRGB, true>(pm, lds);
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, col = lane & 15;
    const float fmul = (float)(1 << g);                          // frequency 2^g of this lane group

    const int64_t per_block = (MLP_BLOCK / 64) * MLP_CB * 16, step = (int64_t)nblocks * per_block;
    auto load_in = [&](int64_t wbase, MlpCol& A, MlpCol& B) {
        const int64_t pa = min(wbase + col, (int64_t)cnt - 1), pb = min(wbase + 16 + col, (int64_t)cnt - 1);
#pragma unroll
        for (int s = 0; s < EMB_STEPS; ++s) { A.eb[s] = emb[(int64_t)(4 * s + g) * cap + pa]; B.eb[s] = emb[(int64_t)(4 * s + g) * cap + pb]; }
#pragma unroll
        for (int c = 0; c < 3; ++c) { A.dv[c] = ds[(int64_t)c * stride + pa]; B.dv[c] = ds[(int64_t)c * stride + pb]; }
    };
    int64_t wbase = (int64_t)vblock * per_block + (int64_t)wv * MLP_CB * 16;
    if (wbase >= cnt) return;
    MlpCol A, B;
    float nA_eb[EMB_STEPS], nB_eb[EMB_STEPS], nA_dv[3], nB_dv[3];      // inputs of the NEXT tile: loaded a whole tile ahead
    load_in(wbase, A, B);
    for (; wbase < cnt; wbase += step) {
        {
            MlpCol tA, tB;
            load_in(min(wbase + step, (int64_t)cnt - 1), tA, tB);         // (clamped: the values are unused past the last tile)
#pragma unroll
            for (int s = 0; s < EMB_STEPS; ++s) { nA_eb[s] = tA.eb[s]; nB_eb[s] = tB.eb[s]; }
#pragma unroll
            for (int c = 0; c < 3; ++c) { nA_dv[c] = tA.dv[c]; nB_dv[c] = tB.dv[c]; }
        }
        st_occ1(lds, lane, g, A);
        MLP_FENCE();
        st_occ1(lds, lane, g, B); st_act(A);
        mlp_interleave<20, 4>();
        MLP_FENCE();
        st_occ2(lds, lane, g, A); st_act(B);
        mlp_interleave<16, 5>();
        MLP_FENCE();
        st_occ2(lds, lane, g, B); st_rgb_in(A, g, fmul);
        mlp_interleave<16, 4>();
        MLP_FENCE();
        st_rgb1(lds, lane, g, A); st_rgb_in(B, g, fmul);
        mlp_interleave<30, 1>();
        MLP_FENCE();
        st_rgb1(lds, lane, g, B); st_act(A);
        mlp_interleave<64, 1>();
        MLP_FENCE();
        float4 rA, rB;
        if (NRGB == 3) {
            st_rgb2(lds, lane, g, A); st_act(B);
            mlp_interleave<64, 1>();
            MLP_FENCE();
            st_rgb2(lds, lane, g, B); st_act(A);
            mlp_interleave<64, 1>();
            MLP_FENCE();
            rA = st_head(lds, g, A); st_act(B);
            MLP_FENCE();
            rB = st_head(lds, g, B);
        } else {
            rA = st_head(lds, g, A); st_act(B);
            MLP_FENCE();
            rB = st_head(lds, g, B);
        }
        const int64_t pa = wbase + col, pb = wbase + 16 + col;
        if (g == 0) {
            if (pa < cnt) { if (raw_direct) raw_direct[pa] = rA; else raws[(int64_t)l_slot[pa] * INVR_NUM_PARTS + part] = rA; }
            if (pb < cnt) { if (raw_direct) raw_direct[pb] = rB; else raws[(int64_t)l_slot[pb] * INVR_NUM_PARTS + part] = rB; }
        }
#pragma unroll
        for (int s = 0; s < EMB_STEPS; ++s) { A.eb[s] = nA_eb[s]; B.eb[s] = nB_eb[s]; }
#pragma unroll
        for (int c = 0; c < 3; ++c) { A.dv[c] = nA_dv[c]; B.dv[c] = nB_dv[c]; }
    }
}


template <int NRGB>
__global__ __launch_bounds__(MLP_BLOCK, 3) void k_part_mlp(PartMlpDev pm, const float* __restrict__ emb,
                                                        const float* __restrict__ ds, int64_t stride,
                                                        const int32_t* __restrict__ l_slot,
                                                        const int32_t* __restrict__ count, int64_t cap,
                                                        float4* __restrict__ raws, int part, float4* __restrict__ raw_direct) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    mlp_part<NRGB>(lds, pm, emb, ds, stride, l_slot, *count, cap, raws, part, raw_direct, (int)blockIdx.x, (int)gridDim.x);
}

// all five parts in one persistent launch (see k_part_encode_rs_all): the weights of the next part are staged
// into the same LDS image when a workgroup has finished its share of the previous one
__global__ __launch_bounds__(MLP_BLOCK, 3) void k_part_mlp_all(MlpAllArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    // Every workgroup serves ONE part: the parts get contiguous ranges of workgroups in proportion to their tile counts
    // (device-side counts), so a workgroup stages one LDS weight image instead of five (5 x ~6 us — a fifth of the kernel on a
    // 1/8 shard) and the per-part ceil() shares of a walk over all parts disappear.  A part too small for a range of its own
    // is taken along by the workgroup where its range would start.
    const int per_block = (MLP_BLOCK / 64) * MLP_CB * 16, G = (int)gridDim.x, b = (int)blockIdx.x;
    int64_t tiles[INVR_NUM_PARTS], total = 0;          // tile counts weighted by the part's cost per pair (22.1 : 14.0 kFLOP = 8 : 5)
    for (int p = 0; p < INVR_NUM_PARTS; ++p) {
        tiles[p] = (int64_t)((a.counts[p] + per_block - 1) / per_block) * (a.pm[p].rgb.n_linear == 3 ? 8 : 5);
        total += tiles[p];
    }
    if (total == 0) return;
    int64_t cum = 0;
    for (int p = 0; p < INVR_NUM_PARTS; ++p) {
        const int start = (int)(cum * G / total), end = (int)((cum + tiles[p]) * G / total);
        cum += tiles[p];
        if (tiles[p] == 0) continue;
        int vb, nb;
        if (end > start) { if (b < start || b >= end) continue; vb = b - start; nb = end - start; }
        else { if (b != min(start, G - 1)) continue; vb = 0; nb = 1; }
        if (a.pm[p].rgb.n_linear == 3)
            mlp_part<3>(lds, a.pm[p], a.emb[p], a.ds[p], a.stride, a.l_slot[p], a.counts[p], a.cap, a.raws, p, nullptr, vb, nb);
        else
            mlp_part<2>(lds, a.pm[p], a.emb[p], a.ds[p], a.stride, a.l_slot[p], a.counts[p], a.cap, a.raws, p, nullptr, vb, nb);
    }
}

static bool part_mlp_supported(const PartMlpDev& pm) {
    const MlpDev& o = pm.occ;
    const MlpDev& r = pm.rgb;
    return o.n_linear == 2 && o.dims[0] == 19 && o.dims[1] == HID && o.dims[2] == 17 &&
           (r.n_linear == 2 || r.n_linear == 3) && r.dims[0] == 70 && r.dims[1] == HID &&
           r.dims[r.n_linear] == 3 && (r.n_linear == 2 || r.dims[2] == HID) &&
           pm.n_freq == 4 && pm.latent_dim == 8 && pm.geo_dim == 16;
}

int launch_part_mlp_all(const MlpAllArgs& a, hipStream_t st) {
    for (int p = 0; p < INVR_NUM_PARTS; ++p)
        if (!part_mlp_supported(a.pm[p])) {
            invr_set_error("part MLP kernel supports occ 19-64-17 and rgb 70-64(-64)-3 with 4 view-dir frequencies, latent 8, geo feature 16");
            return 1;
        }
    const int64_t per_block = (MLP_BLOCK / 64) * MLP_CB * 16;
    int64_t tiles = cdiv(a.cap, per_block);
    unsigned grid = (unsigned)(tiles < 256 * 3 ? (tiles > 0 ? tiles : 1) : 256 * 3);
    hipLaunchKernelGGL(k_part_mlp_all, dim3(grid), dim3(MLP_BLOCK), 0, st, a);
    INVR_LAUNCH_CHECK();
    return 0;
}

int launch_part_mlp(const PartMlpDev& pm, const float* emb, const float* d_soa, int64_t stride,
                    const int32_t* l_slot, const int32_t* count, int64_t cap, float4* raws, int part,
                    float4* raw_direct, hipStream_t st) {
    const MlpDev& o = pm.occ;
    const MlpDev& r = pm.rgb;
    bool ok = o.n_linear == 2 && o.dims[0] == 19 && o.dims[1] == HID && o.dims[2] == 17 &&
              (r.n_linear == 2 || r.n_linear == 3) && r.dims[0] == 70 && r.dims[1] == HID &&
              r.dims[r.n_linear] == 3 && (r.n_linear == 2 || r.dims[2] == HID) &&
              pm.n_freq == 4 && pm.latent_dim == 8 && pm.geo_dim == 16;
    if (!ok) {
        invr_set_error("part MLP kernel supports occ 19-64-17 and rgb 70-64(-64)-3 with 4 view-dir frequencies, latent 8, geo feature 16");
        return 1;
    }
    const int64_t per_block = (MLP_BLOCK / 64) * MLP_CB * 16;
    int64_t tiles = cdiv(cap, per_block);
    static int wpb = getenv("INVR_MLP_BPC") ? atoi(getenv("INVR_MLP_BPC")) : 3;
    unsigned grid = (unsigned)(tiles < 256 * wpb ? (tiles > 0 ? tiles : 1) : 256 * wpb);
    if (r.n_linear == 3)
        hipLaunchKernelGGL(k_part_mlp<3>, dim3(grid), dim3(MLP_BLOCK), 0, st, pm, emb, d_soa, stride, l_slot, count, cap, raws, part, raw_direct);
    else
        hipLaunchKernelGGL(k_part_mlp<2>, dim3(grid), dim3(MLP_BLOCK), 0, st, pm, emb, d_soa, stride, l_slot, count, cap, raws, part, raw_direct);
    INVR_LAUNCH_CHECK();
    return 0;
}
