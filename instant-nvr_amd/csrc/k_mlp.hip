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
#if MLP_BF16
    mlp_bf16x8 sh[2], sm[2], sl[2];   // the 16 inputs of the next 64-wide layer (k-slots s = 8 kb + j) as bf16 hi / mid / lo terms
#endif
};

#if MLP_BF16
// One 64 -> 64 (or 64-slot -> 64) layer of a column block on the bf16 matrix pipe: six products per (m-tile, k-block), smallest first.
__device__ __forceinline__ void bf16x3_layer(const float* lds_w, const float* lds_b, int lane, int g, const MlpCol& c, f32x4* out) {
    const mlp_bf16x8* w = reinterpret_cast<const mlp_bf16x8*>(lds_w);
#pragma unroll
    for (int mo = 0; mo < 4; ++mo) {
        out[mo] = bias4(lds_b, mo, g);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const mlp_bf16x8 ah = w[(0 * 8 + mo * 2 + kb) * 64 + lane], am = w[(1 * 8 + mo * 2 + kb) * 64 + lane], al = w[(2 * 8 + mo * 2 + kb) * 64 + lane];
            out[mo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, c.sm[kb], out[mo], 0, 0, 0);
            out[mo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, c.sl[kb], out[mo], 0, 0, 0);
            out[mo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, c.sh[kb], out[mo], 0, 0, 0);
            out[mo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, c.sm[kb], out[mo], 0, 0, 0);
            out[mo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, c.sh[kb], out[mo], 0, 0, 0);
            out[mo] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, c.sh[kb], out[mo], 0, 0, 0);
        }
    }
}
// the hidden activations c.h (k-slot s <-> h[s >> 2][s & 3]) as the split inputs of the next layer
__device__ __forceinline__ void bf16x3_split_h(MlpCol& c) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = c.h[2 * kb + (j >> 2)][j & 3];
        bf16_split3x8(v, c.sh[kb], c.sm[kb], c.sl[kb]);
    }
}
#endif

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
// activation of a colour-MLP layer whose output feeds another 64-wide layer (rgb1 of the three-layer nets): + the bf16 split
__device__ __forceinline__ void st_act_split(MlpCol& c) {
    st_act(c);
#if MLP_BF16
    bf16x3_split_h(c);
#endif
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
#if MLP_BF16
    {   // the 16 k-slots of rgb layer 1 (s < 5: embedding, s >= 5: k5) as split bf16 inputs
        float v0[8] = {c.eb[0], c.eb[1], c.eb[2], c.eb[3], c.eb[4], c.k5[0], c.k5[1], c.k5[2]};
        float v1[8] = {c.k5[3], c.k5[4], c.k5[5], c.k5[6], c.k5[7], c.k5[8], c.k5[9], c.k5[10]};
        bf16_split3x8(v0, c.sh[0], c.sm[0], c.sl[0]);
        bf16_split3x8(v1, c.sh[1], c.sm[1], c.sl[1]);
    }
#endif
}
__device__ __forceinline__ void st_rgb1(const float* lds, int lane, int g, MlpCol& c) {
#if MLP_BF16
    bf16x3_layer(lds + O_W_RGB1, lds + O_B_RGB1, lane, g, c, c.h);
    return;
#endif
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
#if MLP_BF16
    {
        f32x4 o[4];
        bf16x3_layer(lds + O_W_RGB2, lds + O_B_RGB2, lane, g, c, o);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) c.h[mt] = o[mt];
        return;
    }
#endif
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
// matrix / vector instruction mix of the colour-MLP regions below (bf16 x 3: 48 MFMAs of 16 cycles + the split of the next inputs)
#if MLP_BF16
#define RGB_MFMAS 48
#define RGB_V 3
#define RGB_MFMAS_IN 48
#define RGB_V_IN 3
#else
#define RGB_MFMAS 64
#define RGB_V 1
#define RGB_MFMAS_IN 30
#define RGB_V_IN 1
#endif
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
                                         const float* __restrict__ ds, int64_t stride,
                                         int cnt, int64_t cap, float4* __restrict__ raw_direct, const int vblock, const int nblocks) {
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
        mlp_interleave<RGB_MFMAS_IN, RGB_V_IN>();
        MLP_FENCE();
        st_rgb1(lds, lane, g, B);
        if (NRGB == 3) st_act_split(A); else st_act(A);
        mlp_interleave<RGB_MFMAS, RGB_V>();
        MLP_FENCE();
        float4 rA, rB;
        if (NRGB == 3) {
            st_rgb2(lds, lane, g, A); st_act_split(B);
            mlp_interleave<RGB_MFMAS, RGB_V>();
            MLP_FENCE();
            st_rgb2(lds, lane, g, B); st_act(A);
            mlp_interleave<RGB_MFMAS, 1>();
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
            if (pa < cnt) raw_direct[pa] = rA;
            if (pb < cnt) raw_direct[pb] = rB;
        }
#pragma unroll
        for (int s = 0; s < EMB_STEPS; ++s) { A.eb[s] = nA_eb[s]; B.eb[s] = nB_eb[s]; }
#pragma unroll
        for (int c = 0; c < 3; ++c) { A.dv[c] = nA_dv[c]; B.dv[c] = nB_dv[c]; }
    }
}


// One part's MLPs over a dense pair list, every pair fully evaluated: the stage-level entry points (invr_part_field_fwd /
// invr_part_mlp_fwd).  The render path runs the two phases below instead.
template <int NRGB>
__global__ __launch_bounds__(MLP_BLOCK, 3) void k_part_mlp(PartMlpDev pm, const float* __restrict__ emb,
                                                        const float* __restrict__ ds, int64_t stride,
                                                        const int32_t* __restrict__ count, int64_t cap,
                                                        float4* __restrict__ raw_direct) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    mlp_part<NRGB>(lds, pm, emb, ds, stride, *count, cap, raw_direct, (int)blockIdx.x, (int)gridDim.x);
}

// ---- render path -------------------------------------------------------------------------------------------------------
// TPoseHuman.forward keeps, per survivor, only the raw of the part with the largest occupancy (first maximum;
// inb_part_network_multiassign.py:253-256): the colour MLP of every other evaluated pair is dead work — 17.5 k (body, head) /
// 9.3 k (leg, arms) of a pair's 22.1 k / 14.0 kFLOP, for 49 % of the pairs of the bench frame.  So the part MLPs run in two
// phases with identical results:
//   k_part_occ_all    every listed pair: 19 -> 64 -> 17, occupancy to occp[p][pair], the 16 geometry features to feat[p][pair]
//   k_winner_lists    per survivor the arg-max over {listed occupancies, far constants, 0 for unflagged parts} -> wsel[slot];
//                     the winning LISTED pairs (+ the far-constant pair of every part) as per-part lists, segmented by slot group
//   k_part_rgb_all    70 -> 64 (-> 64) -> 3 on the winners only; [rgb, occ] to rgbw[slot] (far constants: rgbw[lcap + p])
// A pair's arithmetic does not depend on its column or tile, so both phases reproduce the one-kernel results bit for bit.

#define OCC_LDS_FLOATS O_W_RGB1
__device__ __forceinline__ void occ_part(float* lds, const PartMlpDev& pm, const float* __restrict__ emb, int cnt, int64_t cap,
                                         float* __restrict__ occp, float4* __restrict__ feat, const int vblock, const int nblocks) {
    const int64_t per_block = (MLP_BLOCK / 64) * MLP_CB * 16, step = (int64_t)nblocks * per_block;
    if ((int64_t)vblock * per_block >= cnt) return;
    __syncthreads();                                   // previous part's weights no longer in use
    stage_weights<2, true, 1>(pm, lds);
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, col = lane & 15;
    auto load_in = [&](int64_t wbase, float* ea, float* eb) {
        const int64_t pa = min(wbase + col, (int64_t)cnt - 1), pb = min(wbase + 16 + col, (int64_t)cnt - 1);
#pragma unroll
        for (int s = 0; s < EMB_STEPS; ++s) { ea[s] = emb[(int64_t)(4 * s + g) * cap + pa]; eb[s] = emb[(int64_t)(4 * s + g) * cap + pb]; }
    };
    int64_t wbase = (int64_t)vblock * per_block + (int64_t)wv * MLP_CB * 16;
    if (wbase >= cnt) return;
    MlpCol A, B;
    float nA[EMB_STEPS], nB[EMB_STEPS];                // inputs of the NEXT tile: loaded a whole tile ahead
    load_in(wbase, A.eb, B.eb);
    for (; wbase < cnt; wbase += step) {
        load_in(min(wbase + step, (int64_t)cnt - 1), nA, nB);
        st_occ1(lds, lane, g, A);
        MLP_FENCE();
        st_occ1(lds, lane, g, B); st_act(A);
        mlp_interleave<20, 4>();
        MLP_FENCE();
        st_occ2(lds, lane, g, A); st_act(B);
        mlp_interleave<16, 5>();
        MLP_FENCE();
        st_occ2(lds, lane, g, B);
        MLP_FENCE();
        const int64_t pa = wbase + col, pb = wbase + 16 + col;
        if (pa < cnt) { feat[pa * 4 + g] = make_float4(A.feat[0], A.feat[1], A.feat[2], A.feat[3]); if (g == 0) occp[pa] = A.occ; }
        if (pb < cnt) { feat[pb * 4 + g] = make_float4(B.feat[0], B.feat[1], B.feat[2], B.feat[3]); if (g == 0) occp[pb] = B.occ; }
#pragma unroll
        for (int s = 0; s < EMB_STEPS; ++s) { A.eb[s] = nA[s]; B.eb[s] = nB[s]; }
    }
}

// Every workgroup serves ONE part: the parts get contiguous ranges of workgroups in proportion to their (cost-weighted) tile
// counts (device-side counts), so a workgroup stages one LDS weight image instead of five and the per-part ceil() shares of a
// walk over all parts disappear.  A part too small for a range of its own is taken along by the workgroup where its range
// would start.  -> (vb, nb) of this workgroup inside part p's range, false if it does not serve p.
__device__ __forceinline__ bool part_range(const int64_t* tiles, int64_t total, int p, int b, int G, int& vb, int& nb) {
    int64_t cum = 0;
    for (int q = 0; q < p; ++q) cum += tiles[q];
    const int start = (int)(cum * G / total), end = (int)((cum + tiles[p]) * G / total);
    if (tiles[p] == 0) return false;
    if (end > start) { if (b < start || b >= end) return false; vb = b - start; nb = end - start; }
    else { if (b != min(start, G - 1)) return false; vb = 0; nb = 1; }
    return true;
}

__global__ __launch_bounds__(MLP_BLOCK, 4) void k_part_occ_all(MlpAllArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[OCC_LDS_FLOATS];
    const int per_block = (MLP_BLOCK / 64) * MLP_CB * 16, G = (int)gridDim.x, b = (int)blockIdx.x;
    int64_t tiles[INVR_NUM_PARTS], total = 0;
    for (int p = 0; p < INVR_NUM_PARTS; ++p) { tiles[p] = (a.counts[p] + per_block - 1) / per_block; total += tiles[p]; }
    if (total == 0) return;
    for (int p = 0; p < INVR_NUM_PARTS; ++p) {
        int vb, nb;
        if (!part_range(tiles, total, p, b, G, vb, nb)) continue;
        occ_part(lds, a.pm[p], a.emb[p], a.counts[p], a.cap, a.occp[p], a.feat[p], vb, nb);
    }
}

// The merge of one survivor = arg-max of the occupancies over the five parts, zeros for unflagged parts, the first maximum wins
// (inb_part_network_multiassign.py:229-256 with cfg.aggr == ""), evaluated here from the phase-1 occupancies; the survivors whose
// winner is a LISTED pair are what the colour MLP still has to see.  One workgroup per group of PAIR_GROUP survivor slots — the
// layout of k_pair_lists: the pairs of group g sit at [off_g, off_g + gcount[g][p]) of part p's list, off_g = the counts of the
// groups before it — which writes its winners, ascending, to wl[p][off_g ...) and their number to wcnt[g][p]: no global offsets,
// no atomics; k_part_rgb_all walks the segments.  The last group appends the far-constant pair of every part.
//
// cfg.aggr == 'mean' (:236-239; InvrScene::aggr = INVR_AGGR_MEAN): the merged raw is the mean over the five parts of (rgb, occ), zeros
// for unflagged parts, so every listed pair needs its colour.  The phases then run as
//   k_part_occ_all -> k_all_lists (wl = identity, wcnt = gcount: every pair "wins") -> k_part_rgb_all ([rgb, occ] per PAIR to
//   feat[p][pair * 4]) -> k_winner_lists<true>: the same rank walk as the arg-max merge, but it sums the listed / far-constant
//   values of a survivor in part order, divides by 5 and writes rgbw[slot] (wsel = 0: "read rgbw[slot]").
#define WL_BLOCK 512         // x WL_PER = PAIR_GROUP: one pass per group (the kernel is a chain of dependent round trips)
#define WL_PER 8
// cfg.aggr == 'dist' (:240-244, MODE 2): the 'mean' flow with the parts weighted by F.normalize(1 / (part_dist + 1e-5)) (L2 over the five
// parts, eps 1e-12) instead of 1 / 5 — part_dist of every (slot, part) from k_knn_pdist.  cfg.aggr == 'mindist' (:245-251, MODE 3): the
// arg-max flow with the winner = the part of smallest part_dist (first minimum), whatever its occupancy — its listed pair, its far
// constant, or zeros when that part is not flagged for the survivor.
template <int MODE>          // 0 = max occupancy, 1 = mean, 2 = dist, 3 = mindist (InvrScene::aggr)
__global__ __launch_bounds__(WL_BLOCK) void k_winner_lists(Workspace w) {
    constexpr bool MEAN = MODE == 1 || MODE == 2;
    __shared__ int s_cnt[WL_BLOCK / 64][INVR_NUM_PARTS];
    __shared__ int s_red[WL_BLOCK / 64][INVR_NUM_PARTS];
    const int na = w.counters[CNT_ACTIVE];
    const int64_t g = blockIdx.x, g_last = (max(na, 1) - 1) / PAIR_GROUP;
    if (g > g_last) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int base[INVR_NUM_PARTS], wbase[INVR_NUM_PARTS], cidx[INVR_NUM_PARTS];
    {
        int acc[INVR_NUM_PARTS] = {0, 0, 0, 0, 0};
        for (int64_t q = threadIdx.x; q < g; q += WL_BLOCK)
#pragma unroll
            for (int p = 0; p < INVR_NUM_PARTS; ++p) acc[p] += w.gcount[q * INVR_NUM_PARTS + p];
#pragma unroll
        for (int p = 0; p < INVR_NUM_PARTS; ++p) {
            acc[p] = __builtin_amdgcn_readlane(wave_incl_sum_i(acc[p]), 63);
            if (lane == 0) s_red[wv][p] = acc[p];
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < INVR_NUM_PARTS; ++p) {
            base[p] = 0;
            for (int k = 0; k < WL_BLOCK / 64; ++k) base[p] += s_red[k][p];
            wbase[p] = 0;
            cidx[p] = w.counters[CNT_PAIRS + p] - 1;               // list index of the part's far-constant pair
        }
    }
    const int pbase0[INVR_NUM_PARTS] = {base[0], base[1], base[2], base[3], base[4]};
    for (int64_t t0 = g * PAIR_GROUP; t0 < min((g + 1) * (int64_t)PAIR_GROUP, (int64_t)na); t0 += WL_BLOCK * WL_PER) {
        const int64_t s0 = t0 + (int64_t)threadIdx.x * WL_PER;
        unsigned long long fb = 0ull, ffb = 0ull;
        if (s0 + WL_PER <= na) {
            fb = *reinterpret_cast<const unsigned long long*>(w.pflags + s0);
            ffb = *reinterpret_cast<const unsigned long long*>(w.farflags + s0);
        } else
            for (int k = 0; k < WL_PER; ++k) if (s0 + k < na) { fb |= (unsigned long long)w.pflags[s0 + k] << (8 * k); ffb |= (unsigned long long)w.farflags[s0 + k] << (8 * k); }
        // list index of this thread's first pair of every part: ranks inside the thread, the wave, the block (= k_pair_lists)
        int pos[INVR_NUM_PARTS];
#pragma unroll
        for (int p = 0; p < INVR_NUM_PARTS; ++p) {
            const int mine = __popcll(fb & (0x0101010101010101ull << p));
            const int x = wave_incl_sum_i(mine);
            pos[p] = x - mine;
            if (lane == 63) s_cnt[wv][p] = x;
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < INVR_NUM_PARTS; ++p) {
            pos[p] += base[p];
            for (int k = 0; k < WL_BLOCK / 64; ++k) { const int c = s_cnt[k][p]; if (k < wv) pos[p] += c; base[p] += c; }
        }
        __syncthreads();                 // s_cnt is reused below
        // the merge of the thread's WL_PER survivors.  All occupancies are fetched first, unconditionally (an unflagged / far
        // (survivor, part) reads the part's far constant): loads under per-part conditions would be 40 serial round trips
        unsigned long long wb = 0ull, sel8 = 0ull;
        int widx[WL_PER], pidx[WL_PER][INVR_NUM_PARTS];
        float oc[WL_PER][INVR_NUM_PARTS];
#pragma unroll
        for (int k = 0; k < WL_PER; ++k)
#pragma unroll
            for (int p = 0; p < INVR_NUM_PARTS; ++p) {
                const bool listed = (fb >> (8 * k + p)) & 1ull;
                pidx[k][p] = listed ? pos[p] : cidx[p];
                pos[p] += listed ? 1 : 0;
            }
        if (MEAN) {
            // raws.mean(dim=1) over (Na, P, 4) with zeros for unflagged parts: the per-pair values in part order, / P
#pragma unroll 2
            for (int k = 0; k < WL_PER; ++k) {
                const unsigned fl = (unsigned)(fb >> (8 * k)) & 0xffu, ff = (unsigned)(ffb >> (8 * k)) & 0xffu;
                float4 v[INVR_NUM_PARTS];
#pragma unroll
                for (int p = 0; p < INVR_NUM_PARTS; ++p) v[p] = w.feat[p][(int64_t)pidx[k][p] * 4];      // (unflagged: the part's far constant, dropped below)
                float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
                if (MODE == 2) {
                    // torch.sum(raws * F.normalize(1.0 / (part_dist + 1e-5), dim=-1)[..., None], dim=1)
                    float inv[INVR_NUM_PARTS], n2 = 0.0f;
                    const int64_t sl = min(s0 + k, (int64_t)na - 1);
#pragma unroll
                    for (int p = 0; p < INVR_NUM_PARTS; ++p) { inv[p] = 1.0f / (w.pdist[sl * INVR_NUM_PARTS + p] + 1e-5f); n2 += inv[p] * inv[p]; }
                    const float den = fmaxf(sqrtf(n2), 1e-12f);
#pragma unroll
                    for (int p = 0; p < INVR_NUM_PARTS; ++p) {
                        const float wp = inv[p] / den;
                        if ((fl | ff) & (1u << p)) { sum.x += v[p].x * wp; sum.y += v[p].y * wp; sum.z += v[p].z * wp; sum.w += v[p].w * wp; }
                    }
                    if (s0 + k < na) {
                        w.rgbw[s0 + k] = sum;
                        w.wsel[s0 + k] = ((fl | ff) & 0x1fu) ? (uint8_t)0 : (uint8_t)255;
                    }
                    continue;
                }
#pragma unroll
                for (int p = 0; p < INVR_NUM_PARTS; ++p)
                    if ((fl | ff) & (1u << p)) { sum.x += v[p].x; sum.y += v[p].y; sum.z += v[p].z; sum.w += v[p].w; }
                const float np = (float)INVR_NUM_PARTS;
                if (s0 + k < na) {
                    w.rgbw[s0 + k] = make_float4(sum.x / np, sum.y / np, sum.z / np, sum.w / np);
                    w.wsel[s0 + k] = ((fl | ff) & 0x1fu) ? (uint8_t)0 : (uint8_t)255;
                }
            }
            continue;
        }
#pragma unroll
        for (int k = 0; k < WL_PER; ++k)
#pragma unroll
            for (int p = 0; p < INVR_NUM_PARTS; ++p) oc[k][p] = w.occp[p][pidx[k][p]];
#pragma unroll
        for (int k = 0; k < WL_PER; ++k) {
            const unsigned fl = (unsigned)(fb >> (8 * k)) & 0xffu, ff = (unsigned)(ffb >> (8 * k)) & 0xffu;
            float best = 0.0f;
            unsigned bsel = 255u;
            int bidx = 0;
            if (MODE == 3) {
                // part_dist.argmin(dim=1): the first minimum (a NaN, from a part with fewer than 4 vertices, wins as in torch)
                const int64_t sl = min(s0 + k, (int64_t)na - 1);
                int bp = 0;
                float bd = w.pdist[sl * INVR_NUM_PARTS];
#pragma unroll
                for (int p = 1; p < INVR_NUM_PARTS; ++p) {
                    const float d = w.pdist[sl * INVR_NUM_PARTS + p];
                    if (!(bd != bd) && (d < bd || d != d)) { bd = d; bp = p; }
                }
#pragma unroll
                for (int p = 0; p < INVR_NUM_PARTS; ++p)
                    if (p == bp) {
                        if (fl & (1u << p)) bsel = (unsigned)p;
                        else if (ff & (1u << p)) bsel = 8u + (unsigned)p;
                        bidx = pidx[k][p];
                    }
            } else {
#pragma unroll
            for (int p = 0; p < INVR_NUM_PARTS; ++p) {
                float c = 0.0f;
                unsigned sel = 255u;
                if (fl & (1u << p)) { c = oc[k][p]; sel = (unsigned)p; }
                else if (ff & (1u << p)) { c = oc[k][p]; sel = 8u + (unsigned)p; }
                if (p == 0 || c > best) { best = c; bsel = sel; bidx = pidx[k][p]; }
            }
            }
            widx[k] = bidx;
            sel8 |= (unsigned long long)bsel << (8 * k);
            if (bsel < (unsigned)INVR_NUM_PARTS) wb |= 1ull << (8 * k + bsel);
        }
        if (s0 + WL_PER <= na) *reinterpret_cast<unsigned long long*>(w.wsel + s0) = sel8;
        else for (int k = 0; k < WL_PER; ++k) if (s0 + k < na) w.wsel[s0 + k] = (uint8_t)(sel8 >> (8 * k));
        // winner ranks, then the lists
        int wpos[INVR_NUM_PARTS];
#pragma unroll
        for (int p = 0; p < INVR_NUM_PARTS; ++p) {
            const int mine = __popcll(wb & (0x0101010101010101ull << p));
            const int x = wave_incl_sum_i(mine);
            wpos[p] = x - mine;
            if (lane == 63) s_cnt[wv][p] = x;
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < INVR_NUM_PARTS; ++p) {
            int q = pbase0[p] + wbase[p] + wpos[p];
            for (int k = 0; k < WL_BLOCK / 64; ++k) { const int c = s_cnt[k][p]; if (k < wv) q += c; wbase[p] += c; }
#pragma unroll
            for (int k = 0; k < WL_PER; ++k)
                if ((wb >> (8 * k + p)) & 1ull) w.wl[p][q++] = widx[k];
        }
        __syncthreads();                 // s_cnt is reused by the next tile
    }
    if (!MEAN && threadIdx.x < INVR_NUM_PARTS) {
        const int p = threadIdx.x;
        int nw = 0, off = 0, ci = 0;
#pragma unroll
        for (int q = 0; q < INVR_NUM_PARTS; ++q) if (q == p) { nw = wbase[q]; off = pbase0[q]; ci = cidx[q]; }
        if (g == g_last) w.wl[p][off + nw++] = ci;               // the far-constant pair: always evaluated
        w.wcnt[g * INVR_NUM_PARTS + p] = nw;
    }
}

// cfg.aggr == 'mean': every listed pair (and the far-constant pair that ends a part's list) goes through the colour MLP.
__global__ __launch_bounds__(256) void k_all_lists(Workspace w) {
    const int na = w.counters[CNT_ACTIVE];
    const int g_last = (max(na, 1) - 1) / PAIR_GROUP;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (int64_t)gridDim.x * blockDim.x;
#pragma unroll
    for (int p = 0; p < INVR_NUM_PARTS; ++p) {
        const int n = w.counters[CNT_PAIRS + p];
        for (int64_t i = t; i < n; i += nt) w.wl[p][i] = (int)i;
    }
    for (int64_t i = t; i < (int64_t)(g_last + 1) * INVR_NUM_PARTS; i += nt)
        w.wcnt[i] = w.gcount[i] + (i / INVR_NUM_PARTS == g_last ? 1 : 0);
}

// Phase 2.  A wave's unit of work is a tile of 32 winners (two 16-pair column blocks) of ONE segment (slot group); tile t of a
// part -> (segment, offset) through a wave-cooperative cursor: 64 segments per window (lane = segment), tile / pair prefix sums
// by wave scans, the window advanced as the wave's tile index grows — no per-part scan kernel, no LDS, any number of groups.
struct SegCursor {
    int g0, tiles_before, pairs_before;      // window start; tiles / pairs of the segments before it
    int wc, tl, ti, pe;                      // this lane's segment: winners, tiles, inclusive tile prefix, exclusive pair prefix
    int win_tiles, win_pairs;
};
__device__ __forceinline__ void seg_window(SegCursor& c, const int32_t* __restrict__ wcnt, const int32_t* __restrict__ gcount,
                                           int part, int g_last, int lane) {
    const int gi = c.g0 + lane;
    const bool ok = gi <= g_last;
    c.wc = ok ? wcnt[(int64_t)gi * INVR_NUM_PARTS + part] : 0;
    const int gc = ok ? gcount[(int64_t)gi * INVR_NUM_PARTS + part] : 0;
    c.tl = (c.wc + 31) >> 5;
    const int x = wave_incl_sum_i(c.tl), y = wave_incl_sum_i(gc);
    c.ti = x; c.pe = y - gc;
    c.win_tiles = __shfl(x, 63); c.win_pairs = __shfl(y, 63);
}
// -> first tile of the segment that holds tile T, the segment's offset in the lists and its winner count
__device__ __forceinline__ void seg_locate(SegCursor& c, int T, const int32_t* __restrict__ wcnt, const int32_t* __restrict__ gcount,
                                           int part, int g_last, int lane, int& seg_tile0, int& sb, int& sn) {
    while (T >= c.tiles_before + c.win_tiles && c.g0 + 64 <= g_last) {
        c.tiles_before += c.win_tiles; c.pairs_before += c.win_pairs; c.g0 += 64;
        seg_window(c, wcnt, gcount, part, g_last, lane);
    }
    const unsigned long long m = __ballot(c.tiles_before + c.ti > T);
    const int L = m ? __ffsll((long long)m) - 1 : 63;
    seg_tile0 = c.tiles_before + __shfl(c.ti - c.tl, L);
    sb = c.pairs_before + __shfl(c.pe, L);
    sn = __shfl(c.wc, L);
}

struct RgbIn { float eb[EMB_STEPS]; float dv[3]; float4 ft; };

template <int NRGB, bool MEAN>
__device__ __forceinline__ void rgb_part(float* lds, const MlpAllArgs& a, const int part, const int g_last, const int total_tiles,
                                         const int vblock, const int nblocks) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, col = lane & 15;
    if (vblock * (MLP_BLOCK / 64) >= total_tiles) return;
    __syncthreads();                                   // previous part's weights no longer in use
    stage_weights<NRGB, true, 2>(a.pm[part], lds);
    __syncthreads();
    const float fmul = (float)(1 << g);                          // frequency 2^g of this lane group
    const float* __restrict__ emb = a.emb[part];
    const float* __restrict__ ds = a.ds[part];
    const float4* __restrict__ feat = a.feat[part];
    float4* featw = a.feat[part];                                // (aggr 'mean' writes the pair's result back into the pair's first float4:
                                                                 //  read by this wave a tile earlier, by no other wave)
    const float* __restrict__ occp = a.occp[part];
    const int32_t* __restrict__ l_slot = a.l_slot[part];
    const int32_t* __restrict__ wl = a.wl[part];
    const int64_t cap = a.cap;
    const int step = nblocks * (MLP_BLOCK / 64);
    int T = vblock * (MLP_BLOCK / 64) + wv;
    if (T >= total_tiles) return;
    SegCursor cur;
    cur.g0 = 0; cur.tiles_before = 0; cur.pairs_before = 0;
    seg_window(cur, a.wcnt, a.gcount, part, g_last, lane);
    // pair indices of tile T (-1 = column beyond the segment's winners; the loads then read the segment's last winner)
    auto tile_pairs = [&](int T_, int& ia, int& ib, bool& va, bool& vb_) {
        int t0, sb, sn;
        seg_locate(cur, T_, a.wcnt, a.gcount, part, g_last, lane, t0, sb, sn);
        const int ja = (T_ - t0) * 32 + col, jb = ja + 16;
        va = ja < sn; vb_ = jb < sn;
        ia = wl[sb + min(ja, sn - 1)]; ib = wl[sb + min(jb, sn - 1)];
    };
    auto load_in = [&](int pa, int pb, RgbIn& A, RgbIn& B) {
#pragma unroll
        for (int s = 0; s < EMB_STEPS; ++s) { A.eb[s] = emb[(int64_t)(4 * s + g) * cap + pa]; B.eb[s] = emb[(int64_t)(4 * s + g) * cap + pb]; }
#pragma unroll
        for (int c = 0; c < 3; ++c) { A.dv[c] = ds[(int64_t)c * a.stride + pa]; B.dv[c] = ds[(int64_t)c * a.stride + pb]; }
        A.ft = feat[(int64_t)pa * 4 + g]; B.ft = feat[(int64_t)pb * 4 + g];
    };
    // software pipeline: inputs one tile ahead, pair indices two tiles ahead (a list read and the gathers it feeds are two
    // dependent round trips)
    int ia, ib, n_ia, n_ib, nn_ia = 0, nn_ib = 0;
    bool va, vb, n_va, n_vb, nn_va = false, nn_vb = false;
    RgbIn inA, inB, nA, nB;
    tile_pairs(T, ia, ib, va, vb);
    load_in(ia, ib, inA, inB);
    n_ia = ia; n_ib = ib; n_va = n_vb = false;
    if (T + step < total_tiles) tile_pairs(T + step, n_ia, n_ib, n_va, n_vb);
    for (; T < total_tiles; T += step) {
        load_in(n_ia, n_ib, nA, nB);                              // (tile T + step; unused past the last tile)
        nn_ia = n_ia; nn_ib = n_ib; nn_va = nn_vb = false;
        if (T + 2 * step < total_tiles) tile_pairs(T + 2 * step, nn_ia, nn_ib, nn_va, nn_vb);
        const int slotA = l_slot[ia], slotB = l_slot[ib];         // consumed by the stores at the end of the tile
        const float occA = occp[ia], occB = occp[ib];
        MlpCol A, B;
#pragma unroll
        for (int s = 0; s < EMB_STEPS; ++s) { A.eb[s] = inA.eb[s]; B.eb[s] = inB.eb[s]; }
#pragma unroll
        for (int c = 0; c < 3; ++c) { A.dv[c] = inA.dv[c]; B.dv[c] = inB.dv[c]; }
        A.feat[0] = inA.ft.x; A.feat[1] = inA.ft.y; A.feat[2] = inA.ft.z; A.feat[3] = inA.ft.w;
        B.feat[0] = inB.ft.x; B.feat[1] = inB.ft.y; B.feat[2] = inB.ft.z; B.feat[3] = inB.ft.w;
        A.occ = occA; B.occ = occB;
        st_rgb_in(A, g, fmul);
        MLP_FENCE();
        st_rgb1(lds, lane, g, A); st_rgb_in(B, g, fmul);
        mlp_interleave<RGB_MFMAS_IN, RGB_V_IN>();
        MLP_FENCE();
        st_rgb1(lds, lane, g, B);
        if (NRGB == 3) st_act_split(A); else st_act(A);
        mlp_interleave<RGB_MFMAS, RGB_V>();
        MLP_FENCE();
        float4 rA, rB;
        if (NRGB == 3) {
            st_rgb2(lds, lane, g, A); st_act_split(B);
            mlp_interleave<RGB_MFMAS, RGB_V>();
            MLP_FENCE();
            st_rgb2(lds, lane, g, B); st_act(A);
            mlp_interleave<RGB_MFMAS, 1>();
            MLP_FENCE();
            rA = st_head(lds, g, A); st_act(B);
            MLP_FENCE();
            rB = st_head(lds, g, B);
        } else {
            rA = st_head(lds, g, A); st_act(B);
            MLP_FENCE();
            rB = st_head(lds, g, B);
        }
        if (g == 0) {
            if (MEAN) {                                           // 'mean': per pair (k_winner_lists<true> sums them per survivor)
                if (va) featw[(int64_t)ia * 4] = rA;
                if (vb) featw[(int64_t)ib * 4] = rB;
            } else {
                if (va) a.rgbw[slotA == (int)(cap - 1) ? cap + part : (int64_t)slotA] = rA;
                if (vb) a.rgbw[slotB == (int)(cap - 1) ? cap + part : (int64_t)slotB] = rB;
            }
        }
        inA = nA; inB = nB;
        ia = n_ia; ib = n_ib; va = n_va; vb = n_vb;
        n_ia = nn_ia; n_ib = nn_ib; n_va = nn_va; n_vb = nn_vb;
    }
}

#ifndef RGB_WPS
#define RGB_WPS (MLP_BF16 ? 2 : 3)          // workgroups per CU the colour kernel is compiled / launched for (bf16 x 3: 59 KB of LDS each)
#endif
template <bool MEAN>
__global__ __launch_bounds__(MLP_BLOCK, RGB_WPS) void k_part_rgb_all(MlpAllArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    __shared__ int s_tot[INVR_NUM_PARTS];
    const int G = (int)gridDim.x, b = (int)blockIdx.x, lane = threadIdx.x & 63;
    const int na = a.n_active[0], g_last = (max(na, 1) - 1) / PAIR_GROUP;
    if (threadIdx.x < INVR_NUM_PARTS) s_tot[threadIdx.x] = 0;
    __syncthreads();
    {   // winner tiles per part (every workgroup sums the segment counts itself)
        int acc[INVR_NUM_PARTS] = {0, 0, 0, 0, 0};
        for (int q = threadIdx.x; q <= g_last; q += MLP_BLOCK)
#pragma unroll
            for (int p = 0; p < INVR_NUM_PARTS; ++p) acc[p] += (a.wcnt[(int64_t)q * INVR_NUM_PARTS + p] + 31) >> 5;
#pragma unroll
        for (int p = 0; p < INVR_NUM_PARTS; ++p) {
            acc[p] = __builtin_amdgcn_readlane(wave_incl_sum_i(acc[p]), 63);
            if (lane == 0 && acc[p]) atomicAdd(&s_tot[p], acc[p]);
        }
    }
    __syncthreads();
    int ntile[INVR_NUM_PARTS];
    int64_t tiles[INVR_NUM_PARTS], total = 0;          // tile counts weighted by the part's colour-MLP cost per pair (17.5 : 9.3 kFLOP)
    for (int p = 0; p < INVR_NUM_PARTS; ++p) {
        ntile[p] = s_tot[p];
        tiles[p] = (int64_t)((ntile[p] + (MLP_BLOCK / 64) - 1) / (MLP_BLOCK / 64)) * (a.pm[p].rgb.n_linear == 3 ? 15 : 8);
        total += tiles[p];
    }
    if (total == 0) return;
    for (int p = 0; p < INVR_NUM_PARTS; ++p) {
        int vb, nb;
        if (!part_range(tiles, total, p, b, G, vb, nb)) continue;
        if (a.pm[p].rgb.n_linear == 3) rgb_part<3, MEAN>(lds, a, p, g_last, ntile[p], vb, nb);
        else rgb_part<2, MEAN>(lds, a, p, g_last, ntile[p], vb, nb);
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

int launch_part_mlp_all(const MlpAllArgs& a, const Workspace& w, hipStream_t st) {
    for (int p = 0; p < INVR_NUM_PARTS; ++p)
        if (!part_mlp_supported(a.pm[p])) {
            invr_set_error("part MLP kernel supports occ 19-64-17 and rgb 70-64(-64)-3 with 4 view-dir frequencies, latent 8, geo feature 16");
            return 1;
        }
    const int64_t per_block = (MLP_BLOCK / 64) * MLP_CB * 16;
    int64_t tiles = cdiv(a.cap, per_block);
    unsigned grid_occ = (unsigned)(tiles < 256 * 4 ? (tiles > 0 ? tiles : 1) : 256 * 4);
    unsigned grid_rgb = (unsigned)(tiles < 256 * RGB_WPS ? (tiles > 0 ? tiles : 1) : 256 * RGB_WPS);
    hipLaunchKernelGGL(k_part_occ_all, dim3(grid_occ), dim3(MLP_BLOCK), 0, st, a);
    INVR_LAUNCH_CHECK();
    if (a.aggr == INVR_AGGR_MEAN || a.aggr == INVR_AGGR_DIST) {
        int64_t lt = cdiv(a.cap, 256);
        hipLaunchKernelGGL(k_all_lists, dim3((unsigned)(lt < 1024 ? (lt > 0 ? lt : 1) : 1024)), dim3(256), 0, st, w);
        INVR_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_part_rgb_all<true>, dim3(grid_rgb), dim3(MLP_BLOCK), 0, st, a);
        INVR_LAUNCH_CHECK();
        if (a.aggr == INVR_AGGR_MEAN) hipLaunchKernelGGL(k_winner_lists<1>, dim3((unsigned)w.n_groups), dim3(WL_BLOCK), 0, st, w);
        else hipLaunchKernelGGL(k_winner_lists<2>, dim3((unsigned)w.n_groups), dim3(WL_BLOCK), 0, st, w);
        INVR_LAUNCH_CHECK();
        return 0;
    }
    if (a.aggr == INVR_AGGR_MINDIST) hipLaunchKernelGGL(k_winner_lists<3>, dim3((unsigned)w.n_groups), dim3(WL_BLOCK), 0, st, w);
    else
    hipLaunchKernelGGL(k_winner_lists<0>, dim3((unsigned)w.n_groups), dim3(WL_BLOCK), 0, st, w);
    INVR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_part_rgb_all<false>, dim3(grid_rgb), dim3(MLP_BLOCK), 0, st, a);
    INVR_LAUNCH_CHECK();
    return 0;
}

int launch_part_mlp(const PartMlpDev& pm, const float* emb, const float* d_soa, int64_t stride,
                    const int32_t* count, int64_t cap, float4* raw_direct, hipStream_t st) {
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
        hipLaunchKernelGGL(k_part_mlp<3>, dim3(grid), dim3(MLP_BLOCK), 0, st, pm, emb, d_soa, stride, count, cap, raw_direct);
    else
        hipLaunchKernelGGL(k_part_mlp<2>, dim3(grid), dim3(MLP_BLOCK), 0, st, pm, emb, d_soa, stride, count, cap, raw_direct);
    INVR_LAUNCH_CHECK();
    return 0;
}
