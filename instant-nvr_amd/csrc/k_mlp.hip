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

template <int NRGB>
__device__ __forceinline__ void mlp_part(float* lds, const PartMlpDev& pm, const float* __restrict__ emb,
                                         const float* __restrict__ ds, int64_t stride, const int32_t* __restrict__ l_slot,
                                         int cnt, int64_t cap, float4* __restrict__ raws, int part,
                                         float4* __restrict__ raw_direct) {
    if ((int64_t)blockIdx.x * (MLP_BLOCK / 64) * MLP_CB * 16 >= cnt) return;
    __syncthreads();                                   // previous part's weights no longer in use
    stage_weights<NRGB>(pm, lds);
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, col = lane & 15;
    // per-lane constant k-slots of the [d, latent, pad] block of the rgb input
    const float* lat = pm.rgb_latent + pm.latent_index[0] * pm.latent_dim;
    const float misc0_lat = lat[0];                              // e = 3 (g == 3)
    const float misc1 = lat[1 + g];                              // e = 4+g
    const float misc2 = g < 3 ? lat[5 + g] : 0.0f;               // e = 8+g ; e = 11 is padding
    const float fmul = (float)(1 << g);                          // frequency 2^g of this lane group

    const int64_t per_block = (MLP_BLOCK / 64) * MLP_CB * 16;
    for (int64_t t0 = (int64_t)blockIdx.x * per_block; t0 < cnt; t0 += (int64_t)gridDim.x * per_block) {
        const int64_t wbase = t0 + (int64_t)wv * MLP_CB * 16;
        if (wbase >= cnt) continue;
        int64_t pair[MLP_CB];
        float eb[MLP_CB][EMB_STEPS];
        float dv[MLP_CB][3];
#pragma unroll
        for (int cb = 0; cb < MLP_CB; ++cb) {
            pair[cb] = min(wbase + cb * 16 + col, (int64_t)cnt - 1);
#pragma unroll
            for (int s = 0; s < EMB_STEPS; ++s) eb[cb][s] = emb[(int64_t)(4 * s + g) * cap + pair[cb]];
#pragma unroll
            for (int c = 0; c < 3; ++c) dv[cb][c] = ds[(int64_t)c * stride + pair[cb]];
        }
        // ---- occ layer 1: 20 -> 64
        f32x4 h[MLP_CB][4];
#pragma unroll
        for (int cb = 0; cb < MLP_CB; ++cb)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) h[cb][mt] = bias4(lds + O_B_OCC1, mt, g);
#pragma unroll
        for (int s = 0; s < EMB_STEPS; ++s)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const float a = lds[O_W_OCC1 + (s * 4 + mt) * 64 + lane];
#pragma unroll
                for (int cb = 0; cb < MLP_CB; ++cb) h[cb][mt] = mfma4(a, eb[cb][s], h[cb][mt]);
            }
#pragma unroll
        for (int cb = 0; cb < MLP_CB; ++cb)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) h[cb][mt] = softplus4(h[cb][mt]);
        // ---- occ layer 2: features 1..16 on MFMA, logit 0 on VALU
        f32x4 feat[MLP_CB];
        float occ[MLP_CB];
#pragma unroll
        for (int cb = 0; cb < MLP_CB; ++cb) {
            feat[cb] = bias4(lds + O_B_OCC2, 0, g);
            float lg = head_dot(h[cb], lds + O_V_OCC, g) + lds[O_V_OCC + 64];
            occ[cb] = one_minus_exp_neg(softplus_f(lg));          // 1 - exp(-softplus(h0))  (:52)
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float a = lds[O_W_OCC2 + s * 64 + lane];
#pragma unroll
            for (int cb = 0; cb < MLP_CB; ++cb) feat[cb] = mfma4(a, h[cb][s >> 2][s & 3], feat[cb]);
        }
        // ---- rgb layer 1: 72 -> 64
#pragma unroll
        for (int cb = 0; cb < MLP_CB; ++cb)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) h[cb][mt] = bias4(lds + O_B_RGB1, mt, g);
        float kb[MLP_CB][RGB1_STEPS];
#pragma unroll
        for (int cb = 0; cb < MLP_CB; ++cb) {
#pragma unroll
            for (int s = 0; s < EMB_STEPS; ++s) kb[cb][s] = eb[cb][s];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float sn, cs;
                sincos_hw(dv[cb][c] * fmul, &sn, &cs);
                kb[cb][5 + 2 * c] = sn;
                kb[cb][6 + 2 * c] = cs;
            }
            kb[cb][11] = g == 0 ? dv[cb][0] : (g == 1 ? dv[cb][1] : (g == 2 ? dv[cb][2] : misc0_lat));
            kb[cb][12] = misc1;
            kb[cb][13] = misc2;
#pragma unroll
            for (int r = 0; r < 4; ++r) kb[cb][14 + r] = feat[cb][r];
        }
#pragma unroll
        for (int s = 0; s < RGB1_STEPS; ++s)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const float a = lds[O_W_RGB1 + (s * 4 + mt) * 64 + lane];
#pragma unroll
                for (int cb = 0; cb < MLP_CB; ++cb) h[cb][mt] = mfma4(a, kb[cb][s], h[cb][mt]);
            }
#pragma unroll
        for (int cb = 0; cb < MLP_CB; ++cb)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) h[cb][mt] = softplus4(h[cb][mt]);
        // ---- rgb layer 2: 64 -> 64 (body, head)
        if (NRGB == 3) {
            f32x4 h2[MLP_CB][4];
#pragma unroll
            for (int cb = 0; cb < MLP_CB; ++cb)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) h2[cb][mt] = bias4(lds + O_B_RGB2, mt, g);
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const float a = lds[O_W_RGB2 + (s * 4 + mt) * 64 + lane];
#pragma unroll
                    for (int cb = 0; cb < MLP_CB; ++cb) h2[cb][mt] = mfma4(a, h[cb][s >> 2][s & 3], h2[cb][mt]);
                }
#pragma unroll
            for (int cb = 0; cb < MLP_CB; ++cb)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) h[cb][mt] = softplus4(h2[cb][mt]);
        }
        // ---- rgb head 64 -> 3, sigmoid; store [rgb, occ]
#pragma unroll
        for (int cb = 0; cb < MLP_CB; ++cb) {
            float o[3];
#pragma unroll
            for (int c = 0; c < 3; ++c)
                o[c] = sigmoid_f(head_dot(h[cb], lds + O_V_OUT + c * 64, g) + lds[O_V_OUT + 3 * 64 + c]);
            const int64_t pi = wbase + cb * 16 + col;
            if (g == 0 && pi < cnt) {
                float4 r = make_float4(o[0], o[1], o[2], occ[cb]);
                if (raw_direct) raw_direct[pi] = r;
                else raws[(int64_t)l_slot[pi] * INVR_NUM_PARTS + part] = r;
            }
        }
    }
}


template <int NRGB>
__global__ __launch_bounds__(MLP_BLOCK, 3) void k_part_mlp(PartMlpDev pm, const float* __restrict__ emb,
                                                        const float* __restrict__ ds, int64_t stride,
                                                        const int32_t* __restrict__ l_slot,
                                                        const int32_t* __restrict__ count, int64_t cap,
                                                        float4* __restrict__ raws, int part, float4* __restrict__ raw_direct) {
    __shared__ float lds[LDS_FLOATS];
    mlp_part<NRGB>(lds, pm, emb, ds, stride, l_slot, *count, cap, raws, part, raw_direct);
}

// all five parts in one persistent launch (see k_part_encode_rs_all): the weights of the next part are staged
// into the same LDS image when a workgroup has finished its share of the previous one
__global__ __launch_bounds__(MLP_BLOCK, 3) void k_part_mlp_all(MlpAllArgs a) {
    __shared__ float lds[LDS_FLOATS];
    for (int p = 0; p < INVR_NUM_PARTS; ++p) {
        if (a.pm[p].rgb.n_linear == 3)
            mlp_part<3>(lds, a.pm[p], a.emb[p], a.ds[p], a.stride, a.l_slot[p], a.counts[p], a.cap, a.raws, p, nullptr);
        else
            mlp_part<2>(lds, a.pm[p], a.emb[p], a.ds[p], a.stride, a.l_slot[p], a.counts[p], a.cap, a.raws, p, nullptr);
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
