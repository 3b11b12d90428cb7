// Shared pieces of the part-MLP kernels (k_mlp.hip forward, k_mlp_bwd.hip backward): LDS weight image,
// K-order maps, MFMA / activation helpers.  See k_mlp.hip for the layout description.
#pragma once
#include "pipeline.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MLP_BLOCK 256
#define MLP_CB 2                 // 16-pair column blocks per wave iteration
#define HID 64
#define EMB_STEPS 5              // 20 / 4
#define RGB1_STEPS 18            // 72 / 4
#define RGB1F_STEPS 16           // forward kernels: the 8 latent inputs are constant per call and folded into the bias (64 / 4)

// The colour MLP's two 64-wide layers (70 -> 64 and 64 -> 64) in the FORWARD kernels run on the bf16 matrix pipe at fp32 accuracy:
// weights and activations are split into three bf16 terms (hi + mid + lo = the fp32 value exactly: 3 x 8 significand bits), and the six
// leading products hi hi, hi mid, mid hi, hi lo, lo hi, mid mid (the dropped ones are <= 2^-24 relative) run as v_mfma_f32_16x16x32_bf16
// with fp32 accumulation: 48 MFMAs of 16 cycles per 16-pair tile and layer instead of 64 fp32 MFMAs of 32 cycles.  Measured on the
// layer alone (tools/mlp_layer_microbench.hip, splitting and activation included): 0.58x the time, error against float64 7.4e-7 (fp32
// MFMA: 8.7e-7).  -DMLP_BF16=0 builds the fp32 form (the A/B switch of profiles/r4_bf16_mlp.md).
#ifndef MLP_BF16
#define MLP_BF16 1
#endif
typedef __bf16 mlp_bf16x8 __attribute__((ext_vector_type(8)));
#define RGB_BF_FLOATS (3 * 4 * 2 * 64 * 4)           // [split 3][m-tile 4][k-block 2][lane 64] x 8 bf16 = 24 KB per layer
#define RGB1_LDS (MLP_BF16 && RGB_BF_FLOATS > RGB1_STEPS * 4 * 64 ? RGB_BF_FLOATS : RGB1_STEPS * 4 * 64)
#define RGB2_LDS (MLP_BF16 && RGB_BF_FLOATS > 16 * 4 * 64 ? RGB_BF_FLOATS : 16 * 4 * 64)

// x = hi + mid + lo exactly (round-to-nearest conversions; the remainders are exact fp32 subtractions)
__device__ __forceinline__ void bf16_split3(float x, __bf16& hi, __bf16& mid, __bf16& lo) {
    hi = (__bf16)x;
    const float r1 = x - (float)hi;
    mid = (__bf16)r1;
    lo = (__bf16)(r1 - (float)mid);
}
__device__ __forceinline__ void bf16_split3x8(const float* v, mlp_bf16x8& hi, mlp_bf16x8& mid, mlp_bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { __bf16 a, b, c; bf16_split3(v[j], a, b, c); hi[j] = a; mid[j] = b; lo[j] = c; }
}

// LDS carve (floats)
#define O_W_OCC1 0                                   // 5*4*64
#define O_B_OCC1 (O_W_OCC1 + EMB_STEPS * 4 * 64)     // 64
#define O_W_OCC2 (O_B_OCC1 + 64)                     // 16*64   (feature rows 1..16)
#define O_B_OCC2 (O_W_OCC2 + 16 * 64)                // 16
#define O_V_OCC (O_B_OCC2 + 16)                      // 4*16    (row 0, slot order) + bias
#define O_W_RGB1 (O_V_OCC + 64 + 4)                  // 18*4*64 floats (fp32 image) | RGB_BF_FLOATS (bf16 x 3 image); 16-byte aligned
#define O_B_RGB1 (O_W_RGB1 + RGB1_LDS)               // 64
#define O_W_RGB2 (O_B_RGB1 + 64)                     // 16*4*64 | RGB_BF_FLOATS
#define O_B_RGB2 (O_W_RGB2 + RGB2_LDS)               // 64
#define O_V_OUT (O_B_RGB2 + 64)                      // 3*4*16 + 3(+1)
#define LDS_FLOATS (O_V_OUT + 3 * 64 + 4)

// source column of rgb layer-1 for k-slot (step s, lane group g); -1 = zero padding.
// rgb input = [emb 0..18 | dirPE 19..45 | feat 46..61 | latent 62..69] (part_base_network.py:57)
// dirPE = [d, sin(2^0 d), cos(2^0 d), sin(2^1 d), ...] (freq_embedder.py:20-31)
__device__ __forceinline__ int rgb1_col(int s, int g) {
    if (s < 5) { int e = 4 * s + g; return e < 19 ? e : -1; }
    if (s < 11) { int u = s - 5, comp = u >> 1, fn = u & 1; return 19 + 3 + g * 6 + fn * 3 + comp; }
    if (s < 14) { int e = 4 * (s - 11) + g; return e < 3 ? 19 + e : (e < 11 ? 62 + (e - 3) : -1); }
    return 46 + 4 * g + (s - 14);
}
// forward kernels (LOG2DOM staging): [emb 0..18 + pad | sin / cos 24 | d 3 + pad | feat 16] = 16 k-steps; the latent code is the
// same for every pair of a call (one frame), so W[:, 62..69] . latent is added to the layer's bias when the weights are staged —
// 8 of the part's 108 / 172 MFMAs per 16-pair tile less.
__device__ __forceinline__ int rgb1f_col(int s, int g) {
    if (s < 5) { int e = 4 * s + g; return e < 19 ? e : -1; }
    if (s < 11) { int u = s - 5, comp = u >> 1, fn = u & 1; return 19 + 3 + g * 6 + fn * 3 + comp; }
    if (s == 11) return g < 3 ? 19 + g : -1;
    return 46 + 4 * g + (s - 12);
}
// hidden unit held by lane group g for k-step s of a 64-wide hidden layer
__device__ __forceinline__ int hid_col(int s, int g) { return 16 * (s >> 2) + 4 * g + (s & 3); }

// LOG2DOM (forward-only kernels): the four m-tile weights of a k-step sit next to each other per lane ([k-step][lane][m-tile];
// occ layer 2, one m-tile: four k-steps), so ONE ds_read_b128 feeds four MFMAs — LDS instructions take issue slots like VALU
// does, and the SIMD's issue is what bounds these kernels.  The hidden activations are kept in the log2 domain, u = log2(1 + exp2(z log2e)) = softplus(z) /
// ln2, and the two scale factors are folded into the staged weights — a layer that feeds a Softplus is scaled by log2e (weights
// and bias), a layer that consumes Softplus outputs by ln2; for a hidden-to-hidden layer the two cancel exactly (only its bias is
// scaled).  The activation then costs {min, exp2, add, log2} = 4 single-issue VALU per value instead of 4.5 issue slots with packed
// multiplies, which are expensive beside MFMAs (MI355X_MICROARCH.md "price of one filler beside MFMAs").
template <int NRGB, bool LOG2DOM = false, int WHAT = 3>   // NRGB: number of rgb linears, 2 (70-64-3) or 3 (70-64-64-3); WHAT: 1 = occ MLP, 2 = rgb MLP, 3 = both
__device__ void stage_weights(const PartMlpDev& pm, float* lds) {
    const float s_in = LOG2DOM ? INVR_LOG2E : 1.0f, s_out = LOG2DOM ? INVR_LN2 : 1.0f;
    const float* W0 = pm.occ.w[0]; const float* W1 = pm.occ.w[1];
    const float* R0 = pm.rgb.w[0]; const float* R1 = pm.rgb.w[1]; const float* R2 = pm.rgb.w[NRGB - 1];
    if (WHAT & 1) {
        for (int t = threadIdx.x; t < EMB_STEPS * 4 * 64; t += MLP_BLOCK) {
            int ln = t & 63, mt = (t >> 6) & 3, s = t >> 8, g = ln >> 4, i = ln & 15;
            int col = 4 * s + g;
            lds[O_W_OCC1 + (LOG2DOM ? (s * 64 + ln) * 4 + mt : t)] = col < 19 ? W0[(16 * mt + i) * 19 + col] * s_in : 0.0f;
        }
        for (int t = threadIdx.x; t < 16 * 64; t += MLP_BLOCK) {
            int ln = t & 63, s = t >> 6, g = ln >> 4, i = ln & 15;
            lds[O_W_OCC2 + (LOG2DOM ? ((s >> 2) * 64 + ln) * 4 + (s & 3) : t)] = W1[(1 + i) * HID + hid_col(s, g)] * s_out;
        }
    }
    if ((WHAT & 2) && LOG2DOM && MLP_BF16) {
        // bf16 x 3 images: element j of (m-tile mo, k-block kb, lane (g, i)) = W[16 mo + i][column of k-slot (s = 8 kb + j, g)]
        mlp_bf16x8* w1 = reinterpret_cast<mlp_bf16x8*>(lds + O_W_RGB1);
        mlp_bf16x8* w2 = reinterpret_cast<mlp_bf16x8*>(lds + O_W_RGB2);
        for (int t = threadIdx.x; t < (NRGB == 3 ? 2 : 1) * 4 * 2 * 64; t += MLP_BLOCK) {
            const int layer = t >> 9, ln = t & 63, kb = (t >> 6) & 1, mo = (t >> 7) & 3, g = ln >> 4, i = ln & 15;
            float w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (layer == 0) { const int col = rgb1f_col(8 * kb + j, g); w[j] = col >= 0 ? R0[(16 * mo + i) * 70 + col] * s_in : 0.0f; }
                else w[j] = R1[(16 * mo + i) * HID + hid_col(8 * kb + j, g)];
            }
            mlp_bf16x8 vh, vm, vl;
            bf16_split3x8(w, vh, vm, vl);
            mlp_bf16x8* dst = layer == 0 ? w1 : w2;
            dst[(0 * 8 + mo * 2 + kb) * 64 + ln] = vh;
            dst[(1 * 8 + mo * 2 + kb) * 64 + ln] = vm;
            dst[(2 * 8 + mo * 2 + kb) * 64 + ln] = vl;
        }
    } else if (WHAT & 2) {
        for (int t = threadIdx.x; t < (LOG2DOM ? RGB1F_STEPS : RGB1_STEPS) * 4 * 64; t += MLP_BLOCK) {
            int ln = t & 63, mt = (t >> 6) & 3, s = t >> 8, g = ln >> 4, i = ln & 15;
            int col = LOG2DOM ? rgb1f_col(s, g) : rgb1_col(s, g);
            lds[O_W_RGB1 + (LOG2DOM ? (s * 64 + ln) * 4 + mt : t)] = col >= 0 ? R0[(16 * mt + i) * 70 + col] * s_in : 0.0f;
        }
        if (NRGB == 3)
            for (int t = threadIdx.x; t < 16 * 4 * 64; t += MLP_BLOCK) {
                int ln = t & 63, mt = (t >> 6) & 3, s = t >> 8, g = ln >> 4, i = ln & 15;
                lds[O_W_RGB2 + (LOG2DOM ? (s * 64 + ln) * 4 + mt : t)] = R1[(16 * mt + i) * HID + hid_col(s, g)];
            }
    }
    for (int t = threadIdx.x; t < 64; t += MLP_BLOCK) {
        int g = t >> 4, u = t & 15;                              // slot order: [g][mt*4+r]
        int hc = 16 * (u >> 2) + 4 * g + (u & 3);
        if (WHAT & 1) {
            lds[O_B_OCC1 + t] = pm.occ.b[0][t] * s_in;
            lds[O_V_OCC + t] = W1[hc] * s_out;                   // occ logit row 0
        }
        if (WHAT & 2) {
            float b1 = pm.rgb.b[0][t];
            if (LOG2DOM) {                                       // + W[:, latent] . latent (the latent block of the rgb input)
                const float* lat = pm.rgb_latent + pm.latent_index[0] * pm.latent_dim;
#pragma unroll
                for (int j = 0; j < 8; ++j) b1 = fmaf(R0[t * 70 + 62 + j], lat[j], b1);
            }
            lds[O_B_RGB1 + t] = b1 * s_in;
            if (NRGB == 3) lds[O_B_RGB2 + t] = pm.rgb.b[1][t] * s_in;      // (rgb2 weights: ln2 * log2e = 1, unscaled)
#pragma unroll
            for (int c = 0; c < 3; ++c) lds[O_V_OUT + c * 64 + t] = R2[c * HID + hc] * s_out;
        }
    }
    if ((WHAT & 1) && threadIdx.x < 16) lds[O_B_OCC2 + threadIdx.x] = pm.occ.b[1][1 + threadIdx.x];
    if (threadIdx.x == 0) {
        if (WHAT & 1) lds[O_V_OCC + 64] = pm.occ.b[1][0];
        if (WHAT & 2) for (int c = 0; c < 3; ++c) lds[O_V_OUT + 3 * 64 + c] = pm.rgb.b[NRGB - 1][c];
    }
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// Softplus of the four accumulator values of an MFMA tile.  The matrix pipe and the VALU do not overlap on a SIMD in
// these kernels (their busy times add up, profiles/), so VALU issue slots are kernel time: this form is
// ln2 * log2(1 + exp2(x log2e)) with the multiplies / add as packed fp32 (v_pk_mul_f32 / v_pk_add_f32, two values per
// issue) — 4.5 issue slots per value instead of the 7 of max(x,0) + ln2 log2(1 + exp2(-|x| log2e)) (softplus_f).
// Absolute error <= 1 ulp of the result (1.4e-6 at x = 20, 2e-7 for |x| <= 2), the same class as torch's
// log1p(exp(x)); x log2e is clamped at 126 so that huge pre-activations return x (1 +- 1e-7) instead of inf.
typedef float mlp_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 softplus4(f32x4 v) {
    mlp_v2f a = {v[0], v[1]}, b = {v[2], v[3]};
    a = a * INVR_LOG2E;
    b = b * INVR_LOG2E;
    mlp_v2f ea = {exp2_raw(fminf(a.x, 126.0f)), exp2_raw(fminf(a.y, 126.0f))};
    mlp_v2f eb = {exp2_raw(fminf(b.x, 126.0f)), exp2_raw(fminf(b.y, 126.0f))};
    ea = ea + 1.0f;
    eb = eb + 1.0f;
    mlp_v2f la = {log2_raw(ea.x), log2_raw(ea.y)}, lb = {log2_raw(eb.x), log2_raw(eb.y)};
    la = la * INVR_LN2;
    lb = lb * INVR_LN2;
    f32x4 r;
    r[0] = la.x; r[1] = la.y; r[2] = lb.x; r[3] = lb.y;
    return r;
}
// log2-domain Softplus of an MFMA tile whose pre-activation already carries the log2e factor (stage_weights<.., true>):
// u = log2(1 + exp2(a)); softplus = ln2 * u is folded into the consumer's weights.  a is clamped at 126 (exp2 overflow).
__device__ __forceinline__ f32x4 softplus4_log2(f32x4 v) {
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = log2_raw(1.0f + exp2_raw(fminf(v[k], 126.0f)));
    return r;
}
__device__ __forceinline__ f32x4 bias4(const float* b, int mt, int g) {
    const float* p = b + 16 * mt + 4 * g;
    f32x4 r; r[0] = p[0]; r[1] = p[1]; r[2] = p[2]; r[3] = p[3];
    return r;
}
// scalar-FMA form of head_dot (forward kernels: packed fp32 ops beside MFMAs cost more than their slot)
__device__ __forceinline__ float head_dot_s(const f32x4* h, const float* wv, int g) {
    const float* w = wv + g * 16;
    float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        a0 = fmaf(w[mt * 4], h[mt][0], a0); a1 = fmaf(w[mt * 4 + 1], h[mt][1], a1);
        a0 = fmaf(w[mt * 4 + 2], h[mt][2], a0); a1 = fmaf(w[mt * 4 + 3], h[mt][3], a1);
    }
    float r = a0 + a1;
    r += __shfl_xor(r, 16);
    r += __shfl_xor(r, 32);
    return r;
}
// dot of the 16 hidden values this lane holds with slot-ordered head weights, summed over the 4 lane groups
__device__ __forceinline__ float head_dot(const f32x4* h, const float* wv, int g) {
    // two products per issue slot (v_pk_fma_f32); the slot-ordered weights of this lane group are 16 contiguous floats
    const mlp_v2f* w2 = reinterpret_cast<const mlp_v2f*>(wv + g * 16);
    mlp_v2f acc = {0.0f, 0.0f};
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const mlp_v2f lo = {h[mt][0], h[mt][1]}, hi = {h[mt][2], h[mt][3]};
        acc = __builtin_elementwise_fma(w2[mt * 2], lo, acc);
        acc = __builtin_elementwise_fma(w2[mt * 2 + 1], hi, acc);
    }
    float r = acc.x + acc.y;
    r += __shfl_xor(r, 16);
    r += __shfl_xor(r, 32);
    return r;
}
