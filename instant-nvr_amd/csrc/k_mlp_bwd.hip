// Row f1: backward of the two part MLPs (part_base_network.Network.forward after the encoder,
// part_base_network.py:44-63) on the fp32 matrix cores.
//
// One wave = one 16-pair column block at a time.  The forward is recomputed in registers exactly as in
// k_part_mlp (same LDS weight image, same K orders), then the data path runs backwards:
//   rgb head (VALU) -> [rgb2^T] -> rgb1^T -> occ2^T (+ logit row on VALU) -> occ1^T
// The transposed products g_in = W^T g_z reuse the FORWARD weight image: the A operand of output row i and
// k-step (mt', r') is W[out = 16 mt' + 4 g + r'][in = c(i)], which sits at a computable index of that image
// (4-way bank conflicts, cheap next to the MFMA).  Choosing the row -> input-column map c(i) per product puts
// every result directly into the layout its consumer needs (feature gradients in the `feat` layout, embedding
// gradients in k-slot order), so no cross-lane traffic is needed, and — as in the forward — the accumulators of
// one product are the B operands of the next.
// Weight gradients dW = g_z^T a_in have K = number of pairs: the kernel writes g_z and a_in of every layer as
// row-major (n, dim) matrices and the host reduces them with slab-batched GEMMs (autograd.py); bias gradients
// are column sums of g_z; the latent-code gradient is accumulated here.
#include "mlp_common.h"

// MlpBwdOut (pipeline.h): g_emb (20,n) SoA, rows 0..18 written (k-slot order = embedding column); gz (5, n_pad, 64) and
// a (5, n_pad, 72): per layer [0 occ1, 1 occ2, 2 rgb1, 3 rgb2 (3-linear colour nets only), 4 rgb head] the gradient
// w.r.t. the layer's pre-activation / output and the layer's input, row-major and zero-padded by the caller, so that ONE
// batched GEMM gz^T a over 2048-row slabs yields all weight gradients of the part (bias gradients = column sums of gz).
// The rgb1 input is stored in k-slot order (column 4 s + g, rgb1_col in mlp_common.h).  g_latent (8) is accumulated
// with atomics (pre-zeroed by the caller).

// Phase timers (profiling builds only, -DMLPB_PROF: tools/exp_mlpb_prof.py; not part of libinvr.so)
#ifdef MLPB_PROF
__device__ unsigned long long g_mlpb_prof[16];
extern "C" int invr_debug_mlpb_prof(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mlpb_prof), sizeof(g_mlpb_prof)) != hipSuccess) return 1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_mlpb_prof), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#define MP_DECL long long mp_t0 = clock64(); long long mp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define MP(i) { const long long mp_now = clock64(); mp_acc[i] += mp_now - mp_t0; mp_t0 = mp_now; }
#define MP_FLUSH if ((threadIdx.x & 63) == 0) { for (int mp_i = 0; mp_i < 8; ++mp_i) atomicAdd(&g_mlpb_prof[mp_i], (unsigned long long)mp_acc[mp_i]); atomicAdd(&g_mlpb_prof[8], 1ull); }
#else
#define MP_DECL
#define MP(i)
#define MP_FLUSH
#endif

__device__ __forceinline__ f32x4 dsoftplus4(f32x4 gin, f32x4 act) {    // softplus'(z) = sigmoid(z) = 1 - exp(-softplus(z))
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = gin[k] * (1.0f - exp2_raw(-act[k] * INVR_LOG2E));
    return r;
}
__device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }

template <int NRGB>
__global__ __launch_bounds__(MLP_BLOCK, 1) void k_part_mlp_bwd(PartMlpDev pm, const float* __restrict__ emb,
                                                            const float* __restrict__ ds, int64_t n_host, int64_t stride,
                                                            const int32_t* __restrict__ count,
                                                            const float4* __restrict__ g_raw, const int32_t* __restrict__ l_slot,
                                                            int part, MlpBwdOut o) {
    // n pairs (device count when `count` is given: the training pipeline never learns it on the host); SoA inputs /
    // outputs have `stride` entries per row; g_raw is (n,4) or, with l_slot, the per-(slot, part) array the merge wrote
    const int64_t n = count ? (int64_t)*count : n_host;
    __shared__ float lds[LDS_FLOATS];
    if ((int64_t)blockIdx.x * ((MLP_BLOCK / 64) * 16) >= n) return;       // (the grid is sized from an upper bound: no staging for nothing)
    MP_DECL
    stage_weights<NRGB>(pm, lds);
    __syncthreads();
    MP(0)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, col = lane & 15;
    const int i = col;                                       // as the row index of A operands
    const float* lat = pm.rgb_latent + pm.latent_index[0] * pm.latent_dim;
    const float misc0_lat = lat[0];
    const float misc1 = lat[1 + g];
    const float misc2 = g < 3 ? lat[5 + g] : 0.0f;
    const float fmul = (float)(1 << g);
    f32x4 lat_acc = {0.f, 0.f, 0.f, 0.f};
#define G(l) (o.gz + (int64_t)(l) * o.n_pad * 64)
#define A(l) (o.a + (int64_t)(l) * o.n_pad * 72)

    const int64_t per_block = (MLP_BLOCK / 64) * 16;
    if (n <= 0) return;
    for (int64_t t0 = (int64_t)blockIdx.x * per_block + (int64_t)wv * 16; t0 < n; t0 += (int64_t)gridDim.x * per_block) {
        const int64_t pair = min(t0 + col, n - 1);
        const bool live = t0 + col < n;
        MP(7)
        // ---------------- forward recompute (k_part_mlp, one column block) ----------------
        float eb[EMB_STEPS], dv[3];
#pragma unroll
        for (int s = 0; s < EMB_STEPS; ++s) eb[s] = emb[(int64_t)(4 * s + g) * stride + pair];
#pragma unroll
        for (int c = 0; c < 3; ++c) dv[c] = ds[(int64_t)c * stride + pair];
        f32x4 h1[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) h1[mt] = bias4(lds + O_B_OCC1, mt, g);
#ifdef MLPB_PROF
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        MP(1)
#pragma unroll
        for (int s = 0; s < EMB_STEPS; ++s)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) h1[mt] = mfma4(lds[O_W_OCC1 + (s * 4 + mt) * 64 + lane], eb[s], h1[mt]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) h1[mt] = softplus4(h1[mt]);
        f32x4 feat = bias4(lds + O_B_OCC2, 0, g);
        const float lg = head_dot(h1, lds + O_V_OCC, g) + lds[O_V_OCC + 64];
        const float occ = one_minus_exp_neg(softplus_f(lg));
#pragma unroll
        for (int s = 0; s < 16; ++s) feat = mfma4(lds[O_W_OCC2 + s * 64 + lane], h1[s >> 2][s & 3], feat);
        float kb[RGB1_STEPS];
#pragma unroll
        for (int s = 0; s < EMB_STEPS; ++s) kb[s] = eb[s];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float sn, cs;
            sincos_hw(dv[c] * fmul, &sn, &cs);
            kb[5 + 2 * c] = sn;
            kb[6 + 2 * c] = cs;
        }
        kb[11] = g == 0 ? dv[0] : (g == 1 ? dv[1] : (g == 2 ? dv[2] : misc0_lat));
        kb[12] = misc1;
        kb[13] = misc2;
#pragma unroll
        for (int r = 0; r < 4; ++r) kb[14 + r] = feat[r];
        if (live) {
#pragma unroll
            for (int s = 0; s < RGB1_STEPS; ++s) A(2)[pair * 72 + 4 * s + g] = kb[s];
#pragma unroll
            for (int s = 0; s < EMB_STEPS; ++s)
                if (4 * s + g < 19) A(0)[pair * 72 + 4 * s + g] = eb[s];
        }
        f32x4 hr1[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) hr1[mt] = bias4(lds + O_B_RGB1, mt, g);
#pragma unroll
        for (int s = 0; s < RGB1_STEPS; ++s)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) hr1[mt] = mfma4(lds[O_W_RGB1 + (s * 4 + mt) * 64 + lane], kb[s], hr1[mt]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) hr1[mt] = softplus4(hr1[mt]);
        f32x4 hl[4];                                           // last hidden rgb activation
        if (NRGB == 3) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) hl[mt] = bias4(lds + O_B_RGB2, mt, g);
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) hl[mt] = mfma4(lds[O_W_RGB2 + (s * 4 + mt) * 64 + lane], hr1[s >> 2][s & 3], hl[mt]);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) hl[mt] = softplus4(hl[mt]);
        } else {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) hl[mt] = hr1[mt];
        }
        float rgb[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb[c] = sigmoid_f(head_dot(hl, lds + O_V_OUT + c * 64, g) + lds[O_V_OUT + 3 * 64 + c]);

        MP(2)
        // ---------------- backward ----------------
        const float4 gr = live ? (l_slot ? g_raw[(int64_t)l_slot[pair] * INVR_NUM_PARTS + part] : g_raw[pair]) : make_float4(0.f, 0.f, 0.f, 0.f);
        float go[3] = {gr.x * rgb[0] * (1.0f - rgb[0]), gr.y * rgb[1] * (1.0f - rgb[1]), gr.z * rgb[2] * (1.0f - rgb[2])};
        const float g_lg = gr.w * (1.0f - occ) * occ;            // occ = 1 - exp(-softplus(lg)): d occ / d lg = (1 - occ) occ
        // rgb head^T (VALU): g_hl[hid] = sum_c Wout[c][hid] go[c]
        f32x4 gz[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = 0.0f;
#pragma unroll
                for (int c = 0; c < 3; ++c) a = fmaf(lds[O_V_OUT + c * 64 + g * 16 + mt * 4 + r], go[c], a);
                gz[mt][r] = a;
            }
            gz[mt] = dsoftplus4(gz[mt], hl[mt]);
        }
        if (live) {
            if (g == 0) { G(4)[pair * 64] = go[0]; G(4)[pair * 64 + 1] = go[1]; G(4)[pair * 64 + 2] = go[2]; }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                store4(G(NRGB == 3 ? 3 : 2) + pair * 64 + 16 * mt + 4 * g, gz[mt]);
                store4(A(4) + pair * 72 + 16 * mt + 4 * g, hl[mt]);
            }
        }
        if (NRGB == 3) {                                          // rgb2^T: g_hr1 = W2^T gz ; gz <- g_hr1 * softplus'(.)
            f32x4 acc[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) acc[mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mtp = 0; mtp < 4; ++mtp)
#pragma unroll
                for (int rp = 0; rp < 4; ++rp)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
                        acc[mi] = mfma4(lds[O_W_RGB2 + ((4 * mi + (i & 3)) * 4 + mtp) * 64 + (i >> 2) * 16 + 4 * g + rp], gz[mtp][rp], acc[mi]);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) gz[mi] = dsoftplus4(acc[mi], hr1[mi]);
            if (live) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    store4(G(2) + pair * 64 + 16 * mt + 4 * g, gz[mt]);
                    store4(A(3) + pair * 72 + 16 * mt + 4 * g, hr1[mt]);
                }
            }
        }
        MP(3)
        // rgb1^T: embedding slots (2 tiles, row 4g+r of tile mi <-> k-slot (s = 4 mi + r, g)), feature tile (row = feature
        // index), latent tile (row = latent index)
        f32x4 ge[2], gfeat = {0.f, 0.f, 0.f, 0.f}, glat = {0.f, 0.f, 0.f, 0.f};
        ge[0] = ge[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int e_lat = i + 3;
#pragma unroll
        for (int mtp = 0; mtp < 4; ++mtp)
#pragma unroll
            for (int rp = 0; rp < 4; ++rp) {
                const float b = gz[mtp][rp];
                const int tail = mtp * 64 + (i >> 2) * 16 + 4 * g + rp;
                ge[0] = mfma4(lds[O_W_RGB1 + ((i & 3) * 4) * 64 + tail], b, ge[0]);
                ge[1] = mfma4((i & 3) == 0 ? lds[O_W_RGB1 + (4 * 4) * 64 + tail] : 0.0f, b, ge[1]);
                gfeat = mfma4(lds[O_W_RGB1 + ((14 + (i & 3)) * 4) * 64 + tail], b, gfeat);
                glat = mfma4(i < 8 ? lds[O_W_RGB1 + ((11 + (e_lat >> 2)) * 4 + mtp) * 64 + (e_lat & 3) * 16 + 4 * g + rp] : 0.0f, b, glat);
            }
        lat_acc += glat;
        // occ layer 2^T: g_h1 = W1[1..16]^T g_feat + W1[0] g_lg ; then softplus'
        f32x4 gh[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
            for (int r = 0; r < 4; ++r) gh[mi][r] = lds[O_V_OCC + g * 16 + mi * 4 + r] * g_lg;
#pragma unroll
            for (int rp = 0; rp < 4; ++rp)
                gh[mi] = mfma4(lds[O_W_OCC2 + (4 * mi + (i & 3)) * 64 + (i >> 2) * 16 + 4 * g + rp], gfeat[rp], gh[mi]);
            gh[mi] = dsoftplus4(gh[mi], h1[mi]);
        }
        if (live) {
            float* q = G(1) + pair * 64;
            if (g == 0) q[0] = g_lg;
#pragma unroll
            for (int r = 0; r < 4; ++r) q[1 + 4 * g + r] = gfeat[r];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                store4(G(0) + pair * 64 + 16 * mt + 4 * g, gh[mt]);
                store4(A(1) + pair * 72 + 16 * mt + 4 * g, h1[mt]);
            }
        }
        // occ layer 1^T into the same embedding-slot tiles
#pragma unroll
        for (int mtp = 0; mtp < 4; ++mtp)
#pragma unroll
            for (int rp = 0; rp < 4; ++rp) {
                const float b = gh[mtp][rp];
                const int tail = mtp * 64 + (i >> 2) * 16 + 4 * g + rp;
                ge[0] = mfma4(lds[O_W_OCC1 + ((i & 3) * 4) * 64 + tail], b, ge[0]);
                ge[1] = mfma4((i & 3) == 0 ? lds[O_W_OCC1 + (4 * 4) * 64 + tail] : 0.0f, b, ge[1]);
            }
        if (live) {
#pragma unroll
            for (int s = 0; s < EMB_STEPS; ++s)
                if (4 * s + g < 19) o.g_emb[(int64_t)(4 * s + g) * stride + pair] = ge[s >> 2][s & 3];
        }
        MP(4)
    }
    // latent-code gradient: rows 4g+r (< 8) of the latent tile, summed over this wave's pairs and the 16 columns
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float v = lat_acc[r];
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) v += __shfl_xor(v, d);
        if (col == 0 && g < 2) atomicAdd(o.g_latent + (o.latent_full ? pm.latent_index[0] * pm.latent_dim : 0) + 4 * g + r, v);
    }
    MP(5)
    MP_FLUSH
}

#undef G
#undef A

int launch_part_mlp_bwd(const PartMlpDev& pm, const float* emb_soa, const float* d_soa, int64_t n, int64_t stride,
                        const int32_t* count, const float* g_raw, const int32_t* l_slot, int part, const MlpBwdOut& o, hipStream_t st) {
    if (n == 0) return 0;                              // n = the pair count, or its upper bound when `count` (device) is given
    const int64_t per_block = (MLP_BLOCK / 64) * 16;
    int64_t tiles = cdiv(n, per_block);
    unsigned grid = (unsigned)(tiles < 256 * 2 ? tiles : 256 * 2);
    if (pm.rgb.n_linear == 3)
        hipLaunchKernelGGL(k_part_mlp_bwd<3>, dim3(grid), dim3(MLP_BLOCK), 0, st, pm, emb_soa, d_soa, n, stride, count,
                           reinterpret_cast<const float4*>(g_raw), l_slot, part, o);
    else
        hipLaunchKernelGGL(k_part_mlp_bwd<2>, dim3(grid), dim3(MLP_BLOCK), 0, st, pm, emb_soa, d_soa, n, stride, count,
                           reinterpret_cast<const float4*>(g_raw), l_slot, part, o);
    INVR_LAUNCH_CHECK();
    return 0;
}
