// K3 + K4: linear-blend-skinning inverse warp to the canonical big pose and the hash-encoded
// residual deformer, evaluated per flagged (point, part) pair.
// Replaces Network.pose_points_to_tpose_points (inb_part_network_multiassign.py:77-120) with
// get_inverse_blend_params / get_blend_params / torch_inverse_3x3 / pose_points_to_tpose_points /
// tpose_points_to_pose_points / pose_dirs_to_tpose_dirs / tpose_dirs_to_pose_dirs
// (lib/utils/blend_utils.py:395-487, 293-317) and Deformer.forward
// (lib/networks/deformers/uv_deformer.py:23-45: UV-volume trilinear -> (u,v,t) -> 8-level F=2
// grid encoder -> 19-32-32-3 Softplus MLP -> 0.05*tanh).
//
// One thread per pair.  The 24 joint matrices and the deformer MLP weights are wave-uniform:
// they are read through the scalar path (s_load) and used as SGPR operands of v_fmac.
// The deformer tables are 0.34 MB (L2 resident).
#include <stdlib.h>
#include "pipeline.h"
#include "grid_generic.h"
#include "front_bodies.h"

#define WARP_BLOCK 128

// adjugate / (det + eps)  (blend_utils.py:293-317)
__device__ __forceinline__ void inverse3x3(const Mat34& M, float* inv) {
    const float a = M.m[0], b = M.m[1], c = M.m[2];
    const float d = M.m[4], e = M.m[5], f = M.m[6];
    const float g = M.m[8], h = M.m[9], i = M.m[10];
    const float m00 = e * i - f * h, m01 = d * i - f * g, m02 = d * h - e * g;
    const float m10 = b * i - c * h, m11 = a * i - c * g, m12 = a * h - b * g;
    const float m20 = b * f - c * e, m21 = a * f - c * d, m22 = a * e - b * d;
    const float det = a * m00 - b * m01 + c * m02;
    const float den = det + 1.1920928955078125e-07f;      // torch.finfo(float32).eps
    inv[0] = m00 / den;  inv[1] = -m10 / den; inv[2] = m20 / den;
    inv[3] = -m01 / den; inv[4] = m11 / den;  inv[5] = -m21 / den;
    inv[6] = m02 / den;  inv[7] = -m12 / den; inv[8] = m22 / den;
}

__device__ __forceinline__ void warp_with_mats(const Mat34& Aw, const Mat34& Bw, const float* pp, const float* pd, float* xb, float* db);

__device__ __forceinline__ void warp_point(const float* __restrict__ A, const float* __restrict__ big_A, const float* bw,
                                           const float* pp, const float* pd, float* xb, float* db) {
    Mat34 Aw, Bw;
    blend_mats(A, bw, Aw);
    blend_mats(big_A, bw, Bw);
    warp_with_mats(Aw, Bw, pp, pd, xb, db);
}

// canonical big-pose point / direction from the blended pose matrix Aw and big-pose matrix Bw
__device__ __forceinline__ void warp_with_mats(const Mat34& Aw, const Mat34& Bw, const float* pp, const float* pd, float* xb, float* db) {
    float inv[9];
    inverse3x3(Aw, inv);
    const float x0 = pp[0] - Aw.m[3], x1 = pp[1] - Aw.m[7], x2 = pp[2] - Aw.m[11];
    float xt[3], dt[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        xt[r] = inv[r * 3] * x0 + inv[r * 3 + 1] * x1 + inv[r * 3 + 2] * x2;           // R_inv . (x - t)
        dt[r] = inv[r * 3] * pd[0] + inv[r * 3 + 1] * pd[1] + inv[r * 3 + 2] * pd[2];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        xb[r] = Bw.m[r * 4] * xt[0] + Bw.m[r * 4 + 1] * xt[1] + Bw.m[r * 4 + 2] * xt[2] + Bw.m[r * 4 + 3];
        db[r] = Bw.m[r * 4] * dt[0] + Bw.m[r * 4 + 1] * dt[1] + Bw.m[r * 4 + 2] * dt[2];
    }
}

// Deformer.forward for one canonical point (uv_deformer.py:31-38)
__device__ __forceinline__ void deform_point(const SceneDev& s, const GridDev& dg, const float* __restrict__ W0,
                                             const float* __restrict__ B0, const float* __restrict__ W1,
                                             const float* __restrict__ B1, const float* __restrict__ W2,
                                             const float* __restrict__ B2, const float* xb, float* resd) {
    float uvt[3];
    sample_volume_dev<2>(s.tuv, 0, xb[0], xb[1], xb[2], uvt);
    uvt[2] = s.frame_dim[0];
    float feat[19];
    grid_encode_concat<8, 2>(dg, uvt, feat);
    float h1[32], h2[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        float acc = B0[j];
#pragma unroll
        for (int i = 0; i < 19; ++i) acc = fmaf(W0[j * 19 + i], feat[i], acc);
        h1[j] = softplus_f(acc);
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        float acc = B1[j];
#pragma unroll
        for (int i = 0; i < 32; ++i) acc = fmaf(W1[j * 32 + i], h1[i], acc);
        h2[j] = softplus_f(acc);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float acc = B2[j];
#pragma unroll
        for (int i = 0; i < 32; ++i) acc = fmaf(W2[j * 32 + i], h2[i], acc);
        resd[j] = 0.05f * tanhf(acc);
    }
}

// ---- dense variant (invr_warp_deform): every (point, part) --------------------------------------
__global__ __launch_bounds__(WARP_BLOCK) void k_warp_dense(SceneDev s, GridDev dg, const float* __restrict__ A,
                                                           const float* __restrict__ big_A, const float* __restrict__ W0,
                                                           const float* __restrict__ B0, const float* __restrict__ W1,
                                                           const float* __restrict__ B1, const float* __restrict__ W2,
                                                           const float* __restrict__ B2, const float* pose_pts,
                                                           const float* pose_dirs, const float* bw, const uint8_t* flag,
                                                           int64_t n, float* tpose, float* tdirs, float* resd) {
    int64_t q = (int64_t)blockIdx.x * WARP_BLOCK + threadIdx.x;     // pair index = point*P + part
    if (q >= n * INVR_NUM_PARTS) return;
    int64_t i = q / INVR_NUM_PARTS;
    float b[INVR_NUM_JOINTS];
#pragma unroll
    for (int j = 0; j < INVR_NUM_JOINTS; ++j) b[j] = bw[q * INVR_NUM_JOINTS + j];
    float xb[3], db[3], r[3] = {0.f, 0.f, 0.f};
    warp_point(A, big_A, b, pose_pts + i * 3, pose_dirs + i * 3, xb, db);
    if (flag[q]) deform_point(s, dg, W0, B0, W1, B1, W2, B2, xb, r);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        tpose[q * 3 + c] = xb[c] + r[c];
        tdirs[q * 3 + c] = db[c];
        resd[q * 3 + c] = r[c];
    }
}

int launch_warp_deform_dense(const SceneDev& s, const GridDev& dg, const MlpDev& dm, const float* pose_pts,
                             const float* pose_dirs, const float* bw, const uint8_t* flag, int64_t n,
                             float* tpose, float* tdirs, float* resd, hipStream_t st) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_warp_dense, dim3((unsigned)cdiv(n * INVR_NUM_PARTS, WARP_BLOCK)), dim3(WARP_BLOCK), 0, st,
                       s, dg, s.A, s.big_A, dm.w[0], dm.b[0], dm.w[1], dm.b[1], dm.w[2], dm.b[2], pose_pts, pose_dirs, bw, flag, n,
                       tpose, tdirs, resd);
    INVR_LAUNCH_CHECK();
    return 0;
}

// ---- pipeline variant: walks the per-part pair lists --------------------------------------------
// Two kernels so that neither carries the other's live state (LBS: 24 blend weights + two 3x4
// matrices; deformer: 19 features + two 32-wide hidden layers): each fits well under 128 VGPRs and
// runs at >= 4 waves/SIMD instead of the 2 waves/SIMD (256 VGPRs) of the fused form.
// Per-vertex pre-blended matrices (built once per frame, k_vertex_mats): bw @ A with bw = sum_k w_k pbw[nn_k]
// is linear in the skinning rows, so  A_bw = sum_k w_k M_A[nn_k]  with  M_A[v] = sum_j pbw[v][j] A_j  (and the same
// for big_A).  A pair then needs 4 x 96 B gathers and 96 FMAs instead of 4 x 96 B gathers, the 24-wide blend and
// 576 FMAs against 576 scalar matrix entries (which the compiler could only keep by spilling SGPRs into VGPR
// lanes: 2.2 k v_readlane / v_writelane per pair made the old kernel VALU-bound at 0.32 ms).
__global__ __launch_bounds__(VMAT_BLOCK) void k_vertex_mats(SceneDev s, KnnIndex ix, const float* __restrict__ A,
                                                           const float* __restrict__ big_A) {
    vertex_mats_body(s, ix, A, big_A, (int)blockIdx.y, (int)(blockIdx.x * VMAT_BLOCK + threadIdx.x));
}

int launch_vertex_mats(const RenderArgs& a, const Workspace& w, hipStream_t st) {
    const int m = a.scene.M < w.knn.mpad ? a.scene.M : w.knn.mpad;
    hipLaunchKernelGGL(k_vertex_mats, dim3((unsigned)cdiv(m, VMAT_BLOCK), INVR_NUM_PARTS), dim3(VMAT_BLOCK), 0, st, a.scene, w.knn,
                       a.scene.A, a.scene.big_A);
    INVR_LAUNCH_CHECK();
    return 0;
}

// canonical point / direction of pair i of part p's list
__device__ __forceinline__ void warp_pair(const RenderArgs& a, const Workspace& w, const float4* __restrict__ vm, const int p, const int64_t i,
                                          float* xb, float* db) {
    const int slot = w.l_slot[p][i];
    const int4 nn = reinterpret_cast<const int4*>(w.l_nn[p])[slot];       // (stored per survivor slot by k_knn_pairs)
    const float4 wt = reinterpret_cast<const float4*>(w.l_w[p])[slot];
    const float4* r0 = vm + (int64_t)nn.x * 6;
    const float4* r1 = vm + (int64_t)nn.y * 6;
    const float4* r2 = vm + (int64_t)nn.z * 6;
    const float4* r3 = vm + (int64_t)nn.w * 6;
    Mat34 Aw, Bw;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const float4 v0 = r0[j], v1 = r1[j], v2 = r2[j], v3 = r3[j];
        float* o = j < 3 ? Aw.m + j * 4 : Bw.m + (j - 3) * 4;
        o[0] = fmaf(wt.w, v3.x, fmaf(wt.z, v2.x, fmaf(wt.y, v1.x, wt.x * v0.x)));
        o[1] = fmaf(wt.w, v3.y, fmaf(wt.z, v2.y, fmaf(wt.y, v1.y, wt.x * v0.y)));
        o[2] = fmaf(wt.w, v3.z, fmaf(wt.z, v2.z, fmaf(wt.y, v1.z, wt.x * v0.z)));
        o[3] = fmaf(wt.w, v3.w, fmaf(wt.z, v2.w, fmaf(wt.y, v1.w, wt.x * v0.w)));
    }
    float pp[3], pd[3];
    sample_pose_point(a, w.active_idx[slot], pp[0], pp[1], pp[2], nullptr, pd);
    warp_with_mats(Aw, Bw, pp, pd, xb, db);
    if (!a.scene.tpose_viewdir) {                       // cfg.tpose_viewdir False: world view dir
        int64_t ray = w.active_idx[slot] / a.S;
        const float* vd = a.wpts ? a.wdirs : a.ray_d;
        db[0] = vd[ray * 3]; db[1] = vd[ray * 3 + 1]; db[2] = vd[ray * 3 + 2];
    }
}

__global__ __launch_bounds__(WARP_BLOCK) void k_warp_pairs(RenderArgs a, Workspace w) {
    const int p = blockIdx.y;
    const int cnt = w.counters[CNT_PAIRS + p];
    const float4* __restrict__ vm = w.knn.vmat + (int64_t)p * w.knn.mpad * 6;
    for (int64_t i = (int64_t)blockIdx.x * WARP_BLOCK + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * WARP_BLOCK) {
        float xb[3], db[3];
        warp_pair(a, w, vm, p, i, xb, db);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            w.l_x[p][c * w.lcap + i] = xb[c];                // init_bigpose
            w.l_d[p][c * w.lcap + i] = db[c];
        }
    }
}


// Residual deformer of the pair lists on the fp32 matrix cores (same D^T = W . X^T orientation as
// k_part_mlp: 16 pairs = the N columns of v_mfma_f32_16x16x4_f32, accumulators of one layer are the B
// operands of the next).  A wave handles 64 pairs per iteration:
//   1. lane j: canonical point of pair j, trilinear (u,v) from the UV volume
//   2. four 16-pair tiles; in tile cb lane (g = lane>>4, col = lane&15) encodes levels 2g and 2g+1 of
//      pair cb*16+col (16 float2 gathers per lane, no index math duplicated between lanes) — the K order
//      of layer 1 is permuted to match: k-slot (s<4, g) = feature 3 + 2*(2g + s/2) + s%2, (4, g) = uvt[g]
//   3. 19(20) -> 32 -> 32 on MFMA (10 + 16 instructions per tile), 32 -> 3 head as VALU dots, 0.05*tanh
//   4. lane j = cb*16+col takes the result of "its" pair back and writes tpose / resd, coalesced.
typedef float dfx4 __attribute__((ext_vector_type(4)));
#define DF_BLOCK 256
#define DF_O_W1 0                       // 5 k-steps * 2 m-tiles * 64 lanes
#define DF_O_W2 (DF_O_W1 + 5 * 2 * 64)  // 8 * 2 * 64
#define DF_O_B1 (DF_O_W2 + 8 * 2 * 64)  // 32
#define DF_O_B2 (DF_O_B1 + 32)          // 32
#define DF_O_V (DF_O_B2 + 32)           // 3 * 32, slot order [c][g*8 + mt*4 + r]
#define DF_O_B3 (DF_O_V + 96)           // 3 (+1)
#define DF_LDS (DF_O_B3 + 4)

__device__ __forceinline__ int df_col(int s, int g) { return s < 4 ? 3 + 2 * (2 * g + (s >> 1)) + (s & 1) : (g < 3 ? g : -1); }

struct LaneLevel2 { const float2* tab; int res; float cell; bool hashed; };

// one level of the 8x2 deformer grid for this lane's point: same arithmetic and accumulation order as
// grid_level_lookup + grid_encode_concat (part_base_embedder.py:115-159)
__device__ __forceinline__ void lane_level_f2(const GridDev& dg, const LaneLevel2& L, float x, float y, float z, float& f0, float& f1) {
    int c0x, c1x, c0y, c1y, c0z, c1z;
    float tx, ty, tz;
    level_corners(x, L.cell, L.res, c0x, c1x, tx);
    level_corners(y, L.cell, L.res, c0y, c1y, ty);
    level_corners(z, L.cell, L.res, c0z, c1z, tz);
    float2 v[8];
    if (L.hashed) {
        const uint64_t hx[2] = {(uint64_t)(uint32_t)c0x, (uint64_t)(uint32_t)c1x};
        const uint64_t hy[2] = {(uint64_t)(uint32_t)c0y * HASH_P1, (uint64_t)(uint32_t)c1y * HASH_P1};
        const uint64_t hz[2] = {(uint64_t)(uint32_t)c0z * HASH_P2, (uint64_t)(uint32_t)c1z * HASH_P2};
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = L.tab[grid_hash_mod(hx[k >> 2] ^ hy[(k >> 1) & 1] ^ hz[k & 1], dg)];
    } else {
        const unsigned ures = (unsigned)L.res;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            v[k] = L.tab[((unsigned)((k & 4) ? c1x : c0x) * ures + (unsigned)((k & 2) ? c1y : c0y)) * ures + (unsigned)((k & 1) ? c1z : c0z)];
    }
    const float ux = 1.0f - tx, uy = 1.0f - ty, uz = 1.0f - tz;
    f0 = 0.0f; f1 = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float wk = ((k & 4) ? tx : ux) * ((k & 2) ? ty : uy) * ((k & 1) ? tz : uz);
        f0 = fmaf(wk, v[k].x, f0);
        f1 = fmaf(wk, v[k].y, f1);
    }
}

template <int DF_CB>                    // 16-pair tiles in flight per wave (register budget)
__global__ __launch_bounds__(DF_BLOCK) void k_deform_pairs(RenderArgs a, Workspace w, GridDev dg,
                                                           const float* __restrict__ W0, const float* __restrict__ B0,
                                                           const float* __restrict__ W1, const float* __restrict__ B1,
                                                           const float* __restrict__ W2, const float* __restrict__ B2) {
    __shared__ float lds[DF_LDS];
    const int p = blockIdx.y;
    const int cnt = w.counters[CNT_PAIRS + p];
    if ((int64_t)blockIdx.x * DF_BLOCK >= cnt) return;
    for (int t = threadIdx.x; t < 5 * 2 * 64; t += DF_BLOCK) {
        const int ln = t & 63, mt = (t >> 6) & 1, s = t >> 7, g = ln >> 4, i = ln & 15, col = df_col(s, g);
        lds[DF_O_W1 + t] = col >= 0 ? W0[(16 * mt + i) * 19 + col] : 0.0f;
    }
    for (int t = threadIdx.x; t < 8 * 2 * 64; t += DF_BLOCK) {
        const int ln = t & 63, mt = (t >> 6) & 1, s = t >> 7, g = ln >> 4, i = ln & 15;
        lds[DF_O_W2 + t] = W1[(16 * mt + i) * 32 + 16 * (s >> 2) + 4 * g + (s & 3)];
    }
    if (threadIdx.x < 32) {
        const int t = threadIdx.x, g = t >> 3, u = t & 7, hc = 16 * (u >> 2) + 4 * g + (u & 3);
        lds[DF_O_B1 + t] = B0[t];
        lds[DF_O_B2 + t] = B1[t];
#pragma unroll
        for (int c = 0; c < 3; ++c) lds[DF_O_V + c * 32 + t] = W2[c * 32 + hc];
        if (t < 3) lds[DF_O_B3 + t] = B2[t];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, col = lane & 15;
    LaneLevel2 LA, LB;                   // levels 2g and 2g+1
    LA.tab = LB.tab = nullptr; LA.res = LB.res = 2; LA.cell = LB.cell = 1.0f; LA.hashed = LB.hashed = false;
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        LaneLevel2 t;
        t.hashed = l >= dg.start_hash;
        t.res = dg.res[l];
        t.cell = dg.cell[l];
        const float* tb = dg.separate_dense ? (t.hashed ? dg.hash + (int64_t)(l - dg.start_hash) * dg.T * 2 : dg.dense + dg.dense_off[l] * 2)
                                            : dg.hash + (int64_t)l * dg.T * 2;
        t.tab = reinterpret_cast<const float2*>(tb);
        if (l == 2 * g) LA = t;
        if (l == 2 * g + 1) LB = t;
    }
    const float gb0 = dg.bounds[0], gb1 = dg.bounds[1], gb2 = dg.bounds[2];
    const float ge0 = dg.bounds[3] - gb0, ge1 = dg.bounds[4] - gb1, ge2 = dg.bounds[5] - gb2;
    const float tn = (a.scene.frame_dim[0] - gb2) / ge2;                 // uvt[2] = frame_dim, normalised (:112)
    float* lx = w.l_x[p];
    float* lr = w.l_r[p];

    for (int64_t base = (int64_t)blockIdx.x * DF_BLOCK + wv * 64; base < cnt; base += (int64_t)gridDim.x * DF_BLOCK) {
        const int64_t i = min(base + lane, (int64_t)cnt - 1);
        float xb[3], uv[2];
#pragma unroll
        for (int c = 0; c < 3; ++c) xb[c] = lx[c * w.lcap + i];
        sample_volume_dev<2>(a.scene.tuv, 0, xb[0], xb[1], xb[2], uv);
        const float un = (uv[0] - gb0) / ge0, vn = (uv[1] - gb1) / ge1;
        float r3[3] = {0.f, 0.f, 0.f};
#pragma unroll 1
        for (int half = 0; half < 4 / DF_CB; ++half) {
        float eb[DF_CB][5];
#pragma unroll
        for (int cb = 0; cb < DF_CB; ++cb) {
            const int src = (half * DF_CB + cb) * 16 + col;
            const float uu = __shfl(un, src), vv = __shfl(vn, src);
            lane_level_f2(dg, LA, uu, vv, tn, eb[cb][0], eb[cb][1]);
            lane_level_f2(dg, LB, uu, vv, tn, eb[cb][2], eb[cb][3]);
            eb[cb][4] = g == 0 ? uu : (g == 1 ? vv : (g == 2 ? tn : 0.0f));
        }
        // ---- layer 1: 20 -> 32
        dfx4 h[DF_CB][2];
#pragma unroll
        for (int cb = 0; cb < DF_CB; ++cb)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) h[cb][mt][r] = lds[DF_O_B1 + 16 * mt + 4 * g + r];
#pragma unroll
        for (int s = 0; s < 5; ++s)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const float aw = lds[DF_O_W1 + (s * 2 + mt) * 64 + lane];
#pragma unroll
                for (int cb = 0; cb < DF_CB; ++cb) h[cb][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw, eb[cb][s], h[cb][mt], 0, 0, 0);
            }
#pragma unroll
        for (int cb = 0; cb < DF_CB; ++cb)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) h[cb][mt][r] = softplus_f(h[cb][mt][r]);
        // ---- layer 2: 32 -> 32
        dfx4 h2[DF_CB][2];
#pragma unroll
        for (int cb = 0; cb < DF_CB; ++cb)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) h2[cb][mt][r] = lds[DF_O_B2 + 16 * mt + 4 * g + r];
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const float aw = lds[DF_O_W2 + (s * 2 + mt) * 64 + lane];
#pragma unroll
                for (int cb = 0; cb < DF_CB; ++cb) h2[cb][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw, h[cb][s >> 2][s & 3], h2[cb][mt], 0, 0, 0);
            }
        // ---- head 32 -> 3, 0.05 * tanh; lane j = cb*16+col keeps the result of pair j
#pragma unroll
        for (int cb = 0; cb < DF_CB; ++cb) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) h2[cb][mt][r] = softplus_f(h2[cb][mt][r]);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float acc = 0.0f;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc = fmaf(lds[DF_O_V + c * 32 + g * 8 + mt * 4 + r], h2[cb][mt][r], acc);
                acc += __shfl_xor(acc, 16);
                acc += __shfl_xor(acc, 32);
                if (g == half * DF_CB + cb) r3[c] = acc + lds[DF_O_B3 + c];
            }
        }
        }
        if (base + lane < cnt) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float r = 0.05f * tanhf(r3[c]);
                lx[c * w.lcap + i] = xb[c] + r;                  // tpose = init_bigpose + resd (:111)
                lr[c * w.lcap + i] = r;
            }
        }
    }
}

// ---- deformer with per-frame t-slices ---------------------------------------------------------------
// The deformer's third grid coordinate is frame_dim (uv_deformer.py:33-34): ONE value for the whole call.
// So per level the z corner pair and its weight are the same for every point, and the 3-D grid collapses
// to a 2-D (u,v) table  S_l[cx][cy] = (1-tz) row(cx,cy,c0z) + tz row(cx,cy,c1z)  (hashed levels included:
// the slice of a hashed level is materialised densely).  sum_l res_l^2 = 2959 entries (24 KB) for the
// reference's 8 levels — built once per call by k_deform_slice and held in LDS, so the 64 L1-line gathers
// per pair that bounded the kernel become 32 ds_read_b64.  (u,v) index math stays the reference's.
__global__ void k_deform_slice(GridDev dg, DfSliceInfo si, const float* __restrict__ frame_dim, float2* __restrict__ out) {
    deform_slice_body(dg, si, frame_dim, out, (int)(blockIdx.x * blockDim.x + threadIdx.x));
}

struct LaneSlice { int off, res; float cell; };

__device__ __forceinline__ void lane_level_slice(const float2* S, const LaneSlice& L, float x, float y, float& f0, float& f1) {
    int c0x, c1x, c0y, c1y;
    float tx, ty;
    // (the hardware division on purpose: the reciprocal form of common.h:div_exact measured SLOWER in this kernel, 0.46 -> 0.50 ms
    // for the stage — its branch breaks the MFMA / VALU interleave of the column blocks; round 5: the range test hoisted to ONE
    // wave-uniform branch per tile with both forms of the tile body, 170 instead of 168 registers = 2 instead of 3 waves per SIMD:
    // stage 0.495 -> 0.513 ms, gpurun_out/r5l — not kept)
    level_corners(x, L.cell, L.res, c0x, c1x, tx);
    level_corners(y, L.cell, L.res, c0y, c1y, ty);
    // (24-bit multiplies: 0 <= c < res <= the slice budget's 64 — v_mul_lo_u32 is a quarter-rate instruction)
    const float2* row0 = S + L.off + (int)__umul24((unsigned)c0x, (unsigned)L.res);
    const float2* row1 = S + L.off + (int)__umul24((unsigned)c1x, (unsigned)L.res);
    const float2 s00 = row0[c0y], s01 = row0[c1y], s10 = row1[c0y], s11 = row1[c1y];
    const float ux = 1.0f - tx, uy = 1.0f - ty;
    const float w00 = ux * uy, w01 = ux * ty, w10 = tx * uy, w11 = tx * ty;
    f0 = fmaf(w11, s11.x, fmaf(w10, s10.x, fmaf(w01, s01.x, w00 * s00.x)));
    f1 = fmaf(w11, s11.y, fmaf(w10, s10.y, fmaf(w01, s01.y, w00 * s00.y)));
}

// The hidden activations of this kernel are kept in the log2 domain, u = log2(1 + exp2(z log2e)) = softplus(z) / ln2, with the two
// scale factors folded into the staged weights as in the part MLPs (mlp_common.h): a layer that feeds a Softplus is scaled by log2e
// (weights and bias), a layer that consumes Softplus outputs by ln2 — for the hidden-to-hidden layer the two cancel, only its bias is
// scaled.  {min, exp2, add, log2} = 4 instructions per value instead of the 7 of softplus_f; 64 values per pair.
__device__ __forceinline__ float softplus_log2(float a) { return log2_raw(1.0f + exp2_raw(fminf(a, 126.0f))); }

#ifndef DF_WPE
#define DF_WPE 3
#endif
template <int DF_CB, bool VSMALL>        // VSMALL: the UV volume qualifies for 24-bit index math (volume_is_small, common.h)
__global__ __launch_bounds__(DF_BLOCK) __attribute__((amdgpu_waves_per_eu(DF_WPE, DF_WPE))) void k_deform_pairs_slice(RenderArgs a, Workspace w, GridDev dg, DfSliceInfo si,
                                                                 const float* __restrict__ W0, const float* __restrict__ B0,
                                                                 const float* __restrict__ W1, const float* __restrict__ B1,
                                                                 const float* __restrict__ W2, const float* __restrict__ B2) {
    extern __shared__ __attribute__((aligned(16))) float lds[];          // [MLP weights | slices]
    const int p = blockIdx.y;
    const int cnt = w.counters[CNT_PAIRS + p];
    if ((int64_t)blockIdx.x * DF_BLOCK >= cnt) return;
    float2* S = reinterpret_cast<float2*>(lds + DF_LDS);
    for (int t = threadIdx.x; t < si.off[8]; t += DF_BLOCK) S[t] = w.dslice[t];
    for (int t = threadIdx.x; t < 5 * 2 * 64; t += DF_BLOCK) {
        const int ln = t & 63, mt = (t >> 6) & 1, s = t >> 7, g = ln >> 4, i = ln & 15, col = df_col(s, g);
        lds[DF_O_W1 + t] = col >= 0 ? W0[(16 * mt + i) * 19 + col] * INVR_LOG2E : 0.0f;      // log2-domain activations, see below
    }
    for (int t = threadIdx.x; t < 8 * 2 * 64; t += DF_BLOCK) {
        const int ln = t & 63, mt = (t >> 6) & 1, s = t >> 7, g = ln >> 4, i = ln & 15;
        lds[DF_O_W2 + t] = W1[(16 * mt + i) * 32 + 16 * (s >> 2) + 4 * g + (s & 3)];
    }
    if (threadIdx.x < 32) {
        const int t = threadIdx.x, g = t >> 3, u = t & 7, hc = 16 * (u >> 2) + 4 * g + (u & 3);
        lds[DF_O_B1 + t] = B0[t] * INVR_LOG2E;
        lds[DF_O_B2 + t] = B1[t] * INVR_LOG2E;                    // (layer 2's weights: ln2 * log2e = 1, unscaled)
#pragma unroll
        for (int c = 0; c < 3; ++c) lds[DF_O_V + c * 32 + t] = W2[c * 32 + hc] * INVR_LN2;
        if (t < 3) lds[DF_O_B3 + t] = B2[t];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, col = lane & 15;
    LaneSlice LA = {0, 2, 1.0f}, LB = {0, 2, 1.0f};             // levels 2g and 2g+1
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        const LaneSlice t = {si.off[l], dg.res[l], dg.cell[l]};
        if (l == 2 * g) LA = t;
        if (l == 2 * g + 1) LB = t;
    }
    const float gb0 = dg.bounds[0], gb1 = dg.bounds[1], gb2 = dg.bounds[2];
    const float ge0 = dg.bounds[3] - gb0, ge1 = dg.bounds[4] - gb1, ge2 = dg.bounds[5] - gb2;
    const float tn = (a.scene.frame_dim[0] - gb2) / ge2;
    float* lx = w.l_x[p];
    float* lr = w.l_r[p];

    for (int64_t base = (int64_t)blockIdx.x * DF_BLOCK + wv * 64; base < cnt; base += (int64_t)gridDim.x * DF_BLOCK) {
        const int64_t i = min(base + lane, (int64_t)cnt - 1);
        float xb[3], uv[2];
#pragma unroll
        for (int c = 0; c < 3; ++c) xb[c] = lx[c * w.lcap + i];
        sample_volume_dev<2, VSMALL>(a.scene.tuv, 0, xb[0], xb[1], xb[2], uv);
        const float un = (uv[0] - gb0) / ge0, vn = (uv[1] - gb1) / ge1;
        float r3[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int half = 0; half < 4 / DF_CB; ++half) {
            float eb[DF_CB][5];
#pragma unroll
            for (int cb = 0; cb < DF_CB; ++cb) {
                const int src = (half * DF_CB + cb) * 16 + col;
                const float uu = __shfl(un, src), vv = __shfl(vn, src);
                lane_level_slice(S, LA, uu, vv, eb[cb][0], eb[cb][1]);
                lane_level_slice(S, LB, uu, vv, eb[cb][2], eb[cb][3]);
                eb[cb][4] = g == 0 ? uu : (g == 1 ? vv : (g == 2 ? tn : 0.0f));
            }
            dfx4 h[DF_CB][2];
#pragma unroll
            for (int cb = 0; cb < DF_CB; ++cb)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[cb][mt][r] = lds[DF_O_B1 + 16 * mt + 4 * g + r];
#pragma unroll
            for (int s = 0; s < 5; ++s)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const float aw = lds[DF_O_W1 + (s * 2 + mt) * 64 + lane];
#pragma unroll
                    for (int cb = 0; cb < DF_CB; ++cb) h[cb][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw, eb[cb][s], h[cb][mt], 0, 0, 0);
                }
#pragma unroll
            for (int cb = 0; cb < DF_CB; ++cb)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[cb][mt][r] = softplus_log2(h[cb][mt][r]);
            dfx4 h2[DF_CB][2];
#pragma unroll
            for (int cb = 0; cb < DF_CB; ++cb)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h2[cb][mt][r] = lds[DF_O_B2 + 16 * mt + 4 * g + r];
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const float aw = lds[DF_O_W2 + (s * 2 + mt) * 64 + lane];
#pragma unroll
                    for (int cb = 0; cb < DF_CB; ++cb) h2[cb][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw, h[cb][s >> 2][s & 3], h2[cb][mt], 0, 0, 0);
                }
#pragma unroll
            for (int cb = 0; cb < DF_CB; ++cb) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h2[cb][mt][r] = softplus_log2(h2[cb][mt][r]);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float acc = 0.0f;
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc = fmaf(lds[DF_O_V + c * 32 + g * 8 + mt * 4 + r], h2[cb][mt][r], acc);
                    acc += __shfl_xor(acc, 16);
                    acc += __shfl_xor(acc, 32);
                    if (g == half * DF_CB + cb) r3[c] = acc + lds[DF_O_B3 + c];
                }
            }
        }
        if (base + lane < cnt) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float r = 0.05f * tanhf(r3[c]);
                lx[c * w.lcap + i] = xb[c] + r;                  // tpose = init_bigpose + resd (:111)
                lr[c * w.lcap + i] = r;
            }
        }
    }
}

// depends on the grid and frame_dim only (not on the pair lists): launched on the side stream beside the KNN
int launch_deform_slice(const RenderArgs& a, const Workspace& w, const GridDev& dg, hipStream_t st) {
    DfSliceInfo si;
    if (!deform_slices_fit(dg, si, deform_cb())) return 0;
    hipLaunchKernelGGL(k_deform_slice, dim3((unsigned)cdiv(si.off[8], 256)), dim3(256), 0, st, dg, si, a.scene.frame_dim, w.dslice);
    INVR_LAUNCH_CHECK();
    return 0;
}

// ---- backward of the deformer's grid through the same t-slices -------------------------------------------------------------
// feature (l, f) of a point = sum over its four (u, v) corners of w_uv * S_l[corner][f],  S_l = (1-tz) row(., ., c0z) + tz row(., ., c1z)
// with ONE tz per call.  So the table gradient is the transposed two-step: every workgroup scatter-adds w_uv * g into a 24 KB LDS
// image of the slices (2959 float2 entries for the reference's 8 levels) over its share of the points, then sends each non-zero
// entry to its two table rows with the weights (1-tz, tz).  The generic level-outer backward (k_grid_encode_bwd_rt) zeroed and
// flushed a 132 KB LDS image of a whole 3-D level slice per level and workgroup: 242 us per iteration for ~1e5 points; this one
// 125 us with one workgroup per CU — bound by the LDS atomics of the coarse levels (all points of a workgroup on 16..50 entries;
// reducing wave-uniform cells with shuffles first did not help: the UV coordinates of neighbouring points are unrelated).
#define DSB_BLOCK 1024
__global__ __launch_bounds__(DSB_BLOCK) void k_deform_slice_bwd(GridDev dg, DfSliceInfo si, const float* __restrict__ frame_dim,
                                                                const float* __restrict__ uvt, const float* __restrict__ gfeat,
                                                                int64_t n_host, const int32_t* __restrict__ count, int feat_dim,
                                                                float* __restrict__ g_dense, float* __restrict__ g_hash) {
    extern __shared__ __attribute__((aligned(16))) float2 gS[];
    const int64_t n = count ? (int64_t)*count : n_host;
    const int total = si.off[dg.L];
    for (int e = threadIdx.x; e < total; e += DSB_BLOCK) gS[e] = make_float2(0.f, 0.f);
    __syncthreads();
    const float gb0 = dg.bounds[0], gb1 = dg.bounds[1];
    const float ge0 = dg.bounds[3] - gb0, ge1 = dg.bounds[4] - gb1;
    const int off = dg.include_input ? 3 : 0;
    for (int64_t i = (int64_t)blockIdx.x * DSB_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * DSB_BLOCK) {
        const float un = (uvt[i * 3] - gb0) / ge0, vn = (uvt[i * 3 + 1] - gb1) / ge1;          // :112
        const float* go = gfeat + i * feat_dim + off;
#pragma unroll 2
        for (int l = 0; l < 8; ++l) {
            const float g0 = go[2 * l], g1 = go[2 * l + 1];
            const int res = dg.res[l];
            int c0x, c1x, c0y, c1y;
            float tx, ty;
            level_corners(un, dg.cell[l], res, c0x, c1x, tx);
            level_corners(vn, dg.cell[l], res, c0y, c1y, ty);
            const float ux = 1.0f - tx, uy = 1.0f - ty;
            float* r0 = reinterpret_cast<float*>(gS + si.off[l] + c0x * res);
            float* r1 = reinterpret_cast<float*>(gS + si.off[l] + c1x * res);
            const float w00 = ux * uy, w01 = ux * ty, w10 = tx * uy, w11 = tx * ty;
            atomicAdd(r0 + 2 * c0y, w00 * g0); atomicAdd(r0 + 2 * c0y + 1, w00 * g1);
            atomicAdd(r0 + 2 * c1y, w01 * g0); atomicAdd(r0 + 2 * c1y + 1, w01 * g1);
            atomicAdd(r1 + 2 * c0y, w10 * g0); atomicAdd(r1 + 2 * c0y + 1, w10 * g1);
            atomicAdd(r1 + 2 * c1y, w11 * g0); atomicAdd(r1 + 2 * c1y + 1, w11 * g1);
        }
    }
    __syncthreads();
    const float tn = (frame_dim[0] - dg.bounds[2]) / (dg.bounds[5] - dg.bounds[2]);
    for (int e = threadIdx.x; e < total; e += DSB_BLOCK) {
        const float2 g = gS[e];
        if (g.x == 0.0f && g.y == 0.0f) continue;
        int l = 0;
        while (e >= si.off[l + 1]) ++l;
        const int res = dg.res[l];
        int c0z, c1z;
        float tz;
        level_corners(tn, dg.cell[l], res, c0z, c1z, tz);
        const int idx = e - si.off[l], cx = idx / res, cy = idx - cx * res;
        const bool hashed = l >= dg.start_hash;
        float* tb = dg.separate_dense ? (hashed ? g_hash + (int64_t)(l - dg.start_hash) * dg.T * 2 : g_dense + dg.dense_off[l] * 2)
                                      : g_hash + (int64_t)l * dg.T * 2;
        unsigned r0, r1;
        if (hashed) {
            const uint64_t hxy = (uint64_t)(uint32_t)cx ^ ((uint64_t)(uint32_t)cy * HASH_P1);
            r0 = grid_hash_mod(hxy ^ ((uint64_t)(uint32_t)c0z * HASH_P2), dg);
            r1 = grid_hash_mod(hxy ^ ((uint64_t)(uint32_t)c1z * HASH_P2), dg);
        } else {
            r0 = ((unsigned)cx * (unsigned)res + (unsigned)cy) * (unsigned)res + (unsigned)c0z;
            r1 = ((unsigned)cx * (unsigned)res + (unsigned)cy) * (unsigned)res + (unsigned)c1z;
        }
        const float uz = 1.0f - tz;
        unsafeAtomicAdd(tb + (int64_t)r0 * 2, uz * g.x); unsafeAtomicAdd(tb + (int64_t)r0 * 2 + 1, uz * g.y);
        unsafeAtomicAdd(tb + (int64_t)r1 * 2, tz * g.x); unsafeAtomicAdd(tb + (int64_t)r1 * 2 + 1, tz * g.y);
    }
}

// -> 0 launched, 1 error, -1 not applicable (the slices do not fit / not the 8 x 2 concat grid): use the generic backward
int launch_deform_slice_bwd(const GridDev& dg, const float* frame_dim, const float* uvt, const float* gfeat, int64_t n_max,
                            const int32_t* count, float* g_dense, float* g_hash, hipStream_t st) {
    DfSliceInfo si;
    if (!deform_slices_fit(dg, si, deform_cb()) || dg.F != 2 || dg.sum || !dg.include_input) return -1;
    if (n_max == 0) return 0;
    const size_t lds_bytes = (size_t)si.off[8] * sizeof(float2);
    const int64_t tiles = cdiv(n_max, DSB_BLOCK);
    static const int gmax = getenv("INVR_DSB_GRID") ? atoi(getenv("INVR_DSB_GRID")) : 256;
    const unsigned grid = (unsigned)(tiles < gmax ? (tiles > 0 ? tiles : 1) : gmax);
    hipLaunchKernelGGL(k_deform_slice_bwd, dim3(grid), dim3(DSB_BLOCK), lds_bytes, st, dg, si, frame_dim, uvt, gfeat, n_max, count,
                       3 + 16, g_dense, g_hash);
    INVR_LAUNCH_CHECK();
    return 0;
}

int launch_warp_pairs(const RenderArgs& a, const Workspace& w, const GridDev& dg, const MlpDev& dm, hipStream_t st) {
    int64_t tiles = cdiv(w.lcap, WARP_BLOCK);
    unsigned gx = (unsigned)(tiles < 1024 ? (tiles > 0 ? tiles : 1) : 1024);
    hipLaunchKernelGGL(k_warp_pairs, dim3(gx, INVR_NUM_PARTS), dim3(WARP_BLOCK), 0, st, a, w);
    INVR_LAUNCH_CHECK();
    int64_t dtiles = cdiv(w.lcap, DF_BLOCK);
    unsigned dgx = (unsigned)(dtiles < 512 ? (dtiles > 0 ? dtiles : 1) : 512);
    const int cbv = deform_cb();
    DfSliceInfo si;
    if (deform_slices_fit(dg, si, cbv)) {                 // slices built by launch_deform_slice
        const size_t lds_bytes = (size_t)DF_LDS * sizeof(float) + (size_t)si.off[8] * sizeof(float2);
        const bool vs = volume_is_small(a.scene.tuv);
        auto kern = cbv == 1 ? (vs ? k_deform_pairs_slice<1, true> : k_deform_pairs_slice<1, false>)
                             : (vs ? k_deform_pairs_slice<2, true> : k_deform_pairs_slice<2, false>);
        hipLaunchKernelGGL(kern, dim3(dgx, INVR_NUM_PARTS), dim3(DF_BLOCK), lds_bytes, st, a, w, dg, si, dm.w[0], dm.b[0],
                           dm.w[1], dm.b[1], dm.w[2], dm.b[2]);
        INVR_LAUNCH_CHECK();
        return 0;
    }
    if (cbv % 10 == 1)
        hipLaunchKernelGGL(k_deform_pairs<1>, dim3(dgx, INVR_NUM_PARTS), dim3(DF_BLOCK), 0, st, a, w, dg, dm.w[0], dm.b[0], dm.w[1],
                           dm.b[1], dm.w[2], dm.b[2]);
    else
        hipLaunchKernelGGL(k_deform_pairs<2>, dim3(dgx, INVR_NUM_PARTS), dim3(DF_BLOCK), 0, st, a, w, dg, dm.w[0], dm.b[0], dm.w[1],
                           dm.b[1], dm.w[2], dm.b[2]);
    INVR_LAUNCH_CHECK();
    return 0;
}

// ---- stand-alone deformer on arbitrary canonical points (invr_deform_fwd) ------------------------
__global__ __launch_bounds__(WARP_BLOCK) void k_deform_points(SceneDev s, GridDev dg, const float* __restrict__ W0,
                                                              const float* __restrict__ B0, const float* __restrict__ W1,
                                                              const float* __restrict__ B1, const float* __restrict__ W2,
                                                              const float* __restrict__ B2, const float* __restrict__ pts,
                                                              int64_t n, float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * WARP_BLOCK + threadIdx.x;
    if (i >= n) return;
    float xb[3] = {pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2]}, r[3];
    deform_point(s, dg, W0, B0, W1, B1, W2, B2, xb, r);
    out[i * 3] = r[0]; out[i * 3 + 1] = r[1]; out[i * 3 + 2] = r[2];
}

int launch_deform_points(const SceneDev& s, const GridDev& dg, const MlpDev& dm, const float* pts, int64_t n, float* out, hipStream_t st) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_deform_points, dim3((unsigned)cdiv(n, WARP_BLOCK)), dim3(WARP_BLOCK), 0, st, s, dg, dm.w[0], dm.b[0],
                       dm.w[1], dm.b[1], dm.w[2], dm.b[2], pts, n, out);
    INVR_LAUNCH_CHECK();
    return 0;
}
