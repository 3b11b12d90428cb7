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
#include "pipeline.h"
#include "grid_generic.h"

#define WARP_BLOCK 128

struct Mat34 { float m[12]; };   // rows 0..2 of a 4x4: [R | t]

__device__ __forceinline__ void blend_mats(const float* __restrict__ A, const float* bw, Mat34& o) {
#pragma unroll
    for (int e = 0; e < 12; ++e) o.m[e] = 0.0f;
#pragma unroll
    for (int j = 0; j < INVR_NUM_JOINTS; ++j)
#pragma unroll
        for (int e = 0; e < 12; ++e) o.m[e] = fmaf(bw[j], A[j * 16 + e], o.m[e]);     // bw @ A.view(24,16)
}

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

__device__ __forceinline__ void warp_point(const float* __restrict__ A, const float* __restrict__ big_A, const float* bw,
                                           const float* pp, const float* pd, float* xb, float* db) {
    Mat34 Aw, Bw;
    blend_mats(A, bw, Aw);
    blend_mats(big_A, bw, Bw);
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
__global__ __launch_bounds__(WARP_BLOCK) void k_warp_pairs(RenderArgs a, Workspace w, const float* __restrict__ A,
                                                           const float* __restrict__ big_A) {
    const int p = blockIdx.y;
    const int cnt = w.counters[CNT_PAIRS + p];
    const float* __restrict__ pb = a.scene.part_pbw + (int64_t)p * a.scene.M * INVR_NUM_JOINTS;
    for (int64_t i = (int64_t)blockIdx.x * WARP_BLOCK + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * WARP_BLOCK) {
        const int slot = w.l_slot[p][i];
        const int4 nn = reinterpret_cast<const int4*>(w.l_nn[p])[i];
        const float4 wt = reinterpret_cast<const float4*>(w.l_w[p])[i];
        float b[INVR_NUM_JOINTS];
        const float4* r0 = reinterpret_cast<const float4*>(pb + (int64_t)nn.x * INVR_NUM_JOINTS);
        const float4* r1 = reinterpret_cast<const float4*>(pb + (int64_t)nn.y * INVR_NUM_JOINTS);
        const float4* r2 = reinterpret_cast<const float4*>(pb + (int64_t)nn.z * INVR_NUM_JOINTS);
        const float4* r3 = reinterpret_cast<const float4*>(pb + (int64_t)nn.w * INVR_NUM_JOINTS);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float4 v0 = r0[j], v1 = r1[j], v2 = r2[j], v3 = r3[j];
            b[j * 4 + 0] = v0.x * wt.x + v1.x * wt.y + v2.x * wt.z + v3.x * wt.w;     // einsum('ijkl,ijk->ijl')
            b[j * 4 + 1] = v0.y * wt.x + v1.y * wt.y + v2.y * wt.z + v3.y * wt.w;
            b[j * 4 + 2] = v0.z * wt.x + v1.z * wt.y + v2.z * wt.z + v3.z * wt.w;
            b[j * 4 + 3] = v0.w * wt.x + v1.w * wt.y + v2.w * wt.z + v3.w * wt.w;
        }
        float pp[3], pd[3], xb[3], db[3];
        sample_pose_point(a, w.active_idx[slot], pp[0], pp[1], pp[2], nullptr, pd);
        warp_point(A, big_A, b, pp, pd, xb, db);
        if (!a.scene.tpose_viewdir) {                       // cfg.tpose_viewdir False: world view dir
            int64_t ray = w.active_idx[slot] / a.S;
            const float* vd = a.wpts ? a.wdirs : a.ray_d;
            db[0] = vd[ray * 3]; db[1] = vd[ray * 3 + 1]; db[2] = vd[ray * 3 + 2];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            w.l_x[p][c * w.lcap + i] = xb[c];                // init_bigpose
            w.l_d[p][c * w.lcap + i] = db[c];
        }
    }
}

__global__ __launch_bounds__(WARP_BLOCK) void k_deform_pairs(RenderArgs a, Workspace w, GridDev dg,
                                                             const float* __restrict__ W0, const float* __restrict__ B0,
                                                             const float* __restrict__ W1, const float* __restrict__ B1,
                                                             const float* __restrict__ W2, const float* __restrict__ B2) {
    const int p = blockIdx.y;
    const int cnt = w.counters[CNT_PAIRS + p];
    for (int64_t i = (int64_t)blockIdx.x * WARP_BLOCK + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * WARP_BLOCK) {
        float xb[3], r[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) xb[c] = w.l_x[p][c * w.lcap + i];
        deform_point(a.scene, dg, W0, B0, W1, B1, W2, B2, xb, r);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            w.l_x[p][c * w.lcap + i] = xb[c] + r[c];     // tpose = init_bigpose + resd (:111)
            w.l_r[p][c * w.lcap + i] = r[c];
        }
    }
}

int launch_warp_pairs(const RenderArgs& a, const Workspace& w, const GridDev& dg, const MlpDev& dm, hipStream_t st) {
    int64_t tiles = cdiv(w.lcap, WARP_BLOCK);
    unsigned gx = (unsigned)(tiles < 1024 ? (tiles > 0 ? tiles : 1) : 1024);
    hipLaunchKernelGGL(k_warp_pairs, dim3(gx, INVR_NUM_PARTS), dim3(WARP_BLOCK), 0, st, a, w, a.scene.A, a.scene.big_A);
    INVR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_deform_pairs, dim3(gx, INVR_NUM_PARTS), dim3(WARP_BLOCK), 0, st, a, w, dg, dm.w[0], dm.b[0], dm.w[1],
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
