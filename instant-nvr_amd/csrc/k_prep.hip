// Row f4: per-frame scene-tensor preparation on the device.
//  * invr_rigid_transformation : batch_rodrigues + get_rigid_transformation
//    (lib/utils/if_nerf/if_nerf_data_utils.py:523-577): axis-angle poses -> the 24 LBS matrices `A`,
//    float64 arithmetic like NumPy, float32 result.
//  * invr_pack_parts : the per-part KNN reference sets of Dataset.__getitem__
//    (lib/datasets/h36m/tpose_dataset.py:570-600): stable partition of the posed vertices / skinning
//    weights by part id, lengths2, and the per-part canonical bounds (min/max of tpose -/+ bbox_overlap).
#include "common.h"

__global__ void k_rigid_transformation(const double* __restrict__ poses, const double* __restrict__ joints,
                                       const int32_t* __restrict__ parents, float* __restrict__ A) {
    __shared__ double local[24][4][4];
    __shared__ double chain[24][4][4];
    const int j = threadIdx.x;
    if (j < 24) {
        // batch_rodrigues (:523-542)
        const double px = poses[j * 3] + 1e-8, py = poses[j * 3 + 1] + 1e-8, pz = poses[j * 3 + 2] + 1e-8;
        const double angle = sqrt(px * px + py * py + pz * pz);
        const double rx = poses[j * 3] / angle, ry = poses[j * 3 + 1] / angle, rz = poses[j * 3 + 2] / angle;
        const double c = cos(angle), s = sin(angle);
        const double K[3][3] = {{0.0, -rz, ry}, {rz, 0.0, -rx}, {-ry, rx, 0.0}};
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                double kk = 0.0;
                for (int m = 0; m < 3; ++m) kk += K[a][m] * K[m][b];
                local[j][a][b] = (a == b ? 1.0 : 0.0) + s * K[a][b] + (1.0 - c) * kk;
            }
        const int par = parents[j];
        for (int a = 0; a < 3; ++a) local[j][a][3] = joints[j * 3 + a] - (j > 0 ? joints[par * 3 + a] : 0.0);   // rel_joints
        local[j][3][0] = local[j][3][1] = local[j][3][2] = 0.0;
        local[j][3][3] = 1.0;
    }
    __syncthreads();
    if (j == 0) {                                       // kinematic chain (:558-563)
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) chain[0][a][b] = local[0][a][b];
        for (int i = 1; i < 24; ++i) {
            const int par = parents[i];
            for (int a = 0; a < 4; ++a)
                for (int b = 0; b < 4; ++b) {
                    double acc = 0.0;
                    for (int m = 0; m < 4; ++m) acc += chain[par][a][m] * local[i][m][b];
                    chain[i][a][b] = acc;
                }
        }
    }
    __syncthreads();
    if (j < 24) {                                       // remove the rest-pose joint (:566-570)
        for (int a = 0; a < 4; ++a) {
            double rel = 0.0;
            for (int b = 0; b < 3; ++b) rel += chain[j][a][b] * joints[j * 3 + b];      // joints_homogen has w = 0
            for (int b = 0; b < 4; ++b) A[(j * 4 + a) * 4 + b] = (float)(b == 3 ? chain[j][a][3] - rel : chain[j][a][b]);
        }
    }
}

int launch_rigid_transformation(const double* poses, const double* joints, const int32_t* parents, float* A, hipStream_t st) {
    hipLaunchKernelGGL(k_rigid_transformation, dim3(1), dim3(64), 0, st, poses, joints, parents, A);
    INVR_LAUNCH_CHECK();
    return 0;
}

// one workgroup per part: ordered (stable) compaction of the vertices with parts[v] == p
#define PACK_T 1024
__global__ __launch_bounds__(PACK_T) void k_pack_parts(const float* __restrict__ ppts, const float* __restrict__ weights,
                                                       const int64_t* __restrict__ parts, const float* __restrict__ tpose,
                                                       int n_verts, int n_w, int stride, float overlap,
                                                       float* __restrict__ part_pts, float* __restrict__ part_pbw,
                                                       int64_t* __restrict__ lengths2, float* __restrict__ bounds) {
    __shared__ int wsum[PACK_T / 64];
    __shared__ int carry_s;
    __shared__ float red[6][PACK_T / 64];
    const int p = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    __syncthreads();
    for (int base = 0; base < n_verts; base += PACK_T) {
        const int v = base + threadIdx.x;
        const bool mine = v < n_verts && parts[v] == p;
        const unsigned long long m = __ballot(mine);
        if (lane == 0) wsum[wv] = __popcll(m);
        __syncthreads();
        int off = carry_s;
        for (int k = 0; k < wv; ++k) off += wsum[k];
        if (mine) {
            const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
            for (int a = 0; a < 3; ++a) {
                part_pts[((int64_t)p * stride + pos) * 3 + a] = ppts[v * 3 + a];
                lo[a] = fminf(lo[a], tpose[v * 3 + a]);
                hi[a] = fmaxf(hi[a], tpose[v * 3 + a]);
            }
            for (int k = 0; k < n_w; ++k) part_pbw[((int64_t)p * stride + pos) * n_w + k] = weights[(int64_t)v * n_w + k];
        }
        __syncthreads();
        if (threadIdx.x == 0) { int t = 0; for (int k = 0; k < PACK_T / 64; ++k) t += wsum[k]; carry_s += t; }
        __syncthreads();
    }
    const int len = carry_s;
    // zero padding behind the part's vertices (the reference allocates zeros, :579-580)
    for (int64_t e = (int64_t)len * 3 + threadIdx.x; e < (int64_t)stride * 3; e += PACK_T) part_pts[(int64_t)p * stride * 3 + e] = 0.0f;
    for (int64_t e = (int64_t)len * n_w + threadIdx.x; e < (int64_t)stride * n_w; e += PACK_T) part_pbw[(int64_t)p * stride * n_w + e] = 0.0f;
    for (int a = 0; a < 3; ++a) {
        for (int d = 32; d >= 1; d >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], d)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], d)); }
        if (lane == 0) { red[a][wv] = lo[a]; red[3 + a][wv] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        lengths2[p] = len;
        for (int a = 0; a < 3; ++a) {
            float l = red[a][0], h = red[3 + a][0];
            for (int k = 1; k < PACK_T / 64; ++k) { l = fminf(l, red[a][k]); h = fmaxf(h, red[3 + a][k]); }
            bounds[p * 6 + a] = l - overlap;            // :588-589
            bounds[p * 6 + 3 + a] = h + overlap;
        }
    }
}

int launch_pack_parts(const float* ppts, const float* weights, const int64_t* parts, const float* tpose, int n_verts, int n_w,
                      int stride, float overlap, float* part_pts, float* part_pbw, int64_t* lengths2, float* bounds, hipStream_t st) {
    hipLaunchKernelGGL(k_pack_parts, dim3(INVR_NUM_PARTS), dim3(PACK_T), 0, st, ppts, weights, parts, tpose, n_verts, n_w, stride,
                       overlap, part_pts, part_pbw, lengths2, bounds);
    INVR_LAUNCH_CHECK();
    return 0;
}
