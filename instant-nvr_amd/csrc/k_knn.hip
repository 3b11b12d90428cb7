// K2: per-part 4-nearest-neighbour skinning against the posed SMPL vertices.
// Replaces pts_knn_blend_weights_multiassign_batch -> sample_blend_closest_points -> knn_points
// (lib/utils/blend_utils.py:817-825, 741-763, 732-738; pytorch3d brute-force KNN, squared L2,
// first lengths2[p] vertices of part p).
//
// One thread per query point, 256 points per workgroup; the part's vertices stream through LDS in
// 2048-vertex tiles and are read as wave-uniform (broadcast) ds_read_b128, so the inner loop is
// 3 sub + 3 mul/add + 1 compare per vertex with a rarely-taken sorted-insert branch.
// NOTE (reference quirk that parity depends on): the weights are normalised by (sum + 1e-8)
// (blend_utils.py:748), so for a part further than ~0.5 m the gaussian weights underflow against
// the epsilon, the "weighted distance" tends to 0 and the pair IS flagged (dist < smpl_thresh) with
// near-zero blend weights.  Far parts therefore cannot be culled by distance.
#include "pipeline.h"

#define KNN_BLOCK 256
#define KNN_TILE 2048
#define KNN_K 4
#define KNN_EPS 1e-8f
// 2 * radius**2 with radius = 0.075 (blend_utils.py:741,747), evaluated in double like Python does
#define KNN_TWO_R2 ((float)(2.0 * 0.075 * 0.075))

struct Top4 {
    float d[KNN_K];
    int i[KNN_K];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int k = 0; k < KNN_K; ++k) { d[k] = __builtin_inff(); i[k] = 0; }
    }
    __device__ __forceinline__ void push(float v, int idx) {
        if (v < d[3]) {
            if (v < d[2]) {
                d[3] = d[2]; i[3] = i[2];
                if (v < d[1]) {
                    d[2] = d[1]; i[2] = i[1];
                    if (v < d[0]) { d[1] = d[0]; i[1] = i[0]; d[0] = v; i[0] = idx; }
                    else { d[1] = v; i[1] = idx; }
                } else { d[2] = v; i[2] = idx; }
            } else { d[3] = v; i[3] = idx; }
        }
    }
};

// exact 4-NN of (px,py,pz) among verts[0..len) of one part; all threads of the block must call.
__device__ __forceinline__ void knn_scan_part(const float* __restrict__ verts, int len, float px, float py,
                                              float pz, Top4& t, float4* sv) {
    t.init();
    for (int base = 0; base < len; base += KNN_TILE) {
        int m = min(KNN_TILE, len - base);
        __syncthreads();
        for (int j = threadIdx.x; j < m; j += KNN_BLOCK) {
            const float* v = verts + (int64_t)(base + j) * 3;
            sv[j] = make_float4(v[0], v[1], v[2], 0.0f);
        }
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < m; ++j) {
            float4 v = sv[j];
            float dx = px - v.x, dy = py - v.y, dz = pz - v.z;
            float d2 = dx * dx + dy * dy + dz * dz;       // ((p1-p2)**2).sum(-1)
            t.push(d2, base + j);
        }
    }
}

// gaussian weights + weighted distance (blend_utils.py:745-749)
__device__ __forceinline__ float knn_weights(const Top4& t, float* w) {
    float d[KNN_K], s = 0.0f;
#pragma unroll
    for (int k = 0; k < KNN_K; ++k) {
        d[k] = sqrtf(t.d[k]);                              // cast_knn_points: dists.sqrt()
        w[k] = expf(-(d[k] * d[k]) / KNN_TWO_R2);
        s += w[k];
    }
    float den = s + KNN_EPS, dist = 0.0f;
#pragma unroll
    for (int k = 0; k < KNN_K; ++k) {
        w[k] = w[k] / den;
        dist += d[k] * w[k];
    }
    return dist;
}

// ---- dense variant: every (point, part) -> bw (n,P,24), dist (n,P) -----------------------------
__global__ __launch_bounds__(KNN_BLOCK) void k_knn_dense(SceneDev s, const float* pose_pts, int64_t n, float* bw, float* dist) {
    __shared__ float4 sv[KNN_TILE];
    int64_t i = (int64_t)blockIdx.x * KNN_BLOCK + threadIdx.x;
    bool live = i < n;
    float px = 0, py = 0, pz = 0;
    if (live) { px = pose_pts[i * 3]; py = pose_pts[i * 3 + 1]; pz = pose_pts[i * 3 + 2]; }
    for (int p = 0; p < INVR_NUM_PARTS; ++p) {
        int len = (int)s.lengths2[p];
        Top4 t;
        knn_scan_part(s.part_pts + (int64_t)p * s.M * 3, len, px, py, pz, t, sv);
        if (!live) continue;
        float w[KNN_K];
        float ds = knn_weights(t, w);
        dist[i * INVR_NUM_PARTS + p] = ds;
        float* o = bw + (i * INVR_NUM_PARTS + p) * INVR_NUM_JOINTS;
        const float* pb = s.part_pbw + (int64_t)p * s.M * INVR_NUM_JOINTS;
#pragma unroll
        for (int j = 0; j < INVR_NUM_JOINTS; ++j) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < KNN_K; ++k) acc += pb[(int64_t)t.i[k] * INVR_NUM_JOINTS + j] * w[k];
            o[j] = acc;
        }
    }
}

int launch_knn_blend_dense(const SceneDev& s, const float* pose_pts, int64_t n, float* bw, float* dist, hipStream_t st) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_knn_dense, dim3((unsigned)cdiv(n, KNN_BLOCK)), dim3(KNN_BLOCK), 0, st, s, pose_pts, n, bw, dist);
    INVR_LAUNCH_CHECK();
    return 0;
}

// ---- pipeline variant ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_part_aabb(SceneDev s, float* aabb) {
    int p = blockIdx.x;
    int len = (int)s.lengths2[p];
    const float* v = s.part_pts + (int64_t)p * s.M * 3;
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int j = threadIdx.x; j < len; j += 64)
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], v[j * 3 + a]); hi[a] = fmaxf(hi[a], v[j * 3 + a]); }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int d = 32; d >= 1; d >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], d)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], d)); }
    if (threadIdx.x == 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) { aabb[p * 6 + a] = lo[a]; aabb[p * 6 + 3 + a] = hi[a]; }
}

__global__ __launch_bounds__(KNN_BLOCK) void k_knn_pairs(RenderArgs a, Workspace w) {
    __shared__ float4 sv[KNN_TILE];
    const int na = w.counters[CNT_ACTIVE];
    const int lane = threadIdx.x & 63;
    for (int64_t tile = blockIdx.x; tile * KNN_BLOCK < na; tile += gridDim.x) {
        int64_t slot = tile * KNN_BLOCK + threadIdx.x;
        bool live = slot < na;
        float px = 0, py = 0, pz = 0;
        if (live) sample_pose_point(a, w.active_idx[slot], px, py, pz, nullptr, nullptr);
        unsigned flags = 0;
        for (int p = 0; p < INVR_NUM_PARTS; ++p) {
            int len = (int)a.scene.lengths2[p];
            Top4 t;
            knn_scan_part(a.scene.part_pts + (int64_t)p * a.scene.M * 3, len, px, py, pz, t, sv);
            float wt[KNN_K];
            float ds = knn_weights(t, wt);
            bool flag = live && ds < a.scene.thresh;          // pflag (inb_part_network_multiassign.py:90)
            unsigned long long m = __ballot(flag);
            if (m) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&w.counters[CNT_PAIRS + p], __popcll(m));
                base = __shfl(base, 0);
                if (flag) {
                    int64_t pos = base + __popcll(m & ((1ull << lane) - 1ull));
                    w.l_slot[p][pos] = (int32_t)slot;
                    reinterpret_cast<int4*>(w.l_nn[p])[pos] = make_int4(t.i[0], t.i[1], t.i[2], t.i[3]);
                    reinterpret_cast<float4*>(w.l_w[p])[pos] = make_float4(wt[0], wt[1], wt[2], wt[3]);
                    flags |= 1u << p;
                }
            }
        }
        if (live) w.pflags[slot] = (uint8_t)flags;
    }
}

int launch_knn_pairs(const RenderArgs& a, const Workspace& w, hipStream_t st) {
    hipLaunchKernelGGL(k_part_aabb, dim3(INVR_NUM_PARTS), dim3(64), 0, st, a.scene, w.part_aabb);
    INVR_LAUNCH_CHECK();
    int64_t tiles = cdiv(w.cap, KNN_BLOCK);
    unsigned grid = (unsigned)(tiles < 256 * 8 ? (tiles > 0 ? tiles : 1) : 256 * 8);
    hipLaunchKernelGGL(k_knn_pairs, dim3(grid), dim3(KNN_BLOCK), 0, st, a, w);
    INVR_LAUNCH_CHECK();
    return 0;
}
