// K2: per-part 4-nearest-neighbour skinning against the posed SMPL vertices.
// Replaces pts_knn_blend_weights_multiassign_batch -> sample_blend_closest_points -> knn_points
// (lib/utils/blend_utils.py:817-825, 741-763, 732-738; pytorch3d brute-force KNN, squared L2,
// first lengths2[p] vertices of part p) and the pflag test (inb_part_network_multiassign.py:90).
//
// REFERENCE QUIRK the design is built around: the gaussian weights are normalised by
// (sum + 1e-8) (blend_utils.py:748).  With w = exp(-d^2 / 0.01125):
//   * d1 (nearest vertex) < thresh            -> dist ~ d1            -> flagged   ("near")
//   * thresh*1.01 <= d1 <= ~0.47 m            -> dist >= d1*w1/(w1+1e-8) >= thresh -> NOT flagged
//   * d1 >~ 0.5 m                             -> weights underflow against the epsilon, dist -> 0
//                                                -> flagged with tiny blend weights  ("band")
//   * d1 > 0.68 m                             -> every normalised weight <= exp(-0.68^2/0.01125)/1e-8
//                                                = 1.4e-10, sum s <= 5.6e-10.  Then |A_bw| <= 1.1e-9,
//                                                R_inv = adj/(det+eps_f32) <= 5e-12, x_t <= 5e-11 and the
//                                                canonical point |x_b| <= s*|t_big| <= 1.1e-9 m, view
//                                                direction <= 1e-20: below the fp32 resolution of every
//                                                coordinate the encoders / UV volume form from it (half an
//                                                ulp of the 0.3..1.2 m box offsets is 1.5e-8..6e-8) -> the
//                                                pair's field value is a per-part CONSTANT (x_b = 0, d = 0)
// So every sample evaluates ~3 parts in the reference, most of them "far" pairs that all collapse
// onto the same canonical point.  This kernel classifies each (point, part) with cluster bounds,
// runs the exact 4-NN only where the result can matter (near / band), marks far pairs with a bit
// (the merge kernel substitutes the per-part constant, evaluated once per frame through the very
// same warp/encode/MLP kernels from an appended zero-weight pair) and drops provably unflagged
// pairs.  Exactness: near/band pairs are bit-faithful brute-force results; far pairs differ from
// the reference by <= 1.1e-9 m in the canonical point (see above).
//
// Data layout: a per-frame prepare kernel (k_part_prepare, side stream) Morton-sorts each part's vertices — a counting sort on
// the 12 leading Morton bits in LDS, every element ranking itself inside its bucket — and writes them pair-interleaved,
// {x0,x1,y0,y1}{z0,z1,row0,row1} per two vertices, with one {AABB, first vertex} record per 64-vertex cluster and four AABBs of its
// 16-vertex sub-clusters; k_knn_voxel_class classifies the live lattice cells of the distance volume per part (far / provably
// unflagged / undecided + candidate-cluster mask + 4th-nearest bound).  The query kernel (k_knn_pairs) holds the WHOLE index in
// LDS (persistent 1024-thread workgroups, one per CU); a wave draws tickets of 64 survivors, one thread per point: every vertex /
// record read is a wave-uniform (broadcast) ds_read_b128, clusters and sub-clusters are pruned per wave with
// lb(box) <= current 4th-best, distances of two vertices per packed-fp32 instruction, top-4 kept as 64-bit (distance, row) keys.
#include <stdlib.h>
#include "pipeline.h"
#include "front_bodies.h"

#define KNN_BLOCK 256
#define KNN_K 4
#define KNN_EPS 1e-8f
// 2 * radius**2 with radius = 0.075 (blend_utils.py:741,747), evaluated in double like Python does
#define KNN_TWO_R2 ((float)(2.0 * 0.075 * 0.075))
#define KNN_TWO_R2_F 0.01125f
#define KNN_DFAR2 0.4624f        // (0.68 m)^2, see the header comment: valid while the entries of A / big_A are <= 2 in magnitude
// The bound above scales with the largest entry M of the frame's A / big_A matrices (|A_bw| <= s M, |x_b| <= s |t_big|): the far
// distance actually used is derived per frame by k_part_prepare from the matrices themselves — 0.68 m for M <= 2, the distance at
// which 4 exp(-d^2 / 0.01125) / 1e-8 * M falls to the same 1.1e-9 beyond that, +inf (no folding) for non-finite / absurd
// matrices — and read by the kernels from ix.dfar2.
__device__ __forceinline__ float knn_far_dist2(float m_abs) {
    if (!(m_abs <= 1e6f)) return __builtin_inff();
    if (m_abs <= 2.0f) return KNN_DFAR2;
    return fmaxf(KNN_DFAR2, KNN_TWO_R2_F * logf(4.0f * m_abs / 1.12e-17f));
}

// Sorted 4-best list on 64-bit keys (squared distance bits << 32 | vertex row): non-negative floats
// order like their bit patterns, so one unsigned compare orders by (distance, row) and the result
// does not depend on the order in which vertices are visited (exact distance ties do occur at
// ~1e-7 per pair).  +inf / row 0 is both the empty marker and the cluster padding sentinel.
struct Top4 {
    unsigned long long k[KNN_K];
    float d[KNN_K];
    int i[KNN_K];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < KNN_K; ++j) k[j] = 0x7F80000000000000ull;
    }
    // Sorted insert as a min/max network on the keys READ AS DOUBLES: the high word of a key is the bit pattern of a non-negative
    // float <= +inf, so every key is a finite non-negative double and doubles of that kind order like their bit patterns —
    // v_min_f64 / v_max_f64 are 64-bit unsigned min / max here (fp64 denormals are on by default, no NaN pattern can occur).
    // 7 full-rate instructions, no compare, no select, no branch; the compare-and-select form was 5 v_cmp_u64 + 14 v_cndmask
    // + wait states per insert and made the inserts the largest item of the sweep.
    static __device__ __forceinline__ double kmin(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
    static __device__ __forceinline__ double kmax(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
    __device__ __forceinline__ void push_net(float v, int idx) {
        double x = __longlong_as_double((long long)(((unsigned long long)__float_as_uint(v) << 32) | (unsigned)idx));
        const double k0 = __longlong_as_double((long long)k[0]), k1 = __longlong_as_double((long long)k[1]),
                     k2 = __longlong_as_double((long long)k[2]), k3 = __longlong_as_double((long long)k[3]);
        const double n0 = kmin(k0, x); x = kmax(k0, x);
        const double n1 = kmin(k1, x); x = kmax(k1, x);
        const double n2 = kmin(k2, x); x = kmax(k2, x);
        const double n3 = kmin(k3, x);
        k[0] = (unsigned long long)__double_as_longlong(n0); k[1] = (unsigned long long)__double_as_longlong(n1);
        k[2] = (unsigned long long)__double_as_longlong(n2); k[3] = (unsigned long long)__double_as_longlong(n3);
    }
    // guarded form for the brute-force kernels (most vertices fail the guard: one compare, usually wave-skipped branch)
    __device__ __forceinline__ void push(float v, int idx) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)idx;
        if (key < k[3]) push_net(v, idx);
    }
    __device__ __forceinline__ float worst() const { return __uint_as_float((unsigned)(k[3] >> 32)); }
    __device__ __forceinline__ void finish() {
#pragma unroll
        for (int j = 0; j < KNN_K; ++j) { d[j] = __uint_as_float((unsigned)(k[j] >> 32)); i[j] = (int)(unsigned)k[j]; }
    }
};

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

// ---- dense brute-force variant (invr_knn_blend): every (point, part) -> bw (n,P,24), dist (n,P) ----
#define KNN_TILE 2048
__global__ __launch_bounds__(KNN_BLOCK) void k_knn_dense(SceneDev s, const float* pose_pts, int64_t n, float* bw, float* dist,
                                                        int32_t* nn_out, float* d2_out, float* w_out) {
    __shared__ float4 sv[KNN_TILE];
    int64_t i = (int64_t)blockIdx.x * KNN_BLOCK + threadIdx.x;
    bool live = i < n;
    float px = 0, py = 0, pz = 0;
    if (live) { px = pose_pts[i * 3]; py = pose_pts[i * 3 + 1]; pz = pose_pts[i * 3 + 2]; }
    for (int p = 0; p < INVR_NUM_PARTS; ++p) {
        int len = (int)s.lengths2[p];
        const float* verts = s.part_pts + (int64_t)p * s.M * 3;
        Top4 t;
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
                t.push(dx * dx + dy * dy + dz * dz, base + j);                 // ((p1-p2)**2).sum(-1)
            }
        }
        if (!live) continue;
        t.finish();
        float w[KNN_K];
        float ds = knn_weights(t, w);
        dist[i * INVR_NUM_PARTS + p] = ds;
        if (nn_out) {                                             // invr_knn_neighbors: the neighbours themselves, in (distance,row) order
#pragma unroll
            for (int k = 0; k < KNN_K; ++k) {
                nn_out[(i * INVR_NUM_PARTS + p) * KNN_K + k] = t.i[k];
                d2_out[(i * INVR_NUM_PARTS + p) * KNN_K + k] = t.d[k];
                w_out[(i * INVR_NUM_PARTS + p) * KNN_K + k] = w[k];
            }
        }
        if (!bw) continue;
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

int launch_knn_blend_dense(const SceneDev& s, const float* pose_pts, int64_t n, float* bw, float* dist, int32_t* nn, float* d2,
                           float* w, hipStream_t st) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_knn_dense, dim3((unsigned)cdiv(n, KNN_BLOCK)), dim3(KNN_BLOCK), 0, st, s, pose_pts, n, bw, dist, nn, d2, w);
    INVR_LAUNCH_CHECK();
    return 0;
}

// Phase timers of k_knn_pairs (tools/knn_phase_prof.sh builds a second library with -DKNN_PROF; not part of libinvr.so)
#ifdef KNN_PROF
__device__ unsigned long long g_knn_prof[32];
extern "C" int invr_debug_knn_prof(unsigned long long* out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_knn_prof), sizeof(g_knn_prof)) != hipSuccess) return 1;
    if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_knn_prof), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#define KP_DECL long long kp_t0 = clock64(); long long kp_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define KP(i) { const long long kp_now = clock64(); kp_acc[i] += kp_now - kp_t0; kp_t0 = kp_now; }
#define KP_CNT(i) { kp_acc[i] += 1; }
#define KP_MARK const long long kp_mark = kp_acc[11];
#define KP_BAND(c) { if (c) kp_acc[7] += kp_acc[11] - kp_mark; }
#define KP_FLUSH if ((threadIdx.x & 63) == 0) { for (int kp_i = 0; kp_i < 16; ++kp_i) atomicAdd(&g_knn_prof[kp_i], (unsigned long long)kp_acc[kp_i]); }
#define KP_FLUSH_AT(base) if (threadIdx.x == 0 && blockIdx.x == 0) { for (int kp_i = 0; kp_i < 12; ++kp_i) atomicAdd(&g_knn_prof[(base) + kp_i], (unsigned long long)kp_acc[kp_i]); }
#define KP_SCAN_ARG , long long* kp_acc
#define KP_SCAN_PASS , kp_acc
#define KP_INS(c) { if (__ballot(c)) kp_acc[12] += 1; }      // pair records whose insert branch the wave executes
#else
#define KP_DECL
#define KP(i)
#define KP_CNT(i)
#define KP_MARK
#define KP_BAND(c)
#define KP_FLUSH
#define KP_FLUSH_AT(base)
#define KP_SCAN_ARG
#define KP_SCAN_PASS
#define KP_INS(c)
#endif

// ---- per-frame prepare: Morton sort + clusters ------------------------------------------------------
#define PREP_T 1024
#define PREP_MAX 8192

__device__ __forceinline__ unsigned spread6(unsigned v) {       // 6 bits -> every third bit
    v &= 63u;
    v = (v | (v << 8)) & 0x0300F;
    v = (v | (v << 4)) & 0x030C3;
    v = (v | (v << 2)) & 0x09249;
    return v;
}

// (body: one PREP_T-thread workgroup per part p; prep_lds = [keys PREP_MAX | counting-sort output PREP_MAX | 4096 buckets] = 80 KB)
__device__ __forceinline__ void part_prepare_body(const SceneDev& s, const KnnIndex& ix, const int p, unsigned* prep_lds) {
    unsigned* keys = prep_lds;
    unsigned* sorted = prep_lds + PREP_MAX;
    __shared__ float red[6][PREP_T / 64];
    const int len = min((int)s.lengths2[p], PREP_MAX);
    const float* v = s.part_pts + (int64_t)p * s.M * 3;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    KP_DECL
    // 1. part AABB (the thread's <= 8 vertices stay in registers for step 2: one global round trip instead of two)
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    float vr[PREP_MAX / PREP_T][3];
#pragma unroll
    for (int k = 0; k < PREP_MAX / PREP_T; ++k) {
        const int j = threadIdx.x + k * PREP_T;
#pragma unroll
        for (int a = 0; a < 3; ++a) vr[k][a] = j < len ? v[j * 3 + a] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < PREP_MAX / PREP_T; ++k)
        if (threadIdx.x + k * PREP_T < len)
#pragma unroll
            for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], vr[k][a]); hi[a] = fmaxf(hi[a], vr[k][a]); }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int d = 32; d >= 1; d >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], d)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], d)); }
        if (lane == 0) { red[a][wv] = lo[a]; red[3 + a][wv] = hi[a]; }
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int k = 0; k < PREP_T / 64; ++k) { lo[a] = fminf(lo[a], red[a][k]); hi[a] = fmaxf(hi[a], red[3 + a][k]); }
    if (threadIdx.x == 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) { ix.part_aabb[p * 6 + a] = lo[a]; ix.part_aabb[p * 6 + 3 + a] = hi[a]; }
    if (p == 0 && wv == 0) {          // far-fold distance of this frame from the largest |entry| of A / big_A (header comment)
        float m = 0.0f;
        for (int j = lane; j < INVR_NUM_JOINTS * 16; j += 64) {
            const float x = fabsf(s.A[j]), y = fabsf(s.big_A[j]);
            m = (x != x || y != y) ? __builtin_inff() : fmaxf(m, fmaxf(x, y));
        }
        for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
        if (lane == 0) ix.dfar2[0] = knn_far_dist2(m);
    }
    KP(0)
    // 2. Morton keys (6 bits / axis) | original index (13 bits)
#pragma unroll
    for (int k = 0; k < PREP_MAX / PREP_T; ++k) {
        const int j = threadIdx.x + k * PREP_T;
        if (j < len) {
            unsigned q[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float e = hi[a] - lo[a];
                float u = e > 0.f ? (vr[k][a] - lo[a]) / e : 0.f;
                q[a] = (unsigned)fminf(fmaxf(u * 64.0f, 0.0f), 63.0f);
            }
            unsigned m = (spread6(q[0]) << 2) | (spread6(q[1]) << 1) | spread6(q[2]);
            keys[j] = (m << 13) | (unsigned)j;
        }
    }
    __syncthreads();
    KP(1)
    // 3. sort by key: counting sort on the 12 leading Morton bits (4096 buckets for <= 8192 vertices: ~1 vertex per bucket),
    //    then every element takes its rank inside its bucket.  (A 1024-thread bitonic sort of 4096 keys needs 78
    //    barrier-separated passes: 40 of this kernel's 70 us, and the kernel is the head of the frame's critical path for
    //    small frames / ray shards.)  The result is the same total order: keys are unique (they end in the vertex index).
    {
        unsigned* hist = sorted + PREP_MAX;              // [4096] bucket counts -> start offsets
        __shared__ int wsum[PREP_T / 64];
        for (int j = threadIdx.x; j < 4096; j += PREP_T) hist[j] = 0;
        __syncthreads();
        for (int j = threadIdx.x; j < len; j += PREP_T) atomicAdd(&hist[keys[j] >> 19], 1u);     // key = morton18 << 13 | index
        __syncthreads();
        // exclusive scan of the 4096 counts: 4 per thread, wave scan, block scan
        unsigned c4[4], tot = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { c4[k] = hist[threadIdx.x * 4 + k]; tot += c4[k]; }
        unsigned x = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { unsigned y = __shfl_up(x, d); if (lane >= d) x += y; }
        if (lane == 63) wsum[wv] = (int)x;
        __syncthreads();
        unsigned woff = 0;
        for (int k = 0; k < wv; ++k) woff += (unsigned)wsum[k];
        unsigned run = woff + x - tot;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) { hist[threadIdx.x * 4 + k] = run; run += c4[k]; }
        __syncthreads();
        // scatter through per-bucket cursors (the cursor of bucket b ends at the start of bucket b+1)
        for (int j = threadIdx.x; j < len; j += PREP_T) {
            const unsigned key = keys[j];
            sorted[atomicAdd(&hist[key >> 19], 1u)] = key;
        }
        __syncthreads();
        // order inside the buckets: every element finds its rank among the keys of its own bucket [end(b-1), end(b)) — a few
        // independent LDS reads per thread.  (One thread insertion-sorting a whole bucket was a serial chain of dependent LDS
        // round trips on the fullest bucket — the vertices lie on a surface, occupied buckets hold up to ~30 — and 70 % of this
        // kernel, which heads the frame's critical path for small frames / ray shards.)
        for (int j = threadIdx.x; j < len; j += PREP_T) {
            const unsigned key = sorted[j];
            const unsigned b = key >> 19;
            const int lo_b = b ? (int)hist[b - 1] : 0, hi_b = (int)hist[b];
            int rank = 0;
            for (int q = lo_b; q < hi_b; ++q) rank += sorted[q] < key ? 1 : 0;
            keys[lo_b + rank] = key;
        }
        __syncthreads();
    }
    KP(2)
    // 4. sorted vertices, pair-interleaved {x0,x1,y0,y1} {z0,z1,n0,n1} with n = |v|^2 (packed-fp32 distance math reads two
    //    vertices per register pair; n feeds the sweep's |v|^2 - 2 q.v prefilter), their rows inside the part as 16-bit entries of
    //    a separate array (only read on an insert), clusters of 64 and their four 16-vertex sub-clusters
    const int64_t voff = (int64_t)p * ix.mpad;
    const int lenp = (len + 63) & ~63;
    float* svf = reinterpret_cast<float*>(ix.sverts + voff);
    for (int j = threadIdx.x; j < lenp; j += PREP_T) {
        // sentinel: infinitely far, never enters a top-4 -> clusters are always scanned as 64
        float x = 1e30f, y = 1e30f, z = 1e30f;              // squared distance and |v|^2 overflow to +inf
        int o = 0;
        if (j < len) { o = (int)(keys[j] & 8191u); x = v[o * 3]; y = v[o * 3 + 1]; z = v[o * 3 + 2]; }
        float* b = svf + (j >> 1) * 8 + (j & 1);
        b[0] = x; b[2] = y; b[4] = z; b[6] = (x * x + y * y) + z * z;
        ix.srow[voff + j] = (uint16_t)o;                    // (PREP_MAX = 8192 rows: 13 bits)
    }
    KP(3)
    const int ncl = (len + 63) >> 6;
    for (int c = wv; c < ncl; c += PREP_T / 64) {
        int j = c * 64 + lane;
        float x[3];
        bool ok = j < len;
        int o = ok ? (int)(keys[j] & 8191u) : 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) x[a] = v[o * 3 + a];
        float clo[3], chi[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            clo[a] = ok ? x[a] : __builtin_inff();
            chi[a] = ok ? x[a] : -__builtin_inff();
            for (int d = 1; d <= 8; d <<= 1) { clo[a] = fminf(clo[a], __shfl_xor(clo[a], d)); chi[a] = fmaxf(chi[a], __shfl_xor(chi[a], d)); }
        }
        const int64_t co = (int64_t)p * ix.cpad + c;
        if ((lane & 15) == 0) {
            ix.sub[co * 8 + (lane >> 4) * 2 + 0] = make_float4(clo[0], clo[1], clo[2], 0.f);
            ix.sub[co * 8 + (lane >> 4) * 2 + 1] = make_float4(chi[0], chi[1], chi[2], 0.f);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a)
            for (int d = 16; d <= 32; d <<= 1) { clo[a] = fminf(clo[a], __shfl_xor(clo[a], d)); chi[a] = fmaxf(chi[a], __shfl_xor(chi[a], d)); }
        if (lane == 0) {
            ix.cl[co * 3 + 0] = make_float4(clo[0], clo[1], clo[2], 0.f);
            ix.cl[co * 3 + 1] = make_float4(chi[0], chi[1], chi[2], 0.f);
            ix.cl[co * 3 + 2] = make_float4(x[0], x[1], x[2], 0.f);     // lane 0 = first vertex of the cluster
        }
    }
    KP(4)
    KP_CNT(8)
    KP_FLUSH_AT(16)
}

#define PREP_LDS_BYTES ((size_t)(2 * PREP_MAX + 4096) * sizeof(unsigned))
__global__ __launch_bounds__(PREP_T) void k_part_prepare(SceneDev s, KnnIndex ix) {
    extern __shared__ unsigned prep_lds[];
    part_prepare_body(s, ix, (int)blockIdx.x, prep_lds);
}

// wave-uniform 16-byte LDS read that stays a ds_read_b128 (256 B/clk): when .w is unused the compiler narrows
// the load to ds_read_b96, which runs at 96 B/clk (MI355X_MICROARCH.md LDS table) — 2x the LDS time
__device__ __forceinline__ float4 lds_ld4(const float4* p) {
    float4 v = *p;
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
    return v;
}

__device__ __forceinline__ float aabb_dist2(float px, float py, float pz, float4 lo, float4 hi) {
    float ex = fmaxf(fmaxf(lo.x - px, px - hi.x), 0.0f);
    float ey = fmaxf(fmaxf(lo.y - py, py - hi.y), 0.0f);
    float ez = fmaxf(fmaxf(lo.z - pz, pz - hi.z), 0.0f);
    return ex * ex + ey * ey + ez * ez;
}

typedef float v2f __attribute__((ext_vector_type(2)));

// 16 vertices = 8 pair records.  Straight-line batches of 4 pairs: 8 wave-uniform (broadcast) 16-byte LDS
// reads are issued together; the squared distances of two vertices are formed with packed fp32 ops
// (v_pk_add/mul_f32 — each component is the same IEEE op sequence as ((p1-p2)**2).sum(-1)); one fp32
// compare against the current 4th best guards the exact 64-bit-key inserts of both.
__device__ __forceinline__ void scan_sub16(const float4* sv, const unsigned* rw, v2f px, v2f py, v2f pz, Top4& t KP_SCAN_ARG) {
#pragma unroll
    for (int m0 = 0; m0 < 8; m0 += 4) {
        float4 A[4], B[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { A[k] = sv[(m0 + k) * 2]; B[k] = sv[(m0 + k) * 2 + 1]; }   // wave-uniform addresses
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const v2f dx = px - (v2f){A[k].x, A[k].y}, dy = py - (v2f){A[k].z, A[k].w}, dz = pz - (v2f){B[k].x, B[k].y};
            const v2f d2 = (dx * dx + dy * dy) + dz * dz;
            KP_INS(fminf(d2.x, d2.y) <= t.worst())
            if (fminf(d2.x, d2.y) <= t.worst()) {
                const unsigned r = rw[m0 + k];                                     // rows of the two vertices, 16 bits each
                t.push_net(d2.x, (int)(r & 0xFFFFu));
                t.push_net(d2.y, (int)(r >> 16));
            }
        }
    }
}

// The sweep's form of scan_sub16 (round 6).  |q - v|^2 = |v|^2 - 2 q.v + |q|^2: with n = |v|^2 in the record, s = n + a.v (a = -2q,
// three packed FMAs for two vertices) decides against thr = worst - |q|^2 + delta whether a vertex CAN be among the lane's four
// nearest; only then the exact squared distance — the reference's ((p1 - p2)**2).sum(-1) operation for operation — is formed and
// inserted, so the kept keys are those of scan_sub16 bit for bit (a vertex whose exact key is below the lane's fourth is never
// skipped; pushing a vertex that is not is a no-op of the min / max network).
// Error budget, u = 2^-24, M = |q| + max |v| of the part: n carries <= 3u |v|^2, the FMA chain <= 3u (|v|^2 + 2 |q||v|), Q = fl(|q|^2)
// <= 3u |q|^2, the exact path |D - T| <= 5u T, forming thr <= 4u M^2  ->  <= 19u M^2 in total; delta = 32u M^2 = 2^-19 M^2
// (1.2e-5 m^2 for M = 2.5 m: 1-5 % of a near pair's 4th-best squared distance, nothing for a band pair's).  A sentinel vertex
// (n = +inf) gives s = +inf: it passes only while thr is still +inf, and its exact distance +inf leaves the list unchanged as before.
// 5 instead of 11 VALU instructions per pair record that holds no candidate (74 % of the records, tools/knn_order_model.py counters).
__device__ __forceinline__ void scan_sub16_pf(const float4* sv, const unsigned* rw, v2f ax, v2f ay, v2f az, v2f px, v2f py, v2f pz,
                                              const float qd, float& thr, Top4& t KP_SCAN_ARG) {
#pragma unroll
    for (int m0 = 0; m0 < 8; m0 += 4) {
        float4 A[4], B[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { A[k] = sv[(m0 + k) * 2]; B[k] = sv[(m0 + k) * 2 + 1]; }   // wave-uniform addresses
        v2f s[4];                                // the four records' chains interleaved: no wait state between dependent packed FMAs
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] = __builtin_elementwise_fma(az, (v2f){B[k].x, B[k].y}, (v2f){B[k].z, B[k].w});
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] = __builtin_elementwise_fma(ay, (v2f){A[k].z, A[k].w}, s[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] = __builtin_elementwise_fma(ax, (v2f){A[k].x, A[k].y}, s[k]);
        // (the empty statement keeps the twelve FMAs in front of the first test — the optimizer sinks each chain to its use otherwise;
        // v_min_f32 spelled out: fminf of a value that came through the statement is canonicalized first, two more instructions per
        // record — s is never a NaN: finite operands, or +inf from a sentinel's |v|^2)
        asm volatile("" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float smin;
            asm("v_min_f32 %0, %1, %2" : "=v"(smin) : "v"(s[k].x), "v"(s[k].y));
            if (smin <= thr) {
                const v2f dx = px - (v2f){A[k].x, A[k].y}, dy = py - (v2f){A[k].z, A[k].w}, dz = pz - (v2f){B[k].x, B[k].y};
                const v2f d2 = (dx * dx + dy * dy) + dz * dz;
                KP_INS(fminf(d2.x, d2.y) <= t.worst())
                if (fminf(d2.x, d2.y) <= t.worst()) {
                    const unsigned r = rw[m0 + k];                                 // rows of the two vertices, 16 bits each
                    t.push_net(d2.x, (int)(r & 0xFFFFu));
                    t.push_net(d2.y, (int)(r >> 16));
                    thr = t.worst() - qd;
                }
            }
        }
    }
}

// The whole KNN index (Morton-sorted vertices + cluster records of all 5 parts, ~116 KB for SMPL's
// 6890 vertices) is staged ONCE per workgroup into LDS (160 KB/CU on MI355X) by persistent
// 1024-thread workgroups (one per CU, 4 waves/SIMD); every vertex / cluster record is then a
// wave-uniform (broadcast) ds_read_b128.  (The scalar-cache path was tried first: with a working set
// of ~7x the 16 KB scalar cache its miss path throttled the kernel to ~30 % VALU utilisation.)
#define KNN_T 1024
#define KNN_LDS_MAX_V 7680          // float4 vertex slots (123 KB) + 2 bytes of row per slot + 11 record float4 per cluster must fit 160 KB

#define KNN_LDS_HDR 8               // float4: the LDS carve table (offsets / lengths of the five parts)
#define KNN_LDS_FLOAT4 (KNN_LDS_HDR + KNN_LDS_MAX_V + KNN_LDS_MAX_V / 8 + (KNN_LDS_MAX_V / 64 + INVR_NUM_PARTS) * 11)

// Fallback when the posed vertex sets of the five parts do not fit the LDS-resident index together (> 8192 vertex slots, e.g.
// SMPL-X): brute force per (survivor, part) over LDS tiles of the part's vertices — the arithmetic of k_knn_dense, the outputs of
// k_knn_pairs (far pairs by the exact nearest distance > 0.68 m; neighbours / weights of flagged pairs stored at the survivor's
// slot).  Called by k_knn_pairs (same launch) when its LDS carve does not fit; sv = KNN_TILE float4 of LDS.
template <int BLOCK>
__device__ void knn_pairs_bf(const RenderArgs& a, const Workspace& w, float4* sv) {
    const int na = w.counters[CNT_ACTIVE];
    const float dfar2 = w.knn.dfar2[0];
    for (int64_t tile = blockIdx.x; tile * BLOCK < na; tile += gridDim.x) {
        const int64_t slot = tile * BLOCK + threadIdx.x;
        const bool live = slot < na;
        float px = 0, py = 0, pz = 0;
        if (live) sample_pose_point(a, w.active_idx[slot], px, py, pz, nullptr, nullptr);
        unsigned flags = 0, farflags = 0;
        for (int p = 0; p < INVR_NUM_PARTS; ++p) {
            const int len = (int)a.scene.lengths2[p];
            if (len < KNN_K) continue;
            const float* verts = a.scene.part_pts + (int64_t)p * a.scene.M * 3;
            Top4 t;
            t.init();
            for (int base = 0; base < len; base += KNN_TILE) {
                const int m = min(KNN_TILE, len - base);
                __syncthreads();
                for (int j = threadIdx.x; j < m; j += BLOCK) {
                    const float* v = verts + (int64_t)(base + j) * 3;
                    sv[j] = make_float4(v[0], v[1], v[2], 0.0f);
                }
                __syncthreads();
#pragma unroll 4
                for (int j = 0; j < m; ++j) {
                    const float4 v = sv[j];
                    const float dx = px - v.x, dy = py - v.y, dz = pz - v.z;
                    t.push(dx * dx + dy * dy + dz * dz, base + j);
                }
            }
            t.finish();
            float wt[KNN_K];
            const float ds = knn_weights(t, wt);
            const bool far = live && t.d[0] > dfar2;
            const bool hit = live && !far && ds < a.scene.thresh;
            if (far) farflags |= 1u << p;
            const unsigned long long fb = __ballot(far);
            if ((threadIdx.x & 63) == 0 && fb) atomicAdd(&w.counters[CNT_FAR + p], __popcll(fb));
            const unsigned long long hb = __ballot(hit);
            if ((threadIdx.x & 63) == 0 && hb) atomicAdd(&w.gcount[slot / PAIR_GROUP * INVR_NUM_PARTS + p], __popcll(hb));
            if (hit) {
                flags |= 1u << p;
                reinterpret_cast<int4*>(w.l_nn[p])[slot] = make_int4(t.i[0], t.i[1], t.i[2], t.i[3]);
                reinterpret_cast<float4*>(w.l_w[p])[slot] = make_float4(wt[0], wt[1], wt[2], wt[3]);
            }
        }
        if (live) {
            w.pflags[slot] = (uint8_t)flags;
            w.farflags[slot] = (uint8_t)farflags;
        }
    }
}

struct KnnLds { int voff[INVR_NUM_PARTS], coff[INVR_NUM_PARTS], soff[INVR_NUM_PARTS], len[INVR_NUM_PARTS]; };

// Outputs per survivor slot: pflags / farflags bytes and, for every flagged part, the 4 neighbour rows and weights at
// l_nn[p][slot] / l_w[p][slot]; k_pair_lists then builds the dense per-part lists of flagged slots.  (The lists used to be
// appended here, aggregated per 1024-point workgroup tile: the two barriers per tile made every wave wait for the slowest of
// the 16 — 64 consecutive survivors are one depth slab of a few rays (ray-major order: depth segments of a few rays), and they differ widely in how many parts they
// come near — 29 % of the kernel's wave time on a whole frame and 51 % on a 1/8 shard, tools/knn_phase_prof.py.  Now a wave
// never waits for another one.)
// KNN_DBG: ablation switch of the profiling builds only (tools/knn_phase_prof.sh: -DKNN_DBG=1 no exact scans, 2 seed cluster
// only, 4 the sweep's box tests without its vertex scans, 8 vertex scans whose prefilter never passes — WRONG results); the shipped
// library is compiled without it.
#ifndef KNN_DBG
#define KNN_DBG 0
#endif
#ifndef KNN_SUB4
#define KNN_SUB4 2
#endif
#ifdef KNN_WPE          // experiment: a register budget that leaves room for another kernel's wave beside the workgroup's four per SIMD
__global__ __launch_bounds__(KNN_T) __attribute__((amdgpu_waves_per_eu(KNN_WPE, KNN_WPE))) void k_knn_pairs(RenderArgs a, Workspace w) {
#else
__global__ __launch_bounds__(KNN_T) void k_knn_pairs(RenderArgs a, Workspace w) {
#endif
    constexpr int dbg = KNN_DBG;
    // all LDS is dynamic (a static __shared__ in front would misalign the float4 region, guide G17): vertices, then cluster records
    extern __shared__ __attribute__((aligned(16))) float4 lds_raw[];
    float4* lds = lds_raw + KNN_LDS_HDR;
    int* s_carve = reinterpret_cast<int*>(lds_raw);          // [voff | coff | soff | len] x 5: read per part by the (rolled) part loop
    const KnnIndex& ix = w.knn;
    KP_DECL
    // LDS carve from the (device-resident) part lengths: [vertices of part 0..4 | records of part 0..4]
    KnnLds L;
    int roff = 0;
    {
        int off = 0;
        for (int p = 0; p < INVR_NUM_PARTS; ++p) {
            L.len[p] = min((int)a.scene.lengths2[p], PREP_MAX);
            L.voff[p] = off;
            off += (L.len[p] + 63) / 64 * 64;
        }
        if (off > KNN_LDS_MAX_V) {                       // does not fit the LDS-resident design
            knn_pairs_bf<KNN_T>(a, w, lds);
            return;
        }
        roff = off;                                      // the 16-bit rows of all vertex slots: slot j of part p at 2-byte entry voff[p] + j
        off += off / 8;
        for (int p = 0; p < INVR_NUM_PARTS; ++p) { L.coff[p] = off; off += (L.len[p] + 63) / 64 * 3; }
        for (int p = 0; p < INVR_NUM_PARTS; ++p) { L.soff[p] = off; off += (L.len[p] + 63) / 64 * 8; }
        if (threadIdx.x == 0)
            for (int p = 0; p < INVR_NUM_PARTS; ++p) {
                s_carve[p] = L.voff[p]; s_carve[5 + p] = L.coff[p]; s_carve[10 + p] = L.soff[p]; s_carve[15 + p] = L.len[p];
            }
    }
    // stage vertices and cluster records
    for (int p = 0; p < INVR_NUM_PARTS; ++p) {
        const int len = L.len[p], ncl = (len + 63) >> 6;
        for (int j = threadIdx.x; j < ncl * 64; j += KNN_T) lds[L.voff[p] + j] = ix.sverts[(int64_t)p * ix.mpad + j];
        for (int j = threadIdx.x; j < ncl * 8; j += KNN_T)            // (64 rows of 2 bytes = 8 float4 per cluster; mpad is a multiple of 64)
            lds[roff + L.voff[p] / 8 + j] = reinterpret_cast<const float4*>(ix.srow + (int64_t)p * ix.mpad)[j];
        for (int j = threadIdx.x; j < ncl * 3; j += KNN_T) lds[L.coff[p] + j] = ix.cl[(int64_t)p * ix.cpad * 3 + j];
#if KNN_SUB4
        // the four sub-cluster boxes of a cluster TRANSPOSED, {lo.x[4], lo.y[4], lo.z[4], hi.x[4], hi.y[4], hi.z[4]} (6 of the cluster's 8
        // float4 slots): the sweep tests all four with packed-fp32 arithmetic, two boxes per instruction
        for (int j = threadIdx.x; j < ncl * 6; j += KNN_T) {
            const int c = j / 6, r = j - c * 6, ax = r % 3, hi = r / 3;
            const float* g = reinterpret_cast<const float*>(ix.sub + (int64_t)p * ix.cpad * 8 + c * 8 + hi) + ax;     // box s at + s * 8 floats
            lds[L.soff[p] + c * 8 + r] = make_float4(g[0], g[8], g[16], g[24]);
        }
#else
        for (int j = threadIdx.x; j < ncl * 8; j += KNN_T) lds[L.soff[p] + j] = ix.sub[(int64_t)p * ix.cpad * 8 + j];
#endif
    }
    __syncthreads();
    KP(6)
    const int na = w.counters[CNT_ACTIVE];
    const float dfar2 = ix.dfar2[0];                        // far-fold distance^2 of this frame (k_part_prepare)
    const int lane = threadIdx.x & 63;
    int far_cnt[INVR_NUM_PARTS] = {0, 0, 0, 0, 0};          // far pairs seen by this wave (statistics), flushed once at the end
    // dynamic scheduling, one ticket per wave and 64 survivors: the cost of 64 points varies ~10x with how many clusters they
    // have to sweep.  A wave's first ticket is its index; further tickets come from 16 counters on separate cache lines
    // (workgroup b draws from counter b % 16, which owns the tickets = b mod 16): one counter for all 4096 waves serialised
    // the returned atomics at ~15 ns each and, ordered in front of the wave's loads, slowed every phase of the kernel.
    const int n_wave = (int)gridDim.x * (KNN_T / 64);
    const int n_cls = min(16, (int)gridDim.x), my_cls = (int)blockIdx.x % n_cls;
    int32_t* my_counter = w.counters + CNT_TICKETS + my_cls * 32;
    int ticket = (int)blockIdx.x * (KNN_T / 64) + (int)(threadIdx.x >> 6);
    constexpr int tsize = 64;
    while ((int64_t)ticket * tsize < na) {
        KP_CNT(8)
        const int64_t slot = (int64_t)ticket * tsize + lane;
        const bool live = lane < tsize && slot < na;
        float px = 0, py = 0, pz = 0;
        if (live) sample_pose_point(a, w.active_idx[slot], px, py, pz, nullptr, nullptr);
        unsigned flags = 0, farflags = 0;
        // class of the lattice cell the point lies in (0 = undecided for every part when outside / disabled)
        int vcell = -1;
        if (ix.voxcls && live) {
            const VolDev& v = a.scene.pbw;
            const float ux = (px - v.bounds[0]) / (v.bounds[3] - v.bounds[0]) * (float)(v.dx - 1);
            const float uy = (py - v.bounds[1]) / (v.bounds[4] - v.bounds[1]) * (float)(v.dy - 1);
            const float uz = (pz - v.bounds[2]) / (v.bounds[5] - v.bounds[2]) * (float)(v.dz - 1);
            if (ux >= 0.0f && uy >= 0.0f && uz >= 0.0f && ux <= (float)(v.dx - 1) && uy <= (float)(v.dy - 1) && uz <= (float)(v.dz - 1))
                vcell = ((int)ux * v.dy + (int)uy) * v.dz + (int)uz;
        }
        // the cell's records of all five parts in ONE round trip (15 independent loads): the classes were written by the kernel
        // that ran just before this one on another XCD's L2, so a wave's first ticket misses all the way to memory — and fetched
        // part by part inside the loop that was up to 15 DEPENDENT misses (a 1/8 shard has ~1 ticket per wave: 45 % of its
        // kernel time).  Class 0 = the cell was never classified (not live; a survivor on a cell face can land there): its
        // mask / bound are stale and are not used; class 3 = classified, undecided.
        unsigned cell_cls[INVR_NUM_PARTS];
        unsigned long long cell_mask[INVR_NUM_PARTS];
        float cell_u2[INVR_NUM_PARTS];
#pragma unroll
        for (int p = 0; p < INVR_NUM_PARTS; ++p) {
            cell_cls[p] = 0u; cell_mask[p] = ~0ull; cell_u2[p] = __builtin_inff();
            if (vcell >= 0) {
                cell_cls[p] = ix.voxcls[(int64_t)vcell * INVR_NUM_PARTS + p];
                if (ix.voxmask) { cell_mask[p] = ix.voxmask[(int64_t)vcell * INVR_NUM_PARTS + p]; cell_u2[p] = ix.voxu2[(int64_t)vcell * INVR_NUM_PARTS + p]; }
            }
        }
        KP(1)
        // NOT unrolled: five copies of the classification + sweep made the kernel 70 KB of code — more than the 64 KB instruction
        // cache two CUs share — and its waves, spread over the copies, stalled on instruction fetch
#pragma unroll 1
        for (int p = 0; p < INVR_NUM_PARTS; ++p) {
            KP(2)
            const int len = __builtin_amdgcn_readfirstlane(s_carve[15 + p]);
            const int L_voff = __builtin_amdgcn_readfirstlane(s_carve[p]), L_coff = __builtin_amdgcn_readfirstlane(s_carve[5 + p]),
                      L_soff = __builtin_amdgcn_readfirstlane(s_carve[10 + p]);
            if (len < KNN_K) continue;                     // reference: inf distances -> NaN dist -> unflagged
            // (p is wave-uniform: the selects below are four scalar-conditioned moves per value)
            const unsigned c2 = p == 0 ? cell_cls[0] : p == 1 ? cell_cls[1] : p == 2 ? cell_cls[2] : p == 3 ? cell_cls[3] : cell_cls[4];
            const bool undecided = c2 == 0u || c2 == 3u;
            if (__ballot(live && undecided) == 0) {                 // every live lane sits in a decided cell
                if (c2 == 1) farflags |= 1u << p;
                continue;
            }
            const float* bb = ix.part_aabb + p * 6;
            const float lbp = aabb_dist2(px, py, pz, make_float4(bb[0], bb[1], bb[2], 0.f), make_float4(bb[3], bb[4], bb[5], 0.f));
            if (__ballot(live && !(lbp > dfar2)) == 0) {          // whole wave far from this part
                if (live) farflags |= 1u << p;
                continue;
            }
            const int ncl = (len + 63) >> 6;
            const float4* cl = lds + L_coff;                       // records {lo, hi, rep}
            const float4* sv = lds + L_voff;
            const unsigned* rw = reinterpret_cast<const unsigned*>(lds + roff) + L_voff / 2;      // two 16-bit rows per pair record
            // candidate clusters of the wave: OR of the lattice-cell masks of its undecided lanes (all ones when a lane is
            // outside the lattice, the masks are off or the part has more than 64 clusters).  Lanes in decided cells take the
            // cell's class and request nothing.
            const bool maybe = live && undecided;
            unsigned long long mym = 0ull;
            if (maybe && c2 == 3u) mym = p == 0 ? cell_mask[0] : p == 1 ? cell_mask[1] : p == 2 ? cell_mask[2] : p == 3 ? cell_mask[3] : cell_mask[4];
            else if (maybe) mym = ~0ull;
            // (every lane of the wave is active here: `live` / `maybe` are predicates, the control flow around is wave-uniform)
            const unsigned long long M = ((unsigned long long)wave_or_u32((unsigned)(mym >> 32)) << 32) | wave_or_u32((unsigned)mym);
            const bool full = M == ~0ull || ncl > 64;
            const unsigned long long Mc = ncl >= 64 ? M : (M & ((1ull << ncl) - 1ull));
            // bounds on the nearest-vertex distance from the cluster records
            float lb2 = __builtin_inff(), ub2 = __builtin_inff();
            int seed = 0;
            if (full) {
                for (int c0 = 0; c0 < ncl; c0 += 4) {
                    float4 rec[12];
#pragma unroll
                    for (int k = 0; k < 12; ++k) rec[k] = lds_ld4(cl + min(c0 + k / 3, ncl - 1) * 3 + (k % 3));    // wave-uniform
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float4 r = rec[k * 3 + 2];
                        const float dx = px - r.x, dy = py - r.y, dz = pz - r.z;
                        const float u = dx * dx + dy * dy + dz * dz;
                        if (u < ub2) { ub2 = u; seed = min(c0 + k, ncl - 1); }
                        lb2 = fminf(lb2, aabb_dist2(px, py, pz, rec[k * 3], rec[k * 3 + 1]));
                    }
                }
            } else {
                for (unsigned long long it = Mc; it; it &= it - 1ull) {
                    const int c = __ffsll((long long)it) - 1;
                    const float4 klo = lds_ld4(cl + c * 3), khi = lds_ld4(cl + c * 3 + 1), r = lds_ld4(cl + c * 3 + 2);
                    const float dx = px - r.x, dy = py - r.y, dz = pz - r.z;
                    const float u = dx * dx + dy * dy + dz * dz;
                    if (u < ub2) { ub2 = u; seed = c; }
                    lb2 = fminf(lb2, aabb_dist2(px, py, pz, klo, khi));
                }
            }
            const bool is_far = maybe ? lb2 > dfar2 : (live && c2 == 1);
            const bool unflagged = maybe ? (lb2 >= a.scene.near_hi2 && ub2 <= a.scene.band_lo2) : (c2 == 2);
            const bool scan = live && !is_far && !unflagged;
            if (live && is_far) farflags |= 1u << p;
            if (__ballot(scan) == 0 || (dbg & 1)) continue;
            KP(2)
            KP_CNT(9)
            KP_MARK
            // exact 4-NN: seed with the cluster of the wave's first scanning lane, then a pruned sweep that
            // walks outwards from the seed in Morton order (neighbouring indices are mostly neighbouring
            // patches, so the 4th-best bound tightens early); clusters are pruned as a whole and then per
            // 16-vertex sub-cluster
            Top4 t;
            t.init();
            if (!full && scan) {
                // lattice cell bound: at least 4 vertices of the part lie within sqrt(u2) of every point of the cell, so four
                // placeholder entries (u2, row 0xFFFFFFFF) prune from the first vertex on and are all evicted by real ones
                const unsigned long long ph = ((unsigned long long)__float_as_uint(p == 0 ? cell_u2[0] : p == 1 ? cell_u2[1] : p == 2 ? cell_u2[2] : p == 3 ? cell_u2[3] : cell_u2[4]) << 32) | 0xFFFFFFFFull;
#pragma unroll
                for (int j = 0; j < KNN_K; ++j) t.k[j] = ph;
            }
            const float4* sb = lds + L_soff;
            const v2f px2 = {px, px}, py2 = {py, py}, pz2 = {pz, pz};
            const int seed_c = __builtin_amdgcn_readlane(seed, __ffsll((long long)__ballot(scan)) - 1);
            if (full) {                                  // no bound to start from: the seed cluster is scanned unconditionally
#pragma unroll 1
                for (int s4 = 0; s4 < 4; ++s4) scan_sub16(sv + seed_c * 64 + s4 * 16, rw + seed_c * 32 + s4 * 8, px2, py2, pz2, t KP_SCAN_PASS);
            }
            // prefilter constants of this (lane, part): a = -2 q, qd = |q|^2 - delta with delta = 2^-19 (|q| + max |v|)^2 (scan_sub16_pf)
            const v2f ax2 = {-2.0f * px, -2.0f * px}, ay2 = {-2.0f * py, -2.0f * py}, az2 = {-2.0f * pz, -2.0f * pz};
            const float q2 = (px * px + py * py) + pz * pz;
            const float vmx = fmaxf(fabsf(bb[0]), fabsf(bb[3])), vmy = fmaxf(fabsf(bb[1]), fabsf(bb[4])), vmz = fmaxf(fabsf(bb[2]), fabsf(bb[5]));
            const float mm = sqrtf(q2) + sqrtf((vmx * vmx + vmy * vmy) + vmz * vmz);
            const float qd = q2 - mm * mm * 1.9073486328125e-06f;
            // (lanes that do not scan this part — decided cell, far, provably unflagged — never pass the prefilter: they used to keep a
            // top-4 of their own and pulled the wave into the insert branch for it)
            float thr = (scan && !(dbg & 8)) ? t.worst() - qd : -__builtin_inff();
#pragma unroll 1
            for (int k = full ? 1 : 0; (seed_c + k < ncl || seed_c - k >= 0) && !(dbg & 2); ++k) {
#pragma unroll 1
                for (int side = 0; side < (k ? 2 : 1); ++side) {
                    const int c = side ? seed_c - k : seed_c + k;
                    if (c < 0 || c >= ncl) continue;
                    if (!full && !((Mc >> c) & 1ull)) continue;                   // no lane of the wave can have a neighbour there
                    const bool need = scan && aabb_dist2(px, py, pz, lds_ld4(cl + c * 3), lds_ld4(cl + c * 3 + 1)) <= t.worst();
                    if (__ballot(need) == 0) continue;
                    KP_CNT(10)
#if KNN_SUB4
                    // the box distances of the cluster's four sub-clusters at once: six record reads in flight instead of four round
                    // trips of two, two boxes per packed instruction — each value the operations of aabb_dist2 in its order, so a
                    // sub-cluster is scanned exactly when the one-at-a-time test (against the 4th-best of that moment) scans it
                    float lbs[4];
                    {
                        const float4 LX = lds_ld4(sb + c * 8), LY = lds_ld4(sb + c * 8 + 1), LZ = lds_ld4(sb + c * 8 + 2);
                        const float4 HX = lds_ld4(sb + c * 8 + 3), HY = lds_ld4(sb + c * 8 + 4), HZ = lds_ld4(sb + c * 8 + 5);
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const v2f lx = h ? (v2f){LX.z, LX.w} : (v2f){LX.x, LX.y}, hx = h ? (v2f){HX.z, HX.w} : (v2f){HX.x, HX.y};
                            const v2f ly = h ? (v2f){LY.z, LY.w} : (v2f){LY.x, LY.y}, hy = h ? (v2f){HY.z, HY.w} : (v2f){HY.x, HY.y};
                            const v2f lz = h ? (v2f){LZ.z, LZ.w} : (v2f){LZ.x, LZ.y}, hz = h ? (v2f){HZ.z, HZ.w} : (v2f){HZ.x, HZ.y};
                            const v2f a0 = lx - px2, b0 = px2 - hx, a1 = ly - py2, b1 = py2 - hy, a2 = lz - pz2, b2 = pz2 - hz;
                            const v2f ex = {fmaxf(fmaxf(a0.x, b0.x), 0.0f), fmaxf(fmaxf(a0.y, b0.y), 0.0f)};
                            const v2f ey = {fmaxf(fmaxf(a1.x, b1.x), 0.0f), fmaxf(fmaxf(a1.y, b1.y), 0.0f)};
                            const v2f ez = {fmaxf(fmaxf(a2.x, b2.x), 0.0f), fmaxf(fmaxf(a2.y, b2.y), 0.0f)};
                            const v2f l2 = (ex * ex + ey * ey) + ez * ez;
                            lbs[2 * h] = l2.x; lbs[2 * h + 1] = l2.y;
                        }
                    }
                    // (rolled: four inlined copies of the scan are +40 % code in a kernel that sits at the instruction cache's size;
                    // s4 is wave-uniform, the select is three scalar-conditioned moves)
#if KNN_SUB4 == 2
#pragma unroll
#else
#pragma unroll 1
#endif
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const float lb_s = s4 == 0 ? lbs[0] : s4 == 1 ? lbs[1] : s4 == 2 ? lbs[2] : lbs[3];
                        const bool need_s = need && lb_s <= t.worst();
                        if (__ballot(need_s) == 0) continue;
                        KP_CNT(11)
                        if (dbg & 4) continue;
                        scan_sub16_pf(sv + c * 64 + s4 * 16, rw + c * 32 + s4 * 8, ax2, ay2, az2, px2, py2, pz2, qd, thr, t KP_SCAN_PASS);
                    }
#else
#pragma unroll 1
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const bool need_s = need && aabb_dist2(px, py, pz, lds_ld4(sb + c * 8 + s4 * 2), lds_ld4(sb + c * 8 + s4 * 2 + 1)) <= t.worst();
                        if (__ballot(need_s) == 0) continue;
                        KP_CNT(11)
                        scan_sub16_pf(sv + c * 64 + s4 * 16, rw + c * 32 + s4 * 8, ax2, ay2, az2, px2, py2, pz2, qd, thr, t KP_SCAN_PASS);
                    }
#endif
                }
            }
            KP(3)
            t.finish();
            KP_BAND(__ballot(scan && t.d[0] < a.scene.near_hi2) == 0ull)      // (profiling builds: sub-cluster scans of part scans without a near lane)
#pragma unroll
            for (int j = 0; j < KNN_K; ++j) t.i[j] = min(t.i[j] & 0x7FFFFFFF, len - 1);      // (a surviving placeholder would be a bug; never index out of the part)
            float wt[KNN_K];
            const float ds = knn_weights(t, wt);
            if (scan && ds < a.scene.thresh) {                        // pflag (inb_part_network_multiassign.py:90)
                flags |= 1u << p;
                reinterpret_cast<int4*>(w.l_nn[p])[slot] = make_int4(t.i[0], t.i[1], t.i[2], t.i[3]);
                reinterpret_cast<float4*>(w.l_w[p])[slot] = make_float4(wt[0], wt[1], wt[2], wt[3]);
            }
            KP(4)
        }
        KP(2)
        int my_cnt = 0;                              // lane p < 5: pairs of this ticket flagged for part p
#pragma unroll
        for (int p = 0; p < INVR_NUM_PARTS; ++p) {
            far_cnt[p] += __popcll(__ballot((farflags >> p) & 1u));
            const int c = __popcll(__ballot((flags >> p) & 1u));
            if (lane == p) my_cnt = c;
        }
        if (my_cnt) atomicAdd(&w.gcount[(slot - lane) / PAIR_GROUP * INVR_NUM_PARTS + lane], my_cnt);      // (64 | PAIR_GROUP; not returned)
        if (live) {
            w.pflags[slot] = (uint8_t)flags;
            w.farflags[slot] = (uint8_t)farflags;
        }
        KP(5)
        int next = 0;
        if (lane == 0) next = atomicAdd(my_counter, 1);
        ticket = n_wave + __builtin_amdgcn_readfirstlane(next) * n_cls + my_cls;
        KP(0)
    }
    if (lane == 0)
#pragma unroll
        for (int p = 0; p < INVR_NUM_PARTS; ++p)
            if (far_cnt[p]) atomicAdd(&w.counters[CNT_FAR + p], far_cnt[p]);
    KP_FLUSH
    // The 16 waves of a workgroup leave together (round 6).  A workgroup fills its CU's LDS and its four SIMDs' registers (4 x 128), so
    // no wave of another kernel ever runs beside this kernel's packed-fp32 sweep — except in the tail, when waves that ran out of
    // tickets exited and freed their registers while their neighbours were still sweeping.  Packed-fp32 sequences beside waves of
    // other kernels are what produced the wrong lanes of profiles/r6_replay_mismatch.md in k_warp_pairs; this kernel never showed
    // them (0 differing neighbour rows / weights in 30,000 stressed frames), and the barrier keeps it that way by construction.
    __syncthreads();
}

// The per-part pair lists from the flag bytes k_knn_pairs left per survivor: l_slot[p][0..count) = the survivors flagged for
// part p, ASCENDING (a deterministic order: consecutive pairs are neighbours in space, k_cull.hip); neighbours and weights stay
// where the KNN wrote them — at the survivor's slot.  One workgroup per group of PAIR_GROUP slots; its list offsets are the sums
// of the per-group counts the KNN accumulated (gcount) over the groups before it — no atomics here (device-scope atomics on one
// address serialise at tens of ns each on the 8-XCD part: 1167 claims per part cost this kernel 90 us).  The workgroup of the
// last group appends one zero-weight pair per part behind the real ones — its field value is the constant every far pair of
// that part takes (header comment); it lives in the extra slot `cap` — and exports the counters (no extra launches).
#define PL_BLOCK 256
#define PL_PER 8
__global__ __launch_bounds__(PL_BLOCK) void k_pair_lists(Workspace w, int32_t* __restrict__ stats) {
    __shared__ int s_cnt[PL_BLOCK / 64][INVR_NUM_PARTS];
    __shared__ int s_red[PL_BLOCK / 64][INVR_NUM_PARTS];
    const int na = w.counters[CNT_ACTIVE];
    const int64_t g = blockIdx.x, g_last = (max(na, 1) - 1) / PAIR_GROUP;
    if (g > g_last) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // list offsets of this group = counts of the groups before it
    int base[INVR_NUM_PARTS];
    {
        int acc[INVR_NUM_PARTS] = {0, 0, 0, 0, 0};
        for (int64_t q = threadIdx.x; q < g; q += PL_BLOCK)
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
            for (int k = 0; k < PL_BLOCK / 64; ++k) base[p] += s_red[k][p];
        }
    }
    for (int64_t t0 = g * PAIR_GROUP; t0 < min((g + 1) * (int64_t)PAIR_GROUP, (int64_t)na); t0 += PL_BLOCK * PL_PER) {
        // thread -> PL_PER consecutive survivors (one 8-byte load of flag bytes); ranks: inside the thread, the wave, the block
        const int64_t s0 = t0 + (int64_t)threadIdx.x * PL_PER;
        unsigned long long fb = 0ull;
        if (s0 + PL_PER <= na) fb = *reinterpret_cast<const unsigned long long*>(w.pflags + s0);
        else for (int k = 0; k < PL_PER; ++k) if (s0 + k < na) fb |= (unsigned long long)w.pflags[s0 + k] << (8 * k);
        int pre[INVR_NUM_PARTS];
#pragma unroll
        for (int p = 0; p < INVR_NUM_PARTS; ++p) {
            const int mine = __popcll(fb & (0x0101010101010101ull << p));
            const int x = wave_incl_sum_i(mine);
            pre[p] = x - mine;
            if (lane == 63) s_cnt[wv][p] = x;
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < INVR_NUM_PARTS; ++p) {
            int pos = base[p] + pre[p];
            for (int k = 0; k < PL_BLOCK / 64; ++k) { const int c = s_cnt[k][p]; if (k < wv) pos += c; base[p] += c; }
            for (int k = 0; k < PL_PER; ++k)
                if ((fb >> (8 * k + p)) & 1ull) w.l_slot[p][pos++] = (int32_t)(s0 + k);
        }
        __syncthreads();                 // s_cnt is reused by the next tile
    }
    if (g != g_last) return;
    // epilogue: base[] = the totals
    if (threadIdx.x < INVR_NUM_PARTS) {
        const int p = threadIdx.x;
        int tot = 0;
#pragma unroll
        for (int q = 0; q < INVR_NUM_PARTS; ++q) if (q == p) tot = base[q];
        w.l_slot[p][tot] = (int32_t)w.cap;
        reinterpret_cast<int4*>(w.l_nn[p])[w.cap] = make_int4(0, 0, 0, 0);
        reinterpret_cast<float4*>(w.l_w[p])[w.cap] = make_float4(0.f, 0.f, 0.f, 0.f);
        w.counters[CNT_PAIRS + p] = tot + 1;
        if (p == 0) w.active_idx[w.cap] = 0;
    }
    __syncthreads();
    if (stats && threadIdx.x < INVR_STATS_LEN) stats[threadIdx.x] = threadIdx.x < CNT_LEN ? w.counters[threadIdx.x] : 0;
}

// Per-frame classification of the distance-volume lattice cells (side stream, after k_part_prepare): for every cell
// and part, from the box-to-box distance to the cluster AABBs and the cell-centre distance to the cluster
// representatives, whether EVERY point of the cell is a far pair (lower bound > 0.68 m) or provably unflagged
// (lower bound >= near_hi and some vertex within band_lo) — the same two tests k_knn_pairs applies per point, so a
// definite cell class is exactly what the per-point classification would conclude; undecided cells (class 0) and
// points outside the lattice run the per-point cluster loop.  17x fewer cells than survivors on the bench frame.
// Four lanes per (live cell, part): with one thread per item the kernel was a single latency-bound wave per SIMD (13 % VALU
// busy, 50 us whatever the frame size, and on the critical path of small frames / ray shards: the KNN waits for it) — every
// loop over the part's clusters is split over the quad and reduced with two shuffles; all reductions are order-independent
// (min, OR, lexicographic (distance, id) top-3, the 4th-smallest of a multiset), so the classes, masks and bounds are the
// one-thread kernel's bit for bit.
#define VC_BLOCK 256
#define VC_Q 4
__device__ __forceinline__ float quad_min(float x) { x = fminf(x, __shfl_xor(x, 1)); return fminf(x, __shfl_xor(x, 2)); }

// (body: workgroup bx of gx of part p)
__device__ __forceinline__ void voxel_class_body(const SceneDev& s, const KnnIndex& ix, const int32_t* __restrict__ n_live_dev,
                                                 const int bx, const int n_bx, const int p) {
    const VolDev& v = s.pbw;
    // the live cells — a corner below the cull threshold, 7 % of the lattice on the bench frame; all other cells are never looked
    // up — were listed by k_cull_cells
    const int n_live = n_live_dev[0];
    const float dfar2 = ix.dfar2[0];
    const int per_block = VC_BLOCK / VC_Q;
    if (bx * per_block >= n_live) return;
    // cluster {lo, hi, rep} and sub-cluster {lo, hi} records of this block's part, staged once
    __shared__ float4 s_cl[(PREP_MAX / 64) * 3];
    __shared__ float4 s_sub[64 * 8];                               // (only read for parts with <= 64 clusters: the mask path)
    const int len = min((int)s.lengths2[p], PREP_MAX), ncl = (len + 63) >> 6;
    for (int j = threadIdx.x; j < ncl * 3; j += VC_BLOCK) s_cl[j] = ix.cl[(int64_t)p * ix.cpad * 3 + j];
    if (ncl <= 64)
        for (int j = threadIdx.x; j < ncl * 8; j += VC_BLOCK) s_sub[j] = ix.sub[(int64_t)p * ix.cpad * 8 + j];
    __syncthreads();
    const int q = threadIdx.x & (VC_Q - 1);
    const int qbase = (threadIdx.x & 63) & ~(VC_Q - 1);            // lane of the quad's first thread inside its wave
    for (int e = bx * per_block + (int)(threadIdx.x >> 2); e < n_live; e += n_bx * per_block) {
        const int idx = ix.live_cells[e];
        // the cell's box, inflated: the point -> cell map of k_knn_pairs is approximate
        auto cell_box = [&](int cell, float* blo, float* bhi) {
            const int z0 = cell % v.dz, y0 = (cell / v.dz) % v.dy, x0 = cell / (v.dz * v.dy);
            const int c0[3] = {x0, y0, z0}, dims[3] = {v.dx, v.dy, v.dz};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float b0 = v.bounds[a], ext = v.bounds[3 + a] - b0, den = (float)(dims[a] - 1);
                blo[a] = b0 + ext * ((float)c0[a] / den) - 1e-4f;
                bhi[a] = b0 + ext * ((float)min(c0[a] + 1, dims[a] - 1) / den) + 1e-4f;
            }
        };
        float lo[3], hi[3], ce[3], h2 = 0.0f;
        cell_box(idx, lo, hi);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            ce[a] = 0.5f * (lo[a] + hi[a]);
            h2 += 0.25f * (hi[a] - lo[a]) * (hi[a] - lo[a]);
        }
        const float h = sqrtf(h2) * 1.0001f;
        // bounds on the nearest-vertex distance of any point of the cell: box-to-box (lower), centre-to-representative + half
        // diagonal (upper) — the two tests k_knn_pairs applies per point
        float lb2 = __builtin_inff(), ub2 = __builtin_inff();
#pragma unroll 1
        for (int c = q; c < ncl; c += VC_Q) {
            const float4 klo = s_cl[c * 3], khi = s_cl[c * 3 + 1];
            const float4 rep = s_cl[c * 3 + 2];
            const float gx = fmaxf(fmaxf(klo.x - hi[0], lo[0] - khi.x), 0.0f);
            const float gy = fmaxf(fmaxf(klo.y - hi[1], lo[1] - khi.y), 0.0f);
            const float gz = fmaxf(fmaxf(klo.z - hi[2], lo[2] - khi.z), 0.0f);
            lb2 = fminf(lb2, (gx * gx + gy * gy + gz * gz) * 0.9999f);
            const float dx = ce[0] - rep.x, dy = ce[1] - rep.y, dz = ce[2] - rep.z;
            const float u = sqrtf(dx * dx + dy * dy + dz * dz) + h;
            ub2 = fminf(ub2, u * u * 1.0001f);
        }
        lb2 = quad_min(lb2);
        ub2 = quad_min(ub2);
        unsigned pc = 0;
        if (len >= KNN_K) {
            if (lb2 > dfar2) pc = 1;
            else if (lb2 >= s.near_hi2 && ub2 <= s.band_lo2) pc = 2;
        }
        if (q == 0) ix.voxcls[(int64_t)idx * INVR_NUM_PARTS + p] = (uint8_t)(pc ? pc : 3u);      // 3 = classified, undecided (0 = never classified)
        if (!ix.voxmask) continue;
        // undecided cell: which clusters can hold one of the 4 nearest vertices of ANY point x of the cell?  d4(x) <=
        // D4(centre) + h, so only clusters whose box comes within that of the cell box; the cluster of x's nearest
        // vertex is always among them, so min-over-candidates of the per-point box bounds equals the min over all.
        unsigned long long mask = ~0ull;
        float u2_out = __builtin_inff();
        if (pc == 0 && len >= KNN_K && ncl <= 64 && s.thresh < 1e8f) {      // (dense stress mode: every cell holds survivors, masks off)
            // upper bound of the 4th-nearest distance from the cell centre: every 16-vertex sub-cluster box with >= 4 real
            // vertices holds 4 vertices within the distance to its farthest corner.  The three sub-clusters with the smallest
            // farthest-corner distance, ties to the smaller id: per lane over its clusters, then merged over the quad.
            float b0 = __builtin_inff(), b1 = b0, b2 = b0;
            int i0 = 0x7fffffff, i1 = i0, i2 = i0;
            auto ins3 = [&](float f2, int id) {
                if (f2 < b2 || (f2 == b2 && id < i2)) {
                    b2 = f2; i2 = id;
                    if (b2 < b1 || (b2 == b1 && i2 < i1)) { const float tf = b1; b1 = b2; b2 = tf; const int ti = i1; i1 = i2; i2 = ti; }
                    if (b1 < b0 || (b1 == b0 && i1 < i0)) { const float tf = b0; b0 = b1; b1 = tf; const int ti = i0; i0 = i1; i1 = ti; }
                }
            };
#pragma unroll 1
            for (int c = q; c < ncl; c += VC_Q) {
                // a sub-cluster's farthest-corner distance is at least the centre's distance to the cluster box: clusters beyond
                // the lane's current third-best cannot enter
                const float4 klo = s_cl[c * 3], khi = s_cl[c * 3 + 1];
                const float qx = fmaxf(fmaxf(klo.x - ce[0], ce[0] - khi.x), 0.0f);
                const float qy = fmaxf(fmaxf(klo.y - ce[1], ce[1] - khi.y), 0.0f);
                const float qz = fmaxf(fmaxf(klo.z - ce[2], ce[2] - khi.z), 0.0f);
                if (qx * qx + qy * qy + qz * qz > b2) continue;
                for (int s4 = 0; s4 < 4; ++s4) {
                    if (len - (c * 64 + s4 * 16) < KNN_K) continue;
                    const float4 slo = s_sub[c * 8 + s4 * 2], shi = s_sub[c * 8 + s4 * 2 + 1];
                    const float fx = fmaxf(fabsf(ce[0] - slo.x), fabsf(ce[0] - shi.x));
                    const float fy = fmaxf(fabsf(ce[1] - slo.y), fabsf(ce[1] - shi.y));
                    const float fz = fmaxf(fabsf(ce[2] - slo.z), fabsf(ce[2] - shi.z));
                    ins3((fx * fx + fy * fy + fz * fz) * 1.0001f, c * 4 + s4);
                }
            }
            {   // merge the four lanes' triples (every lane ends with the quad's three best)
                const float mb[3] = {b0, b1, b2};
                const int mi[3] = {i0, i1, i2};
                b0 = b1 = b2 = __builtin_inff();
                i0 = i1 = i2 = 0x7fffffff;
#pragma unroll
                for (int src = 0; src < VC_Q; ++src)
#pragma unroll
                    for (int t3 = 0; t3 < 3; ++t3) ins3(__shfl(mb[t3], qbase + src), __shfl(mi[t3], qbase + src));
            }
            float k3 = b0;                  // the smallest farthest-corner distance of any sub-cluster
            {   // tighten: the 4th-smallest exact distance among the (up to 48) vertices of those three sub-clusters is still an
                // upper bound of the 4th-nearest distance from the centre, usually the exact one.  Lane q takes vertices q, q+4, ..
                float e0 = __builtin_inff(), e1 = e0, e2 = e0, e3 = e0;
                auto ins4 = [&](float d) {
                    if (d < e3) {
                        e3 = d;
                        if (e3 < e2) { const float tf = e2; e2 = e3; e3 = tf; }
                        if (e2 < e1) { const float tf = e1; e1 = e2; e2 = tf; }
                        if (e1 < e0) { const float tf = e0; e0 = e1; e1 = tf; }
                    }
                };
                const int ids[3] = {i0, i1, i2};
                // one sub-cluster at a time (8 loads in flight): all 24 loads up front cost 96 registers, and this body shares its
                // launch — and with it its register allocation — with the cull flags, which need 8 waves per SIMD (k_front_cull)
#pragma unroll 1
                for (int t3 = 0; t3 < 3; ++t3) {
                    if (ids[t3] == 0x7fffffff) continue;
                    // the lane's 4 vertices of the sub-cluster sit in pair records (q>>1) + 2m
                    const float4* rec = ix.sverts + (int64_t)p * ix.mpad + ids[t3] * 16;
                    float4 ra[4];
                    float2 rb[4];                                  // {z0, z1} of the pair record's second half
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        ra[m] = rec[2 * ((q >> 1) + 2 * m)];
                        rb[m] = *reinterpret_cast<const float2*>(rec + 2 * ((q >> 1) + 2 * m) + 1);
                    }
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const float4 A = ra[m];
                        const float2 B = rb[m];
                        const float vx = (q & 1) ? A.y : A.x, vy = (q & 1) ? A.w : A.z, vz = (q & 1) ? B.y : B.x;
                        const float dx = ce[0] - vx, dy = ce[1] - vy, dz = ce[2] - vz;
                        ins4(dx * dx + dy * dy + dz * dz);
                    }
                }
                const float me[4] = {e0, e1, e2, e3};
                e0 = e1 = e2 = e3 = __builtin_inff();
#pragma unroll
                for (int src = 0; src < VC_Q; ++src)
#pragma unroll
                    for (int t4 = 0; t4 < 4; ++t4) ins4(__shfl(me[t4], qbase + src));
                k3 = fminf(k3, e3 * 1.0001f);
            }
            const float u = sqrtf(k3) + h;
            const float u2 = u * u * 1.0002f;
            u2_out = u2;
            unsigned mlo = 0u, mhi = 0u;
            // (the box is formed again from the cell index, behind a register fence: six registers less across the phases above —
            // this body shares its launch and its register allocation with the cull flags, k_front_cull)
            int idx2 = idx;
            asm volatile("" : "+v"(idx2));
            float lo2[3], hi2[3];
            cell_box(idx2, lo2, hi2);
#pragma unroll 1
            for (int c = q; c < ncl; c += VC_Q) {
                const float4 klo = s_cl[c * 3], khi = s_cl[c * 3 + 1];
                const float gx = fmaxf(fmaxf(klo.x - hi2[0], lo2[0] - khi.x), 0.0f);
                const float gy = fmaxf(fmaxf(klo.y - hi2[1], lo2[1] - khi.y), 0.0f);
                const float gz = fmaxf(fmaxf(klo.z - hi2[2], lo2[2] - khi.z), 0.0f);
                if ((gx * gx + gy * gy + gz * gz) * 0.9999f <= u2) { if (c < 32) mlo |= 1u << c; else mhi |= 1u << (c - 32); }
            }
            mlo |= __shfl_xor(mlo, 1); mlo |= __shfl_xor(mlo, 2);
            mhi |= __shfl_xor(mhi, 1); mhi |= __shfl_xor(mhi, 2);
            mask = ((unsigned long long)mhi << 32) | mlo;
        }
        if (q == 0) {
            ix.voxmask[(int64_t)idx * INVR_NUM_PARTS + p] = mask;
            ix.voxu2[(int64_t)idx * INVR_NUM_PARTS + p] = u2_out;
        }
    }
}

__global__ __launch_bounds__(VC_BLOCK) void k_knn_voxel_class(SceneDev s, KnnIndex ix, const int32_t* __restrict__ n_live_dev) {
    voxel_class_body(s, ix, n_live_dev, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y);
}

// Workgroups per part.  The body walks the LIVE cells (a device-side count, ~7 % of the lattice) with a grid stride, so the grid is sized
// for an eighth of the lattice: a workgroup that finds nothing to do still costs ~2 ns of dispatch, and a grid for every cell was 10 k of
// them (25 us of the 37 us k_front_cull took on a 1/8 ray shard, profiles/r4_front_chain.md).
static unsigned voxel_class_grid(const VolDev& v) {
    const int64_t cells = (int64_t)v.dx * v.dy * v.dz, items = (VC_BLOCK / VC_Q) * 8;
    const int64_t g = cdiv(cells, items);
    return (unsigned)(g < 1 ? 1 : (g < 512 ? g : 512));
}

int launch_knn_voxel_class(const RenderArgs& a, const Workspace& w, hipStream_t st) {
    hipLaunchKernelGGL(k_knn_voxel_class, dim3(voxel_class_grid(a.scene.pbw), INVR_NUM_PARTS), dim3(VC_BLOCK), 0, st,
                       a.scene, w.knn, w.counters + CNT_LIVE);
    INVR_LAUNCH_CHECK();
    return 0;
}

// The per-frame KNN index only depends on the posed vertices, not on the rays.
int launch_knn_prepare(const RenderArgs& a, const Workspace& w, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        INVR_HIP(hipFuncSetAttribute((const void*)k_part_prepare, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PREP_LDS_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL(k_part_prepare, dim3(INVR_NUM_PARTS), dim3(PREP_T), PREP_LDS_BYTES, st, a.scene, w.knn);
    INVR_LAUNCH_CHECK();
    return 0;
}

// ---- the front of a frame as two launches on ONE stream ------------------------------------------------------------------------
// Everything in front of the KNN used to be two chains joined by events: {cell mask -> cull flags -> scan -> compaction} on the
// caller's stream, {index build -> lattice classes -> vertex matrices -> deformer slices} on a library stream.  Under hipGraph replay
// every edge between the two chains costs 10-17 us on this runtime (the kernel behind a cross-stream dependency starts that much
// after its predecessor ends; kernels that follow each other on one stream start back to back): 74 us per frame, a fifth of a 1/8
// ray shard (gpurun_out/r4c).  Independent kernels now share a LAUNCH instead: workgroup ranges of one grid run different bodies.
//   k_front_scene : [5 x index build | cull cell mask + live-cell list | per-vertex matrices | deformer t-slices]   (scene only)
//   k_front_cull  : [lattice-cell classes of the live cells x 5 parts | cull flags of the ray-samples]
// Both bodies of a launch run side by side on the chip; the launch lasts as long as its longest range.
struct FrontSceneArgs {
    SceneDev s; KnnIndex ix;
    float thresh_hi; uint8_t* cullmask; int32_t* n_live;          // cell mask (n_cells_wg == 0: no mask this frame)
    GridDev dg; DfSliceInfo si; float2* dslice;                   // deformer slices (n_slice_wg == 0: they do not fit)
    int n_cells_wg, n_vmat_wg, n_slice_wg, vmat_m;
};
__global__ __launch_bounds__(PREP_T) void k_front_scene(FrontSceneArgs f) {
    extern __shared__ unsigned prep_lds[];
    int b = (int)blockIdx.x;
    if (b < INVR_NUM_PARTS) { part_prepare_body(f.s, f.ix, b, prep_lds); return; }
    b -= INVR_NUM_PARTS;
    if (b < f.n_cells_wg) {
        cull_cells_body(f.s.pbw, f.thresh_hi, f.cullmask, f.ix.live_cells, f.n_live, f.ix.voxcls, b * PREP_T + (int)threadIdx.x);
        return;
    }
    b -= f.n_cells_wg;
    if (b < f.n_vmat_wg * INVR_NUM_PARTS) {
        vertex_mats_body(f.s, f.ix, f.s.A, f.s.big_A, b / f.n_vmat_wg, (b % f.n_vmat_wg) * PREP_T + (int)threadIdx.x);
        return;
    }
    b -= f.n_vmat_wg * INVR_NUM_PARTS;
    deform_slice_body(f.dg, f.si, f.s.frame_dim, f.dslice, b * PREP_T + (int)threadIdx.x);
}

// -> *have_cells = 1 if the cell mask / live-cell list were built (launch_cull_cells' conditions)
int launch_front_scene(const RenderArgs& a, const Workspace& w, const GridDev& dg, int* have_cells, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        INVR_HIP(hipFuncSetAttribute((const void*)k_front_scene, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PREP_LDS_BYTES));
        attr_set = true;
    }
    FrontSceneArgs f;
    f.s = a.scene; f.ix = w.knn;
    const VolDev& v = a.scene.pbw;
    const int64_t cells = (int64_t)v.dx * v.dy * v.dz;
    static const bool no_mask = getenv("INVR_NO_CULLMASK") != nullptr;
    *have_cells = !(cells > CULL_MASK_MAX || cells > VOXMASK_MAX_CELLS || no_mask);
    f.thresh_hi = a.scene.thresh * (1.0f + 1e-5f); f.cullmask = w.cullmask; f.n_live = w.counters + CNT_LIVE;
    f.n_cells_wg = *have_cells ? (int)cdiv(cells, PREP_T) : 0;
    f.vmat_m = a.scene.M < w.knn.mpad ? a.scene.M : w.knn.mpad;
    f.n_vmat_wg = (int)cdiv(f.vmat_m, PREP_T);
    f.dg = dg; f.dslice = w.dslice;
    f.n_slice_wg = deform_slices_fit(dg, f.si, deform_cb()) ? (int)cdiv(f.si.off[8], PREP_T) : 0;
    const unsigned grid = (unsigned)(INVR_NUM_PARTS + f.n_cells_wg + f.n_vmat_wg * INVR_NUM_PARTS + f.n_slice_wg);
    hipLaunchKernelGGL(k_front_scene, dim3(grid), dim3(PREP_T), PREP_LDS_BYTES, st, f);
    INVR_LAUNCH_CHECK();
    return 0;
}

// cullmask OR-dilated by one cell in every direction (27 bytes per cell, L2-resident): d1[c] = 0 means no cell within +-1 of c can hold a survivor
__global__ __launch_bounds__(256) void k_dilate_mask(int dx, int dy, int dz, const uint8_t* __restrict__ mask, uint8_t* __restrict__ d1) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= dx * dy * dz) return;
    const int z0 = i % dz, y0 = (i / dz) % dy, x0 = i / (dz * dy);
    unsigned any = 0;
    for (int x = max(x0 - 1, 0); x <= min(x0 + 1, dx - 1); ++x)
        for (int y = max(y0 - 1, 0); y <= min(y0 + 1, dy - 1); ++y)
#pragma unroll
            for (int dzz = -1; dzz <= 1; ++dzz) {
                const int z = min(max(z0 + dzz, 0), dz - 1);
                any |= mask[(x * dy + y) * dz + z];
            }
    d1[i] = any ? 1 : 0;
}

template <bool FAST, bool RAY4>
__attribute__((amdgpu_waves_per_eu(8, 8)))
__global__ __launch_bounds__(CULL_BLOCK) void k_front_cull(RenderArgs a, Workspace w, double inv_S, float lin_step, int vc_gx) {
    static_assert(VC_BLOCK == CULL_BLOCK, "the two bodies share a launch");
    const int b = (int)blockIdx.x;
    if (b < vc_gx * INVR_NUM_PARTS) { voxel_class_body(a.scene, w.knn, w.counters + CNT_LIVE, b % vc_gx, vc_gx, b / vc_gx); return; }
    // (one workgroup per tile.  A fixed grid of 2048 workgroups walking the tiles measured 352 us against 155 us: the tile's body is a
    // chain of dependent round trips, and workgroups that start together walk it in lockstep — profiles/r4_cull_experiments.md)
    cull_flag_body<true, FAST, RAY4>(a, w, inv_S, lin_step, (int64_t)(b - vc_gx * INVR_NUM_PARTS));
}

// lattice classes + cull flags in one launch; returns 0 and sets *done = 0 when the frame does not take the masked cull
// (launch_cull's condition), in which case the caller runs launch_cull alone
int launch_front_cull(const RenderArgs& a, const Workspace& w, int* done, hipStream_t st) {
    const VolDev& v = a.scene.pbw;
    const int64_t cells = (int64_t)v.dx * v.dy * v.dz, nb = cdiv(a.N, CULL_TILE);
    static const bool no_voxcls = getenv("INVR_NO_VOXCLS") != nullptr;
    *done = 0;
    if (!(a.N >= 4 * cells) || no_voxcls) return 0;
    const unsigned gx = voxel_class_grid(v);
    const double inv_S = 1.0 / (double)a.S;
    const float lin_step = 1.0f / (float)(a.S - 1);
    const bool fast = !a.wpts && !a.jitter && a.N < (1ll << 31) && cells * v.c < (1ll << 31) && a.S >= 2 &&
                      v.dx <= 1024 && v.dy <= 1024 && v.dz <= 1024;      // (front_bodies.h: the pre-test's error bound)
    const unsigned grid = gx * INVR_NUM_PARTS + (unsigned)nb;
    static const bool no_d1 = getenv("INVR_NO_CULL_D1") != nullptr;
    if (fast && (a.S & 3) == 0 && !a.z_vals && !no_d1) {
        // the cell mask dilated by one cell (k_dilate_mask, ~3 us between the two front launches): a thread of the RAY4 cull drops
        // its four samples on ONE look-up when the whole segment provably stays inside dead cells (front_bodies.h)
        Workspace w2 = w;
        w2.use_d1 = 1;
        hipLaunchKernelGGL(k_dilate_mask, dim3((unsigned)cdiv(cells, 256)), dim3(256), 0, st, v.dx, v.dy, v.dz, w.cullmask, w.cullmask_d1);
        INVR_LAUNCH_CHECK();
        hipLaunchKernelGGL((k_front_cull<true, true>), dim3(grid), dim3(CULL_BLOCK), 0, st, a, w2, inv_S, lin_step, (int)gx);
    } else
    if (fast && (a.S & 3) == 0) hipLaunchKernelGGL((k_front_cull<true, true>), dim3(grid), dim3(CULL_BLOCK), 0, st, a, w, inv_S, lin_step, (int)gx);
    else if (fast) hipLaunchKernelGGL((k_front_cull<true, false>), dim3(grid), dim3(CULL_BLOCK), 0, st, a, w, inv_S, lin_step, (int)gx);
    else hipLaunchKernelGGL((k_front_cull<false, false>), dim3(grid), dim3(CULL_BLOCK), 0, st, a, w, inv_S, lin_step, (int)gx);
    INVR_LAUNCH_CHECK();
    *done = 1;
    return 0;
}

// cfg.aggr in {'dist', 'mindist'} (inb_part_network_multiassign.py:240-251) merges by `part_dist` = the KNN's weighted distance of
// EVERY (survivor, part) (blend_utils.py:749 through :817-825; Network.forward hands pbw[..., -1:] down, :88-91,161) — also of the
// parts the pruned search never scans (provably unflagged, far).  These non-default modes take it from a brute-force pass over the
// survivors: the arithmetic of k_knn_dense, the point of a slot from the ray list.  A part with fewer than 4 vertices gives NaN as in
// the reference (inf distances, zero weights: inf * 0).
#define PD_BLOCK 256
__global__ __launch_bounds__(PD_BLOCK) void k_knn_pdist(RenderArgs a, Workspace w) {
    __shared__ float4 sv[KNN_TILE];
    const int na = w.counters[CNT_ACTIVE];
    for (int64_t tile = blockIdx.x; tile * PD_BLOCK < na; tile += gridDim.x) {
        const int64_t slot = tile * PD_BLOCK + threadIdx.x;
        const bool live = slot < na;
        float px = 0, py = 0, pz = 0;
        if (live) sample_pose_point(a, w.active_idx[slot], px, py, pz, nullptr, nullptr);
        for (int p = 0; p < INVR_NUM_PARTS; ++p) {
            const int len = (int)a.scene.lengths2[p];
            const float* verts = a.scene.part_pts + (int64_t)p * a.scene.M * 3;
            Top4 t;
            t.init();
            for (int base = 0; base < len; base += KNN_TILE) {
                const int m = min(KNN_TILE, len - base);
                __syncthreads();
                for (int j = threadIdx.x; j < m; j += PD_BLOCK) {
                    const float* v = verts + (int64_t)(base + j) * 3;
                    sv[j] = make_float4(v[0], v[1], v[2], 0.0f);
                }
                __syncthreads();
#pragma unroll 4
                for (int j = 0; j < m; ++j) {
                    const float4 v = sv[j];
                    const float dx = px - v.x, dy = py - v.y, dz = pz - v.z;
                    t.push(dx * dx + dy * dy + dz * dz, base + j);
                }
            }
            t.finish();
            float wt[KNN_K];
            const float ds = knn_weights(t, wt);          // (fewer than 4 vertices: +inf entries remain -> inf * 0 = NaN, as the reference)
            if (live) w.pdist[slot * INVR_NUM_PARTS + p] = ds;
        }
    }
}

int launch_knn_pdist(const RenderArgs& a, const Workspace& w, hipStream_t st) {
    const int64_t tiles = cdiv(w.cap, PD_BLOCK);
    hipLaunchKernelGGL(k_knn_pdist, dim3((unsigned)(tiles < 2048 ? (tiles > 0 ? tiles : 1) : 2048)), dim3(PD_BLOCK), 0, st, a, w);
    INVR_LAUNCH_CHECK();
    return 0;
}

int launch_knn_pairs(const RenderArgs& a, const Workspace& w, int32_t* stats, hipStream_t st) {
    const size_t lds_bytes = (size_t)KNN_LDS_FLOAT4 * sizeof(float4);      // ~150 KB: one workgroup per CU
    static bool attr_set = false;
    if (!attr_set) {
        INVR_HIP(hipFuncSetAttribute((const void*)k_knn_pairs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        attr_set = true;
    }
    int64_t tiles = cdiv(w.cap, 64);              // tickets of 64 survivors, taken by single waves
    unsigned grid = (unsigned)(tiles < 256 * 16 ? (tiles > 0 ? cdiv(tiles, 16) : 1) : 256);
    hipLaunchKernelGGL(k_knn_pairs, dim3(grid), dim3(KNN_T), lds_bytes, st, a, w);
    INVR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pair_lists, dim3((unsigned)w.n_groups), dim3(PL_BLOCK), 0, st, w, stats);
    INVR_LAUNCH_CHECK();
    return 0;
}
