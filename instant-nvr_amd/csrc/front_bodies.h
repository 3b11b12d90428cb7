// Device bodies shared by the stand-alone kernels (k_cull.hip, k_warp.hip) and by the two fused front-of-frame launches of
// k_knn.hip (k_front_scene: KNN index build + cull cell mask + per-vertex matrices + deformer t-slices as workgroup ranges of ONE
// launch; k_front_cull: lattice-cell classification + cull flags as workgroup ranges of one launch).  A fork / join through a side
// stream costs 10-17 us per edge under hipGraph replay on this runtime (gpurun_out/r4c: 74 us of a 0.56 ms ray shard), launches
// that follow each other on one stream start back to back — so independent small kernels share a launch instead of a stream.
#pragma once
#include <stdlib.h>
#include "pipeline.h"

#define CULL_BLOCK 256
#define CULL_PER 4                      // ray-samples per thread
#define CULL_TILE (CULL_BLOCK * CULL_PER)

// Per-frame cell mask of the distance volume: the trilinear value of a sample is a convex combination of the 8
// corners of its cell (weights in [0,1], sum 1 within 4e-7), so a cell whose corners are all >= thresh*(1+1e-5)
// cannot hold a survivor — 93 % of the samples of the bench frame then skip the 8 taps.  Cell (x0,y0,z0) pairs
// with corner x1 = min(x0+1, dx-1) exactly as the border-clamped sampler does, so there are dx*dy*dz cells.
// The live cells are also appended to a list (wave-aggregated: one atomic per wave), which the KNN's per-cell classification
// (k_knn_voxel_class, side stream) walks instead of the whole lattice.
// (body: `i` = the thread's cell; called with whole waves by k_cull_cells and by the scene-setup launch k_front_scene, k_knn.hip)
__device__ __forceinline__ void cull_cells_body(const VolDev& v, float thresh_hi, uint8_t* __restrict__ mask, int32_t* __restrict__ live,
                                                int32_t* __restrict__ n_live, uint8_t* __restrict__ voxcls, const int i) {
    const bool in = i < v.dx * v.dy * v.dz;
    bool keep = false;
    if (in) {
    const int z0 = i % v.dz, y0 = (i / v.dz) % v.dy, x0 = i / (v.dz * v.dy);
    const int x1 = min(x0 + 1, v.dx - 1), y1 = min(y0 + 1, v.dy - 1), z1 = min(z0 + 1, v.dz - 1);
    float m = __builtin_inff();
    bool nan = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int xx = (k & 4) ? x1 : x0, yy = (k & 2) ? y1 : y0, zz = (k & 1) ? z1 : z0;
        const float d = v.data[(((int64_t)xx * v.dy + yy) * v.dz + zz) * v.c + (v.c - 1)];
        nan = nan || d != d;
        m = fminf(m, d);
    }
    keep = m < thresh_hi || nan;
    mask[i] = keep ? 1 : 0;
    }
    if (live) {
        const unsigned long long b = __ballot(keep);
        if (b) {
            const int lane = threadIdx.x & 63;
            int base = 0;
            if (lane == __ffsll((long long)b) - 1) base = atomicAdd(n_live, __popcll(b));
            base = __shfl(base, __ffsll((long long)b) - 1);
            if (keep) live[base + __popcll(b & ((1ull << lane) - 1ull))] = i;
        }
        // class "undecided" for every cell: k_knn_pairs maps a point to its cell with slightly different arithmetic than the
        // sampler, so a survivor on a cell face may look up a neighbour that is not live — and is never classified
        if (in && voxcls) {
#pragma unroll
            for (int p = 0; p < INVR_NUM_PARTS; ++p) voxcls[(int64_t)i * INVR_NUM_PARTS + p] = 0;
        }
    }
}

// distance channel of the pose-space volume at (px,py,pz): sample_volume_dev<1> (same arithmetic, bit for bit)
// with the cell-mask early-out; returns +inf for samples in masked-out cells (they fail pn < thresh either way)
// The frame path's pre-test: a sample whose cell is masked out — 92 % of the bench frame's — is recognised ahead of the three exact
// quotients.  Its cell comes from ONE multiply per axis by pre[c] = RN(1 / extent) x (d - 1): within 3e-7 x d of the exact lattice
// coordinate (the exact form rounds five times, this one three), so unless the coordinate lies within CULL_PRE_DELTA of a lattice plane
// (or was clamped: fraction 0) the cell IS the exact path's cell and a clear mask byte rejects the sample as the exact path would; the
// others (~1.2 %) and every sample of a live cell are CANDIDATES for the exact path.  -> the cell; sure = the cell is the exact path's.
#define CULL_PRE_DELTA 2e-3f            // >> 3e-7 x 1024 (the host admits d <= 1024 per axis for the pre-test)
__device__ __forceinline__ int cull_pre_cell(const VolDev& v, float px, float py, float pz, const float* pre, const float* bnd, bool& sure) {
    const float ax = fminf(fmaxf((px - bnd[0]) * pre[0], 0.0f), (float)(v.dx - 1));        // (NaN -> 0 -> fraction 0 -> candidate)
    const float ay = fminf(fmaxf((py - bnd[1]) * pre[1], 0.0f), (float)(v.dy - 1));
    const float az = fminf(fmaxf((pz - bnd[2]) * pre[2], 0.0f), (float)(v.dz - 1));
    const float cx = floorf(ax), cy = floorf(ay), cz = floorf(az);
    sure = fabsf((ax - cx) - 0.5f) < 0.5f - CULL_PRE_DELTA && fabsf((ay - cy) - 0.5f) < 0.5f - CULL_PRE_DELTA &&
           fabsf((az - cz) - 0.5f) < 0.5f - CULL_PRE_DELTA;
    return (int)fmaf(fmaf(cx, (float)v.dy, cy), (float)v.dz, cz);                          // exact: cells <= CULL_MASK_MAX = 2^22; always a valid cell
}

// CHECK = false (the candidates of the two-phase frame path): no mask read — a dead cell's 8 corners are all >= thresh (1 + 1e-5), its
// trilinear value cannot pass `< thresh` (cull_cells_body), so the taps alone give the same decision one dependent round trip earlier.
template <typename IDX, bool CHECK = true>        // IDX = uint32_t when dx*dy*dz*c < 2^31 (host-checked): 64-bit index multiplies are quarter rate
__device__ __forceinline__ float cull_distance(const VolDev& v, const uint8_t* __restrict__ mask, float px, float py, float pz, const float* rext,
                                               const float* bnd) {
    const float b0x = bnd[0], b0y = bnd[1], b0z = bnd[2];          // v.bounds, read once per thread by the caller
    const float b1x = bnd[3], b1y = bnd[4], b1z = bnd[5];
    // (p - b0) / (b1 - b0): the IEEE quotients through the per-thread reciprocals of the three extents (common.h:div_exact)
    float gx = div_exact(px - b0x, b1x - b0x, rext[0]) * 2.0f - 1.0f;
    float gy = div_exact(py - b0y, b1y - b0y, rext[1]) * 2.0f - 1.0f;
    float gz = div_exact(pz - b0z, b1z - b0z, rext[2]) * 2.0f - 1.0f;
    float ix = ((gx + 1.0f) * 0.5f) * (float)(v.dx - 1);
    float iy = ((gy + 1.0f) * 0.5f) * (float)(v.dy - 1);
    float iz = ((gz + 1.0f) * 0.5f) * (float)(v.dz - 1);
    ix = fminf(fmaxf(ix, 0.0f), (float)(v.dx - 1));
    iy = fminf(fmaxf(iy, 0.0f), (float)(v.dy - 1));
    iz = fminf(fmaxf(iz, 0.0f), (float)(v.dz - 1));
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    if (CHECK && !mask[((IDX)x0 * (IDX)v.dy + (IDX)y0) * (IDX)v.dz + (IDX)z0]) return __builtin_inff();
    const float tx = ix - fx, ty = iy - fy, tz = iz - fz;
    const int x1 = min(x0 + 1, v.dx - 1), y1 = min(y0 + 1, v.dy - 1), z1 = min(z0 + 1, v.dz - 1);
    float out = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int xx = (k & 4) ? x1 : x0, yy = (k & 2) ? y1 : y0, zz = (k & 1) ? z1 : z0;
        const float wk = ((k & 4) ? tx : 1.0f - tx) * ((k & 2) ? ty : 1.0f - ty) * ((k & 1) ? tz : 1.0f - tz);
        out = fmaf(wk, v.data[(((IDX)xx * (IDX)v.dy + (IDX)yy) * (IDX)v.dz + (IDX)zz) * (IDX)v.c + (IDX)(v.c - 1)], out);
    }
    return out;
}

// tile of 1024 consecutive ray-samples per workgroup: sub-tile k holds samples base + k*256 + tid,
// one 64-bit survivor mask per (sub-tile, wave): mask word index = tile*16 + k*4 + wave
// FAST (rays, no jitter, N < 2^31, small volume): 32-bit sample / ray / volume indices — the generic path spends a
// third of its instructions on a 64-bit division by S and 64-bit index multiplies (quarter-rate integer ops).  The
// float arithmetic is the same op sequence as sample_pose_point / sample_z / linspace01, bit for bit.
template <bool MASKED, bool FAST, bool RAY4 = false>      // RAY4: FAST && MASKED && S % 4 == 0 (the host's choice)
__device__ __forceinline__ void cull_flag_body(const RenderArgs& a, const Workspace& w, double inv_S, float lin_step, const int64_t tile) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __shared__ int cnt[CULL_PER * (CULL_BLOCK / 64)];
    float rext[3] = {0.f, 0.f, 0.f}, pre[3] = {0.f, 0.f, 0.f}, bnd[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    constexpr bool PRE = MASKED && FAST;
    auto ray_of = [&](unsigned iu, unsigned& ray, unsigned& s) {
        const unsigned S = (unsigned)a.S;
        ray = (unsigned)((double)iu * inv_S);                     // floor(i / S), possibly one too small
        s = iu - ray * S;
        if (s >= S) { ++ray; s -= S; }
    };
    unsigned f_ray = 0u, f_s = 0u;
    const unsigned blk_q = FAST ? (unsigned)CULL_BLOCK / (unsigned)a.S : 0u, blk_r = FAST ? (unsigned)CULL_BLOCK % (unsigned)a.S : 0u;   // (scalar unit)
    // PRE, first thing in the workgroup: the ray loads of the thread's four samples (addresses from the kernel arguments alone) go out
    // before the scene constants are fetched — the workgroup's life is a chain of dependent round trips (measured: halving the
    // instructions of this body left its 155 us unchanged), so the chain is kept short and its links wide.
    float rn[CULL_PER], rf[CULL_PER], o3[CULL_PER][3], d3[CULL_PER][3];
    unsigned ss[CULL_PER];
    bool valid[CULL_PER];
    // RAY4 (S a multiple of 4): a thread takes four CONSECUTIVE samples of one ray — one set of ray loads, and the lattice coordinate is
    // affine in the sample depth, a(z) = A + z B, so the pre-test of a sample is three multiply-adds instead of the whole pose transform
    // (the mask words are built in LDS, bit j = sample j of the tile: any sample-to-thread map gives the same words).
    constexpr bool ray4 = PRE && RAY4;
    if (PRE && ray4) {
        const int64_t i0 = tile * CULL_TILE + 4 * (int64_t)threadIdx.x;
        valid[0] = i0 < a.N;                                              // (N = R S, S % 4 == 0: all four samples or none)
        ray_of((unsigned)min(i0, a.N - 1), f_ray, f_s);
        const unsigned ray = min(f_ray, (unsigned)(a.R - 1));
        rn[0] = a.near[ray]; rf[0] = a.far[ray];
        const float* __restrict__ rd = a.ray_d + (size_t)ray * 3u;
        const float* __restrict__ ro = a.ray_o + (size_t)ray * 3u;
#pragma unroll
        for (int c = 0; c < 3; ++c) { o3[0][c] = ro[c]; d3[0][c] = rd[c]; }
    } else if (PRE) {
#pragma unroll
        for (int k = 0; k < CULL_PER; ++k) {
            const int64_t i = tile * CULL_TILE + k * CULL_BLOCK + threadIdx.x;
            valid[k] = i < a.N;
            if (k == 0) ray_of((unsigned)min(i, a.N - 1), f_ray, f_s);
            else {                                                        // sample i + CULL_BLOCK: (ray, s) advance by a wave-uniform step
                f_ray += blk_q; f_s += blk_r;
                if (f_s >= (unsigned)a.S) { ++f_ray; f_s -= (unsigned)a.S; }
            }
            const unsigned ray = min(f_ray, (unsigned)(a.R - 1));         // (samples beyond N: any ray, result unused)
            ss[k] = f_s;
            rn[k] = a.near[ray]; rf[k] = a.far[ray];
            const float* __restrict__ rd = a.ray_d + (size_t)ray * 3u;    // (one address + immediate offsets per array)
            const float* __restrict__ ro = a.ray_o + (size_t)ray * 3u;
#pragma unroll
            for (int c = 0; c < 3; ++c) { o3[k][c] = ro[c]; d3[k][c] = rd[c]; }
        }
    }
    if (MASKED) {
        const float* pb = a.scene.pbw.bounds;
        const int dd[3] = {a.scene.pbw.dx, a.scene.pbw.dy, a.scene.pbw.dz};
#pragma unroll
        for (int c = 0; c < 6; ++c) bnd[c] = pb[c];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (PRE) pre[c] = __builtin_amdgcn_rcpf(bnd[3 + c] - bnd[c]) * (float)(dd[c] - 1);    // (1 ulp: inside the pre-test's margin; the
            else rext[c] = rcp_for_div(bnd[3 + c] - bnd[c]);                                       //  exact reciprocals: phase 2 only)
        }
    }
    float R[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, Th[3] = {0.f, 0.f, 0.f};
    if (FAST) {          // read before the first store of the kernel: scalar loads, SGPR-resident for all CULL_PER samples
#pragma unroll
        for (int c = 0; c < 9; ++c) R[c] = a.scene.R[c];
#pragma unroll
        for (int c = 0; c < 3; ++c) Th[c] = a.scene.Th[c];
    }
    // FAST: pose-space point of sample s of a ray from the ray's near / far / origin / direction — the op sequence of
    // sample_pose_point / sample_z / linspace01, bit for bit
    auto point_from = [&](float near, float far, const float* ro, const float* rd, unsigned s, float& px, float& py, float& pz, float& z) {
        const float t = ((int)s < a.S / 2) ? lin_step * (float)(int)s : 1.0f - lin_step * (float)(a.S - 1 - (int)s);   // linspace01
        z = near * (1.0f - t) + far * t;                          // sample_z
        const float wx = ro[0] + rd[0] * z, wy = ro[1] + rd[1] * z, wz = ro[2] + rd[2] * z;   // pts = o + d*z
        const float qx = wx - Th[0], qy = wy - Th[1], qz = wz - Th[2];                                         // (p - Th) @ R
        px = qx * R[0] + qy * R[3] + qz * R[6];
        py = qx * R[1] + qy * R[4] + qz * R[7];
        pz = qx * R[2] + qy * R[5] + qz * R[8];
    };
    auto fast_point = [&](unsigned ray, unsigned s, float& px, float& py, float& pz, float& z) {
        const float near = a.near[ray], far = a.far[ray];
        const float* __restrict__ rd = a.ray_d + (size_t)ray * 3u;        // (one address + immediate offsets per array)
        const float* __restrict__ ro = a.ray_o + (size_t)ray * 3u;
        const float o[3] = {ro[0], ro[1], ro[2]}, d[3] = {rd[0], rd[1], rd[2]};
        point_from(near, far, o, d, s, px, py, pz, z);
    };
    if (PRE) {
        // Two phases.  A wave holds 64 consecutive samples of a ray: if one of them needs the exact path (three exact quotients, 8
        // taps) the whole wave walks it — with the exact path inside the sample loop nearly every wave of a frame that shows the body
        // paid it four times.  So: (1) every sample: point + pre-test, the candidates (8 % + the ambiguous 1 %) appended to a list of the
        // workgroup; (2) the list, densely: one lane per candidate, its survivor bit ORed into the tile's 16 mask words in LDS.
        __shared__ unsigned long long s_mask[CULL_TILE / 64];
        __shared__ unsigned short s_cand[CULL_TILE];
        __shared__ int s_ncand;
        if (threadIdx.x < CULL_TILE / 64) s_mask[threadIdx.x] = 0ull;
        if (threadIdx.x == 0) s_ncand = 0;
        __syncthreads();
        // (1) is a chain of dependent round trips per sample — the ray's near / far / origin / direction (first touch: HBM), then the
        // mask byte — and measured latency-bound (instruction count halved: same 155 us), so the four samples of a thread are walked
        // side by side: all ray loads first, then all points and mask loads, then the appends.
        int cell[CULL_PER];
        bool sure[CULL_PER];
        float zz[CULL_PER];
        if (ray4) {
            // lattice coordinate of the ray: a(z) = A + z B, A = (((o - Th) R) - b0) pre, B = (d R) pre.  Against the exact path's point
            // (o + d z - Th) R both forms round a handful of times at magnitudes <= M = |o| + |Th| + far |d|: the two differ by
            // < 1e-6 M metres = 1e-6 M pre cells; a ray for which 2e-6 M pre reaches half of CULL_PRE_DELTA decides nothing here
            // (every sample a candidate).  The bench frame: M = 8.5 m, 32 cells / m: 5e-4 of the 1e-3 allowed.
            const float* o = o3[0];
            const float* d = d3[0];
            const float q0 = o[0] - Th[0], q1 = o[1] - Th[1], q2 = o[2] - Th[2];
            float A[3], B[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                A[c] = ((q0 * R[c] + q1 * R[3 + c] + q2 * R[6 + c]) - bnd[c]) * pre[c];
                B[c] = (d[0] * R[c] + d[1] * R[3 + c] + d[2] * R[6 + c]) * pre[c];
            }
            const float M = (fabsf(o[0]) + fabsf(o[1]) + fabsf(o[2])) + (fabsf(Th[0]) + fabsf(Th[1]) + fabsf(Th[2])) +
                            fmaxf(fabsf(rn[0]), fabsf(rf[0])) * (fabsf(d[0]) + fabsf(d[1]) + fabsf(d[2]));
            const bool ok_ray = 2e-6f * M * fmaxf(fmaxf(pre[0], pre[1]), pre[2]) <= 0.5f * CULL_PRE_DELTA;      // (NaN: false)
            const VolDev& v = a.scene.pbw;
            // One look-up for the thread's whole segment (w.use_d1: k_dilate_mask ran for this frame).  The lattice coordinates of its
            // four samples lie between those of the first and the last (affine in z, z monotone in the sample index, the clamp monotone):
            // within h = half that span of the midpoint's, per axis.  The exact path's coordinate of a sample differs from the affine
            // form by < CULL_PRE_DELTA (ok_ray), so with h + 2 CULL_PRE_DELTA < 1 every sample's EXACT cell is within +-1 of the
            // midpoint's cell on every axis — and if the dilated mask is clear there, none of the four can survive: no candidates.
            bool need = true;
            if (w.use_d1) {
                const int s0 = (int)f_s, s3 = (int)f_s + 3;
                const float t0 = (s0 < a.S / 2) ? lin_step * (float)s0 : 1.0f - lin_step * (float)(a.S - 1 - s0);
                const float t3 = (s3 < a.S / 2) ? lin_step * (float)s3 : 1.0f - lin_step * (float)(a.S - 1 - s3);
                const float z0 = rn[0] * (1.0f - t0) + rf[0] * t0, z3 = rn[0] * (1.0f - t3) + rf[0] * t3;
                const float zm = 0.5f * (z0 + z3), hz = 0.5f * fabsf(z3 - z0);
                const float h = hz * fmaxf(fmaxf(fabsf(B[0]), fabsf(B[1])), fabsf(B[2]));
                const float mx = fminf(fmaxf(fmaf(zm, B[0], A[0]), 0.0f), (float)(v.dx - 1));
                const float my = fminf(fmaxf(fmaf(zm, B[1], A[1]), 0.0f), (float)(v.dy - 1));
                const float mz = fminf(fmaxf(fmaf(zm, B[2], A[2]), 0.0f), (float)(v.dz - 1));
                const int mc = (int)fmaf(fmaf(floorf(mx), (float)v.dy, floorf(my)), (float)v.dz, floorf(mz));
                const bool seg_ok = ok_ray && h + 2.0f * CULL_PRE_DELTA + 1e-3f < 1.0f;          // (NaN: false)
#ifdef CULL_D1_MUTATE          // (test builds only: a WRONG skip rule — the exact survivor-set tests must catch it)
                need = !(seg_ok && w.cullmask[mc] == 0);
#else
                need = !(seg_ok && w.cullmask_d1[mc] == 0);
#endif
            }
            if (!need) {
#pragma unroll
                for (int k = 0; k < CULL_PER; ++k) { zz[k] = 0.0f; sure[k] = true; cell[k] = -1; valid[k] = false; }      // no candidate, no mask read
            } else
#pragma unroll
            for (int k = 0; k < CULL_PER; ++k) {
                const int sk = (int)f_s + k;
                const float t = (sk < a.S / 2) ? lin_step * (float)sk : 1.0f - lin_step * (float)(a.S - 1 - sk);   // linspace01
                const float z = rn[0] * (1.0f - t) + rf[0] * t;           // sample_z (exact: also the z_vals output)
                zz[k] = z;
                const float ax = fminf(fmaxf(fmaf(z, B[0], A[0]), 0.0f), (float)(v.dx - 1));
                const float ay = fminf(fmaxf(fmaf(z, B[1], A[1]), 0.0f), (float)(v.dy - 1));
                const float az = fminf(fmaxf(fmaf(z, B[2], A[2]), 0.0f), (float)(v.dz - 1));
                const float cx = floorf(ax), cy = floorf(ay), cz = floorf(az);
                sure[k] = ok_ray && fabsf((ax - cx) - 0.5f) < 0.5f - CULL_PRE_DELTA && fabsf((ay - cy) - 0.5f) < 0.5f - CULL_PRE_DELTA &&
                          fabsf((az - cz) - 0.5f) < 0.5f - CULL_PRE_DELTA;
                cell[k] = (int)fmaf(fmaf(cx, (float)v.dy, cy), (float)v.dz, cz);
                valid[k] = valid[0];
            }
        } else {
#pragma unroll
        for (int k = 0; k < CULL_PER; ++k) {
            float px, py, pz;
            point_from(rn[k], rf[k], o3[k], d3[k], ss[k], px, py, pz, zz[k]);
            cell[k] = cull_pre_cell(a.scene.pbw, px, py, pz, pre, bnd, sure[k]);
        }
        }
        uint8_t mb[CULL_PER];
#pragma unroll
        for (int k = 0; k < CULL_PER; ++k) mb[k] = cell[k] >= 0 ? w.cullmask[cell[k]] : (uint8_t)0;
        if (a.z_vals) {
#pragma unroll
            for (int k = 0; k < CULL_PER; ++k)
                if (valid[k]) a.z_vals[tile * CULL_TILE + (ray4 ? 4 * (int)threadIdx.x + k : k * CULL_BLOCK + (int)threadIdx.x)] = zz[k];
        }
        // one append per thread: its (up to four) candidates take consecutive list slots — rank inside the wave by a scan of the
        // per-thread counts, one LDS atomic per wave
        bool cand[CULL_PER];
        int nc = 0;
#pragma unroll
        for (int k = 0; k < CULL_PER; ++k) { cand[k] = valid[k] && !(sure[k] && !mb[k]); nc += cand[k] ? 1 : 0; }
        const int incl = wave_incl_sum_i(nc);
        const int tot = __builtin_amdgcn_readlane(incl, 63);
        if (tot) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_ncand, tot);
            base = __builtin_amdgcn_readfirstlane(base);
            int pos = base + incl - nc;
#pragma unroll
            for (int k = 0; k < CULL_PER; ++k)
                if (cand[k]) s_cand[pos++] = (unsigned short)(ray4 ? 4 * (int)threadIdx.x + k : k * CULL_BLOCK + (int)threadIdx.x);
        }
        __syncthreads();
        const int n_cand = s_ncand;
        if ((int)threadIdx.x < n_cand) {
#pragma unroll
            for (int c = 0; c < 3; ++c) rext[c] = rcp_for_div(bnd[3 + c] - bnd[c]);
        }
        for (int c = threadIdx.x; c < n_cand; c += CULL_BLOCK) {
            const unsigned loc = s_cand[c];
            unsigned ray, sm;
            ray_of((unsigned)(tile * CULL_TILE) + loc, ray, sm);
            float px, py, pz, z;
            fast_point(ray, sm, px, py, pz, z);
            const float pn = cull_distance<uint32_t, false>(a.scene.pbw, w.cullmask, px, py, pz, rext, bnd);
            if (pn < a.scene.thresh) atomicOr(&s_mask[loc >> 6], 1ull << (loc & 63u));               // :135
        }
        __syncthreads();
        if (threadIdx.x < CULL_TILE / 64) {
            const unsigned long long m = s_mask[threadIdx.x];
            w.mask[tile * (CULL_TILE / 64) + threadIdx.x] = m;            // word k * 4 + wave: the lanes of that wave's k-th sample
            cnt[threadIdx.x] = __popcll(m);
        }
    } else {
#pragma unroll
    for (int k = 0; k < CULL_PER; ++k) {
        const int64_t i = tile * CULL_TILE + k * CULL_BLOCK + threadIdx.x;
        bool keep = false;
        if (i < a.N) {
            float px, py, pz, z;
            if (FAST) {
                if (k == 0) ray_of((unsigned)i, f_ray, f_s);
                else {
                    f_ray += blk_q; f_s += blk_r;
                    if (f_s >= (unsigned)a.S) { ++f_ray; f_s -= (unsigned)a.S; }
                }
                fast_point(f_ray, f_s, px, py, pz, z);
            } else {
                sample_pose_point(a, i, px, py, pz, &z, nullptr);
            }
            if (a.z_vals) a.z_vals[i] = z;
            float pn;
            if (MASKED) pn = FAST ? cull_distance<uint32_t>(a.scene.pbw, w.cullmask, px, py, pz, rext, bnd)
                                  : cull_distance<int64_t>(a.scene.pbw, w.cullmask, px, py, pz, rext, bnd);
            else sample_volume_dev<1>(a.scene.pbw, a.scene.pbw.c - 1, px, py, pz, &pn);   // distance channel
            keep = pn < a.scene.thresh;                                               // :135
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) {
            w.mask[tile * (CULL_TILE / 64) + k * (CULL_BLOCK / 64) + wv] = m;
            cnt[k * (CULL_BLOCK / 64) + wv] = __popcll(m);
        }
    }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int c = 0;
#pragma unroll
        for (int k = 0; k < CULL_TILE / 64; ++k) c += cnt[k];
        w.block_cnt[tile] = c;
    }
}

// ---- per-vertex pre-blended matrices (k_warp.hip header) ------------------------------------------------
struct Mat34 { float m[12]; };   // rows 0..2 of a 4x4: [R | t]

__device__ __forceinline__ void blend_mats(const float* __restrict__ A, const float* bw, Mat34& o) {
#pragma unroll
    for (int e = 0; e < 12; ++e) o.m[e] = 0.0f;
#pragma unroll
    for (int j = 0; j < INVR_NUM_JOINTS; ++j)
#pragma unroll
        for (int e = 0; e < 12; ++e) o.m[e] = fmaf(bw[j], A[j * 16 + e], o.m[e]);     // bw @ A.view(24,16)
}

#define VMAT_BLOCK 128
__device__ __forceinline__ void vertex_mats_body(const SceneDev& s, const KnnIndex& ix, const float* __restrict__ A,
                                                 const float* __restrict__ big_A, const int p, const int v) {
    if (v >= s.M || v >= ix.mpad) return;          // padding rows behind lengths2[p] are zeros in part_pbw: kept finite
    const float* __restrict__ row = s.part_pbw + ((int64_t)p * s.M + v) * INVR_NUM_JOINTS;
    float b[INVR_NUM_JOINTS];
#pragma unroll
    for (int j = 0; j < INVR_NUM_JOINTS; ++j) b[j] = row[j];
    Mat34 Ma, Mb;
    blend_mats(A, b, Ma);
    blend_mats(big_A, b, Mb);
    float4* o = ix.vmat + ((int64_t)p * ix.mpad + v) * 6;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        o[r] = make_float4(Ma.m[r * 4], Ma.m[r * 4 + 1], Ma.m[r * 4 + 2], Ma.m[r * 4 + 3]);
        o[3 + r] = make_float4(Mb.m[r * 4], Mb.m[r * 4 + 1], Mb.m[r * 4 + 2], Mb.m[r * 4 + 3]);
    }
}

// ---- per-frame t-slices of the deformer grid (k_warp.hip: k_deform_pairs_slice) ---------------------------
struct DfSliceInfo { int off[INVR_MAX_LEVELS + 1]; };

__device__ __forceinline__ void deform_slice_body(const GridDev& dg, const DfSliceInfo& si, const float* __restrict__ frame_dim,
                                                  float2* __restrict__ out, const int e) {
    if (e >= si.off[dg.L]) return;
    int l = 0;
    while (e >= si.off[l + 1]) ++l;
    const int res = dg.res[l];
    const float tn = (frame_dim[0] - dg.bounds[2]) / (dg.bounds[5] - dg.bounds[2]);
    int c0z, c1z;
    float tz;
    level_corners(tn, dg.cell[l], res, c0z, c1z, tz);
    const int idx = e - si.off[l], cx = idx / res, cy = idx - cx * res;
    const bool hashed = l >= dg.start_hash;
    const float* tb = dg.separate_dense ? (hashed ? dg.hash + (int64_t)(l - dg.start_hash) * dg.T * 2 : dg.dense + dg.dense_off[l] * 2)
                                        : dg.hash + (int64_t)l * dg.T * 2;
    const float2* tab = reinterpret_cast<const float2*>(tb);
    unsigned r0, r1;
    if (hashed) {
        const uint64_t hxy = (uint64_t)(uint32_t)cx ^ ((uint64_t)(uint32_t)cy * HASH_P1);
        r0 = grid_hash_mod(hxy ^ ((uint64_t)(uint32_t)c0z * HASH_P2), dg);
        r1 = grid_hash_mod(hxy ^ ((uint64_t)(uint32_t)c1z * HASH_P2), dg);
    } else {
        r0 = ((unsigned)cx * (unsigned)res + (unsigned)cy) * (unsigned)res + (unsigned)c0z;
        r1 = ((unsigned)cx * (unsigned)res + (unsigned)cy) * (unsigned)res + (unsigned)c1z;
    }
    const float2 v0 = tab[r0], v1 = tab[r1];
    const float uz = 1.0f - tz;
    out[e] = make_float2(fmaf(tz, v1.x, uz * v0.x), fmaf(tz, v1.y, uz * v0.y));
}

static bool deform_slices_fit(const GridDev& dg, DfSliceInfo& si, int cbv) {
    si.off[0] = 0;
    for (int l = 0; l < INVR_MAX_LEVELS; ++l) si.off[l + 1] = si.off[l] + (l < dg.L ? dg.res[l] * dg.res[l] : 0);
    return dg.L == 8 && si.off[8] <= DF_SLICE_MAX && cbv < 10;
}
static int deform_cb() {
    static int cbv = getenv("INVR_DF_CB") ? atoi(getenv("INVR_DF_CB")) : 2;
    return cbv;
}
