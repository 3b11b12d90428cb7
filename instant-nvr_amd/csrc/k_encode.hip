// K5: multi-resolution hash-grid encoder.
// Replaces HashEmbedder.forward (lib/networks/embedders/part_base_embedder.py:106-174).
//
// k_part_encode — the HBM-bound kernel of the path — handles the per-part grids
// (16 levels x 16 features, sum over features): ONE WAVE encodes ONE POINT at a time with
//      lane = level*4 + q        (level 0..15, q = which 16-byte quarter of the 64-byte row)
// so every corner fetch is one global_load_dwordx4 per lane and the 4 lanes of a quad read one
// whole, contiguous 64-byte table row: 16 fully-used 64-B segments per wave instruction, 8
// instructions (corners) per point = the 8 KiB per (point,part) pair of SURVEY.md §8(d).
// Each lane keeps a float4 accumulator over the 8 corners (4 FMAs per row), the 16 features are
// summed with two quad-DPP adds, level results are staged per wave in LDS ([k][64 points]) and
// flushed as coalesced 256-byte rows into the SoA embedding buffer the MLP kernel consumes.
// Two points are processed per iteration to keep 16 row fetches in flight per wave.
#include <stdlib.h>
#include "pipeline.h"
#include "grid_generic.h"

// ---- generic thread-per-point encoder (API entry point; any configuration) ---------------------
__global__ void k_grid_encode_rt(GridDev g, const float* xyz, int64_t n, float* out, int out_dim) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x[3];
    grid_normalise(g, xyz + i * 3, x);
    float* o = out + i * out_dim;
    int off = 0;
    if (g.include_input) { o[0] = x[0]; o[1] = x[1]; o[2] = x[2]; off = 3; }
    float tot[16];
    for (int f = 0; f < 16; ++f) tot[f] = 0.0f;
    for (int l = 0; l < g.L; ++l) {
        int64_t rows[8];
        float wts[8];
        const float* tab = grid_level_lookup(g, l, x, rows, wts);
        float acc[16];
        for (int f = 0; f < 16; ++f) acc[f] = 0.0f;
        for (int k = 0; k < 8; ++k)
            for (int f = 0; f < g.F; ++f) acc[f] = fmaf(wts[k], tab[rows[k] * g.F + f], acc[f]);
        if (!g.sum) {
            for (int f = 0; f < g.F; ++f) o[off + l * g.F + f] = acc[f];
        } else if (g.sum_over_features) {
            float s = 0.0f;
            for (int f = 0; f < g.F; ++f) s += acc[f];
            o[off + l] = s;
        } else {
            for (int f = 0; f < g.F; ++f) tot[f] += acc[f];
        }
    }
    if (g.sum && !g.sum_over_features)
        for (int f = 0; f < g.F; ++f) o[off + f] = tot[f];
}

int launch_grid_encode_generic(const GridDev& g, const float* xyz, int64_t n, float* out, hipStream_t st) {
    if (n == 0) return 0;
    if (g.F > 16 || g.L > INVR_MAX_LEVELS) { invr_set_error("grid encoder: F<=16 and L<=16 required"); return 1; }
    int od = (g.sum ? (g.sum_over_features ? g.L : g.F) : g.L * g.F) + (g.include_input ? 3 : 0);
    hipLaunchKernelGGL(k_grid_encode_rt, dim3((unsigned)cdiv(n, 128)), dim3(128), 0, st, g, xyz, n, out, od);
    INVR_LAUNCH_CHECK();
    return 0;
}

int launch_part_encode_bwd(const GridDev& g, const float* xyz, const float* gout, int64_t n, float* g_dense, float* g_hash,
                           float* g_xyz, hipStream_t st);

// ---- generic backward: table gradients (atomic adds) + input gradient ------------------------------
// Contention control (a 64x64 training patch sends ~1e5 pairs through 8-row coarse levels):
//  * sum-over-features grids (the part grids): d out_l / d table[row][f] = w_k * g_l for EVERY f, so
//    only ONE scalar per row is accumulated (into column 0 of the gradient row) and k_expand_rows
//    copies it to the other F-1 columns afterwards: 16x fewer atomics;
//  * levels whose whole table slice fits 16 KB of LDS are accumulated per workgroup in LDS
//    (ds_add_f32) and flushed with one global atomic per touched entry.
#define BWD_BLOCK 1024
#define BWD_LDS_FLOATS 33792           // 132 KB dynamic LDS: one level slice of up to 16.9k rows x 2 features

// Persistent workgroups, LEVEL-outer / tile-inner: for each level the workgroup accumulates the
// gradients of ALL its points in LDS (when the level slice fits) and flushes once, so a level that is
// hammered by every point (the deformer's (u,v,t) input has a constant t: a 2-D slice of each level
// receives everything) costs one global atomic per touched entry per workgroup.
__global__ __launch_bounds__(BWD_BLOCK) void k_grid_encode_bwd_rt(GridDev g, const float* __restrict__ xyz,
                                                                  const float* __restrict__ gout, int64_t n_host,
                                                                  const int32_t* __restrict__ count, int out_dim,
                                                                  float* g_dense, float* g_hash, float* __restrict__ g_xyz) {
    extern __shared__ __attribute__((aligned(16))) float sacc[];
    const int64_t n = count ? (int64_t)*count : n_host;
    const int off = g.include_input ? 3 : 0;
    const bool rowscalar = g.sum && g.sum_over_features;
    const int Fe = rowscalar ? 1 : g.F;                         // accumulated floats per row
    for (int l = 0; l < g.L; ++l) {
        const bool hashed = l >= g.start_hash;
        const int64_t level_rows = hashed ? g.T : (int64_t)g.res[l] * g.res[l] * g.res[l];
        const bool use_lds = level_rows * Fe <= BWD_LDS_FLOATS;             // uniform
        if (use_lds) {
            for (int j = threadIdx.x; j < (int)(level_rows * Fe); j += BWD_BLOCK) sacc[j] = 0.0f;
            __syncthreads();
        }
        float* gtab_level = nullptr;
        for (int64_t i = (int64_t)blockIdx.x * BWD_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * BWD_BLOCK) {
            float x[3];
            grid_normalise(g, xyz + i * 3, x);
            const float* go = gout + i * out_dim;
            int64_t rows[8];
            float wts[8];
            const float* tab = grid_level_lookup(g, l, x, rows, wts);
            float* gtab;                                                    // gradient table aligned with `tab`
            if (g.separate_dense) gtab = hashed ? g_hash + (tab - g.hash) : g_dense + (tab - g.dense);
            else gtab = g_hash + (tab - g.hash);
            gtab_level = gtab;
            int c0, c1;
            float t[3];
            for (int a = 0; a < 3; ++a) level_corners(x[a], g.cell[l], g.res[l], c0, c1, t[a]);
            float gt[3] = {0.f, 0.f, 0.f};
            for (int k = 0; k < 8; ++k) {
                const float wx = (k & 4) ? t[0] : 1.0f - t[0], wy = (k & 2) ? t[1] : 1.0f - t[1], wz = (k & 1) ? t[2] : 1.0f - t[2];
                float dot = 0.0f;                                           // sum_f g_f * v_kf
                for (int f = 0; f < g.F; ++f) {
                    float gf;
                    if (!g.sum) gf = go[off + l * g.F + f];
                    else if (g.sum_over_features) gf = go[off + l];
                    else gf = go[off + f];
                    dot = fmaf(gf, tab[rows[k] * g.F + f], dot);
                    if (!rowscalar) {
                        if (use_lds) atomicAdd(&sacc[rows[k] * g.F + f], wts[k] * gf);
                        else unsafeAtomicAdd(gtab + rows[k] * g.F + f, wts[k] * gf);
                    }
                }
                if (rowscalar) {
                    const float v = wts[k] * go[off + l];
                    if (use_lds) atomicAdd(&sacc[rows[k]], v);
                    else unsafeAtomicAdd(gtab + rows[k] * g.F, v);          // column 0 carries the row scalar
                }
                gt[0] += ((k & 4) ? 1.0f : -1.0f) * wy * wz * dot;
                gt[1] += ((k & 2) ? 1.0f : -1.0f) * wx * wz * dot;
                gt[2] += ((k & 1) ? 1.0f : -1.0f) * wx * wy * dot;
            }
            if (g_xyz)                                                       // same thread owns point i at every level
                for (int a = 0; a < 3; ++a) {
                    const float prev = l == 0 ? (g.include_input ? go[a] : 0.0f) : g_xyz[i * 3 + a];
                    float v = prev + gt[a] / g.cell[l];                      // f = x / cell, t = f - const
                    if (l == g.L - 1) v = v / (g.bounds[3 + a] - g.bounds[a]);
                    g_xyz[i * 3 + a] = v;
                }
        }
        if (use_lds) {
            __syncthreads();
            // all points of a level share the table slice: recompute its base from the level (no point needed)
            float* base;
            if (g.separate_dense) base = hashed ? g_hash + (int64_t)(l - g.start_hash) * g.T * g.F : g_dense + g.dense_off[l] * g.F;
            else base = g_hash + (int64_t)l * g.T * g.F;
            (void)gtab_level;
            for (int j = threadIdx.x; j < (int)(level_rows * Fe); j += BWD_BLOCK) {
                const float v = sacc[j];
                if (v != 0.0f) unsafeAtomicAdd(base + (rowscalar ? (int64_t)j * g.F : (int64_t)j), v);
            }
            __syncthreads();
        }
    }
}

// copy the row scalar in column 0 to the other columns (sum-over-features grids)
__global__ void k_expand_rows(float* __restrict__ gtab, int64_t rows, int F) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float v = gtab[r * F];
    if (v != 0.0f)
        for (int f = 1; f < F; ++f) gtab[r * F + f] = v;
}

int launch_grid_encode_bwd_generic(const GridDev& g, const float* xyz, const float* gout, int64_t n, float* g_dense,
                                   float* g_hash, float* g_xyz, hipStream_t st, const int32_t* count) {
    if (n == 0) return 0;
    int od = (g.sum ? (g.sum_over_features ? g.L : g.F) : g.L * g.F) + (g.include_input ? 3 : 0);
    const bool fast = g.L == 16 && g.F == 16 && g.sum && g.sum_over_features && g.include_input && !count;
    if (fast) { if (launch_part_encode_bwd(g, xyz, gout, n, g_dense, g_hash, g_xyz, st)) return 1; }
    else {
        static bool attr_set = false;
        const size_t lds = (size_t)BWD_LDS_FLOATS * sizeof(float);
        if (!attr_set) {
            if (hipFuncSetAttribute((const void*)k_grid_encode_bwd_rt, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
                invr_set_error("hipFuncSetAttribute(k_grid_encode_bwd_rt) failed");
                return 1;
            }
            attr_set = true;
        }
        int64_t nb = cdiv(n, BWD_BLOCK);
        hipLaunchKernelGGL(k_grid_encode_bwd_rt, dim3((unsigned)(nb < 256 ? nb : 256)), dim3(BWD_BLOCK), lds, st, g, xyz, gout, n, count, od,
                           g_dense, g_hash, g_xyz);
    }
    INVR_LAUNCH_CHECK();
    if (g.sum && g.sum_over_features && g.F > 1) {
        int64_t hrows = (int64_t)(g.separate_dense ? g.L - g.start_hash : g.L) * g.T;
        hipLaunchKernelGGL(k_expand_rows, dim3((unsigned)cdiv(hrows, 256)), dim3(256), 0, st, g_hash, hrows, g.F);
        INVR_LAUNCH_CHECK();
        if (g.separate_dense) {
            int64_t drows = 0;
            for (int l = 0; l < g.start_hash; ++l) drows += (int64_t)g.res[l] * g.res[l] * g.res[l];
            hipLaunchKernelGGL(k_expand_rows, dim3((unsigned)cdiv(drows, 256)), dim3(256), 0, st, g_dense, drows, g.F);
            INVR_LAUNCH_CHECK();
        }
    }
    return 0;
}

// ---- wave-cooperative 16x16 part encoder ----------------------------------------------------------
#define ENC_BLOCK 256
#define ENC_WAVES (ENC_BLOCK / 64)

__device__ __forceinline__ float rdlane(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ float quad_sum(float s) {
    // quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E
    s += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s), 0xB1, 0xF, 0xF, true));
    s += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s), 0x4E, 0xF, 0xF, true));
    return s;
}

struct LaneLevel {          // per-lane constants of "my" level
    const float4* tab;      // level table base + q*16 bytes
    int res;
    float cell;
    bool hashed;
};

template <int SRC>   // broadcast lane SRC of every quad to the whole quad (v_mov_b32_dpp quad_perm:[SRC x4])
__device__ __forceinline__ int quad_bcast_i(int v) {
    return __builtin_amdgcn_mov_dpp(v, SRC * 0x55, 0xF, 0xF, true);
}
template <int SRC>
__device__ __forceinline__ float quad_bcast_f(float v) { return __int_as_float(quad_bcast_i<SRC>(__float_as_int(v))); }

// Issue the 8 row fetches of one point for this lane's level.  The index arithmetic is shared by
// the 4 lanes of a quad (they fetch the 4 quarters of the same rows): lane q does the division /
// truncation / clipping of axis q (q < 3) and the row index of corners 2q and 2q+1 (the z pair);
// everything is exchanged with quad-permute DPP moves, so the int64 hash + prime modulo is evaluated
// exactly once per corner per level instead of four times.
__device__ __forceinline__ void fetch_point(const GridDev& g, const LaneLevel& L, int q, float x, float y, float z,
                                            float4* v, float* wts) {
    int c0, c1;
    float t;
    const float xa = q == 0 ? x : (q == 1 ? y : z);
    level_corners(xa, L.cell, L.res, c0, c1, t);
    const int c0x = quad_bcast_i<0>(c0), c1x = quad_bcast_i<0>(c1);
    const int c0y = quad_bcast_i<1>(c0), c1y = quad_bcast_i<1>(c1);
    const int c0z = quad_bcast_i<2>(c0), c1z = quad_bcast_i<2>(c1);
    const float tx = quad_bcast_f<0>(t), ty = quad_bcast_f<1>(t), tz = quad_bcast_f<2>(t);
    // my two corners: k = 2q (z = c0z) and k = 2q+1 (z = c1z); x bit = q>>1, y bit = q&1
    const int cx = (q & 2) ? c1x : c0x, cy = (q & 1) ? c1y : c0y;
    unsigned r0, r1;
    if (L.hashed) {
        const uint64_t hxy = (uint64_t)(uint32_t)cx ^ ((uint64_t)(uint32_t)cy * HASH_P1);
        r0 = grid_hash_mod(hxy ^ ((uint64_t)(uint32_t)c0z * HASH_P2), g);
        r1 = grid_hash_mod(hxy ^ ((uint64_t)(uint32_t)c1z * HASH_P2), g);
    } else {
        const unsigned rb = ((unsigned)cx * (unsigned)L.res + (unsigned)cy) * (unsigned)L.res;
        r0 = rb + (unsigned)c0z;
        r1 = rb + (unsigned)c1z;
    }
    const float ux = 1.0f - tx, uy = 1.0f - ty, uz = 1.0f - tz;
    unsigned row[8];
    row[0] = (unsigned)quad_bcast_i<0>((int)r0); row[1] = (unsigned)quad_bcast_i<0>((int)r1);
    row[2] = (unsigned)quad_bcast_i<1>((int)r0); row[3] = (unsigned)quad_bcast_i<1>((int)r1);
    row[4] = (unsigned)quad_bcast_i<2>((int)r0); row[5] = (unsigned)quad_bcast_i<2>((int)r1);
    row[6] = (unsigned)quad_bcast_i<3>((int)r0); row[7] = (unsigned)quad_bcast_i<3>((int)r1);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        // weight_k = prod_axis ((1-o) + (2o-1) t)  (:157-158), offsets 000,001,..,111 (x y z)
        wts[k] = ((k & 4) ? tx : ux) * ((k & 2) ? ty : uy) * ((k & 1) ? tz : uz);
        v[k] = L.tab[(size_t)row[k] * 4];                       // 16 floats per row = 4 float4
    }
}

__device__ __forceinline__ float reduce_point(const float4* v, const float* wts) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        a.x = fmaf(wts[k], v[k].x, a.x); a.y = fmaf(wts[k], v[k].y, a.y);
        a.z = fmaf(wts[k], v[k].z, a.z); a.w = fmaf(wts[k], v[k].w, a.w);
    }
    return quad_sum((a.x + a.y) + (a.z + a.w));
}

__device__ __forceinline__ void encode_rows_part(const GridDev& g, const float* __restrict__ xs, int64_t stride, const int cnt, int64_t cap,
                                                 float* __restrict__ emb, float (*semb)[EMB_K][64]) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int level = lane >> 2, q = lane & 3;
    LaneLevel L;
    L.res = g.res[level];
    L.cell = g.cell[level];
    L.hashed = level >= g.start_hash;
    const float* tb = g.separate_dense
        ? (L.hashed ? g.hash + (int64_t)(level - g.start_hash) * g.T * 16 : g.dense + g.dense_off[level] * 16)
        : g.hash + (int64_t)level * g.T * 16;
    L.tab = reinterpret_cast<const float4*>(tb) + q;
    const float b0x = g.bounds[0], b0y = g.bounds[1], b0z = g.bounds[2];
    const float ex = g.bounds[3] - b0x, ey = g.bounds[4] - b0y, ez = g.bounds[5] - b0z;

    for (int64_t tile = (int64_t)blockIdx.x * ENC_WAVES + wv; tile * 64 < cnt; tile += (int64_t)gridDim.x * ENC_WAVES) {
        const int64_t base = tile * 64;
        const int m = (int)min((int64_t)64, cnt - base);
        // lane i holds the normalised coordinates of point base+i (:112)
        const int64_t pi = base + min(lane, m - 1);
        const float xi = (xs[pi] - b0x) / ex;
        const float yi = (xs[stride + pi] - b0y) / ey;
        const float zi = (xs[2 * stride + pi] - b0z) / ez;
        semb[wv][0][lane] = xi; semb[wv][1][lane] = yi; semb[wv][2][lane] = zi;
        semb[wv][EMB_K - 1][lane] = 0.0f;                        // pad column
        for (int j = 0; j < m; j += 2) {
            const int j1 = min(j + 1, m - 1);
            float4 va[8], vb[8];
            float wa[8], wb[8];
            fetch_point(g, L, q, rdlane(xi, j), rdlane(yi, j), rdlane(zi, j), va, wa);
            fetch_point(g, L, q, rdlane(xi, j1), rdlane(yi, j1), rdlane(zi, j1), vb, wb);
            const float sa = reduce_point(va, wa);
            const float sb = reduce_point(vb, wb);
            if (q == 0) {
                semb[wv][3 + level][j] = sa;
                semb[wv][3 + level][j1] = sb;
            }
        }
        // flush [k][point] rows, coalesced.  The LDS region is private to the wave: no workgroup barrier is needed, the lanes of a
        // wave run in lockstep — stated to the compiler as a wave barrier (no instruction; orders the stores above before the loads)
        __builtin_amdgcn_wave_barrier();
        if (lane < m) {
#pragma unroll
            for (int k = 0; k < EMB_K; ++k) emb[(int64_t)k * cap + base + lane] = semb[wv][k][lane];
        }
    }
}

__global__ __launch_bounds__(ENC_BLOCK) void k_part_encode(GridDev g, const float* __restrict__ xs, int64_t stride,
                                                           const int32_t* __restrict__ count, int64_t cap,
                                                           float* __restrict__ emb) {
    __shared__ float semb[ENC_WAVES][EMB_K][64];
    encode_rows_part(g, xs, stride, *count, cap, emb, semb);
}

// the five parts in one launch (blockIdx.y = part): the training forward's five encoder launches are short (a part's 1e4-5e4
// pairs) and latency-bound, side by side they take the time of the largest
__global__ __launch_bounds__(ENC_BLOCK) void k_part_encode_rows_all(EncodeAllArgs a) {
    __shared__ float semb[ENC_WAVES][EMB_K][64];
    const int p = blockIdx.y;
    encode_rows_part(a.g[p], a.xs[p], a.stride, a.counts[p], a.cap, a.emb[p], semb);
}

// ---- wave-cooperative backward of the 16x16 sum-over-features grids ----------------------------------
// Same lane mapping as k_part_encode (lane = level*4 + quarter-row).  For one point and level l:
//   d out_l / d table[row_k][f] = w_k          (all 16 f)  -> ONE scalar per row, accumulated into column 0
//                                                             of the gradient row (k_expand_rows copies it)
//   d out_l / d t_a             = sum_k (+-)(w_b w_c) S_k,  S_k = sum_f table[row_k][f]   (needs the rows again)
// Small dense levels (<= BWD_SMALL_ROWS rows) are accumulated per workgroup in LDS: a training patch
// sends ~1e5 pairs through 8..1000-row coarse levels and would serialise on global atomics.
#ifndef ENCB_EXP
#define ENCB_EXP 0
#endif
#define BWD_SMALL_ROWS 1100
#define BWD_SMALL_FLOATS 6144
#define BWD_CACHE 4096
#define BWD_CACHE_RES 128

// Layout-generic: element (point i, component c) of xyz / gout / g_xyz sits at i*ps + c*cs (AoS (n,3)/(n,19): ps = 3/19,
// cs = 1; the SoA pair lists of the training pipeline: ps = 1, cs = list stride); the point count is `n_host` or, when
// `count` is given, read on the device.  Table gradients go either into column 0 of the full-size gradient tables
// (g_dense / g_hash, expanded by k_expand_rows) or — `rowgrad` — into a compact (rows,) array in invr_grid_row_sums
// order (dense rows, then T per hashed level): the gradient of a sum-over-features table IS one scalar per row.
struct EncBwdIO {
    const float* xyz; int64_t x_ps, x_cs;
    const float* gout; int64_t go_ps, go_cs;
    float* g_xyz; int64_t gx_ps, gx_cs;
    int64_t n_host; const int32_t* count;
    float* g_dense; float* g_hash; float* rowgrad;
};

__global__ __launch_bounds__(ENC_BLOCK) void k_part_encode_bwd(GridDev g, EncBwdIO io) {
    __shared__ float sgo[ENC_WAVES][64][20];          // g_out tile of the wave
    __shared__ float sgx[ENC_WAVES][64][3];           // per point: gradient w.r.t. the normalised coordinate
    __shared__ float ssmall[BWD_SMALL_FLOATS];        // per-workgroup accumulators of the small dense levels
    // write-combining cache of the medium dense levels (res <= BWD_CACHE_RES, too large for ssmall): a training patch sends
    // all its ~5e4 pairs through a few dozen rows of the coarse levels (cells of 3..12 cm against a 10 cm patch); their
    // same-line global atomics from ~3000 waves serialised the kernel (0.8 of 1.1 ms per iteration).  Persistent workgroups
    // accumulate those rows in LDS (direct-mapped {level<<24 | row, sum}; a slot conflict falls back to the global atomic) and
    // flush each touched row once.
    __shared__ unsigned ckey[BWD_CACHE];
    __shared__ float cval[BWD_CACHE];
    const int64_t n = io.count ? (int64_t)*io.count : io.n_host;
    if (n <= 0) return;
    int tp = 64;                                      // points per wave tile: enough waves to fill the GPU, long enough runs to combine
    while (tp > 16 && n / tp < 4096) tp >>= 1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int level = lane >> 2, q = lane & 3;
    LaneLevel L;
    L.res = g.res[level];
    L.cell = g.cell[level];
    L.hashed = level >= g.start_hash;
    const float* tb = g.separate_dense
        ? (L.hashed ? g.hash + (int64_t)(level - g.start_hash) * g.T * 16 : g.dense + g.dense_off[level] * 16)
        : g.hash + (int64_t)level * g.T * 16;
    // gradient base of my level + stride between rows: full-size table (column 0 of a 16-float row) or compact row scalars
    float* gtb;
    int gstride;
    if (io.rowgrad) {
        gtb = io.rowgrad + (g.separate_dense ? (L.hashed ? g.dense_rows + (int64_t)(level - g.start_hash) * g.T : g.dense_off[level])
                                             : (int64_t)level * g.T);
        gstride = 1;
    } else {
        gtb = g.separate_dense ? (L.hashed ? io.g_hash + (tb - g.hash) : io.g_dense + (tb - g.dense)) : io.g_hash + (tb - g.hash);
        gstride = 16;
    }
    L.tab = reinterpret_cast<const float4*>(tb) + q;
    // LDS slot of my level (small dense levels only), laid out back to back
    int small_off = -1, small_total = 0;
    for (int l = 0; l < 16; ++l) {
        const int64_t rows = (int64_t)g.res[l] * g.res[l] * g.res[l];
        const bool small = l < (g.separate_dense ? g.start_hash : 0) && rows <= BWD_SMALL_ROWS && small_total + rows <= BWD_SMALL_FLOATS;
        if (l == level && small) small_off = small_total;
        if (small) small_total += (int)rows;
    }
    for (int j = threadIdx.x; j < small_total; j += ENC_BLOCK) ssmall[j] = 0.0f;
    for (int j = threadIdx.x; j < BWD_CACHE; j += ENC_BLOCK) { ckey[j] = 0xFFFFFFFFu; cval[j] = 0.0f; }
    const bool cached = small_off < 0 && !L.hashed && g.separate_dense && L.res <= BWD_CACHE_RES;
    __syncthreads();
    const float b0x = g.bounds[0], b0y = g.bounds[1], b0z = g.bounds[2];
    const float ex = g.bounds[3] - b0x, ey = g.bounds[4] - b0y, ez = g.bounds[5] - b0z;

    // a wave walks the tp (<= 64) points of its tile one after the other (all 64 lanes on one point): a training
    // patch has only ~3e4 pairs per part, so the tile shrinks until the grid fills the GPU
    for (int64_t tile = (int64_t)blockIdx.x * ENC_WAVES + wv; tile * tp < n; tile += (int64_t)gridDim.x * ENC_WAVES) {
        const int64_t base = tile * tp;
        const int m = (int)min((int64_t)tp, n - base);
        const int64_t pi = base + min(lane, m - 1);
        const float xi = (io.xyz[pi * io.x_ps] - b0x) / ex, yi = (io.xyz[pi * io.x_ps + io.x_cs] - b0y) / ey,
                    zi = (io.xyz[pi * io.x_ps + 2 * io.x_cs] - b0z) / ez;
        if (io.go_cs == 1) {
            for (int e = lane; e < m * 19; e += 64) sgo[wv][e / 19][e % 19] = io.gout[base * io.go_ps + e];      // coalesced AoS tile copy
        } else if (lane < m) {
#pragma unroll
            for (int k = 0; k < 19; ++k) sgo[wv][lane][k] = io.gout[(base + lane) * io.go_ps + k * io.go_cs];   // coalesced SoA rows
        }
        // run-length combining: consecutive pairs of the list are consecutive samples of a ray / neighbouring rays and
        // fall into the same cell of the coarse and middle levels, where a 64x64 training patch puts 1e4..1e5
        // contributions on a few dozen rows — their same-address global atomics serialised the kernel (0.6 ms per
        // part).  Each lane keeps the pending (row, sum) of its 8 corners and only issues an atomic when the row changes.
        unsigned prow[8];
        float pval[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { prow[k] = 0xFFFFFFFFu; pval[k] = 0.0f; }
        auto flush = [&](unsigned r, float vsum) {
            if (small_off >= 0) { atomicAdd(&ssmall[small_off + r], vsum); return; }
            if (cached) {
                const unsigned key = ((unsigned)level << 24) | r;
                const unsigned slot = ((r * 2654435761u) >> 18 ^ (unsigned)level * 977u) & (BWD_CACHE - 1);
                const unsigned old = atomicCAS(&ckey[slot], 0xFFFFFFFFu, key);
                if (old == 0xFFFFFFFFu || old == key) { atomicAdd(&cval[slot], vsum); return; }
            }
#if ENCB_EXP == 1          // experiment (WRONG results): no global atomics — the kernel's floor without them
            return;
#elif ENCB_EXP == 2        // experiment (results only right if one XCD owns the row): the add performed in the issuing XCD's L2
            __hip_atomic_fetch_add(gtb + (size_t)r * gstride, vsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return;
#endif
            unsafeAtomicAdd(gtb + (size_t)r * gstride, vsum);               // column 0 = row scalar
        };
        // the rows and corner weights of point j (quad-shared index math), and its eight row quarters
        auto prep = [&](int j, unsigned* row, float* tq) {
            const float x = rdlane(xi, j), y = rdlane(yi, j), z = rdlane(zi, j);
            int c0, c1;
            float t;
            const float xa = q == 0 ? x : (q == 1 ? y : z);
            level_corners(xa, L.cell, L.res, c0, c1, t);
            const int c0x = quad_bcast_i<0>(c0), c1x = quad_bcast_i<0>(c1);
            const int c0y = quad_bcast_i<1>(c0), c1y = quad_bcast_i<1>(c1);
            const int c0z = quad_bcast_i<2>(c0), c1z = quad_bcast_i<2>(c1);
            tq[0] = quad_bcast_f<0>(t); tq[1] = quad_bcast_f<1>(t); tq[2] = quad_bcast_f<2>(t);
            const int cx = (q & 2) ? c1x : c0x, cy = (q & 1) ? c1y : c0y;
            unsigned r0, r1;
            if (L.hashed) {
                const uint64_t hxy = (uint64_t)(uint32_t)cx ^ ((uint64_t)(uint32_t)cy * HASH_P1);
                r0 = grid_hash_mod(hxy ^ ((uint64_t)(uint32_t)c0z * HASH_P2), g);
                r1 = grid_hash_mod(hxy ^ ((uint64_t)(uint32_t)c1z * HASH_P2), g);
            } else {
                const unsigned rb = ((unsigned)cx * (unsigned)L.res + (unsigned)cy) * (unsigned)L.res;
                r0 = rb + (unsigned)c0z;
                r1 = rb + (unsigned)c1z;
            }
            row[0] = (unsigned)quad_bcast_i<0>((int)r0); row[1] = (unsigned)quad_bcast_i<0>((int)r1);
            row[2] = (unsigned)quad_bcast_i<1>((int)r0); row[3] = (unsigned)quad_bcast_i<1>((int)r1);
            row[4] = (unsigned)quad_bcast_i<2>((int)r0); row[5] = (unsigned)quad_bcast_i<2>((int)r1);
            row[6] = (unsigned)quad_bcast_i<3>((int)r0); row[7] = (unsigned)quad_bcast_i<3>((int)r1);
        };
        // Software pipeline (round 6): the 8 row gathers of point j + 1 are in flight while point j is reduced and scattered.  The
        // wave walks its points one after the other, and each point used to be a full dependent round trip — index math, 8 KB of
        // random 64-byte rows out of a 1 GB table, reductions, atomics: ~8 us per point at two waves per SIMD, which is what the
        // kernel cost for 1e4 pairs as for 5e4 (profiles/r6_training_step.md).
        unsigned rowc[8];
        float tc[3];
        float4 v[8];
        prep(0, rowc, tc);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = L.tab[(size_t)rowc[k] * 4];
        for (int j = 0; j < m; ++j) {
            unsigned rown[8];
            float tn[3];
            float4 vn[8];
            const bool more = j + 1 < m;                  // (wave-uniform)
            if (more) {
                prep(j + 1, rown, tn);
#pragma unroll
                for (int k = 0; k < 8; ++k) vn[k] = L.tab[(size_t)rown[k] * 4];
            }
            const unsigned* row = rowc;
            const float tx = tc[0], ty = tc[1], tz = tc[2];
            const float ux = 1.0f - tx, uy = 1.0f - ty, uz = 1.0f - tz;
            const float gl = sgo[wv][j][3 + level];
            float gtx = 0.f, gty = 0.f, gtz = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float wx = (k & 4) ? tx : ux, wy = (k & 2) ? ty : uy, wz = (k & 1) ? tz : uz;
                const float S = quad_sum((v[k].x + v[k].y) + (v[k].z + v[k].w));
                gtx += ((k & 4) ? 1.0f : -1.0f) * wy * wz * S;
                gty += ((k & 2) ? 1.0f : -1.0f) * wx * wz * S;
                gtz += ((k & 1) ? 1.0f : -1.0f) * wx * wy * S;
                if (q == 0) {
                    const float val = (wx * wy * wz) * gl;
                    if (row[k] == prow[k]) pval[k] += val;
                    else {
                        if (prow[k] != 0xFFFFFFFFu) flush(prow[k], pval[k]);
                        prow[k] = row[k];
                        pval[k] = val;
                    }
                }
            }
            // d out / d x_norm of this level, then summed over the 16 levels (lanes 4 apart)
            float ax = gtx * gl / L.cell, ay = gty * gl / L.cell, az = gtz * gl / L.cell;
#pragma unroll
            for (int d = 4; d < 64; d <<= 1) { ax += __shfl_xor(ax, d); ay += __shfl_xor(ay, d); az += __shfl_xor(az, d); }
            if (lane == 0) { sgx[wv][j][0] = ax; sgx[wv][j][1] = ay; sgx[wv][j][2] = az; }
            if (more) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { rowc[k] = rown[k]; v[k] = vn[k]; }
                tc[0] = tn[0]; tc[1] = tn[1]; tc[2] = tn[2];
            }
        }
        if (q == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (prow[k] != 0xFFFFFFFFu) flush(prow[k], pval[k]);
        }
        __builtin_amdgcn_wave_barrier();              // sgx / sgo are private to the wave: lane 0's stores above before the loads below
        if (io.g_xyz && lane < m) {
            io.g_xyz[(base + lane) * io.gx_ps + 0 * io.gx_cs] = (sgx[wv][lane][0] + sgo[wv][lane][0]) / ex;
            io.g_xyz[(base + lane) * io.gx_ps + 1 * io.gx_cs] = (sgx[wv][lane][1] + sgo[wv][lane][1]) / ey;
            io.g_xyz[(base + lane) * io.gx_ps + 2 * io.gx_cs] = (sgx[wv][lane][2] + sgo[wv][lane][2]) / ez;
        }
        __builtin_amdgcn_wave_barrier();              // ... and those loads before the next tile's stores
    }
    __syncthreads();
    // flush the write-combining cache: one global atomic per touched (level, row)
    for (int j = threadIdx.x; j < BWD_CACHE; j += ENC_BLOCK) {
        const unsigned key = ckey[j];
        if (key == 0xFFFFFFFFu) continue;
        const int l = (int)(key >> 24);
        const unsigned r = key & 0xFFFFFFu;
        float* base = io.rowgrad ? io.rowgrad + g.dense_off[l] : io.g_dense + g.dense_off[l] * 16;
        unsafeAtomicAdd(base + (size_t)r * gstride, cval[j]);
    }
    // flush the small-level accumulators (dense levels are contiguous in g_dense from dense_off[l])
    int off = 0;
    for (int l = 0; l < 16; ++l) {
        const int64_t rows = (int64_t)g.res[l] * g.res[l] * g.res[l];
        const bool small = l < (g.separate_dense ? g.start_hash : 0) && rows <= BWD_SMALL_ROWS && off + rows <= BWD_SMALL_FLOATS;
        if (!small) continue;
        float* gl_tab = io.rowgrad ? io.rowgrad + g.dense_off[l] : io.g_dense + g.dense_off[l] * 16;
        for (int j = threadIdx.x; j < (int)rows; j += ENC_BLOCK) {
            const float vv = ssmall[off + j];
            if (vv != 0.0f) unsafeAtomicAdd(gl_tab + (size_t)j * gstride, vv);
        }
        off += (int)rows;
    }
}

static int launch_part_encode_bwd_io(const GridDev& g, const EncBwdIO& io_in, hipStream_t st) {
    const EncBwdIO& io = io_in;
    const int64_t n = io.n_host;                             // the count or its upper bound (device count given)
    int tp = 64;
    while (tp > 16 && n / tp < 4096) tp >>= 1;
    int64_t tiles = cdiv(n, (int64_t)tp * ENC_WAVES);
    static const int gmax = getenv("INVR_ENCB_GRID") ? atoi(getenv("INVR_ENCB_GRID")) : 512;      // persistent: 2 workgroups per CU (79 KB of LDS each)
    unsigned grid = (unsigned)(tiles < gmax ? (tiles > 0 ? tiles : 1) : gmax);
    hipLaunchKernelGGL(k_part_encode_bwd, dim3(grid), dim3(ENC_BLOCK), 0, st, g, io);
    INVR_LAUNCH_CHECK();
    return 0;
}

int launch_part_encode_bwd(const GridDev& g, const float* xyz, const float* gout, int64_t n, float* g_dense, float* g_hash,
                           float* g_xyz, hipStream_t st) {
    EncBwdIO io{xyz, 3, 1, gout, 19, 1, g_xyz, 3, 1, n, nullptr, g_dense, g_hash, nullptr};
    return launch_part_encode_bwd_io(g, io, st);
}

// training pipeline: SoA pair lists (stride `stride`), device count, compact row-scalar gradient
int launch_part_encode_bwd_lists(const GridDev& g, const float* x_soa, const float* gout_soa, float* gx_soa, int64_t stride,
                                 int64_t n_max, const int32_t* count, float* rowgrad, hipStream_t st) {
    if (g.L != 16 || g.F != 16 || !g.sum || !g.sum_over_features || !g.include_input) {
        invr_set_error("part encoder backward supports n_levels=16, n_features_per_level=16, sum, sum_over_features, include_input");
        return 1;
    }
    EncBwdIO io{x_soa, 1, stride, gout_soa, 1, stride, gx_soa, 1, stride, n_max, count, nullptr, nullptr, rowgrad};
    return launch_part_encode_bwd_io(g, io, st);
}

// ---- inference-only row-sum variant -------------------------------------------------------------------
// With sum && sum_over_features the F features of a row are only ever used through their sum, and the
// sum commutes with the trilinear interpolation: a derived (rows,) table of row sums (rebuilt by the host
// whenever the tables change) shrinks the 1.09 GB of tables to 68 MB — resident in the 256 MB Infinity
// Cache — and each corner fetch to one dword.  Thread-per-pair: the 64 lanes of a wave hold 64 consecutive
// pairs of the list (eval frames: one depth slab of a few neighbouring rays, k_cull.hip; else consecutive samples of a ray), so neighbouring lanes fall into the
// same / adjacent 64-byte lines; the level constants are wave-uniform (scalar registers).
#define RS_BLOCK 256
#ifndef ENC_X2
#define ENC_X2 1          // A/B round 5: encoder 0.553 -> 0.542 ms, frame -1.1 % (gpurun_out/r5h), bit-identical
#endif
#ifndef ENC_MOD1R
#define ENC_MOD1R 1       // A/B round 5: encoder stage 0.534 -> 0.524 ms (gpurun_out/r5o), bit-identical
#endif
#ifndef ENC_HOIST
#define ENC_HOIST 1       // A/B round 5: encoder stage 0.538 -> 0.528 ms (gpurun_out/r5i), bit-identical
#endif
// (Measured and not kept, round 5: both levels' gathers + the next tile's coordinates in flight at once — a prepare / fetch / finish
//  split of the level — 93 instead of 69 VGPRs, 5 instead of 7 waves per SIMD: 0.540 -> 0.588 ms, gpurun_out/r5k.)
// one level of one point through the row-sum table (wave-uniform level l)
// FASTDIV: the caller has established, once per tile of pairs, that every lane's three coordinates are in div_exact's proven range
// and that this level's reciprocal is usable (rcell != 0) — the three quotients then take the 5-instruction reciprocal form without
// div_exact's per-call range test + ballot + branch (nine wave-uniform branches per level pair of a tile otherwise).  Same bits.
__device__ __forceinline__ void level_corners_fast(float x, float cell, float rcell, int res, int& c0, int& c1, float& t) {
    const float f = div_by_rcp(x, cell, rcell);
    const int a = (int)f, b = (int)(f + 1.0f);
    c0 = min(max(a, 0), res - 1);
    c1 = min(max(b, 0), res - 1);
    t = f - (float)c0;
}
// Round 6: the coarse dense level of a workgroup's level pair in LDS (k_part_encode_rs_xcd).  A workgroup of the XCD kernel serves ONE
// level pair {lg, 15 - lg} of one part at a time; when level lg is dense and small — levels 0..6 of the base-2 parts (8 .. 2197 rows)
// and level 0 of the body (4096 rows) — its row-sum table is staged once per part into ENC_LDS_ROWS floats of LDS and its 4 pair
// look-ups per point become ds_read2_b32 instead of vector-memory requests (the kernel is bound by those, not by bytes): for the
// base-2 parts 7 of the 16 levels leave the L1 / L2 request path.  LDSTAB = this level reads g_enc_tab; same arithmetic, same bits.
#define ENC_LDS_ROWS 4096
__shared__ float g_enc_tab[ENC_LDS_ROWS + 2];
template <bool FASTDIV = false, bool LDSTAB = false>
__device__ __forceinline__ float level_rowsum(const GridDev& g, const float* __restrict__ rs, int hstart, int l, float x, float y, float z) {
    const int res = g.res[l];
    const float cell = g.cell[l];
    int c0x, c1x, c0y, c1y, c0z, c1z;
    float tx, ty, tz;
    const float rcell = g.rcell[l];
    if (FASTDIV) {
        level_corners_fast(x, cell, rcell, res, c0x, c1x, tx);
        level_corners_fast(y, cell, rcell, res, c0y, c1y, ty);
        level_corners_fast(z, cell, rcell, res, c0z, c1z, tz);
    } else {
    level_corners(x, cell, rcell, res, c0x, c1x, tx);
    level_corners(y, cell, rcell, res, c0y, c1y, ty);
    level_corners(z, cell, rcell, res, c0z, c1z, tz);
    }
    unsigned row[8];
    const float* tab;
    if (l >= g.start_hash) {
        tab = rs + g.dense_rows + (int64_t)(l - hstart) * g.T;
        // (cx * 1) ^ (cy * P1) ^ (cz * P2) in 64 bits (:132-136).  The second corner of an axis is the first + d, d in {0, 1, 2}
        // (0: both clipped to the same cell; 2: f within half an ulp below an integer, where f + 1.0f rounds up and
        // trunc(f + 1) = trunc(f) + 2 — the reference's float arithmetic, :116): its product is the first's + d * prime — one
        // 64-bit multiply per axis instead of two (v_mul_lo/hi_u32 are quarter-rate instructions)
        const uint64_t hy0 = (uint64_t)(uint32_t)c0y * HASH_P1, hz0 = (uint64_t)(uint32_t)c0z * HASH_P2;
        const int dy = c1y - c0y, dz = c1z - c0z;
        const uint64_t hx[2] = {(uint64_t)(uint32_t)c0x, (uint64_t)(uint32_t)c1x};
        const uint64_t hy[2] = {hy0, hy0 + (dy > 0 ? HASH_P1 : 0ull) + (dy > 1 ? HASH_P1 : 0ull)};
        const uint64_t hz[2] = {hz0, hz0 + (dz > 0 ? HASH_P2 : 0ull) + (dz > 1 ? HASH_P2 : 0ull)};
        if (g.xdelta) {
            // x's hash prime is 1, so the four corners at c1x are the corners at c0x with the low bits m = c0x ^ c1x flipped:
            // X ^ m = X + m - 2 (X & m), a shift of |delta| <= m < 4096 — their reduction mod T is the c0x corner's + delta, folded
            // back into [0, T) (T > 2^14 > 2 |delta|: host-checked, GridDev.xdelta), instead of four
            // more 3-round reductions
            const uint32_t m = (uint32_t)c0x ^ (uint32_t)c1x;
#if ENC_MOD1R
            if (g.mod1r) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint64_t X = hx[0] ^ hy[(k >> 1) & 1] ^ hz[k & 1];
                    const uint32_t r0 = hash_mod24_1r(X, g.mod_k, g.mod_c, (uint32_t)g.T);
                    row[k] = r0;
                    // r0 + delta in (-T, 2 T): both fix-ups as unsigned minima (a negative value is a huge unsigned one)
                    uint32_t r1 = r0 + m - 2u * ((uint32_t)X & m);
                    r1 = min(r1, r1 + (uint32_t)g.T);
                    r1 = min(r1, r1 - (uint32_t)g.T);
                    row[4 + k] = r1;
                }
            } else
#endif
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint64_t X = hx[0] ^ hy[(k >> 1) & 1] ^ hz[k & 1];
                const uint32_t r0 = hash_mod24_2r(X, g.mod_k, g.mod_c, (uint32_t)g.T);
                row[k] = r0;
                uint32_t r1 = r0 + m - 2u * ((uint32_t)X & m);
                r1 = min(r1, r1 + (uint32_t)g.T);
                r1 = min(r1, r1 - (uint32_t)g.T);
                row[4 + k] = r1;
            }
        } else if (g.mod24) {
#pragma unroll
            for (int k = 0; k < 8; ++k) row[k] = hash_mod24(hx[k >> 2] ^ hy[(k >> 1) & 1] ^ hz[k & 1], g.mod_k, g.mod_c, (uint32_t)g.T);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) row[k] = grid_hash_mod(hx[k >> 2] ^ hy[(k >> 1) & 1] ^ hz[k & 1], g);
        }
    } else {
        tab = g.separate_dense ? rs + g.dense_off[l] : rs + (int64_t)l * g.T;
        // cx*res^2 + cy*res + cz (:124-129) with 24-bit multiplies: a dense level has res^3 <= T < 2^31, so res < 1291 and
        // cx*res + cy < res^2 < 2^21
        const unsigned ures = (unsigned)res;
        const unsigned bx0 = __umul24((unsigned)c0x, ures), bx1 = __umul24((unsigned)c1x, ures);
        const unsigned b00 = __umul24(bx0 + (unsigned)c0y, ures), b01 = __umul24(bx0 + (unsigned)c1y, ures);
        const unsigned b10 = __umul24(bx1 + (unsigned)c0y, ures), b11 = __umul24(bx1 + (unsigned)c1y, ures);
        // The two z corners of an (x, y) corner are ADJACENT floats of the row-sum table (row = .. + cz): one 8-byte load of
        // [z0, z0+1], z0 = min(c0z, res-2), instead of two 4-byte loads — the kernel is bound by the number of vector-memory
        // requests (L1 line lookups), not by bytes.  c0z is z0 or z0+1; c1z is c0z or c0z+1 (both res-1 when clipped) — or, when f
        // lies within half an ulp below an integer, c0z+2 (f + 1.0f rounds up, :116): that rare lane reloads its second corner.
        const unsigned z0 = (unsigned)min(c0z, res - 2);
        const bool s0 = (unsigned)c0z != z0, s1 = (unsigned)c1z != z0;       // take the second float of the pair
        const bool far1 = (unsigned)c1z > z0 + 1u;
        const unsigned bb[4] = {b00, b01, b10, b11};
        float vv[8];
        // all four pair loads are issued before the first use; the rare reload of a far second corner comes after them (inside
        // the loop its predicated load made the compiler wait for every pair load in turn: four serial round trips per level)
        float2 pr[4];
        if (LDSTAB) {
#pragma unroll
            for (int j = 0; j < 4; ++j) pr[j] = make_float2(g_enc_tab[bb[j] + z0], g_enc_tab[bb[j] + z0 + 1u]);       // ds_read2_b32
        } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) __builtin_memcpy(&pr[j], tab + bb[j] + z0, sizeof(float2));   // 4-byte aligned 8-byte load (global_load_dwordx2)
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            vv[2 * j] = s0 ? pr[j].y : pr[j].x;
            vv[2 * j + 1] = s1 ? pr[j].y : pr[j].x;
        }
        if (far1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) vv[2 * j + 1] = LDSTAB ? g_enc_tab[bb[j] + (unsigned)c1z] : tab[bb[j] + (unsigned)c1z];
        }
        const float ux = 1.0f - tx, uy = 1.0f - ty, uz = 1.0f - tz;
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k)     // weight_k = prod_axis ((1-o) + (2o-1) t)  (:157-158)
            acc = fmaf(((k & 4) ? tx : ux) * ((k & 2) ? ty : uy) * ((k & 1) ? tz : uz), vv[k], acc);
        return acc;
    }
    float v[8];
#if ENC_X2
    if (g.xdelta) {
        // x's hash prime is 1: the c1x corner of an (y, z) corner pair sits at row r0 + delta, and delta = +-1 for ~60 % of the pairs
        // (always when c0x is even: X ^ 1 = X +- 1).  Those lanes fetch BOTH rows with one 8-byte load at min(r0, r1); the others issue
        // two 4-byte loads.  The kernel is bound by the number of vector-memory requests (L1 line look-ups), not by bytes: 5.5
        // instead of 8 requests per hashed level and pair.  (A wrap through the table end makes |delta| large: two loads.)
        float2 pr[4];
        float s0[4], s1[4];
        bool adj[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t r0 = row[k], r1 = row[4 + k];
            adj[k] = r1 == r0 + 1u || r0 == r1 + 1u;
            pr[k] = make_float2(0.f, 0.f); s0[k] = 0.f; s1[k] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (adj[k]) __builtin_memcpy(&pr[k], tab + min(row[k], row[4 + k]), sizeof(float2));          // global_load_dwordx2, 4-byte aligned
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (!adj[k]) { s0[k] = tab[row[k]]; s1[k] = tab[row[4 + k]]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool up = row[4 + k] > row[k];
            v[k] = adj[k] ? (up ? pr[k].x : pr[k].y) : s0[k];
            v[4 + k] = adj[k] ? (up ? pr[k].y : pr[k].x) : s1[k];
        }
    } else
#endif
    {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = tab[row[k]];
    }
    const float ux = 1.0f - tx, uy = 1.0f - ty, uz = 1.0f - tz;
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k)     // weight_k = prod_axis ((1-o) + (2o-1) t)  (:157-158)
        acc = fmaf(((k & 4) ? tx : ux) * ((k & 2) ? ty : uy) * ((k & 1) ? tz : uz), v[k], acc);
    return acc;
}

__device__ __forceinline__ void encode_rs_part(const GridDev& g, const float* __restrict__ rs, const float* __restrict__ xs,
                                               int64_t stride, int cnt, int64_t cap, float* __restrict__ emb) {
    const float b0x = g.bounds[0], b0y = g.bounds[1], b0z = g.bounds[2];
    const float ex = g.bounds[3] - b0x, ey = g.bounds[4] - b0y, ez = g.bounds[5] - b0z;
    const int hstart = g.separate_dense ? g.start_hash : 0;
    for (int64_t i = (int64_t)blockIdx.x * RS_BLOCK + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * RS_BLOCK) {
        const float x = (xs[i] - b0x) / ex, y = (xs[stride + i] - b0y) / ey, z = (xs[2 * stride + i] - b0z) / ez;   // :112
        emb[i] = x; emb[cap + i] = y; emb[2 * cap + i] = z;
        emb[(int64_t)(EMB_K - 1) * cap + i] = 0.0f;                 // pad column
#pragma unroll 2
        for (int l = 0; l < 16; ++l) emb[(int64_t)(3 + l) * cap + i] = level_rowsum(g, rs, hstart, l, x, y, z);
    }
}

__global__ __launch_bounds__(RS_BLOCK) void k_part_encode_rs(GridDev g, const float* __restrict__ rs,
                                                             const float* __restrict__ xs, int64_t stride,
                                                             const int32_t* __restrict__ count, int64_t cap,
                                                             float* __restrict__ emb) {
    encode_rs_part(g, rs, xs, stride, *count, cap, emb);
}

// all five parts in ONE persistent launch: a workgroup walks its share of part 0, then of part 1, ... without a
// device-wide drain between parts (5 launches of this size spend ~15 us each on ramp-up and tail)
__global__ __launch_bounds__(RS_BLOCK) void k_part_encode_rs_all(EncodeAllArgs a) {
    for (int p = 0; p < INVR_NUM_PARTS; ++p)
        encode_rs_part(a.g[p], a.g[p].row_sums, a.xs[p], a.stride, a.counts[p], a.cap, a.emb[p]);
}

// XCD-partitioned variant: workgroups are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8), each with
// its own 4 MB L2.  Level group lg = blockIdx.x & 7 handles levels {lg, 15 - lg} (rotated per part so that the
// heavier hashed pairs do not always land on the same XCD) of EVERY pair tile: an XCD's L2 then only ever sees one
// eighth of the row-sum tables (8.5 MB of 68 MB) instead of all of them.
template <bool use_lds>
__global__ __launch_bounds__(RS_BLOCK) void k_part_encode_rs_xcd(EncodeAllArgs a) {
    const int xcd = blockIdx.x & 7;
    const int tstride = gridDim.x >> 3;
    int off = 0;                   // tiles of part p are dealt round-robin from where part p-1 stopped (see k_part_mlp_all)
    for (int p = 0; p < INVR_NUM_PARTS; ++p) {
        const GridDev& g = a.g[p];
        const float* __restrict__ rs = g.row_sums;
        const float* __restrict__ xs = a.xs[p];
        float* __restrict__ emb = a.emb[p];
        const int cnt = a.counts[p];
        int64_t tile0 = (int)(blockIdx.x >> 3) - off;
        if (tile0 < 0) tile0 += tstride;
        off = (int)((off + (cnt + RS_BLOCK - 1) / RS_BLOCK) % tstride);
        const int lg = (xcd + 3 * p) & 7;
        const int la = lg, lb = 15 - lg;
        const float b0x = g.bounds[0], b0y = g.bounds[1], b0z = g.bounds[2];
        const float ex = g.bounds[3] - b0x, ey = g.bounds[4] - b0y, ez = g.bounds[5] - b0z;
        const int hstart = g.separate_dense ? g.start_hash : 0;
        // level la in LDS when it is a dense level of at most ENC_LDS_ROWS rows (wave-uniform; res^3 rows, (:124-129))
        const int res_a = g.res[la];
        const bool lds_a = use_lds && la < g.start_hash && (int64_t)res_a * res_a * res_a <= ENC_LDS_ROWS && tile0 * RS_BLOCK < cnt;
        if (use_lds) __syncthreads();                 // (the previous part's look-ups are done before its table is replaced)
        if (lds_a) {
            const float* __restrict__ src = g.separate_dense ? rs + g.dense_off[la] : rs + (int64_t)la * g.T;
            const int rows = res_a * res_a * res_a;
            for (int t = threadIdx.x; t < rows; t += RS_BLOCK) g_enc_tab[t] = src[t];
            if (threadIdx.x < 2) g_enc_tab[rows + threadIdx.x] = 0.0f;          // (the pair read of the last row's z0 = res - 2 ends at rows - 1: never read, kept defined)
        }
        if (use_lds) __syncthreads();
        for (int64_t i = tile0 * RS_BLOCK + threadIdx.x; i < cnt; i += (int64_t)tstride * RS_BLOCK) {
            const float x = (xs[i] - b0x) / ex, y = (xs[a.stride + i] - b0y) / ey, z = (xs[2 * a.stride + i] - b0z) / ez;   // :112
            if (lg < 3) emb[(int64_t)lg * a.cap + i] = lg == 0 ? x : (lg == 1 ? y : z);
            else if (lg == 3) emb[(int64_t)(EMB_K - 1) * a.cap + i] = 0.0f;          // pad column
#if ENC_HOIST
            // div_exact's range test, once per tile instead of once per quotient (the coordinates are the same for every level)
            const bool okp = fabsf(x) > 8.7e-19f && fabsf(x) < 1.0e6f && fabsf(y) > 8.7e-19f && fabsf(y) < 1.0e6f && fabsf(z) > 8.7e-19f && fabsf(z) < 1.0e6f;
            const bool fast = __ballot(!okp) == 0ull;
            if (lds_a) {
                if (fast && g.rcell[la] != 0.0f) emb[(int64_t)(3 + la) * a.cap + i] = level_rowsum<true, true>(g, rs, hstart, la, x, y, z);
                else emb[(int64_t)(3 + la) * a.cap + i] = level_rowsum<false, true>(g, rs, hstart, la, x, y, z);
            } else
            if (fast && g.rcell[la] != 0.0f) emb[(int64_t)(3 + la) * a.cap + i] = level_rowsum<true>(g, rs, hstart, la, x, y, z);
            else emb[(int64_t)(3 + la) * a.cap + i] = level_rowsum<false>(g, rs, hstart, la, x, y, z);
            if (fast && g.rcell[lb] != 0.0f) emb[(int64_t)(3 + lb) * a.cap + i] = level_rowsum<true>(g, rs, hstart, lb, x, y, z);
            else emb[(int64_t)(3 + lb) * a.cap + i] = level_rowsum<false>(g, rs, hstart, lb, x, y, z);
#else
            emb[(int64_t)(3 + la) * a.cap + i] = level_rowsum(g, rs, hstart, la, x, y, z);
            emb[(int64_t)(3 + lb) * a.cap + i] = level_rowsum(g, rs, hstart, lb, x, y, z);
#endif
        }
    }
}

// one quad per table row: 4 x float4 of a 16-feature row (or F/4 lanes for narrower rows), pairwise sums
__global__ void k_row_sums(const float* __restrict__ tab, int64_t rows, int F, float* __restrict__ out) {
    const int per = F / 4;                                           // lanes per row
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t r = t / per;
    if (r >= rows) return;
    const int q = (int)(t - r * per);
    const float4 a = reinterpret_cast<const float4*>(tab)[r * per + q];
    float s = (a.x + a.y) + (a.z + a.w);
    if (per == 4) {
        s = quad_sum(s);
        if (q == 0) out[r] = s;
    } else {
        atomicAdd(out + r, s);                                       // F != 16: rare; out pre-zeroed by the launcher
    }
}

int launch_row_sums(const GridDev& g, float* out, hipStream_t st) {
    const int per = g.F / 4;
    auto run = [&](const float* tab, int64_t rows, float* o) -> int {
        if (rows == 0) return 0;
        if (per != 4) INVR_HIP(hipMemsetAsync(o, 0, rows * sizeof(float), st));
        hipLaunchKernelGGL(k_row_sums, dim3((unsigned)cdiv(rows * per, 256)), dim3(256), 0, st, tab, rows, g.F, o);
        INVR_LAUNCH_CHECK();
        return 0;
    };
    if (g.separate_dense) {
        if (run(g.dense, g.dense_rows, out)) return 1;
        return run(g.hash, (int64_t)(g.L - g.start_hash) * g.T, out + g.dense_rows);
    }
    return run(g.hash, (int64_t)g.L * g.T, out);
}

int launch_part_encode_rows_all(const EncodeAllArgs& a, hipStream_t st) {
    for (int p = 0; p < INVR_NUM_PARTS; ++p) {
        const GridDev& g = a.g[p];
        if (g.L != 16 || g.F != 16 || !g.sum || !g.sum_over_features || !g.include_input) {
            invr_set_error("part encoder kernel supports n_levels=16, n_features_per_level=16, sum, sum_over_features, include_input (got L=%d F=%d)", g.L, g.F);
            return 1;
        }
    }
    int64_t tiles = cdiv(a.cap, 64 * ENC_WAVES);
    unsigned grid = (unsigned)(tiles < 256 * 4 ? (tiles > 0 ? tiles : 1) : 256 * 4);
    hipLaunchKernelGGL(k_part_encode_rows_all, dim3(grid, INVR_NUM_PARTS), dim3(ENC_BLOCK), 0, st, a);
    INVR_LAUNCH_CHECK();
    return 0;
}

int launch_part_encode_all(const EncodeAllArgs& a, hipStream_t st) {
    for (int p = 0; p < INVR_NUM_PARTS; ++p) {
        const GridDev& g = a.g[p];
        if (g.L != 16 || !g.sum || !g.sum_over_features || !g.include_input || !g.row_sums) {
            invr_set_error("merged part encoder needs 16-level sum/sum_over_features grids with row-sum tables");
            return 1;
        }
    }
    int64_t tiles = cdiv(a.cap, RS_BLOCK);
    unsigned grid = (unsigned)(tiles < 256 * 8 ? (tiles > 0 ? tiles : 1) : 256 * 8);
    static const int xcd_mode = getenv("INVR_ENC_XCD") ? atoi(getenv("INVR_ENC_XCD")) : 1;
    if (xcd_mode) {
        unsigned gx = (unsigned)(tiles * 8 < 256 * 8 ? (tiles > 0 ? tiles * 8 : 8) : 256 * 8);      // a multiple of 8
        static const bool no_lds = getenv("INVR_ENC_NOLDS") != nullptr;                 // (A/B switch)
        if (no_lds) hipLaunchKernelGGL(k_part_encode_rs_xcd<false>, dim3(gx), dim3(RS_BLOCK), 0, st, a);
        else hipLaunchKernelGGL(k_part_encode_rs_xcd<true>, dim3(gx), dim3(RS_BLOCK), 0, st, a);
    } else {
        hipLaunchKernelGGL(k_part_encode_rs_all, dim3(grid), dim3(RS_BLOCK), 0, st, a);
    }
    INVR_LAUNCH_CHECK();
    return 0;
}

int launch_part_encode(const GridDev& g, const float* x_soa, int64_t stride, const int32_t* count, int64_t cap,
                       float* emb, hipStream_t st) {
    if (g.L != 16 || g.F != 16 || !g.sum || !g.sum_over_features || !g.include_input) {
        invr_set_error("part encoder kernel supports n_levels=16, n_features_per_level=16, sum, sum_over_features, include_input (got L=%d F=%d)", g.L, g.F);
        return 1;
    }
    int64_t tiles = cdiv(cap, 64 * ENC_WAVES);
    unsigned grid = (unsigned)(tiles < 256 * 8 ? (tiles > 0 ? tiles : 1) : 256 * 8);
    if (g.row_sums) {
        hipLaunchKernelGGL(k_part_encode_rs, dim3(grid), dim3(RS_BLOCK), 0, st, g, g.row_sums, x_soa, stride, count, cap, emb);
        INVR_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(k_part_encode, dim3(grid), dim3(ENC_BLOCK), 0, st, g, x_soa, stride, count, cap, emb);
    INVR_LAUNCH_CHECK();
    return 0;
}
