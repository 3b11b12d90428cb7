// Thread-per-point multi-resolution grid encoder (any L, F, reduction mode): the residual
// deformer's 8x2 grid and the stand-alone invr_grid_encode_fwd entry point.  The per-part 16x16
// grids of the render path use the wave-cooperative kernel in k_encode.hip instead.
// Semantics: HashEmbedder.forward, lib/networks/embedders/part_base_embedder.py:106-174.
#pragma once
#include "common.h"

// rows (in units of table rows, relative to the level's table) and trilinear weights of the 8
// corners of level l around normalised point x; returns the level's table base pointer.
__device__ __forceinline__ const float* grid_level_lookup(const GridDev& g, int l, const float* x, int64_t* rows, float* wts) {
    const int res = g.res[l];
    const float cell = g.cell[l];
    int c0[3], c1[3];
    float t[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) level_corners(x[a], cell, res, c0[a], c1[a], t[a]);
    const bool hashed = l >= g.start_hash;
#pragma unroll
    for (int k = 0; k < 8; ++k) {                           // offsets 000,001,...,111 (x y z, z fastest) :81-88
        const int cx = (k & 4) ? c1[0] : c0[0], cy = (k & 2) ? c1[1] : c0[1], cz = (k & 1) ? c1[2] : c0[2];
        // weight_k = prod_axis ((1-o) + (2o-1) t)  (:157-158)
        const float wx = (k & 4) ? t[0] : 1.0f - t[0], wy = (k & 2) ? t[1] : 1.0f - t[1], wz = (k & 1) ? t[2] : 1.0f - t[2];
        wts[k] = wx * wy * wz;
        if (hashed) rows[k] = grid_hash_mod((uint64_t)(uint32_t)cx ^ ((uint64_t)(uint32_t)cy * HASH_P1) ^ ((uint64_t)(uint32_t)cz * HASH_P2), g);
        else rows[k] = (int64_t)cx * res * res + (int64_t)cy * res + cz;
    }
    if (g.separate_dense) {
        if (hashed) return g.hash + (int64_t)(l - g.start_hash) * g.T * g.F;
        return g.dense + g.dense_off[l] * g.F;
    }
    return g.hash + (int64_t)l * g.T * g.F;                   // single (L,T,F) table (:153-154)
}

__device__ __forceinline__ void grid_normalise(const GridDev& g, const float* xyz, float* x) {
#pragma unroll
    for (int a = 0; a < 3; ++a) x[a] = (xyz[a] - g.bounds[a]) / (g.bounds[3 + a] - g.bounds[a]);   // :112
}

// compile-time L, F; sum=False, include_input=True: out = [x(3), level0 feats(F), level1 ...]
template <int L, int F>
__device__ __forceinline__ void grid_encode_concat(const GridDev& g, const float* xyz, float* out) {
    float x[3];
    grid_normalise(g, xyz, x);
    out[0] = x[0]; out[1] = x[1]; out[2] = x[2];
    // two levels (16 row fetches) in flight at a time: full unrolling hoists all 8*L fetches and
    // costs > 200 VGPRs (2 waves/SIMD) for no extra memory-level parallelism that matters here
#pragma unroll 2
    for (int l = 0; l < L; ++l) {
        int64_t rows[8];
        float wts[8];
        const float* tab = grid_level_lookup(g, l, x, rows, wts);
        float acc[F];
#pragma unroll
        for (int f = 0; f < F; ++f) acc[f] = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float* r = tab + rows[k] * F;
#pragma unroll
            for (int f = 0; f < F; ++f) acc[f] = fmaf(wts[k], r[f], acc[f]);
        }
#pragma unroll
        for (int f = 0; f < F; ++f) out[3 + l * F + f] = acc[f];
    }
}
