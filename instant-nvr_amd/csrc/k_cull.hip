// K0 + K1: ray sampling, world->pose, near-surface cull, ORDERED stream compaction.
// Replaces inb_renderer.py:15-31 (get_wsampling_points) and
// inb_part_network_multiassign.py:128-140 (world->pose, pnorm < smpl_thresh, nonzero, gathers)
// without the host-syncing nonzero: three stream-ordered launches
//   k_cull_flag   : one thread per ray-sample -> 64-bit survivor mask per wave + per-block count
//   k_scan_blocks : exclusive scan of the tile counts inside super-blocks of 1024 tiles + the super-blocks' totals
//   k_compact     : rank = totals of the super-blocks before + tile offset + wave prefix + popcount(mask below lane) -> active list
// The active list is in ray-major / sample-minor order, exactly the order torch.nonzero gives,
// so the train-time (Na*P, .) layouts of resd/tpts/tocc keep the reference's row order.  Eval frames
// (Workspace::ord_rows > 0) use the depth-windowed order of k_scan_blocks_win / k_compact_win below instead.
#include <stdlib.h>
#include "pipeline.h"

#include "front_bodies.h"

__global__ void k_cull_cells(VolDev v, float thresh_hi, uint8_t* __restrict__ mask, int32_t* __restrict__ live, int32_t* __restrict__ n_live,
                             uint8_t* __restrict__ voxcls) {
    cull_cells_body(v, thresh_hi, mask, live, n_live, voxcls, (int)(blockIdx.x * blockDim.x + threadIdx.x));
}

template <bool MASKED, bool FAST, bool RAY4>
__global__ __launch_bounds__(CULL_BLOCK) void k_cull_flag(RenderArgs a, Workspace w, double inv_S, float lin_step) {
    cull_flag_body<MASKED, FAST, RAY4>(a, w, inv_S, lin_step, (int64_t)blockIdx.x);
}

#define SCAN_T 1024
// Exclusive scan of the per-tile survivor counts in two levels WITHOUT a second pass or any inter-workgroup hand-shake: one workgroup
// per SCAN_T tiles writes the prefix inside its super-block (block_off) and the super-block's total (super_tot); k_compact adds the
// totals of the super-blocks before its own (<= 32 wave-uniform scalar loads for the 32 k tiles of a 512x512x128 frame) and its
// first workgroup publishes the survivor count.  (One workgroup scanning all counts in passes of 8192 took 31 us per frame.)
__global__ __launch_bounds__(SCAN_T) void k_scan_blocks(Workspace w, int64_t nb) {
    __shared__ int wsum[SCAN_T / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * SCAN_T + threadIdx.x;
    const int v = i < nb ? w.block_cnt[i] : 0;
    const int x = wave_incl_sum_i(v);
    if (lane == 63) wsum[wv] = x;
    __syncthreads();
    int woff = 0;
    for (int k = 0; k < wv; ++k) woff += wsum[k];
    if (i < nb) w.block_off[i] = woff + x - v;
    if (threadIdx.x == SCAN_T - 1) w.super_tot[blockIdx.x] = woff + x;
}

__global__ __launch_bounds__(CULL_BLOCK) void k_compact(RenderArgs a, Workspace w, int64_t max_active) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long* mk = w.mask + (int64_t)blockIdx.x * (CULL_TILE / 64);
    int off = w.block_off[blockIdx.x];
    {   // + the totals of the super-blocks before this tile's: one load per lane, summed over the wave on the DPP network
        const int64_t sb = (int64_t)(blockIdx.x / SCAN_T);
        int add = 0;
        for (int64_t j0 = 0; j0 < sb; j0 += 64) add += (j0 + lane < sb) ? w.super_tot[j0 + lane] : 0;
        off += __builtin_amdgcn_readlane(wave_incl_sum_i(add), 63);
    }
    if (blockIdx.x == 0 && wv == 0) {
        // the survivor count = the total over all super-blocks; clamped to the workspace capacity (overflow is reported, never written)
        const int64_t ns = ((int64_t)gridDim.x + SCAN_T - 1) / SCAN_T;
        int64_t na = 0;
        for (int64_t j0 = 0; j0 < ns; j0 += 64) {
            const int t = (j0 + lane < ns) ? w.super_tot[j0 + lane] : 0;
            na += __builtin_amdgcn_readlane(wave_incl_sum_i(t), 63);
        }
        if (lane == 0) {
            if (na > max_active) {
                w.counters[CNT_OVERFLOW] = 1;
                na = max_active;
            }
            w.counters[CNT_ACTIVE] = (int)na;
        }
    }
#pragma unroll
    for (int k = 0; k < CULL_PER; ++k) {
        const int64_t i = (int64_t)blockIdx.x * CULL_TILE + k * CULL_BLOCK + threadIdx.x;
        int woff = off;
        for (int j = 0; j < wv; ++j) woff += __popcll(mk[k * (CULL_BLOCK / 64) + j]);
        const unsigned long long m = mk[k * (CULL_BLOCK / 64) + wv];
        const bool keep = (m >> lane) & 1ull;
        const int rank = woff + __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) w.word_off[(int64_t)blockIdx.x * (CULL_TILE / 64) + k * (CULL_BLOCK / 64) + wv] = woff;
        if (i < a.N && keep && rank < max_active) w.active_idx[rank] = (int32_t)i;
        for (int j = 0; j < CULL_BLOCK / 64; ++j) off += __popcll(mk[k * (CULL_BLOCK / 64) + j]);
    }
}

// ---- windowed survivor order (Workspace::ord_rows > 0) ----------------------------------------------------------------------
// A block = ord_rows rays x ord_cols windows of 8 samples = 8192 ray-samples = 8 cull tiles = 128 mask words.  Inside a block the
// survivors are ranked by (window, ray, sample): the exclusive scan of the 1024 byte popcounts in column-major order.  Same three
// launches as the ray-major order: the scan sums 8 tile counts per block, the compaction ranks inside the block.
#define WIN_BLOCK 8192
#define WIN_BYTES (WIN_BLOCK / 8)
__global__ __launch_bounds__(SCAN_T) void k_scan_blocks_win(Workspace w, int64_t nb_tiles, int64_t nblk) {
    __shared__ int wsum[SCAN_T / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * SCAN_T + threadIdx.x;
    int v = 0;
    if (i < nblk)
        for (int t = 0; t < WIN_BLOCK / CULL_TILE; ++t) v += (i * (WIN_BLOCK / CULL_TILE) + t < nb_tiles) ? w.block_cnt[i * (WIN_BLOCK / CULL_TILE) + t] : 0;
    const int x = wave_incl_sum_i(v);
    if (lane == 63) wsum[wv] = x;
    __syncthreads();
    int woff = 0;
    for (int k = 0; k < wv; ++k) woff += wsum[k];
    if (i < nblk) w.block_off[i] = woff + x - v;
    if (threadIdx.x == SCAN_T - 1) w.super_tot[blockIdx.x] = woff + x;
}

__global__ __launch_bounds__(CULL_BLOCK) void k_compact_win(RenderArgs a, Workspace w, int64_t max_active, int64_t n_words) {
    __shared__ unsigned long long smask[WIN_BLOCK / 64];
    __shared__ int soff[WIN_BYTES];
    __shared__ int wsum[CULL_BLOCK / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int base = w.block_off[blockIdx.x];
    {
        const int64_t sb = (int64_t)(blockIdx.x / SCAN_T);
        int add = 0;
        for (int64_t j0 = 0; j0 < sb; j0 += 64) add += (j0 + lane < sb) ? w.super_tot[j0 + lane] : 0;
        base += __builtin_amdgcn_readlane(wave_incl_sum_i(add), 63);
    }
    if (blockIdx.x == 0 && wv == 0) {                  // the survivor count, clamped to the workspace capacity (k_compact)
        const int64_t ns = ((int64_t)gridDim.x + SCAN_T - 1) / SCAN_T;
        int64_t na = 0;
        for (int64_t j0 = 0; j0 < ns; j0 += 64) {
            const int t = (j0 + lane < ns) ? w.super_tot[j0 + lane] : 0;
            na += __builtin_amdgcn_readlane(wave_incl_sum_i(t), 63);
        }
        if (lane == 0) {
            if (na > max_active) {
                w.counters[CNT_OVERFLOW] = 1;
                na = max_active;
            }
            w.counters[CNT_ACTIVE] = (int)na;
        }
    }
    const int64_t word0 = (int64_t)blockIdx.x * (WIN_BLOCK / 64);
    if (threadIdx.x < WIN_BLOCK / 64) smask[threadIdx.x] = word0 + threadIdx.x < n_words ? w.mask[word0 + threadIdx.x] : 0ull;
    __syncthreads();
    // thread t ranks the bytes at column-major positions 4 t .. 4 t + 3 (ord_rows is a multiple of 4: one window, four rays)
    const int rows = w.ord_rows, cols = w.ord_cols;
    const uint8_t* sbytes = reinterpret_cast<const uint8_t*>(smask);
    int fb[4], cnt[4], tot = 0;
    unsigned bits[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int q = threadIdx.x * 4 + e;
        fb[e] = (q % rows) * cols + q / rows;           // byte of (ray q % rows, window q / rows) in the block's flat sample order
        bits[e] = sbytes[fb[e]];
        cnt[e] = __popc(bits[e]);
        tot += cnt[e];
    }
    const int incl = wave_incl_sum_i(tot);
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int off = base + incl - tot;
    for (int k = 0; k < wv; ++k) off += wsum[k];
    const int64_t i0 = (int64_t)blockIdx.x * WIN_BLOCK;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        soff[fb[e]] = off;
        unsigned b = bits[e];
        int slot = off;
        while (b) {
            const int bit = __ffs((int)b) - 1;
            b &= b - 1u;
            if (slot < max_active) w.active_idx[slot] = (int32_t)(i0 + fb[e] * 8 + bit);
            ++slot;
        }
        off += cnt[e];
    }
    __syncthreads();
    const int64_t n_bytes = (a.N + 7) >> 3, byte0 = (int64_t)blockIdx.x * WIN_BYTES;
    for (int j = threadIdx.x; j < WIN_BYTES; j += CULL_BLOCK)
        if (byte0 + j < n_bytes) w.byte_off[byte0 + j] = soff[j];
}

// cell mask of the cull + list of the live cells (for the KNN's lattice classification); 1 = built
int launch_cull_cells(const RenderArgs& a, const Workspace& w, hipStream_t st) {
    const VolDev& v = a.scene.pbw;
    const int64_t cells = (int64_t)v.dx * v.dy * v.dz;
    static const bool no_mask = getenv("INVR_NO_CULLMASK") != nullptr;
    if (cells > CULL_MASK_MAX || cells > VOXMASK_MAX_CELLS || no_mask) return 0;
    hipLaunchKernelGGL(k_cull_cells, dim3((unsigned)cdiv(cells, 256)), dim3(256), 0, st, v, a.scene.thresh * (1.0f + 1e-5f), w.cullmask,
                       w.knn.live_cells, w.counters + CNT_LIVE, w.knn.voxcls);
    if (hipGetLastError() != hipSuccess) return 0;
    return 1;
}

int launch_cull(const RenderArgs& a, const Workspace& w, int64_t max_active, bool have_cells, bool flags_done, hipStream_t st) {
    int64_t nb = cdiv(a.N, CULL_TILE);
    const VolDev& v = a.scene.pbw;
    const int64_t cells = (int64_t)v.dx * v.dy * v.dz;
    const double inv_S = 1.0 / (double)a.S;
    const float lin_step = 1.0f / (float)(a.S - 1);                // linspace01's step, the same IEEE division
    if (flags_done) {
    } else if (have_cells && a.N >= 4 * cells) {        // the mask pays for itself on full frames only
        const bool fast = !a.wpts && !a.jitter && a.N < (1ll << 31) && cells * v.c < (1ll << 31) && a.S >= 2 &&
                      v.dx <= 1024 && v.dy <= 1024 && v.dz <= 1024;      // (front_bodies.h: the pre-test's error bound)
        if (fast && (a.S & 3) == 0) hipLaunchKernelGGL((k_cull_flag<true, true, true>), dim3((unsigned)nb), dim3(CULL_BLOCK), 0, st, a, w, inv_S, lin_step);
        else if (fast) hipLaunchKernelGGL((k_cull_flag<true, true, false>), dim3((unsigned)nb), dim3(CULL_BLOCK), 0, st, a, w, inv_S, lin_step);
        else hipLaunchKernelGGL((k_cull_flag<true, false, false>), dim3((unsigned)nb), dim3(CULL_BLOCK), 0, st, a, w, inv_S, lin_step);
    } else {
        hipLaunchKernelGGL((k_cull_flag<false, false, false>), dim3((unsigned)nb), dim3(CULL_BLOCK), 0, st, a, w, inv_S, lin_step);
    }
    INVR_LAUNCH_CHECK();
    if (w.ord_rows > 0) {
        const int64_t nblk = cdiv(a.N, WIN_BLOCK);
        hipLaunchKernelGGL(k_scan_blocks_win, dim3((unsigned)cdiv(nblk, SCAN_T)), dim3(SCAN_T), 0, st, w, nb, nblk);
        INVR_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_compact_win, dim3((unsigned)nblk), dim3(CULL_BLOCK), 0, st, a, w, max_active, nb * (CULL_TILE / 64));
        INVR_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(k_scan_blocks, dim3((unsigned)cdiv(nb, SCAN_T)), dim3(SCAN_T), 0, st, w, nb);
    INVR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_compact, dim3((unsigned)nb), dim3(CULL_BLOCK), 0, st, a, w, max_active);
    INVR_LAUNCH_CHECK();
    return 0;
}

// ---- stand-alone volume sampler (invr_sample_volume; blend_utils.py:501-555) -------------------
template <int NC>
__global__ void k_sample_volume(VolDev v, int c0, const float* pts, int64_t n, float* out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float o[NC];
    sample_volume_dev<NC>(v, c0, pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2], o);
#pragma unroll
    for (int c = 0; c < NC; ++c) out[i * NC + c] = o[c];
}

int launch_sample_volume(const VolDev& v, int c0, int nc, const float* pts, int64_t n, float* out, hipStream_t st) {
    if (n == 0) return 0;
    dim3 g((unsigned)cdiv(n, 256)), b(256);
    if (nc == 1) hipLaunchKernelGGL(k_sample_volume<1>, g, b, 0, st, v, c0, pts, n, out);
    else if (nc == 2) hipLaunchKernelGGL(k_sample_volume<2>, g, b, 0, st, v, c0, pts, n, out);
    else { invr_set_error("invr_sample_volume: nc must be 1 or 2 (got %d)", nc); return 1; }
    INVR_LAUNCH_CHECK();
    return 0;
}

// ---- stand-alone sampler + world->pose of selected ray-samples (invr_pose_points) ---------------------------------
// inb_renderer.py:15-31 + blend_utils.py:366-382 through sample_pose_point, the device function the render kernels use:
// a caller (tests, the training path) gets bit-identical pose points / directions for any ray-sample index.
__global__ void k_pose_points(RenderArgs a, const int32_t* __restrict__ idx, int64_t n, float* __restrict__ pts, float* __restrict__ dirs) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    float px, py, pz, pd[3];
    sample_pose_point(a, idx ? (int64_t)idx[j] : j, px, py, pz, nullptr, dirs ? pd : nullptr);
    pts[j * 3] = px; pts[j * 3 + 1] = py; pts[j * 3 + 2] = pz;
    if (dirs) { dirs[j * 3] = pd[0]; dirs[j * 3 + 1] = pd[1]; dirs[j * 3 + 2] = pd[2]; }
}

int launch_pose_points(const RenderArgs& a, const int32_t* idx, int64_t n, float* pts, float* dirs, hipStream_t st) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_pose_points, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, a, idx, n, pts, dirs);
    INVR_LAUNCH_CHECK();
    return 0;
}
