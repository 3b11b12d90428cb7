// Row f1: the training iteration of the path as two stream-ordered passes, no host round trip in between.
//
//   invr_train_fwd : geometry (sampling with jitter, cull, KNN skinning, warp, deformer) -> per-part 64-byte-row encoder ->
//                    part MLPs -> merge + compositing (invr_render_fwd's own kernels) -> distortion regulariser
//                    (inb_renderer.py:96-103) -> offset / pair-regulariser terms (inb_trainer.py:45-48,89-92 with
//                    inb_renderer.py:78-94, crit.py:8-18) reduced on the device
//   invr_train_bwd : distortion^T -> compositing^T -> max-occupancy merge^T -> per part {MLP^T (MFMA) -> weight gradients ->
//                    encoder^T (row-scalar table gradients + canonical-point gradient)} -> deformer^T over {listed pairs,
//                    pair-regulariser neighbours} (MLP^T, grid^T) -> weight gradients
//
// Every count (survivors, pairs per part, selected regulariser pairs) stays in device counters; kernels are launched over
// the capacity and read the counts.  The reference does the same work through ~3 k ATen calls, 7 host-syncing nonzero()s
// and dense (Na*P, .) scatter tensors per iteration.
#include <string.h>
#include "pipeline.h"
#include "grid_generic.h"
#include "mlp_common.h"
#include "train.h"

#define TR_BLOCK 256

// ---- inverse map (slot, part) -> position in part's pair list -------------------------------------------------------
__global__ void k_pair_index(Workspace w, TrainWs t) {
    const int p = blockIdx.y;
    const int cnt = w.counters[CNT_PAIRS + p];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * blockDim.x)
        t.pair_of[(int64_t)w.l_slot[p][i] * INVR_NUM_PARTS + p] = (int32_t)i;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    float s = 0.0f;
    for (int k = 0; k < TR_BLOCK / 64; ++k) s += red[k];
    return s;
}

// list position of dense row (slot, part): the pair itself when listed, the part's far constant (last list entry) when far
__device__ __forceinline__ int dense_row_pair(const Workspace& w, const TrainWs& t, int slot, int p) {
    const unsigned fl = w.pflags[slot], ff = w.farflags[slot];
    if (fl & (1u << p)) return t.pair_of[(int64_t)slot * INVR_NUM_PARTS + p];
    if (ff & (1u << p)) return w.counters[CNT_PAIRS + p] - 1;
    return -1;
}

// Offset term sum_rows ||resd|| over the reference's dense (Na*P, 3) tensor (inb_trainer.py:89-92: zero rows of unflagged
// pairs count in the mean) and selection of the pair-regulariser rows |tocc - 0.5| < 0.02 (inb_renderer.py:80-86) with their
// jittered neighbours tpts + (u - 0.5) * 0.01 (inb_part_network_multiassign.py:34-47).  noise: (Na*P, 3) uniform [0,1) per
// dense row, or NULL = no pair regulariser.
__global__ __launch_bounds__(TR_BLOCK) void k_train_terms(Workspace w, TrainWs t, const float* __restrict__ noise) {
    __shared__ float red[TR_BLOCK / 64];
    const int na = w.counters[CNT_ACTIVE];
    const int64_t rows = (int64_t)na * INVR_NUM_PARTS;
    float acc = 0.0f;
    for (int64_t r = (int64_t)blockIdx.x * TR_BLOCK + threadIdx.x; r < rows; r += (int64_t)gridDim.x * TR_BLOCK) {
        const int slot = (int)(r / INVR_NUM_PARTS), p = (int)(r - (int64_t)slot * INVR_NUM_PARTS);
        const int i = dense_row_pair(w, t, slot, p);
        if (i < 0) continue;
        const float r0 = w.l_r[p][i], r1 = w.l_r[p][w.lcap + i], r2 = w.l_r[p][2 * w.lcap + i];
        acc += sqrtf(r0 * r0 + r1 * r1 + r2 * r2);
        if (!noise) continue;
        const float tocc = w.occp[p][i];
        if (fabsf(tocc - 0.5f) < 0.02f) {
            const int k = atomicAdd(&w.counters[CNT_NB], 1);
            if (k < t.NB) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float tp = w.l_x[p][c * w.lcap + i] - w.l_r[p][c * w.lcap + i];       // init_bigpose
                    t.nb_x[(int64_t)k * 3 + c] = tp + (noise[r * 3 + c] - 0.5f) * 0.01f;
                }
                t.nb_ref[k] = (p << 28) | i;
            }
        }
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0 && acc != 0.0f) atomicAdd(&t.terms[TERM_OFFSET_SUM], acc);
    if (blockIdx.x == 0 && threadIdx.x == 0) t.terms[TERM_OFFSET_ROWS] = (float)rows;
}

// ---- thread-per-point deformer (uv_deformer.py:31-38), forward with kept activations --------------------------------
// (BWD: the backward's recompute — activations AND their derivative factors softplus'(z) = sigmoid(z) formed from the pre-activation
// (sigmoid_acc: ~3 ulp RELATIVE for every z).  The earlier form 1 - exp(-softplus(z)) is a cancellation for z < 0: 6e-8 ABSOLUTE error on a
// factor ~e^z, and with Adam's eps = 1e-15 every extra bit of gradient noise flips the sign of more rounding-level steps — the
// deformer's first layer agreed with the float32 oracle on 0.67-0.94 of its elements after three steps, 0.99 now
// (tests/test_gpu_training.py::test_configs3_real_shape_three_steps_vs_oracle_autograd))
template <bool BWD> struct DeformActT { float feat[19]; float h1[32]; float h2[32]; float th[3]; float s1[BWD ? 32 : 1]; float s2[BWD ? 32 : 1]; };
typedef DeformActT<false> DeformAct;

// (the weights through any pointers: the MlpDev's global tensors, or a workgroup's LDS copy — same operations, same order)
template <bool BWD>
__device__ __forceinline__ void deform_fwd_act_w(const SceneDev& s, const GridDev& dg, const float* W0, const float* B0, const float* W1,
                                                 const float* B1, const float* W2, const float* B2, const float* xb, float* uvt,
                                                 DeformActT<BWD>& a) {
    sample_volume_dev<2>(s.tuv, 0, xb[0], xb[1], xb[2], uvt);
    uvt[2] = s.frame_dim[0];
    grid_encode_concat<8, 2>(dg, uvt, a.feat);
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        float acc = B0[j];
#pragma unroll
        for (int i = 0; i < 19; ++i) acc = fmaf(W0[j * 19 + i], a.feat[i], acc);
        a.h1[j] = softplus_f(acc);
        if (BWD) a.s1[j] = sigmoid_acc(acc);
        if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);       // (keeps the weight loads of later neurons from being hoisted: registers)
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        float acc = B1[j];
#pragma unroll
        for (int i = 0; i < 32; ++i) acc = fmaf(W1[j * 32 + i], a.h1[i], acc);
        a.h2[j] = softplus_f(acc);
        if (BWD) a.s2[j] = sigmoid_acc(acc);
        if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float acc = B2[j];
#pragma unroll
        for (int i = 0; i < 32; ++i) acc = fmaf(W2[j * 32 + i], a.h2[i], acc);
        a.th[j] = tanhf(acc);
    }
}
template <bool BWD>
__device__ __forceinline__ void deform_fwd_act(const SceneDev& s, const GridDev& dg, const MlpDev& dm, const float* xb, float* uvt,
                                               DeformActT<BWD>& a) {
    deform_fwd_act_w<BWD>(s, dg, dm.w[0], dm.b[0], dm.w[1], dm.b[1], dm.w[2], dm.b[2], xb, uvt, a);
}

// neighbours of the selected rows through the deformer; pair term of crit.reg_raw_crit (crit.py:8-18):
// || v_nb / (|v_nb| + 1e-8) - v_self / (|v_self| + 1e-8) ||, summed (the mean's divisor n stays on the device)
// (round 6: the MLP's weights from an LDS copy, as k_deform_bwd — they were ~1100 wave-uniform vector loads per thread)
#define PT_O_B0 (32 * 19)
#define PT_O_W1 (PT_O_B0 + 32)
#define PT_O_B1 (PT_O_W1 + 32 * 32)
#define PT_O_W2 (PT_O_B1 + 32)
#define PT_O_B2 (PT_O_W2 + 3 * 32)
__global__ __launch_bounds__(128) void k_pair_term_fwd(SceneDev s, GridDev dg, MlpDev dm, Workspace w, TrainWs t) {
    __shared__ float red[TR_BLOCK / 64];
    __shared__ __attribute__((aligned(16))) float lw[PT_O_B2 + 4];
    const int nsel = min(w.counters[CNT_NB], (int)t.NB);
    float acc = 0.0f;
    if ((int)(blockIdx.x * blockDim.x) < nsel) {            // (block-uniform: workgroups without a row stage nothing)
        for (int k = threadIdx.x; k < 32 * 19; k += blockDim.x) lw[k] = dm.w[0][k];
        for (int k = threadIdx.x; k < 32 * 32; k += blockDim.x) lw[PT_O_W1 + k] = dm.w[1][k];
        if (threadIdx.x < 96) lw[PT_O_W2 + threadIdx.x] = dm.w[2][threadIdx.x];
        if (threadIdx.x < 32) { lw[PT_O_B0 + threadIdx.x] = dm.b[0][threadIdx.x]; lw[PT_O_B1 + threadIdx.x] = dm.b[1][threadIdx.x]; }
        if (threadIdx.x < 3) lw[PT_O_B2 + threadIdx.x] = dm.b[2][threadIdx.x];
    }
    __syncthreads();
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nsel; k += gridDim.x * blockDim.x) {
        float xb[3] = {t.nb_x[(int64_t)k * 3], t.nb_x[(int64_t)k * 3 + 1], t.nb_x[(int64_t)k * 3 + 2]}, uvt[3];
        DeformAct a;
        deform_fwd_act_w<false>(s, dg, lw, lw + PT_O_B0, lw + PT_O_W1, lw + PT_O_B1, lw + PT_O_W2, lw + PT_O_B2, xb, uvt, a);
        float vn[3], vs[3];
        const int ref = t.nb_ref[k], p = ref >> 28, i = ref & 0x0FFFFFFF;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            vn[c] = 0.05f * a.th[c];
            t.nb_r[(int64_t)k * 3 + c] = vn[c];
            vs[c] = w.l_r[p][c * w.lcap + i];
        }
        const float ln = sqrtf(vn[0] * vn[0] + vn[1] * vn[1] + vn[2] * vn[2]) + 1e-8f;
        const float ls = sqrtf(vs[0] * vs[0] + vs[1] * vs[1] + vs[2] * vs[2]) + 1e-8f;
        const float d0 = vn[0] / ln - vs[0] / ls, d1 = vn[1] / ln - vs[1] / ls, d2 = vn[2] / ln - vs[2] / ls;
        acc += sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) red[wv] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float sum = 0.0f;
        for (int q = 0; q < (int)blockDim.x / 64; ++q) sum += red[q];
        if (sum != 0.0f) atomicAdd(&t.terms[TERM_PAIR_SUM], sum);
        if (blockIdx.x == 0) t.terms[TERM_PAIR_ROWS] = (float)nsel;
    }
}

int launch_train_terms(const RenderArgs& a, const Workspace& w, const TrainWs& t, const GridDev& dg, const MlpDev& dm,
                       const float* noise, hipStream_t st) {
    int64_t tiles = cdiv(w.lcap, 256);
    unsigned gx = (unsigned)(tiles < 512 ? (tiles > 0 ? tiles : 1) : 512);
    hipLaunchKernelGGL(k_pair_index, dim3(gx, INVR_NUM_PARTS), dim3(256), 0, st, w, t);
    INVR_LAUNCH_CHECK();
    int64_t rt = cdiv(w.lcap * INVR_NUM_PARTS, TR_BLOCK);
    hipLaunchKernelGGL(k_train_terms, dim3((unsigned)(rt < 1024 ? (rt > 0 ? rt : 1) : 1024)), dim3(TR_BLOCK), 0, st, w, t, noise);
    INVR_LAUNCH_CHECK();
    if (noise) {
        hipLaunchKernelGGL(k_pair_term_fwd, dim3(512), dim3(128), 0, st, a.scene, dg, dm, w, t);
        INVR_LAUNCH_CHECK();
    }
    return 0;
}

// ---- distortion regulariser, backward (inb_renderer.py:96-103): L_r = sum_ij w_i w_j |m_i - m_j|, m constant ----------
// dL_r/dw_i = 2 sum_j w_j |m_i - m_j|
__global__ __launch_bounds__(256) void k_distortion_bwd(const float* __restrict__ weights, const float* __restrict__ z,
                                                        const float* __restrict__ g_dist, int64_t R, int S, float* __restrict__ g_w) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= R) return;
    const float* wr = weights + ray * S;
    const float* zr = z + ray * S;
    const float g = g_dist[ray];
    for (int i = lane; i < S; i += 64) {
        const float mi = (zr[i] + zr[min(i + 1, S - 1)]) / 2.0f;
        float row = 0.0f;
        for (int j = 0; j < S; ++j) {
            const float mj = (zr[j] + zr[min(j + 1, S - 1)]) / 2.0f;
            row = fmaf(wr[j], fabsf(mi - mj), row);
        }
        g_w[ray * S + i] = 2.0f * g * row;
    }
}

// ---- max-occupancy merge, backward (inb_part_network_multiassign.py:229-256 + the scatter of :156-159) ---------------
// The merged raw of a survivor is the (rgb, occ) of its first-maximum-occupancy part; its gradient goes to that (slot, part)
// entry — for a far pair to the part's constant entry (slot = cap), where the contributions of all its far pairs add up.
template <int MODE>          // InvrScene::aggr: 0 = max occupancy (and 3 = mindist: the forward's wsel names the part either way), 1 = mean, 2 = dist
__global__ __launch_bounds__(256) void k_merge_bwd(Workspace w, const float4* __restrict__ g_rawfull, float4* __restrict__ g_raws) {
    constexpr bool MEAN = MODE == 1 || MODE == 2;
    const int na = w.counters[CNT_ACTIVE];
    for (int slot = blockIdx.x * blockDim.x + threadIdx.x; slot < na; slot += gridDim.x * blockDim.x) {
        const float4 g = g_rawfull[w.active_idx[slot]];
        if (MEAN) {
            // cfg.aggr == 'mean' (:236-239): raws.mean(dim=1) hands g / P to every part's entry; listed pairs keep theirs, the far
            // pairs of a part add up in its constant entry, the zeros of unflagged parts have no producer.  'dist' (:240-244): the
            // part's weight normalize(1 / (part_dist + 1e-5)) instead of 1 / P (the weights depend on the geometry alone: no gradient)
            const unsigned fl = w.pflags[slot], ff = w.farflags[slot];
            float wgt[INVR_NUM_PARTS];
            if (MODE == 2) {
                float n2 = 0.0f;
#pragma unroll
                for (int p = 0; p < INVR_NUM_PARTS; ++p) { wgt[p] = 1.0f / (w.pdist[(int64_t)slot * INVR_NUM_PARTS + p] + 1e-5f); n2 += wgt[p] * wgt[p]; }
                const float den = fmaxf(sqrtf(n2), 1e-12f);
#pragma unroll
                for (int p = 0; p < INVR_NUM_PARTS; ++p) wgt[p] = wgt[p] / den;
            } else {
#pragma unroll
                for (int p = 0; p < INVR_NUM_PARTS; ++p) wgt[p] = 1.0f / (float)INVR_NUM_PARTS;
            }
#pragma unroll
            for (int p = 0; p < INVR_NUM_PARTS; ++p) {
                const float s = wgt[p];
                const float4 gp = make_float4(g.x * s, g.y * s, g.z * s, g.w * s);
                g_raws[(int64_t)slot * INVR_NUM_PARTS + p] = (fl & (1u << p)) ? gp : make_float4(0.f, 0.f, 0.f, 0.f);
                if (!(fl & (1u << p)) && (ff & (1u << p))) {
                    float* c = reinterpret_cast<float*>(g_raws + w.cap * INVR_NUM_PARTS + p);
                    unsafeAtomicAdd(c, gp.x); unsafeAtomicAdd(c + 1, gp.y); unsafeAtomicAdd(c + 2, gp.z); unsafeAtomicAdd(c + 3, gp.w);
                }
            }
            continue;
        }
        // the forward's merge (k_winner_lists): p = the listed pair of part p, 8 + p = the far constant of part p, 255 = zeros
        const unsigned sel = w.wsel[slot];
        const int best_p = sel < 16u ? (int)(sel & 7u) : -1;
        const bool best_far = sel >= 8u && sel < 16u;
#pragma unroll
        for (int p = 0; p < INVR_NUM_PARTS; ++p)
            g_raws[(int64_t)slot * INVR_NUM_PARTS + p] = (p == best_p && !best_far) ? g : make_float4(0.f, 0.f, 0.f, 0.f);
        if (best_far) {
            float* c = reinterpret_cast<float*>(g_raws + w.cap * INVR_NUM_PARTS + best_p);
            unsafeAtomicAdd(c, g.x); unsafeAtomicAdd(c + 1, g.y); unsafeAtomicAdd(c + 2, g.z); unsafeAtomicAdd(c + 3, g.w);
        }
    }
}

// ---- weight gradients dW = gz^T a over the rows of a device-counted list, on the matrix cores -------------------------
// gz (rows, ldg) holds the gradient w.r.t. a layer's outputs (O used columns), a (rows, lda) the layer's inputs (I used
// columns).  A wave accumulates a 64 x 80 tile of dW (4 x 5 MFMA tiles; column I is a virtual all-ones input = the bias
// gradient) over its slabs of rows.  Input column j of `a` may be a k-slot of the part MLPs' rgb layer 1 (slot_order: rgb1_col
// maps it to the weight column, < 0 = padding).
// Round 6: a PERSISTENT grid of WG_GRID workgroups of WG_WAVES waves per job.  A wave keeps its tile in registers over all its
// slabs, the workgroup's waves add their tiles into one LDS image, and that image goes to the gradient tensors with ONE atomic per
// element and workgroup.  (One wave per 256-row slab used to add its own tile: 200 slabs x 5 jobs x 5120 elements for the body's
// 51 k pairs of a training patch — a million same-address float atomics, 200 deep per address: 140 - 350 us per part on the
// backward's critical chain for 20 us of matrix work.)
typedef float wg4 __attribute__((ext_vector_type(4)));
#define WG_SLAB 256
#define WG_WAVES 4
#define WG_GRID 48
#define WG_TILE_O 64
#define WG_TILE_I 80
struct WgradJob {
    const float* gz; const float* a;
    float* dW; float* db;
    int32_t slot_order;     // 1: input column j of `a` is k-slot (j >> 2, j & 3) of the part MLPs' rgb layer 1 (rgb1_col)
    int32_t ldg, lda, O, I, ldw;
};
struct WgradJobs { WgradJob j[5]; int n; };

__global__ __launch_bounds__(64 * WG_WAVES) void k_wgrad(WgradJobs jobs, const int32_t* __restrict__ count, int64_t n_host) {
    __shared__ float red[WG_TILE_O * WG_TILE_I];               // the workgroup's dW tile (20 KB)
    const WgradJob J = jobs.j[blockIdx.y];
    const int64_t n = count ? (int64_t)*count : n_host;
    if ((int64_t)blockIdx.x * WG_WAVES * WG_SLAB >= n) return;  // (block-uniform: no slab for any of this workgroup's waves)
    for (int e = threadIdx.x; e < WG_TILE_O * WG_TILE_I; e += 64 * WG_WAVES) red[e] = 0.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
    const int MT = (J.O + 15) >> 4, NT = (J.I + 1 + 15) >> 4;       // +1: the ones column
    wg4 acc[4][5];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 5; ++nt) acc[mt][nt] = (wg4){0.f, 0.f, 0.f, 0.f};
    for (int64_t slab = (int64_t)blockIdx.x * WG_WAVES + wv; slab * WG_SLAB < n; slab += (int64_t)gridDim.x * WG_WAVES) {
        const int64_t r0 = slab * WG_SLAB, r1 = min(r0 + WG_SLAB, n);
        // operands of one k-step (4 rows): loaded one step ahead of the MFMAs that consume them
        auto load = [&](int64_t r, float* av, float* bv) {
            const int64_t row = r + g;
            const bool live = row < r1;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int o = 16 * mt + i;
                av[mt] = (live && mt < MT && o < J.O) ? J.gz[row * J.ldg + o] : 0.0f;
            }
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) {
                const int c = 16 * nt + i;
                bv[nt] = (live && nt < NT) ? (c < J.I ? J.a[row * J.lda + c] : (c == J.I ? 1.0f : 0.0f)) : 0.0f;
            }
        };
        float av[4], bv[5], an[4], bn[5];
        load(r0, av, bv);
        for (int64_t r = r0; r < r1; r += 4) {
            load(r + 4, an, bn);                               // rows >= r1 load as zeros
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                if (mt >= MT) continue;
#pragma unroll
                for (int nt = 0; nt < 5; ++nt) {
                    if (nt >= NT) continue;
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) av[mt] = an[mt];
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) bv[nt] = bn[nt];
        }
    }
    // D[row = 4g + r][col = i] of tile (mt, nt) = dW[16 mt + 4g + r][16 nt + i]: the waves' tiles summed in LDS
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        if (mt >= MT) continue;
#pragma unroll
        for (int nt = 0; nt < 5; ++nt) {
            if (nt >= NT) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[mt][nt][r];
                if (v != 0.0f) atomicAdd(&red[(16 * mt + 4 * g + r) * WG_TILE_I + 16 * nt + i], v);
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < WG_TILE_O * WG_TILE_I; e += 64 * WG_WAVES) {
        const int o = e / WG_TILE_I, c = e - o * WG_TILE_I;
        const float v = red[e];
        if (o >= J.O || v == 0.0f) continue;
        if (c < J.I) {
            const int wc = J.slot_order ? rgb1_col(c >> 2, c & 3) : c;
            if (wc >= 0) unsafeAtomicAdd(J.dW + (int64_t)o * J.ldw + wc, v);
        } else if (c == J.I) {
            unsafeAtomicAdd(J.db + o, v);
        }
    }
}

int launch_wgrad(const WgradJobs& jobs, const int32_t* count, int64_t n_max, hipStream_t st);

// the five linears of one part from the (gz, a) stacks k_part_mlp_bwd wrote: [0 occ1 64x19, 1 occ2 17x64, 2 rgb1 64x70 (input in
// k-slot order), 3 rgb2 64x64 (3-linear colour nets), 4 rgb head 3x64]; dW / db indexed the same way (entry 3 unused
// for 2-linear colour nets)
int launch_part_wgrad(const float* gz, const float* a, int64_t lcap, int n_rgb, float* const* dW, float* const* db,
                      const int32_t* count, hipStream_t st) {
    WgradJobs jobs;
    memset(&jobs, 0, sizeof(jobs));
    const int O[5] = {64, 17, 64, 64, 3}, I[5] = {19, 64, 72, 64, 64}, LDW[5] = {19, 64, 70, 64, 64};
    int n = 0;
    for (int l = 0; l < 5; ++l) {
        if (l == 3 && n_rgb != 3) continue;
        jobs.j[n++] = WgradJob{gz + (int64_t)l * lcap * 64, a + (int64_t)l * lcap * 72, dW[l], db[l], l == 2 ? 1 : 0, 64, 72, O[l], I[l], LDW[l]};
    }
    jobs.n = n;
    return launch_wgrad(jobs, count, lcap, st);
}

int launch_wgrad(const WgradJobs& jobs, const int32_t* count, int64_t n_max, hipStream_t st) {
    if (n_max <= 0 || jobs.n == 0) return 0;
    const int64_t groups = cdiv(n_max, (int64_t)WG_SLAB * WG_WAVES);
    hipLaunchKernelGGL(k_wgrad, dim3((unsigned)(groups < WG_GRID ? groups : WG_GRID), jobs.n), dim3(64 * WG_WAVES), 0, st, jobs, count, n_max);
    INVR_LAUNCH_CHECK();
    return 0;
}

// ---- deformer backward -----------------------------------------------------------------------------------------------
// The list the deformer is differentiated over: [pairs of part 0 | ... | pairs of part 4 | pair-regulariser neighbours].
// k_deform_list_pairs fills point + upstream gradient of the pair entries:
//   g_resd = g_tpose (encoder^T; tpose = init_bigpose + resd, :111) + g_off * resd / ||resd|| * multiplicity
// (multiplicity: the far constant's residual appears in the dense resd rows of all far pairs of its part).
__global__ __launch_bounds__(256) void k_deform_list_pairs(Workspace w, TrainWs t, const float* __restrict__ g_off_sum) {
    const int p = blockIdx.y;
    const int cnt = w.counters[CNT_PAIRS + p];
    int base = 0;
    for (int q = 0; q < p; ++q) base += w.counters[CNT_PAIRS + q];
    const float goff = g_off_sum ? g_off_sum[0] : 0.0f;
    const float far_mult = (float)w.counters[CNT_FAR + p];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += gridDim.x * blockDim.x) {
        float r[3], x[3], gx[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            r[c] = w.l_r[p][c * w.lcap + i];
            x[c] = w.l_x[p][c * w.lcap + i] - r[c];
            gx[c] = t.g_x[p][c * w.lcap + i];
        }
        const float nrm = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        const float mult = (i == cnt - 1) ? far_mult : 1.0f;                    // last entry = the far constant
        const float k = nrm > 0.0f ? goff * mult / nrm : 0.0f;                  // torch.norm backward: 0 at the origin
        const int64_t e = (int64_t)base + i;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            t.d_pts[e * 3 + c] = x[c];
            t.d_g[e * 3 + c] = gx[c] + k * r[c];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && p == INVR_NUM_PARTS - 1)
        w.counters[CNT_DTOT] = base + cnt + min(w.counters[CNT_NB], (int)t.NB);
}

// gradient of the normalised direction v = x / (|x| + eps) (crit.py:10-11): g_x = g_v / (L + eps) - x (g_v . x) / (L (L + eps)^2)
__device__ __forceinline__ void normdir_bwd(const float* x, const float* gv, float* gx) {
    const float L = sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]), Le = L + 1e-8f;
    const float dot = gv[0] * x[0] + gv[1] * x[1] + gv[2] * x[2];
    const float k = L > 0.0f ? dot / (L * Le * Le) : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) gx[c] = gv[c] / Le - x[c] * k;
}

// pair term backward: neighbour entries of the list (point + gradient) and the self pairs' share (atomics: the far constant
// is the self pair of many rows)
__global__ __launch_bounds__(256) void k_pair_term_bwd(Workspace w, TrainWs t, const float* __restrict__ g_pair_sum) {
    const int nsel = min(w.counters[CNT_NB], (int)t.NB);
    int base[INVR_NUM_PARTS + 1];
    base[0] = 0;
    for (int q = 0; q < INVR_NUM_PARTS; ++q) base[q + 1] = base[q] + w.counters[CNT_PAIRS + q];
    const float gp = g_pair_sum ? g_pair_sum[0] : 0.0f;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nsel; k += gridDim.x * blockDim.x) {
        const int ref = t.nb_ref[k], p = ref >> 28, i = ref & 0x0FFFFFFF;
        float vn[3], vs[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { vn[c] = t.nb_r[(int64_t)k * 3 + c]; vs[c] = w.l_r[p][c * w.lcap + i]; }
        const float ln = sqrtf(vn[0] * vn[0] + vn[1] * vn[1] + vn[2] * vn[2]) + 1e-8f;
        const float ls = sqrtf(vs[0] * vs[0] + vs[1] * vs[1] + vs[2] * vs[2]) + 1e-8f;
        float d[3] = {vn[0] / ln - vs[0] / ls, vn[1] / ln - vs[1] / ls, vn[2] / ln - vs[2] / ls};
        const float tn = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        float gvn[3], gvs[3], gn[3], gs[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { gvn[c] = tn > 0.0f ? gp * d[c] / tn : 0.0f; gvs[c] = -gvn[c]; }
        normdir_bwd(vn, gvn, gn);
        normdir_bwd(vs, gvs, gs);
        const int64_t e = (int64_t)base[INVR_NUM_PARTS] + k;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            t.d_pts[e * 3 + c] = t.nb_x[(int64_t)k * 3 + c];
            t.d_g[e * 3 + c] = gn[c];
            unsafeAtomicAdd(&t.d_g[((int64_t)base[p] + i) * 3 + c], gs[c]);
        }
    }
}

// thread-per-entry backward of 0.05 tanh(MLP(grid(uv(x), t))): recomputes the forward, writes the per-layer (gz, a)
// matrices for the weight-gradient GEMMs and (uvt, g_feat) for the grid backward.  Canonical points carry no gradient
// (the warp is gradient-free in the reference, inb_part_network_multiassign.py:87-90).
// Round 6: the MLP's 1795 weights staged in LDS once per workgroup (they were 2200 wave-uniform VECTOR loads per thread: 1 KB of L1
// traffic per 16 bytes of weights), 256 registers per thread instead of 128 + 151 spilled dwords, the row-major (gz, a) rows stored
// as float4 (a lane's row is contiguous: 8 stores instead of 32 per 32-wide row).  Same operations in the same order: same bits.
#define DB_BLOCK 256
#define DB_O_W0 0                       // 32 x 19
#define DB_O_B0 (DB_O_W0 + 32 * 19)
#define DB_O_W1 (DB_O_B0 + 32)          // 32 x 32
#define DB_O_B1 (DB_O_W1 + 32 * 32)
#define DB_O_W2 (DB_O_B1 + 32)          // 3 x 32
#define DB_O_B2 (DB_O_W2 + 3 * 32)
#define DB_O_W0P (DB_O_B2 + 4)          // 32 x 20: W0 with rows padded to 20 (16-byte aligned rows for the W0^T product)
#define DB_LDS (DB_O_W0P + 32 * 20)
__device__ __forceinline__ void store_row4(float* dst, const float* v, int n4) {
#pragma unroll
    for (int k = 0; k < n4; ++k) reinterpret_cast<float4*>(dst)[k] = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
}
__global__ __launch_bounds__(DB_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_deform_bwd(SceneDev s, GridDev dg, MlpDev dm, Workspace w, TrainWs t) {
    __shared__ __attribute__((aligned(16))) float lw[DB_LDS];
    const int n = w.counters[CNT_DTOT];
    if ((int64_t)blockIdx.x * DB_BLOCK >= n) return;
    for (int k = threadIdx.x; k < 32 * 19; k += DB_BLOCK) lw[DB_O_W0 + k] = dm.w[0][k];
    for (int k = threadIdx.x; k < 32 * 20; k += DB_BLOCK) lw[DB_O_W0P + k] = (k % 20) < 19 ? dm.w[0][(k / 20) * 19 + k % 20] : 0.0f;
    for (int k = threadIdx.x; k < 32 * 32; k += DB_BLOCK) lw[DB_O_W1 + k] = dm.w[1][k];
    if (threadIdx.x < 96) lw[DB_O_W2 + threadIdx.x] = dm.w[2][threadIdx.x];
    if (threadIdx.x < 32) { lw[DB_O_B0 + threadIdx.x] = dm.b[0][threadIdx.x]; lw[DB_O_B1 + threadIdx.x] = dm.b[1][threadIdx.x]; }
    if (threadIdx.x < 3) lw[DB_O_B2 + threadIdx.x] = dm.b[2][threadIdx.x];
    __syncthreads();
    const float* W0 = lw + DB_O_W0; const float* W1 = lw + DB_O_W1; const float* W2 = lw + DB_O_W2;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        float xb[3] = {t.d_pts[(int64_t)e * 3], t.d_pts[(int64_t)e * 3 + 1], t.d_pts[(int64_t)e * 3 + 2]}, uvt[3];
        DeformActT<true> a;
        deform_fwd_act_w<true>(s, dg, W0, lw + DB_O_B0, W1, lw + DB_O_B1, W2, lw + DB_O_B2, xb, uvt, a);
        float gz3[4];
#pragma unroll
        for (int c = 0; c < 3; ++c) gz3[c] = t.d_g[(int64_t)e * 3 + c] * 0.05f * (1.0f - a.th[c] * a.th[c]);
        gz3[3] = 0.0f;
        store_row4(t.d_gz3 + (int64_t)e * 4, gz3, 1);
        float gz2[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const float gh = W2[j] * gz3[0] + W2[32 + j] * gz3[1] + W2[64 + j] * gz3[2];
            gz2[j] = gh * a.s2[j];                                               // softplus'(z) = sigmoid(z)
        }
        store_row4(t.d_gz2 + (int64_t)e * 32, gz2, 8);
        store_row4(t.d_a2 + (int64_t)e * 32, a.h2, 8);
        store_row4(t.d_a1 + (int64_t)e * 32, a.h1, 8);
        // W^T products four outputs at a time: one 16-byte LDS read W[j][4 ib .. 4 ib + 3] feeds four accumulators (every accumulator
        // still sums over j ascending: the element-by-element form's bits)
        float gz1[32];
#pragma unroll
        for (int ib = 0; ib < 8; ++ib) {
            float gh[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const float4 wv = *reinterpret_cast<const float4*>(W1 + j * 32 + 4 * ib);
                gh[0] = fmaf(wv.x, gz2[j], gh[0]); gh[1] = fmaf(wv.y, gz2[j], gh[1]);
                gh[2] = fmaf(wv.z, gz2[j], gh[2]); gh[3] = fmaf(wv.w, gz2[j], gh[3]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) gz1[4 * ib + q] = gh[q] * a.s1[4 * ib + q];
            __builtin_amdgcn_sched_barrier(0);
        }
        store_row4(t.d_gz1 + (int64_t)e * 32, gz1, 8);
        float a0[20];
#pragma unroll
        for (int i = 0; i < 19; ++i) a0[i] = a.feat[i];
        a0[19] = 0.0f;
        store_row4(t.d_a0 + (int64_t)e * 20, a0, 5);
#pragma unroll
        for (int ib = 0; ib < 5; ++ib) {
            float gf[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const float4 wv = *reinterpret_cast<const float4*>(lw + DB_O_W0P + j * 20 + 4 * ib);      // (padded rows: column 19 = 0)
                gf[0] = fmaf(wv.x, gz1[j], gf[0]); gf[1] = fmaf(wv.y, gz1[j], gf[1]);
                gf[2] = fmaf(wv.z, gz1[j], gf[2]); gf[3] = fmaf(wv.w, gz1[j], gf[3]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (4 * ib + q < 19) t.d_gfeat[(int64_t)e * 19 + 4 * ib + q] = gf[q];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) t.d_uvt[(int64_t)e * 3 + c] = uvt[c];
    }
}

int launch_deform_bwd(const RenderArgs& a, const Workspace& w, const TrainWs& t, const GridDev& dg, const MlpDev& dm,
                      const float* g_off_sum, const float* g_pair_sum, const DeformGrads& G, hipStream_t st, hipStream_t side,
                      hipEvent_t ev_fork, hipEvent_t ev_join) {
    int64_t tiles = cdiv(w.lcap, 256);
    unsigned gx = (unsigned)(tiles < 512 ? (tiles > 0 ? tiles : 1) : 512);
    hipLaunchKernelGGL(k_deform_list_pairs, dim3(gx, INVR_NUM_PARTS), dim3(256), 0, st, w, t, g_off_sum);
    INVR_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pair_term_bwd, dim3(256), dim3(256), 0, st, w, t, g_pair_sum);
    INVR_LAUNCH_CHECK();
    int64_t et = cdiv(t.DM, DB_BLOCK);
    hipLaunchKernelGGL(k_deform_bwd, dim3((unsigned)(et < 2048 ? (et > 0 ? et : 1) : 2048)), dim3(DB_BLOCK), 0, st, a.scene, dg, dm, w, t);
    INVR_LAUNCH_CHECK();
    WgradJobs jobs;
    memset(&jobs, 0, sizeof(jobs));
    jobs.n = 3;
    jobs.j[0] = WgradJob{t.d_gz1, t.d_a0, G.w[0], G.b[0], 0, 32, 20, 32, 19, 19};
    jobs.j[1] = WgradJob{t.d_gz2, t.d_a1, G.w[1], G.b[1], 0, 32, 32, 32, 32, 32};
    jobs.j[2] = WgradJob{t.d_gz3, t.d_a2, G.w[2], G.b[2], 0, 4, 32, 3, 32, 32};
    // the weight gradients and the grid^T below both read what k_deform_bwd wrote and nothing of each other: with a side stream
    // (round 6) they run side by side — two latency-bound launches of ~100 us each
    const bool fork = side && ev_fork && ev_join;
    if (fork) {
        INVR_HIP(hipEventRecord(ev_fork, st));
        INVR_HIP(hipStreamWaitEvent(side, ev_fork, 0));
    }
    if (launch_wgrad(jobs, w.counters + CNT_DTOT, t.DM, fork ? side : st)) return 1;
    if (fork) INVR_HIP(hipEventRecord(ev_join, side));
    // grid^T: table gradients of the deformer's 8 x 2 grid (the (u,v,t) input carries no gradient)
    int rc = launch_deform_slice_bwd(dg, a.scene.frame_dim, t.d_uvt, t.d_gfeat, t.DM, w.counters + CNT_DTOT, G.dense, G.hash, st);
    if (rc < 0) rc = launch_grid_encode_bwd_generic(dg, t.d_uvt, t.d_gfeat, t.DM, G.dense, G.hash, nullptr, st, w.counters + CNT_DTOT);
    if (fork) INVR_HIP(hipStreamWaitEvent(st, ev_join, 0));
    return rc;
}

// ---- the training objective of NetworkWrapper.forward (inb_trainer.py:40-98, 176-214 with the plain MSE image term) in ONE launch --------
// loss = w_pair pair + w_dist mean(dist) + w_off offset + mean((rgb - gt)^2), in the wrapper's order of additions; also the per-ray
// |rgb - gt| sum (ret['error']) and the psnr statistic.  As torch ops this was ~35 kernels of 2 us with ~10 us of host time between
// them, and as many again in their autograd backward: 1 ms of a 4 ms iteration with an idle GPU (profiles/r4_training_step.md).
// out[8] = {loss, img_loss, psnr, reg_dist, offset_loss, pair_loss, 0, 0}
__global__ __launch_bounds__(1024) void k_train_loss(const float* __restrict__ rgb, const float* __restrict__ gt, const float* __restrict__ dist,
                                                     const float* __restrict__ terms, int64_t n, float w_pair, float w_dist, float w_off,
                                                     int use_pair, float* __restrict__ out, float* __restrict__ err) {
    __shared__ double red[2][1024 / 64];
    double s2 = 0.0, sd = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        float e = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) { const float d = rgb[i * 3 + c] - gt[i * 3 + c]; s2 += (double)(d * d); e += fabsf(d); }
        if (err) err[i] = e;
        if (dist) sd += (double)dist[i];
    }
    for (int d = 32; d >= 1; d >>= 1) { s2 += __shfl_xor(s2, d); sd += __shfl_xor(sd, d); }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red[0][wv] = s2; red[1][wv] = sd; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s2 = sd = 0.0;
        for (int k = 0; k < 1024 / 64; ++k) { s2 += red[0][k]; sd += red[1][k]; }
        const float img = n > 0 ? (float)(s2 / (double)(3 * n)) : 0.0f;
        const float rd = (dist && n > 0) ? (float)(sd / (double)n) : 0.0f;
        const float off = terms[TERM_OFFSET_SUM] / fmaxf(terms[TERM_OFFSET_ROWS], 1.0f);
        const float pair = use_pair ? terms[TERM_PAIR_SUM] / fmaxf(terms[TERM_PAIR_ROWS], 1.0f) : 0.0f;
        float loss = 0.0f;
        if (use_pair) loss = loss + w_pair * pair;
        if (dist) loss = loss + w_dist * rd;
        loss = loss + w_off * off;
        loss = loss + img;
        out[0] = loss; out[1] = img; out[2] = -10.0f * logf(img) / 2.302585092994046f; out[3] = rd; out[4] = off; out[5] = pair;
        out[6] = out[7] = 0.0f;
    }
}

__global__ __launch_bounds__(256) void k_train_loss_bwd(const float* __restrict__ rgb, const float* __restrict__ gt, const float* __restrict__ terms,
                                                        int64_t n, float w_pair, float w_dist, float w_off, int use_pair,
                                                        const float* __restrict__ g_loss, float* __restrict__ g_rgb, float* __restrict__ g_dist,
                                                        float* __restrict__ g_terms) {
    const float gl = g_loss[0];
    const float k_img = n > 0 ? gl * 2.0f / (float)(3 * n) : 0.0f, k_dist = n > 0 ? gl * w_dist / (float)n : 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
#pragma unroll
        for (int c = 0; c < 3; ++c) g_rgb[i * 3 + c] = k_img * (rgb[i * 3 + c] - gt[i * 3 + c]);
        if (g_dist) g_dist[i] = k_dist;
    }
    if (blockIdx.x == 0 && threadIdx.x < TERM_LEN) {
        float g = 0.0f;
        if (threadIdx.x == TERM_OFFSET_SUM) g = gl * w_off / fmaxf(terms[TERM_OFFSET_ROWS], 1.0f);
        if (threadIdx.x == TERM_PAIR_SUM && use_pair) g = gl * w_pair / fmaxf(terms[TERM_PAIR_ROWS], 1.0f);
        g_terms[threadIdx.x] = g;
    }
}

int launch_train_loss(const float* rgb, const float* gt, const float* dist, const float* terms, int64_t n, float w_pair, float w_dist,
                      float w_off, int use_pair, float* out, float* err, hipStream_t st) {
    hipLaunchKernelGGL(k_train_loss, dim3(1), dim3(1024), 0, st, rgb, gt, dist, terms, n, w_pair, w_dist, w_off, use_pair, out, err);
    INVR_LAUNCH_CHECK();
    return 0;
}

int launch_train_loss_bwd(const float* rgb, const float* gt, const float* terms, int64_t n, float w_pair, float w_dist, float w_off,
                          int use_pair, const float* g_loss, float* g_rgb, float* g_dist, float* g_terms, hipStream_t st) {
    const int64_t tiles = cdiv(n > 0 ? n : 1, 256);
    hipLaunchKernelGGL(k_train_loss_bwd, dim3((unsigned)(tiles < 256 ? tiles : 256)), dim3(256), 0, st, rgb, gt, terms, n, w_pair, w_dist, w_off,
                       use_pair, g_loss, g_rgb, g_dist, g_terms);
    INVR_LAUNCH_CHECK();
    return 0;
}

int launch_distortion_bwd(const float* weights, const float* z, const float* g_dist, int64_t R, int S, float* g_w, hipStream_t st) {
    if (R == 0) return 0;
    hipLaunchKernelGGL(k_distortion_bwd, dim3((unsigned)cdiv(R, 4)), dim3(256), 0, st, weights, z, g_dist, R, S, g_w);
    INVR_LAUNCH_CHECK();
    return 0;
}

int launch_merge_bwd(const Workspace& w, int aggr, const float4* g_rawfull, float4* g_raws, hipStream_t st) {
    INVR_HIP(hipMemsetAsync(g_raws + w.cap * INVR_NUM_PARTS, 0, INVR_NUM_PARTS * sizeof(float4), st));      // far-constant row
    int64_t tiles = cdiv(w.cap, 256);
    const dim3 grid((unsigned)(tiles < 1024 ? (tiles > 0 ? tiles : 1) : 1024));
    if (aggr == INVR_AGGR_MEAN) hipLaunchKernelGGL(k_merge_bwd<1>, grid, dim3(256), 0, st, w, g_rawfull, g_raws);
    else if (aggr == INVR_AGGR_DIST) hipLaunchKernelGGL(k_merge_bwd<2>, grid, dim3(256), 0, st, w, g_rawfull, g_raws);
    else hipLaunchKernelGGL(k_merge_bwd<0>, grid, dim3(256), 0, st, w, g_rawfull, g_raws);
    INVR_LAUNCH_CHECK();
    return 0;
}
