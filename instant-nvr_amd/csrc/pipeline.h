// Internal data layout of one invr_render_fwd call: kernel argument blocks and the workspace carve.
#pragma once
#include "common.h"

#define PAIR_GROUP 4096          // survivor slots per workgroup of k_pair_lists
enum { CNT_ACTIVE = 0, CNT_PAIRS = 1 /* ..5 */, CNT_OVERFLOW = 6, CNT_FAR = 7 /* ..11 */, CNT_TICKET = 12,
       CNT_NB = 13 /* selected pair-regulariser rows */, CNT_DTOT = 14 /* entries of the deformer backward list */,
       CNT_LIVE = 15 /* lattice cells that can hold a survivor */, CNT_LEN = 16 /* exported as stats[] */,
              CNT_TICKETS = 32 /* 16 ticket counters of k_knn_pairs, one per 128-byte line */, CNT_ALLOC = 32 + 16 * 32 };

#define CULL_MASK_MAX (4 << 20)   // cells of the per-frame cull mask (bytes); larger distance volumes run unmasked
#define VOXMASK_MAX_CELLS (1 << 20)  // lattice cells with per-part candidate-cluster masks (40 B each)
#define DF_SLICE_MAX 4096  // float2 entries of the deformer's per-frame (u,v) slice tables (32 KB of LDS)
#define EMB_K 20            // 19-d encoder output padded to 5 MFMA k-steps

struct RenderArgs {
    SceneDev scene;
    const float* ray_o;
    const float* ray_d;
    const float* near;
    const float* far;
    const float* jitter;     // (R,S) or null
    const float* wpts;       // points mode (Network.forward): (N,3) world points, S == 1; else null
    const float* wdirs;      // points mode: (N,3) world view directions
    float* z_vals;           // (R,S) or null
    int64_t R;               // rays
    int32_t S;               // samples per ray
    int64_t N;               // R*S
};

// per-frame KNN acceleration index built by k_part_prepare
struct KnnIndex {
    float4* sverts;      // P*mpad : Morton-sorted vertices, two per float4 pair: {x0,x1,y0,y1} {z0,z1,|v0|^2,|v1|^2}
    uint16_t* srow;      // P*mpad : row of every sorted vertex slot inside its part (0 for the padding sentinels)
    float4* sub;         // P*cpad*8 : AABB {min, max} of the four 16-vertex sub-clusters of every cluster
    float4* cl;          // P*cpad*3 : per 64-vertex cluster {AABB min, AABB max, first vertex (an upper
                         //            bound of the nearest distance)}
    float* part_aabb;    // P*6
    float* dfar2;        // 1 : squared distance beyond which a (point, part) pair is folded into the part's far constant, derived per
                         //     frame from the magnitude of A / big_A (k_part_prepare)
    unsigned long long* voxmask;  // per lattice cell and part: bit c set if cluster c can hold one of the 4 nearest vertices of a
                         //            point of the cell (undecided cells of parts with <= 64 clusters; all ones otherwise); NULL = off
    float* voxu2;        // per (lattice cell, part): squared upper bound of the 4th-nearest distance of any point of the cell
    uint8_t* voxcls;     // per (lattice cell, part): 0 never classified (mask / bound stale), 1 far, 2 unflagged, 3 undecided; NULL = off
    int32_t* live_cells; // lattice cells with a corner below the cull threshold (k_cull_cells; count in counters[CNT_LIVE])
    float4* vmat;        // P*mpad*6 : per vertex, rows 0..2 of sum_j pbw[v][j] A_j and of sum_j pbw[v][j] big_A_j
    int32_t mpad, cpad;
};

// All arrays live in the caller-provided workspace (HBM).  cap = max_active; lists and the
// per-slot arrays hold one extra entry (slot `cap`) for the per-part far-pair constant.
struct Workspace {
    unsigned long long* mask;     // ceil(N/64): survivor bit per ray-sample
    int32_t* block_cnt;           // ceil(N/1024): survivors per cull tile (k_cull_flag)
    int32_t* block_off;           // ceil(N/1024): rank of a tile's (ray-major) / of an 8192-sample block's (windowed order) first survivor inside its super-block
    int32_t* super_tot;           // survivors per super-block of 1024 tiles / blocks (k_scan_blocks[_win] -> k_compact[_win])
    int32_t* counters;            // CNT_ALLOC, followed by gcount (one memset clears both)
    int32_t* gcount;              // [ceil(lcap / PAIR_GROUP)][INVR_NUM_PARTS]: flagged pairs per group of PAIR_GROUP survivor slots
    int64_t n_groups;
    int32_t* active_idx;          // cap: ray-sample index of every survivor (ordered)
    int32_t* word_off;            // ceil(N/64): rank of the first survivor of every 64-sample mask word (slot of sample i =
                                  // word_off[i>>6] + popcount(mask[i>>6] below bit i&63); 2 MB instead of a 131 MB slot-per-sample array)
    // Windowed survivor order (eval frames, launch_cull): ord_rows > 0 = the survivors of a block of ord_rows rays are ranked by
    // (8-sample depth window, ray, sample) instead of (ray, sample) — 64 consecutive survivors are then one depth slab of a few
    // neighbouring rays instead of the front AND back segments of those rays, which is what the KNN's per-wave pruning pays for
    // (tools/knn_order_model.py).  slot of sample i = byte_off[i>>3] + popcount(bits of its mask byte below bit i&7).
    int32_t* byte_off;            // ceil(N/8): rank of the first survivor of every 8-sample mask byte (windowed order only)
    int32_t ord_rows, ord_cols;   // rays per block, 8-sample windows per ray (ord_rows * ord_cols = 1024: blocks of 8192 ray-samples); 0 = ray-major
    uint8_t* pflags;              // cap: bit p set if (slot, part p) is flagged and listed
    uint8_t* farflags;            // cap: bit p set if (slot, part p) is a far pair (takes the part constant)
    KnnIndex knn;
    // per-part pair lists, SoA, each of capacity cap
    int32_t* l_slot[INVR_NUM_PARTS];      // cap
    int32_t* l_nn[INVR_NUM_PARTS];        // cap*4 : neighbour rows inside part_pbw[p]
    float* l_w[INVR_NUM_PARTS];           // cap*4 : normalised gaussian weights
    float* l_x[INVR_NUM_PARTS];           // 3*cap : canonical (big-pose + residual) xyz, SoA
    float* l_d[INVR_NUM_PARTS];           // 3*cap : canonical view dir, SoA
    float* l_r[INVR_NUM_PARTS];           // 3*cap : residual deformation (resd), SoA
    float* emb[INVR_NUM_PARTS];           // EMB_K*cap each : encoder output of part p, SoA [k][pair]
    // part MLPs, phase 1 (every listed pair): occupancy + the 16 geometry features; phase 2 (the rgb MLP) only runs for the
    // pair that wins the max-occupancy merge of its survivor (inb_part_network_multiassign.py:253-256 keeps nothing else)
    float* occp[INVR_NUM_PARTS];          // cap : occupancy of every listed pair (list order; last entry = the far constant)
    float4* feat[INVR_NUM_PARTS];         // 4*cap : occ-MLP outputs 1..16 of every listed pair, [pair][16]
    int32_t* wl[INVR_NUM_PARTS];          // cap : winner lists, segmented by slot group: the winners of group g sit at
                                          //       wl[p][pair offset of g ...), wcnt[g][p] of them (pair indices, ascending)
    int32_t* wcnt;                        // [n_groups][INVR_NUM_PARTS]
    uint8_t* wsel;                        // cap : merge result per survivor: p = listed pair of part p wins, 8 + p = far constant of
                                          //       part p wins, 255 = zeros (no flagged part, or an unflagged part 0 ties at 0)
    float4* rgbw;                         // cap + 8 : [rgb, occ] of the winning listed pair per survivor; [lcap + p] = far constant of part p
    uint8_t* cullmask;                    // CULL_MASK_MAX : 1 if the trilinear cell can hold a survivor (k_cull.hip)
    float2* dslice;                       // DF_SLICE_MAX : per-frame t-slices of the deformer grid (k_warp.hip)
    uint8_t* cullmask_d1;                 // CULL_MASK_MAX : cullmask OR-dilated by one cell in every direction (k_dilate_mask; valid when use_d1)
    int32_t use_d1;                       // this frame's k_front_cull may drop a thread's 4-sample segment on one look-up of cullmask_d1
    float* pdist;                         // 5*lcap : cfg.aggr 'dist' / 'mindist' only — the KNN's weighted distance of EVERY (slot, part), [slot][p]
    int64_t cap;                          // max survivors
    int64_t lcap;                         // cap + 1: list / per-slot array capacity (stride of the SoA lists)
};

// ---- z / pose-space point of a ray-sample (inb_renderer.py:17-29, blend_utils.py:366-382) ----
__device__ __forceinline__ float sample_z(float near, float far, int s, int S, const float* jit_row) {
    float t = linspace01(s, S);
    float z = near * (1.0f - t) + far * t;
    if (jit_row) {   // stratified jitter between the mid-points of neighbouring samples (:20-27)
        float tp = linspace01(max(s - 1, 0), S), tn = linspace01(min(s + 1, S - 1), S);
        float zp = near * (1.0f - tp) + far * tp, zn = near * (1.0f - tn) + far * tn;
        float upper = (s == S - 1) ? z : 0.5f * (zn + z);
        float lower = (s == 0) ? z : 0.5f * (z + zp);
        z = lower + (upper - lower) * jit_row[s];
    }
    return z;
}

__device__ __forceinline__ void sample_pose_point(const RenderArgs& a, int64_t i, float& px, float& py,
                                                  float& pz, float* zout, float* pdir) {
    float wx, wy, wz, dx, dy, dz;
    if (a.wpts) {                                                        // Network.forward(wpts, viewdir, ...)
        wx = a.wpts[i * 3]; wy = a.wpts[i * 3 + 1]; wz = a.wpts[i * 3 + 2];
        dx = a.wdirs[i * 3]; dy = a.wdirs[i * 3 + 1]; dz = a.wdirs[i * 3 + 2];
        if (zout) *zout = 0.0f;
    } else {
        int64_t ray = i / a.S;
        int s = (int)(i - ray * a.S);
        float near = a.near[ray], far = a.far[ray];
        float z = sample_z(near, far, s, a.S, a.jitter ? a.jitter + ray * a.S : nullptr);
        if (zout) *zout = z;
        float ox = a.ray_o[ray * 3 + 0], oy = a.ray_o[ray * 3 + 1], oz = a.ray_o[ray * 3 + 2];
        dx = a.ray_d[ray * 3 + 0]; dy = a.ray_d[ray * 3 + 1]; dz = a.ray_d[ray * 3 + 2];
        wx = ox + dx * z; wy = oy + dy * z; wz = oz + dz * z;            // pts = o + d*z
    }
    const float* R = a.scene.R;
    const float* Th = a.scene.Th;
    float qx = wx - Th[0], qy = wy - Th[1], qz = wz - Th[2];             // (p - Th) @ R
    px = qx * R[0] + qy * R[3] + qz * R[6];
    py = qx * R[1] + qy * R[4] + qz * R[7];
    pz = qx * R[2] + qy * R[5] + qz * R[8];
    if (pdir) {                                                          // viewdir @ R
        pdir[0] = dx * R[0] + dy * R[3] + dz * R[6];
        pdir[1] = dx * R[1] + dy * R[4] + dz * R[7];
        pdir[2] = dx * R[2] + dy * R[5] + dz * R[8];
    }
}

// ---- pipeline stage launchers ------------------------------------------------------------------
int launch_cull_cells(const RenderArgs& a, const Workspace& w, hipStream_t st);      // -> 1 if the cell mask / live list were built
int launch_cull(const RenderArgs& a, const Workspace& w, int64_t max_active, bool have_cells, bool flags_done, hipStream_t st);   // flags_done: k_front_cull wrote the masks, only scan + compaction remain
int launch_front_scene(const RenderArgs& a, const Workspace& w, const GridDev& dg, int* have_cells, hipStream_t st);     // k_knn.hip: index + cell mask + vertex matrices + deformer slices, one launch
int launch_front_cull(const RenderArgs& a, const Workspace& w, int* done, hipStream_t st);                               // k_knn.hip: lattice-cell classes + cull flags, one launch
int launch_pose_points(const RenderArgs& a, const int32_t* idx, int64_t n, float* pts, float* dirs, hipStream_t st);
int launch_knn_prepare(const RenderArgs& a, const Workspace& w, hipStream_t st);
int launch_vertex_mats(const RenderArgs& a, const Workspace& w, hipStream_t st);
int launch_knn_voxel_class(const RenderArgs& a, const Workspace& w, hipStream_t st);
int launch_knn_pairs(const RenderArgs& a, const Workspace& w, int32_t* stats, hipStream_t st);     // stats: exported behind the pair lists when given
int launch_knn_pdist(const RenderArgs& a, const Workspace& w, hipStream_t st);      // aggr 'dist' / 'mindist': part_dist of all (survivor, part)
int launch_deform_slice(const RenderArgs& a, const Workspace& w, const GridDev& dg, hipStream_t st);   // per-call t-slices of the deformer grid (no-op when they do not fit)
int launch_warp_pairs(const RenderArgs& a, const Workspace& w, const GridDev& dg, const MlpDev& dm, hipStream_t st);
int launch_part_encode(const GridDev& g, const float* x_soa, int64_t stride, const int32_t* count, int64_t cap,
                       float* emb, hipStream_t st);
struct EncodeAllArgs {            // k_part_encode_rs_all: the five part grids of one render
    GridDev g[INVR_NUM_PARTS];
    const float* xs[INVR_NUM_PARTS];
    float* emb[INVR_NUM_PARTS];
    const int32_t* counts;        // counters + CNT_PAIRS
    int64_t stride, cap;
};
int launch_part_encode_all(const EncodeAllArgs& a, hipStream_t st);
int launch_part_encode_rows_all(const EncodeAllArgs& a, hipStream_t st);   // 64-byte rows (training forward), blockIdx.y = part
struct PartMlpDev {
    MlpDev occ, rgb;
    const float* rgb_latent;
    const int64_t* latent_index;
    int32_t latent_dim, n_freq, geo_dim;
};
int launch_part_mlp(const PartMlpDev& pm, const float* emb, const float* d_soa, int64_t stride,
                    const int32_t* count, int64_t cap, float4* raw_direct, hipStream_t st);
struct MlpBwdOut {                 // k_mlp_bwd.hip; mirrors InvrMlpBwdOut
    float* g_emb; float* gz; float* a; int64_t n_pad; float* g_latent;
    int latent_full;       // 1: g_latent is the whole (num_latent_code, latent_dim) gradient tensor, row latent_index is accumulated
};
int launch_part_mlp_bwd(const PartMlpDev& pm, const float* emb_soa, const float* d_soa, int64_t n, int64_t stride,
                        const int32_t* count, const float* g_raw, const int32_t* l_slot, int part, const MlpBwdOut& o, hipStream_t st);
struct MlpAllArgs {               // k_part_mlp_all
    PartMlpDev pm[INVR_NUM_PARTS];
    const float* emb[INVR_NUM_PARTS];
    const float* ds[INVR_NUM_PARTS];
    const int32_t* l_slot[INVR_NUM_PARTS];
    const int32_t* counts;
    int64_t stride, cap;
    float* occp[INVR_NUM_PARTS];
    float4* feat[INVR_NUM_PARTS];
    const int32_t* wl[INVR_NUM_PARTS];
    const int32_t* wcnt;
    const int32_t* gcount;
    const int32_t* n_active;      // counters + CNT_ACTIVE
    float4* rgbw;                 // [slot]; far constants at [cap + p] (cap = the list capacity lcap here)
    int32_t aggr;                 // InvrScene::aggr
};
// occ phase over every listed pair -> k_winner_lists -> rgb phase over the winners
int launch_part_mlp_all(const MlpAllArgs& a, const Workspace& w, hipStream_t st);
int launch_merge_composite(const RenderArgs& a, const Workspace& w, float* rgb_map, float* acc_map, float* raw,
                           float* occ, float* weights, hipStream_t st);
