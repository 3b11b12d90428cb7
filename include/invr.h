/*
 * invr.h — C ABI of libinvr.so, the MI355X (gfx950) per-ray render path of Instant-NVR.
 *
 * The reference (zju3dv/instant-nvr) has NO native code on this path: the path is ~2.9k ATen
 * calls per 4096-ray chunk issued from Python.  The boundary a maintainer binds
 * is therefore the Python plug-in seam (lib/networks/make_network.py:5-8,
 * lib/networks/renderer/make_renderer.py:5-16, lib/train/trainers/make_trainer.py:4-7) and this
 * header is what the replacement classes call through ctypes.  Each entry point names the
 * reference function(s) it replaces.
 *
 * Conventions
 *  - plain C, no torch types; every pointer marked "dev" is a device (HBM) address; structs are
 *    host memory and are read during the call only (they may be freed when the call returns).
 *  - float32 values, int32/int64 indices as stated.  Tensors are dense, row-major, in the
 *    reference's own layouts (nn.Linear weight = (out,in); hash tables = (rows,F) / (H,T,F)).
 *  - the caller owns all memory.  Kernels use only the explicit workspace
 *    (invr_workspace_bytes) and never allocate.
 *  - all work is enqueued on `stream` (a hipStream_t, passed as void*; NULL = default stream);
 *    no call synchronises the device.
 *  - return 0 on success, non-zero on error; invr_last_error() gives the message (thread-local).
 *    No C++ exception crosses the boundary.
 */
#ifndef INVR_H
#define INVR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define INVR_NUM_PARTS 5
#define INVR_MAX_LEVELS 16
#define INVR_MAX_LINEAR 4
#define INVR_NUM_JOINTS 24

/* One multi-resolution grid encoder: constructor arithmetic of
 * lib/networks/embedders/part_base_embedder.py:13-104 (res, cell size, dense/hash split, prime T). */
typedef struct InvrGrid {
    const float* dense;            /* dev (dense_rows, F) or NULL when !separate_dense            */
    const float* hash;             /* dev (n_hash, T, F); (L, T, F) when !separate_dense          */
    const float* bounds;           /* dev (2,3) min xyz / max xyz (a checkpoint parameter, :50)   */
    int32_t n_levels;              /* L                                                            */
    int32_t n_features;            /* F                                                            */
    int32_t start_hash;            /* first hashed level (:63-68)                                  */
    int32_t separate_dense;        /* :69                                                          */
    int64_t table_len;             /* T = nextprime(2^log2_hashmap_size) (:42)                     */
    int32_t res[INVR_MAX_LEVELS];  /* entries_num (:52)                                            */
    float cell[INVR_MAX_LEVELS];   /* entries_size, float32(1/(res-1)) (:54,57)                    */
    int64_t dense_off[INVR_MAX_LEVELS]; /* first row of level l inside `dense` (:129)             */
    int32_t sum;                   /* :163                                                         */
    int32_t sum_over_features;     /* :164                                                         */
    int32_t include_input;         /* :172                                                         */
    const float* row_sums;         /* dev, optional (NULL = off): inference-only derived table built by
                                      invr_grid_row_sums — one float per table row = sum of its F features.
                                      Only read by the render path for sum && sum_over_features grids,
                                      where sum_f sum_k w_k row_k[f] == sum_k w_k (sum_f row_k[f]); the
                                      caller must rebuild it whenever dense/hash change.              */
} InvrGrid;

/* Softplus MLP: lib/networks/bw_deform/part_base_network.py:11-24 / uv_deformer.py:15-21 */
typedef struct InvrMlp {
    const float* weight[INVR_MAX_LINEAR];  /* dev (dims[i+1], dims[i]) */
    const float* bias[INVR_MAX_LINEAR];    /* dev (dims[i+1])          */
    int32_t dims[INVR_MAX_LINEAR + 1];
    int32_t n_linear;
} InvrMlp;

/* One body-part field: part_base_network.py:31-42 */
typedef struct InvrPart {
    InvrGrid grid;
    InvrMlp occ;                 /* 19 -> 64 -> 17                                                */
    InvrMlp rgb;                 /* 70 -> 64 (-> 64) -> 3                                         */
    const float* rgb_latent;     /* dev (num_latent_code, latent_dim)                              */
    int32_t latent_dim;
    int32_t num_latent_code;
} InvrPart;

/* inb_part_network_multiassign.py:68-75,172-182 */
typedef struct InvrModel {
    InvrPart part[INVR_NUM_PARTS];
    InvrGrid deform_grid;        /* uv_deformer.py:14                                             */
    InvrMlp deform_mlp;          /* 19 -> 32 -> 32 -> 3                                           */
    int32_t n_dir_freq;          /* viewdir_embedder.kwargs.res (4)                                */
    int32_t geo_feature_dim;     /* 16                                                             */
} InvrModel;

/* Per-frame tensors of the collated `batch` dict (lib/datasets/h36m/tpose_dataset.py:454-600),
 * leading batch dim of 1 dropped. */
typedef struct InvrScene {
    const float* R;              /* dev (3,3)                                                      */
    const float* Th;             /* dev (3)                                                        */
    const float* A;              /* dev (24,4,4)                                                   */
    const float* big_A;          /* dev (24,4,4)                                                   */
    const float* pbw;            /* dev (Dx,Dy,Dz,C) blend-weight volume; last channel = distance  */
    int32_t pbw_dims[3];
    int32_t pbw_channels;        /* C (25)                                                         */
    const float* pbounds;        /* dev (2,3)                                                      */
    const float* tuv;            /* dev (Dx',Dy',Dz',2)                                            */
    int32_t tuv_dims[3];
    const float* tbounds;        /* dev (2,3)                                                      */
    const float* part_pts;       /* dev (P,M,3)                                                    */
    const float* part_pbw;       /* dev (P,M,24)                                                   */
    const int64_t* lengths2;     /* dev (P)                                                        */
    int32_t part_stride;         /* M                                                              */
    const float* frame_dim;      /* dev (1)                                                        */
    const int64_t* latent_index; /* dev (1)                                                        */
    float smpl_thresh;           /* cfg.smpl_thresh                                                */
    int32_t tpose_viewdir;       /* cfg.tpose_viewdir                                              */
    float composite_eps;         /* the epsilon of render_weights (net_utils.py:12-15) as inb_renderer.py:72 calls it:
                                  * volume_rendering(rgb, occ, cfg.random_bg) passes the bool as `epsilon`, so 0.0 for every INB
                                  * yaml (random_bg False) and 1.0 with random_bg True                                      */
    int32_t aggr;                /* cfg.aggr, how TPoseHuman.forward merges the five parts' (rgb, occ) of a survivor
                                  * (inb_part_network_multiassign.py:236-256): INVR_AGGR_MAX = '' (the part of largest occupancy, every
                                  * INB yaml), INVR_AGGR_MEAN = 'mean' (mean over the five parts, zeros for unflagged parts)            */
} InvrScene;
#define INVR_AGGR_MAX 0
#define INVR_AGGR_MEAN 1
#define INVR_AGGR_DIST 2      /* 'dist' (:240-244): sum over the parts weighted by normalize(1 / (part_dist + 1e-5))                  */
#define INVR_AGGR_MINDIST 3   /* 'mindist' (:245-251): the (rgb, occ) of the part of smallest part_dist (zeros if that part is unflagged) */

/* Device-side statistics block written by invr_render_fwd (int32[INVR_STATS_LEN]). */
#define INVR_STATS_LEN 16
#define INVR_STAT_ACTIVE 0        /* Na: samples that passed the near-surface cull                 */
#define INVR_STAT_PAIRS 1         /* [1..5]: flagged (point,part) pairs per part                    */
#define INVR_STAT_OVERFLOW 6      /* non-zero if max_active was too small (results truncated)       */

const char* invr_last_error(void);
/* Version of THIS header's ABI: bumped by every incompatible change of a struct or a signature.  A host built against another
 * version must refuse the library (invr._abi.lib() does).  History: 1 = rounds 1-3; 2 = round 4 (InvrScene grew composite_eps / aggr,
 * the Adam entry points take the betas as double, launch-level epsilon of the compositing) + round 5 (invr_part_encode_fwd). */
#define INVR_ABI_VERSION 2
int invr_version(void);
/* sizeof() of the ABI structs as compiled (0 InvrGrid, 1 InvrMlp, 2 InvrPart, 3 InvrModel,
 * 4 InvrScene, 5 InvrWsLayout, 6 InvrMlpBwdOut, 7 InvrAdamTensor): lets a binding verify its struct mirrors. */
size_t invr_sizeof(int32_t which);

/* Bytes of workspace invr_render_fwd needs for n_rays x n_samples with at most max_active
 * samples surviving the cull (pass n_rays*n_samples for the worst case). */
size_t invr_workspace_bytes(int64_t n_rays, int32_t n_samples, int64_t max_active);

/* Renderer.render / get_pixel_value, eval outputs (lib/networks/renderer/inb_renderer.py:53-76,
 * 204-239) over the whole ray list without chunking:
 *   get_wsampling_points (:15-31) -> Network.forward (inb_part_network_multiassign.py:126-168)
 *   -> volume_rendering (lib/utils/net_utils.py:12-44, epsilon = 0).
 * ray_o, ray_d (n_rays,3); near, far (n_rays); jitter (n_rays,n_samples) uniform [0,1) or NULL
 * (NULL = eval / perturb 0).  Outputs: rgb_map (n_rays,3), acc_map (n_rays); optional
 * raw (n_rays*n_samples,4), occ (n_rays*n_samples), weights (n_rays,n_samples), z_vals
 * (n_rays,n_samples), stats (INVR_STATS_LEN int32); pass NULL to skip any optional output. */
int invr_render_fwd(const InvrScene* scene, const InvrModel* model,
                    const float* ray_o, const float* ray_d, const float* near, const float* far,
                    const float* jitter, int64_t n_rays, int32_t n_samples,
                    float* rgb_map, float* acc_map, float* raw, float* occ, float* weights,
                    float* z_vals, int32_t* stats,
                    void* workspace, size_t workspace_bytes, int64_t max_active, void* stream);

/* The gradient-free front half of invr_render_fwd only (sampling, cull, KNN skinning, LBS warp,
 * deformer): fills the pair lists / flags / counters in the workspace (invr_workspace_layout) and
 * z_vals (n_rays,n_samples) (optional).  Used by the training forward, which recomputes the
 * differentiable part on those lists. */
int invr_geometry_fwd(const InvrScene* scene, const InvrModel* model,
                      const float* ray_o, const float* ray_d, const float* near, const float* far,
                      const float* jitter, int64_t n_rays, int32_t n_samples, float* z_vals, int32_t* stats,
                      void* workspace, size_t workspace_bytes, int64_t max_active, void* stream);

/* Network.forward (inb_part_network_multiassign.py:126-168) on arbitrary world points: wpts, viewdir
 * (n,3) -> raw (n,4) = [sigmoid rgb, occ] and occ (n) (NULL to skip), zeros where the point is culled or
 * no part is flagged.  Same pipeline as invr_render_fwd with one sample per "ray". */
size_t invr_field_workspace_bytes(int64_t n_points, int64_t max_active);
int invr_field_fwd(const InvrScene* scene, const InvrModel* model, const float* wpts, const float* viewdir,
                   int64_t n_points, float* raw, float* occ, int32_t* stats,
                   void* workspace, size_t workspace_bytes, int64_t max_active, void* stream);

/* Byte offsets of the arrays invr_render_fwd leaves in the workspace (for callers that need the
 * intermediate pair lists: the train-time outputs resd / tpts / tocc of
 * inb_part_network_multiassign.py:162-165 and the backward pass).  Lists are SoA with stride `lcap`
 * (= max_active + 1; entry / slot `max_active` is the per-part far-pair constant). */
typedef struct InvrWsLayout {
    int64_t cap, lcap;
    int64_t counters;                      /* int32[16]: INVR_STAT_* layout                        */
    int64_t active_idx;                    /* int32[lcap]: ray-sample index of each survivor slot (ray-major for training-mode calls; eval frames
                                            * with a power-of-two sample count: by 8-sample depth window inside blocks of 8192 ray-samples) */
    int64_t word_off;                      /* int32[ceil(N/1024)*16]: rank of the first survivor of each 64-sample mask word (ray-major calls) */
    int64_t mask;                          /* uint64[ceil(N/1024)*16]: survivor bit of ray-sample i = bit i&63 of word i>>6;
                                              slot of a surviving sample = word_off[i>>6] + popcount(lower bits) after a ray-major call,
                                              byte_off[i>>3] + popcount(bits of its mask byte below bit i&7) after a depth-windowed one  */
    int64_t pflags, farflags;              /* uint8[lcap]: bit p = (slot, part p) listed / far     */
    int64_t l_slot[INVR_NUM_PARTS];        /* int32[lcap]: survivor slot of each listed pair, ascending; the last entry is the far constant (slot cap) */
    int64_t l_nn[INVR_NUM_PARTS];          /* int32[lcap*4] indexed by SURVIVOR SLOT: the 4 neighbour rows inside part_pbw[p] of a flagged (slot, part) */
    int64_t l_w[INVR_NUM_PARTS];           /* float[lcap*4] indexed by SURVIVOR SLOT: their normalised gaussian weights                                */
    int64_t l_x[INVR_NUM_PARTS];           /* float[3*lcap] SoA: canonical point incl. residual    */
    int64_t l_d[INVR_NUM_PARTS];           /* float[3*lcap] SoA: canonical view direction          */
    int64_t l_r[INVR_NUM_PARTS];           /* float[3*lcap] SoA: residual (resd)                   */
    int64_t emb[INVR_NUM_PARTS];           /* float[20*lcap] SoA [k][pair]: encoder output (19 values + pad) of every listed pair */
    int64_t occp[INVR_NUM_PARTS];          /* float[lcap]: occupancy of every listed pair (list order; last entry = the part's far constant) */
    int64_t wl[INVR_NUM_PARTS];            /* int32[lcap]: list indices of the pairs that win their survivor's max-occupancy merge (the only
                                              pairs the colour MLP evaluates), segmented by groups of 4096 slots: the winners of group g
                                              start at the list offset of the group's first pair, wcnt[g][p] of them                   */
    int64_t wcnt;                          /* int32[n_groups*5]                                     */
    int64_t wsel;                          /* uint8[lcap]: merge result per survivor: p = listed pair of part p, 8+p = far constant of
                                              part p, 255 = zeros                                                                    */
    int64_t rgbw;                          /* float4[lcap+8]: [rgb, occ] of the winning listed pair per survivor; [lcap+p] = far constant of part p */
    int64_t n_groups;
    int64_t knn_dfar2;                     /* float[1]: squared far-fold distance of the frame (0.4624 = (0.68 m)^2 while |A|, |big_A| entries <= 2) */
    int64_t byte_off;                      /* int32[ceil(N/1024)*128]: rank of the first survivor of each 8-sample mask byte — written by eval
                                              frames (invr_render_fwd without jitter / weights, power-of-two n_samples in 8..1024), whose
                                              survivors are ranked by depth window inside blocks of 8192 ray-samples       */
} InvrWsLayout;
int invr_workspace_layout(int64_t n_rays, int32_t n_samples, int64_t max_active, InvrWsLayout* out);

/* ---- per-stage timing with HIP events on the caller's stream ----------------------------------
 * invr_profile_enable(1) makes every following invr_render_fwd record a hipEvent before/after each
 * stage (on the stream the kernels are launched on).  invr_profile_read() waits for the recorded
 * events, ADDS the elapsed milliseconds of every render since the last read to ms[stage], stores the
 * number of renders in *n_renders and clears the record.  Stages: see INVR_STAGE_*. */
#define INVR_STAGE_CULL 0
#define INVR_STAGE_KNN 1
#define INVR_STAGE_WARP 2
#define INVR_STAGE_ENCODE 3       /* +part (3..7)  */
#define INVR_STAGE_MLP 8          /* +part (8..12) */
#define INVR_STAGE_COMPOSITE 13
#define INVR_NUM_STAGES 14
int invr_profile_enable(int32_t on);
int invr_profile_read(float* ms, int32_t* n_renders);

/* ---- stage-level entry points (each is also a step of invr_render_fwd) ---------------------- */

/* HashEmbedder.forward (part_base_embedder.py:106-174).  xyz (n,3) -> out (n,out_dim). */
int invr_grid_encode_fwd(const InvrGrid* grid, const float* xyz, int64_t n, float* out, void* stream);

/* Backward of invr_grid_encode_fwd (autograd of part_base_embedder.py:106-174): g_out (n,out_dim) ->
 * ACCUMULATES (atomic float adds) into g_dense (dense_rows,F) / g_hash (same shape as grid->hash), both
 * caller-zeroed, and writes g_xyz (n,3) (gradient w.r.t. the un-normalised input; NULL to skip). */
int invr_grid_encode_bwd(const InvrGrid* grid, const float* xyz, const float* g_out, int64_t n,
                         float* g_dense, float* g_hash, float* g_xyz, void* stream);

/* pts_sample_blend_weights / pts_sample_uv (lib/utils/blend_utils.py:501-555): trilinear,
 * border, align_corners.  vol (Dx,Dy,Dz,C) sampled at channels [c0, c0+nc) -> out (n,nc). */
int invr_sample_volume(const float* vol, const int32_t dims[3], int32_t channels, int32_t c0, int32_t nc,
                       const float* bounds, const float* pts, int64_t n, float* out, void* stream);

/* pts_knn_blend_weights_multiassign_batch (blend_utils.py:817-825,741-763): pose_pts (n,3) ->
 * bw (n,P,24), dist (n,P).  Exact brute-force 4-NN per part. */
int invr_knn_blend(const InvrScene* scene, const float* pose_pts, int64_t n, float* bw, float* dist,
                   void* stream);

/* The same brute force, returning the neighbours themselves: nn (n,P,4) int32 rows inside part_pts[p] in ascending
 * (squared distance, row) order, d2 (n,P,4) squared distances, w (n,P,4) normalised gaussian weights (blend_utils.py:745-748),
 * dist (n,P).  The reference for the render pipeline's pruned search (k_knn_pairs), whose per-pair results must be identical. */
int invr_knn_neighbors(const InvrScene* scene, const float* pose_pts, int64_t n, int32_t* nn, float* d2, float* w,
                       float* dist, void* stream);

/* get_wsampling_points (inb_renderer.py:15-31) + world_points_to_pose_points / world_dirs_to_pose_dirs
 * (blend_utils.py:366-382) for selected ray-samples, through the device function the render kernels use (bit-identical
 * points): sample_idx (n) int32 = ray*n_samples + s, or NULL for all n = n_rays*n_samples in order; jitter as in
 * invr_render_fwd -> pose_pts (n,3), pose_dirs (n,3) (NULL to skip). */
int invr_pose_points(const InvrScene* scene, const float* ray_o, const float* ray_d, const float* near, const float* far,
                     const float* jitter, int64_t n_rays, int32_t n_samples, const int32_t* sample_idx, int64_t n,
                     float* pose_pts, float* pose_dirs, void* stream);

/* Network.pose_points_to_tpose_points (inb_part_network_multiassign.py:77-120): LBS inverse warp
 * to the big pose + residual deformer for flagged pairs.  pose_pts, pose_dirs (n,3); bw (n,P,24);
 * flag (n,P) uint8 -> tpose (n,P,3), tdirs (n,P,3), resd (n,P,3) (zeros where !flag). */
int invr_warp_deform(const InvrScene* scene, const InvrModel* model, const float* pose_pts,
                     const float* pose_dirs, const float* bw, const uint8_t* flag, int64_t n,
                     float* tpose, float* tdirs, float* resd, void* stream);

/* part_base_network.Network.forward (part_base_network.py:44-63) for part `pid`:
 * tpts, tdirs (n,3) -> raw (n,4) = [sigmoid rgb, occ].  workspace >= invr_part_field_workspace(n). */
size_t invr_part_field_workspace(int64_t n);
int invr_part_field_fwd(const InvrModel* model, int32_t pid, const int64_t* latent_index,
                        const float* tpts, const float* tdirs, int64_t n, float* raw,
                        void* workspace, size_t workspace_bytes, void* stream);

/* HashEmbedder.forward (part_base_embedder.py:106-174) of one 16-level sum / sum_over_features part grid through the render
 * path's own pair-list encoder kernels (invr_grid_encode_fwd is the generic any-configuration kernel): kernel 0 = the
 * XCD-partitioned row-sum kernel of eval frames, 1 = the 64-byte-row kernel (training forward / eval_row_sums False), 2 = the
 * one-part row-sum kernel; 0 and 2 need grid->row_sums.  xyz (n,3) -> out (n,19) = [normalised xyz, 16 level sums].
 * workspace >= invr_part_encode_workspace(n). */
size_t invr_part_encode_workspace(int64_t n);
int invr_part_encode_fwd(const InvrGrid* grid, const float* xyz, int64_t n, int32_t kernel, float* out,
                         void* workspace, size_t workspace_bytes, void* stream);

/* Deformer.forward without flag (uv_deformer.py:31-38; Network.resd, inb_part_network_multiassign.py:122-124):
 * canonical points (n,3) -> residual (n,3).  Uses scene->tuv/tbounds/frame_dim only. */
int invr_deform_fwd(const InvrScene* scene, const InvrModel* model, const float* pts, int64_t n, float* resd,
                    void* stream);

/* Distortion regulariser of inb_renderer.py:96-103: sum_ij w_i w_j |m_i - m_j| with
 * m = (z_i + z_{i+1})/2 (last repeated).  weights, z_vals (n_rays,n_samples) -> out (n_rays). */
int invr_distortion_fwd(const float* weights, const float* z_vals, int64_t n_rays, int32_t n_samples,
                        float* out, void* stream);

/* volume_rendering (net_utils.py:18-44) with epsilon 0: raw (n_rays,n_samples,4) ->
 * weights (n_rays,n_samples) [optional], rgb_map (n_rays,3), acc_map (n_rays). */
int invr_composite_fwd(const float* raw, int64_t n_rays, int32_t n_samples, float* weights,
                       float* rgb_map, float* acc_map, void* stream);

/* get_rays + get_near_far (lib/utils/if_nerf/if_nerf_data_utils.py:24-38, 92-107): pinhole rays of an
 * H x W image and their near/far against the world AABB.  HOST inputs (read during the call): k_inv =
 * inv(K) (3x3, row-major float64), R (3x3), T (3), cam_o = -R^T T (3), bounds (2,3 float32).  DEVICE
 * outputs over all H*W pixels: ray_d (H*W,3), near, far (H*W), mask (H*W uint8 = mask_at_box; near/far are
 * meaningful where it is 1).  ray_o is cam_o cast to float32 for every ray. */
int invr_generate_rays(const double* k_inv, const double* R, const double* T, const double* cam_o,
                       const float* bounds, int32_t H, int32_t W, float* ray_d, float* near, float* far,
                       uint8_t* mask, void* stream);

/* Inference-only row-sum tables for grids with sum && sum_over_features (part_base_embedder.py:163-165:
 * the F features of a level are only ever used through their sum, which commutes with the trilinear
 * interpolation).  invr_grid_row_sums_len = number of floats: dense rows first, then T per hashed level
 * (L*T for a non-separate table).  invr_grid_row_sums fills `out` (DEVICE) from grid->dense/hash. */
int64_t invr_grid_row_sums_len(const InvrGrid* grid);
int invr_grid_row_sums(const InvrGrid* grid, float* out, void* stream);

/* Training path of the two part MLPs (part_base_network.py:44-63 after the encoder) on SoA inputs:
 * emb_soa (20,n) = the 19 encoder outputs as rows (row 19 = 0), dirs_soa (3,n) canonical view directions.
 * fwd: raw (n,4) = [sigmoid rgb, occ]; count_dev = DEVICE int32 holding n.
 * bwd: recomputes the forward, propagates g_raw (n,4) to the embedding and writes, per layer l = 0 occ layer 1, 1 occ
 * layer 2, 2 rgb layer 1, 3 rgb layer 2 (3-linear colour nets only), 4 rgb head, the gradient w.r.t. the layer's
 * output before the activation into gz[l] (n_pad,64) and the layer's input into a[l] (n_pad,72); the caller zero-fills
 * both.  Weight gradients are dW_l = gz[l]^T a[l] (first out_l rows / in_l columns), bias gradients the column sums of
 * gz[l] — K = n reductions left to the caller's batched GEMM.  a[2] is in the kernel's k-slot order (column 4 s + g,
 * s = 0..17, g = 0..3; rgb1_col in csrc/mlp_common.h). */
typedef struct InvrMlpBwdOut {
    float* g_emb;     /* (20,n) rows 0..18 written */
    float* gz;        /* (5,n_pad,64) */
    float* a;         /* (5,n_pad,72) */
    int64_t n_pad;    /* >= n */
    float* g_latent;  /* (8) accumulated (caller pre-zeroes) */
} InvrMlpBwdOut;
int invr_part_mlp_fwd(const InvrModel* model, int32_t pid, const int64_t* latent_index, const float* emb_soa,
                      const float* dirs_soa, int64_t n, const int32_t* count_dev, float* raw, void* stream);
int invr_part_mlp_bwd(const InvrModel* model, int32_t pid, const int64_t* latent_index, const float* emb_soa,
                      const float* dirs_soa, int64_t n, const float* g_raw, const InvrMlpBwdOut* out, void* stream);

/* Dense Adam step over many tensors in ONE launch (the optimiser of the reference's training loop:
 * lib/train/optimizer.py:13-31 -> torch.optim.Adam, one parameter group per tensor, amsgrad off).
 * `tensors` is a DEVICE array; chunk c covers elements [chunk_index[c]*E, (chunk_index[c]+1)*E) of tensor
 * chunk_tensor[c] with E = invr_adam_chunk_elems(); both chunk tables are DEVICE int32 arrays.  bc1 / bc2_sqrt
 * are the bias corrections 1-beta1^step and sqrt(1-beta2^step) of the tensor's own step count. */
typedef struct InvrAdamTensor {
    float* param;               /* dev, updated in place */
    const float* grad;          /* dev */
    float* exp_avg;             /* dev, updated in place */
    float* exp_avg_sq;          /* dev, updated in place */
    int64_t numel;
    float lr, weight_decay, bc1, bc2_sqrt;
    const float* active;        /* dev float[1] or NULL: when it reads 0 the tensor is SKIPPED this step (no update, no step count) — the
                                   reference's Adam skips tensors without a gradient (a body part with no flagged pair in the batch);
                                   the fused backward reports that per part in InvrTrainGrads.part_active */
    int32_t grad_shift;         /* 0: grad has numel elements.  s > 0: grad is a ROW-SCALAR gradient of numel >> s floats, element i
                                   takes grad[i >> s] (sum-over-features hash tables: invr_train_bwd's compact table gradients) */
    int32_t step;               /* step count kept on the device by invr_adam_advance */
} InvrAdamTensor;
int32_t invr_adam_chunk_elems(void);
/* step += 1 and bc1 / bc2_sqrt = 1 - beta1^step / sqrt(1 - beta2^step) (in double, as torch's host code) for the n entries of a
 * DEVICE tensor table: a training loop uploads the table once and replays {invr_adam_advance, invr_adam_step} every iteration
 * without touching the host (torch.optim.Adam computes the corrections on the host per step and per group). */
int invr_adam_advance(InvrAdamTensor* tensors, int32_t n, double beta1, double beta2, void* stream);
int invr_adam_step(const InvrAdamTensor* tensors, const int32_t* chunk_tensor, const int32_t* chunk_index,
                   int64_t n_chunks, double beta1, double beta2, float eps, void* stream);
/* (the betas are doubles: torch forms 1 - beta in double before the update runs in float — 1 - 0.999f is 1.3e-5 off 0.001) */

/* batch_rodrigues + get_rigid_transformation (lib/utils/if_nerf/if_nerf_data_utils.py:523-577): DEVICE inputs
 * poses (24,3) float64 axis-angle, joints (24,3) float64, parents (24) int32 -> DEVICE A (24,4,4) float32. */
int invr_rigid_transformation(const double* poses, const double* joints, const int32_t* parents, float* A, void* stream);

/* Per-part KNN reference sets of the dataset (lib/datasets/h36m/tpose_dataset.py:570-600), all DEVICE
 * pointers: ppts (V,3) posed vertices, weights (V,n_weights), parts (V) int64 part id, tpose (V,3) canonical
 * vertices -> part_pts (5,stride,3), part_pbw (5,stride,n_weights) (zero padded), lengths2 (5) int64,
 * bounds (5,2,3) = min/max of tpose per part -/+ bbox_overlap.  stride >= V; the caller may then narrow the
 * arrays to max(lengths2) as the reference does. */
int invr_pack_parts(const float* ppts, const float* weights, const int64_t* parts, const float* tpose,
                    int32_t n_verts, int32_t n_weights, int32_t stride, float bbox_overlap,
                    float* part_pts, float* part_pbw, int64_t* lengths2, float* bounds, void* stream);

/* ---- training iteration (row f1): forward + backward of the whole path, stream-ordered, no host round trip -------------
 * Together the two calls are one `loss.backward()`-able evaluation of Renderer.render in train mode
 * (inb_renderer.py:53-103) with NetworkWrapper's regularisers (inb_trainer.py:45-48,84-92) pre-reduced on the device.
 *
 * invr_train_fwd = invr_render_fwd (jitter = the stratified perturbation, 64-byte table rows) plus
 *   dist_loss (n_rays)           : reg_distortion_loss (inb_renderer.py:96-103), NULL to skip
 *   terms (INVR_TERM_LEN floats) : [INVR_TERM_OFFSET_SUM] sum over the reference's dense (Na*P,3) resd rows of ||resd||,
 *                                  [INVR_TERM_OFFSET_ROWS] Na*P (offset_loss = sum / rows, inb_trainer.py:89-92);
 *                                  [INVR_TERM_PAIR_SUM] sum over the selected pair-regulariser rows (|tocc - 0.5| < 0.02,
 *                                  inb_renderer.py:80-86) of crit.reg_raw_crit's per-pair term (crit.py:8-18),
 *                                  [INVR_TERM_PAIR_ROWS] their number n (pair_loss = sum / n when n > 0)
 *   pair_noise (pair_noise_rows,3): uniform [0,1) per dense (slot, part) row, pair_noise_rows >= survivor capacity * 5
 *                                  (compute_val_pair_around_range's torch.rand_like, inb_part_network_multiassign.py:41);
 *                                  NULL = no pair regulariser (cfg.use_pair_reg False)
 * raw (N,4), weights and z_vals (n_rays,n_samples) are REQUIRED: invr_train_bwd reads them back together with the pair
 * lists the forward left in the workspace (same buffer, untouched in between; invr_train_workspace_bytes).
 *
 * invr_train_bwd: gradients of a scalar loss given its gradients w.r.t. rgb_map (n_rays,3), acc_map (n_rays) [NULL = 0],
 * dist_loss (n_rays) [NULL = 0], raw (N,4) [NULL = 0] and the two sums of `terms` (DEVICE float[1] each, NULL = 0).
 * Every gradient is ACCUMULATED into the caller's (pre-zeroed) buffers of InvrTrainGrads; the gradient of a part's
 * sum-over-features hash tables is the compact row-scalar array (one float per table row, invr_grid_row_sums order:
 * d out / d table[row][f] is the same for all 16 f) — 16x less gradient traffic for the optimiser and for a data-parallel
 * all-reduce; invr_expand_row_grad turns it into the dense tensor gradient when one is wanted. */
#define INVR_TERM_OFFSET_SUM 0
#define INVR_TERM_OFFSET_ROWS 1
#define INVR_TERM_PAIR_SUM 2
#define INVR_TERM_PAIR_ROWS 3
#define INVR_TERM_LEN 8
typedef struct InvrPartGrads {
    float* row_grad;                        /* dev (invr_grid_row_sums_len) */
    float* occ_w[INVR_MAX_LINEAR];          /* dev, shapes of InvrMlp.weight / bias */
    float* occ_b[INVR_MAX_LINEAR];
    float* rgb_w[INVR_MAX_LINEAR];
    float* rgb_b[INVR_MAX_LINEAR];
    float* rgb_latent;                      /* dev (num_latent_code, latent_dim): row latent_index receives the gradient */
} InvrPartGrads;
typedef struct InvrTrainGrads {
    InvrPartGrads part[INVR_NUM_PARTS];
    float* deform_dense;                    /* dev, shapes of the deformer grid's dense / hash tables */
    float* deform_hash;
    float* deform_w[INVR_MAX_LINEAR];
    float* deform_b[INVR_MAX_LINEAR];
    float* part_active;                     /* dev float[INVR_NUM_PARTS] or NULL: += 1 for every part that had a flagged (near or far)
                                               pair in this backward — 0 = the reference would not have touched the part's parameters */
} InvrTrainGrads;
size_t invr_train_workspace_bytes(int64_t n_rays, int32_t n_samples, int64_t max_active);
int invr_train_fwd(const InvrScene* scene, const InvrModel* model,
                   const float* ray_o, const float* ray_d, const float* near, const float* far, const float* jitter,
                   int64_t n_rays, int32_t n_samples, const float* pair_noise, int64_t pair_noise_rows,
                   float* rgb_map, float* acc_map, float* raw, float* occ, float* weights, float* z_vals,
                   float* dist_loss, float* terms, int32_t* stats,
                   void* workspace, size_t workspace_bytes, int64_t max_active, void* stream);
int invr_train_bwd(const InvrScene* scene, const InvrModel* model, int64_t n_rays, int32_t n_samples,
                   const float* raw, const float* weights, const float* z_vals,
                   const float* g_rgb_map, const float* g_acc_map, const float* g_dist_loss, const float* g_raw,
                   const float* g_offset_sum, const float* g_pair_sum, const InvrTrainGrads* grads, int32_t stages,
                   void* workspace, size_t workspace_bytes, int64_t max_active, void* stream);
/* `stages` of invr_train_bwd (0 = all): a data-parallel trainer calls HEAD, then the parts one by one — starting the
 * all-reduce of a part's gradients as soon as its kernels are enqueued, beside the remaining backward — then DEFORMER
 * (which needs every part's canonical-point gradient).  HEAD must come first, DEFORMER last. */
#define INVR_BWD_HEAD 1
#define INVR_BWD_PART(p) (2 << (p))
#define INVR_BWD_DEFORMER 64
#define INVR_BWD_ALL 127
/* row-scalar table gradient (invr_grid_row_sums order) -> dense gradients g_dense (dense_rows,F) / g_hash (n_hash,T,F) of the
 * grid's tables (every feature of a row takes the row's scalar); overwrites.  g_dense may be NULL for a non-separate table. */
int invr_expand_row_grad(const InvrGrid* grid, const float* row_grad, float* g_dense, float* g_hash, void* stream);

/* The training objective of NetworkWrapper.forward (lib/train/trainers/inb_trainer.py:40-98, 176-214 with the plain MSE image term,
 * use_lpips False) on the outputs of invr_train_fwd, one launch each way:
 *   loss = w_pair * pair + w_dist * mean(dist) + w_off * offset + mean((rgb_map - rgb_gt)^2)      (the wrapper's order of additions)
 *   offset = terms[OFFSET_SUM] / max(terms[OFFSET_ROWS], 1), pair likewise (use_pair 0: no pair term); dist may be NULL.
 * out8 = {loss, img_loss, psnr = -10 log10(img_loss), reg_dist, offset_loss, pair_loss, 0, 0}; err (n_rays) = sum_c |rgb - gt| or NULL.
 * invr_train_loss_bwd: g_loss (1) -> g_rgb (n_rays,3), g_dist (n_rays) or NULL, g_terms (8) = the gradients invr_train_bwd takes. */
int invr_train_loss_fwd(const float* rgb_map, const float* rgb_gt, const float* dist, const float* terms, int64_t n_rays,
                        float w_pair, float w_dist, float w_off, int32_t use_pair, float* out8, float* err, void* stream);
int invr_train_loss_bwd(const float* rgb_map, const float* rgb_gt, const float* terms, int64_t n_rays, float w_pair, float w_dist,
                        float w_off, int32_t use_pair, const float* g_loss, float* g_rgb, float* g_dist, float* g_terms, void* stream);

/* Backward of invr_composite_fwd: g_rgb_map (n_rays,3), g_acc_map (n_rays) or NULL, g_weights
 * (n_rays,n_samples) or NULL (e.g. from the distortion regulariser) -> g_raw (n_rays,n_samples,4). */
int invr_composite_bwd(const float* raw, const float* g_rgb_map, const float* g_acc_map, const float* g_weights,
                       int64_t n_rays, int32_t n_samples, float* g_raw, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* INVR_H */
