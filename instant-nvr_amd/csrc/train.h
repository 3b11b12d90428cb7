// Training-pass data layout (k_train.hip, invr_train_fwd / invr_train_bwd in invr_abi.hip).
#pragma once
#include "pipeline.h"

enum { TERM_OFFSET_SUM = 0, TERM_OFFSET_ROWS = 1, TERM_PAIR_SUM = 2, TERM_PAIR_ROWS = 3, TERM_LEN = 8 };

// Arrays of one training iteration behind the render workspace (same caller-owned buffer, invr_train_workspace_bytes).
struct TrainWs {
    int32_t* pair_of;      // lcap*P : list position of (slot, part) for listed pairs
    float* terms;          // TERM_LEN : device-side loss reductions (see INVR_TERM_* in invr.h)
    float* nb_x;           // NB*3 : jittered neighbours of the pair-regulariser rows
    float* nb_r;           // NB*3 : their residuals
    int32_t* nb_ref;       // NB : part << 28 | list position of the row's own pair
    float* g_w;            // N : gradient w.r.t. the compositing weights (distortion^T)
    float4* g_rawfull;     // N : gradient w.r.t. the merged per-sample raw
    float4* g_raws;        // lcap*P : gradient w.r.t. the per-(slot, part) raw
    float* g_emb[INVR_NUM_PARTS];   // 20*lcap each : gradient w.r.t. the part's encoder output (own buffers: the parts' chains run concurrently)
    float* g_x[INVR_NUM_PARTS];   // 3*lcap each, SoA : gradient w.r.t. the canonical point (encoder^T)
    float* gz[INVR_NUM_PARTS]; float* a[INVR_NUM_PARTS];   // (5, lcap, 64), (5, lcap, 72) per part : per-layer (output gradient, input)
    // deformer backward list: DM = P*lcap + NB entries
    float* d_pts; float* d_g; float* d_uvt; float* d_gfeat;       // (DM,3) (DM,3) (DM,3) (DM,19)
    float* d_gz1; float* d_gz2; float* d_gz3; float* d_a0; float* d_a1; float* d_a2;   // (DM,32) (DM,32) (DM,4) (DM,20) (DM,32) (DM,32)
    int64_t NB, DM;
};

struct DeformGrads { float* w[3]; float* b[3]; float* dense; float* hash; };

int launch_train_terms(const RenderArgs& a, const Workspace& w, const TrainWs& t, const GridDev& dg, const MlpDev& dm,
                       const float* noise, hipStream_t st);
int launch_distortion_bwd(const float* weights, const float* z, const float* g_dist, int64_t R, int S, float* g_w, hipStream_t st);
int launch_merge_bwd(const Workspace& w, int aggr, const float4* g_rawfull, float4* g_raws, hipStream_t st);
int launch_deform_bwd(const RenderArgs& a, const Workspace& w, const TrainWs& t, const GridDev& dg, const MlpDev& dm,
                      const float* g_off_sum, const float* g_pair_sum, const DeformGrads& G, hipStream_t st, hipStream_t side = nullptr, hipEvent_t ev_fork = nullptr, hipEvent_t ev_join = nullptr);
struct WgradJob;
struct WgradJobs;
int launch_part_wgrad(const float* gz, const float* a, int64_t lcap, int n_rgb, float* const* dW, float* const* db,
                      const int32_t* count, hipStream_t st);
int launch_deform_slice_bwd(const GridDev& dg, const float* frame_dim, const float* uvt, const float* gfeat, int64_t n_max,
                            const int32_t* count, float* g_dense, float* g_hash, hipStream_t st);
int launch_train_loss(const float* rgb, const float* gt, const float* dist, const float* terms, int64_t n, float w_pair, float w_dist,
                      float w_off, int use_pair, float* out, float* err, hipStream_t st);
int launch_train_loss_bwd(const float* rgb, const float* gt, const float* terms, int64_t n, float w_pair, float w_dist, float w_off,
                          int use_pair, const float* g_loss, float* g_rgb, float* g_dist, float* g_terms, hipStream_t st);
