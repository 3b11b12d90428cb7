// C-ABI entry points of libinvr.so (include/invr.h) and the stream-ordered orchestration of one
// render: cull -> KNN pairs -> warp+deform -> per part {encode -> MLP} -> merge+composite.
// No host synchronisation anywhere: survivor / pair counts stay on the device and the consumers
// are persistent grid-stride kernels that read them.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
#include <vector>
#include "pipeline.h"
#include "train.h"

static thread_local char g_err[512] = "";

void invr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* invr_last_error(void) { return g_err; }
extern "C" int invr_version(void) { return INVR_ABI_VERSION; }
extern "C" size_t invr_sizeof(int32_t which) {
    switch (which) {
        case 0: return sizeof(InvrGrid);
        case 1: return sizeof(InvrMlp);
        case 2: return sizeof(InvrPart);
        case 3: return sizeof(InvrModel);
        case 4: return sizeof(InvrScene);
        case 5: return sizeof(InvrWsLayout);
        case 6: return sizeof(InvrMlpBwdOut);
        case 7: return sizeof(InvrAdamTensor);
        case 8: return sizeof(InvrTrainGrads);
        default: return 0;
    }
}

GridDev make_grid_dev(const InvrGrid* g) {
    GridDev d;
    memset(&d, 0, sizeof(d));
    d.dense = g->dense; d.hash = g->hash; d.bounds = g->bounds;
    d.L = g->n_levels; d.F = g->n_features; d.start_hash = g->start_hash; d.separate_dense = g->separate_dense;
    d.T = g->table_len; d.inv_T = 1.0 / (double)g->table_len;
    {   // 32-bit modulo eligibility: T = 2^k + c, x < 2^41 (res <= 8192, prime factor < 2^27)
        int k = 0;
        while ((1ll << (k + 1)) <= d.T) ++k;
        const uint64_t c = (uint64_t)(d.T - (1ll << k));
        d.mod_k = k; d.mod_c = (uint32_t)c; d.mod32 = 0;
        if (k >= 10 && k <= 30 && c > 0) {
            const uint64_t y1 = c * ((1ull << 41) >> k);                  // bound of round 1
            const uint64_t y2 = c * (y1 >> k), y3 = c * (y2 >> k);
            if (y1 < (1ull << 32) && y2 < (1ull << 32) && y3 < (uint64_t)d.T && (y3 >> k) == 0 &&
                (uint64_t)d.T < (1ull << 30))
                d.mod32 = 1;
            // 24-bit multiplies: c and the multiplicands h0 <= 2^41 >> k, h1 <= y1 >> k, h2 <= y2 >> k below 2^24
            d.mod24 = d.mod32 && c < (1u << 24) && ((1ull << 41) >> k) < (1ull << 24) && (y1 >> k) < (1ull << 24) && (y2 >> k) < (1ull << 24);
            // x-corner delta fold (k_encode.hip:level_rowsum): resolutions are <= 8192, so |delta| < 2^13 < T / 2; its c0x corners take
            // the TWO-round reduction (common.h:hash_mod24_2r), valid when the second fold already lands below 2^k (h2 = 0)
            int64_t max_res = 0;
            for (int l = 0; l < g->n_levels && l < INVR_MAX_LEVELS; ++l) max_res = g->res[l] > max_res ? g->res[l] : max_res;
            d.xdelta = d.mod24 && d.T > (1ll << 14) && y2 < (1ull << k) && 2 * max_res <= d.T;      // (one +/- T fix-up: |delta| < 2 res <= T)
            // one-round reduction (common.h:hash_mod24_1r): the keys are XORs of cx, cy * 19349663, cz * 83492791 with c. <= max_res - 1,
            // so they stay below the next power of two above max_res * 83492791
            int xbits = 0;
            while (xbits < 62 && (1ull << xbits) <= (uint64_t)max_res * 83492791ull) ++xbits;
            d.mod1r = d.xdelta && xbits > k && c * ((1ull << xbits) >> k) < 2ull * (uint64_t)d.T && 3ull * (uint64_t)d.T < (1ull << 31);
        }
    }
    for (int l = 0; l < INVR_MAX_LEVELS; ++l) {
        d.res[l] = g->res[l]; d.cell[l] = g->cell[l]; d.dense_off[l] = g->dense_off[l];
        // reciprocal form of x / cell (common.h:div_by_rcp): y = RN(1 / cell), the float nearest to the exact reciprocal — chosen
        // among the double-rounded candidate and its neighbours by the exact residual |1 - cell * y| (a product of two floats is
        // exact in double); not for a divisor whose mantissa is all ones (Markstein's exception) or outside [2^-20, 2^20]
        d.rcell[l] = 0.0f;
        const float b = g->cell[l];
        uint32_t bits;
        memcpy(&bits, &b, 4);
        if (l < g->n_levels && b > 9.6e-7f && b < 1.0e6f && (bits & 0x7fffffu) != 0x7fffffu) {
            const float y0 = (float)(1.0 / (double)b);
            float best = y0;
            double err = fabs(1.0 - (double)b * (double)y0);
            for (float c : {nextafterf(y0, 0.0f), nextafterf(y0, INFINITY)}) {
                const double e = fabs(1.0 - (double)b * (double)c);
                if (e < err) { err = e; best = c; }
            }
            d.rcell[l] = best;
        }
    }
    d.sum = g->sum; d.sum_over_features = g->sum_over_features; d.include_input = g->include_input;
    d.row_sums = (g->sum && g->sum_over_features) ? g->row_sums : nullptr;
    d.dense_rows = 0;
    if (g->separate_dense)
        for (int l = 0; l < g->start_hash && l < INVR_MAX_LEVELS; ++l) d.dense_rows += (int64_t)g->res[l] * g->res[l] * g->res[l];
    return d;
}

MlpDev make_mlp_dev(const InvrMlp* m) {
    MlpDev d;
    memset(&d, 0, sizeof(d));
    for (int i = 0; i < INVR_MAX_LINEAR; ++i) { d.w[i] = m->weight[i]; d.b[i] = m->bias[i]; }
    for (int i = 0; i <= INVR_MAX_LINEAR; ++i) d.dims[i] = m->dims[i];
    d.n_linear = m->n_linear;
    return d;
}

SceneDev make_scene_dev(const InvrScene* s) {
    SceneDev d;
    memset(&d, 0, sizeof(d));
    d.R = s->R; d.Th = s->Th; d.A = s->A; d.big_A = s->big_A;
    d.pbw = VolDev{s->pbw, s->pbounds, s->pbw_dims[0], s->pbw_dims[1], s->pbw_dims[2], s->pbw_channels};
    d.tuv = VolDev{s->tuv, s->tbounds, s->tuv_dims[0], s->tuv_dims[1], s->tuv_dims[2], 2};
    d.part_pts = s->part_pts; d.part_pbw = s->part_pbw; d.lengths2 = s->lengths2; d.M = s->part_stride;
    d.frame_dim = s->frame_dim; d.latent_index = s->latent_index;
    d.thresh = s->smpl_thresh; d.tpose_viewdir = s->tpose_viewdir; d.comp_eps = s->composite_eps; d.aggr = s->aggr;
    // Unflagged band of the nearest-vertex distance d1 (k_knn.hip header): dist >= g(d1) with
    // g(d) = d*w/(w+1e-8), w = exp(-d^2/0.01125); g rises ~d then collapses near 0.48 m.
    auto g = [](double d) { double w = exp(-d * d / (2.0 * 0.075 * 0.075)); return d * w / (w + 1e-8); };
    const double th = (double)s->smpl_thresh;
    d.near_hi2 = __builtin_inff(); d.band_lo2 = 0.f;
    const double lo = th * 1.01;
    if (th > 0 && lo < 0.42 && g(lo) >= th * 1.002 && g(0.42) >= th * 1.05) {
        double a = 0.42, b = 0.8;                       // g is decreasing on [0.42, 0.8]
        for (int it = 0; it < 60; ++it) { double m = 0.5 * (a + b); if (g(m) >= th * 1.05) a = m; else b = m; }
        d.near_hi2 = (float)(lo * lo * 1.0001);
        d.band_lo2 = (float)(a * a * 0.9999);
    }
    return d;
}

static int check_grid(const InvrGrid* g, const char* what) {
    INVR_CHECK(g->n_levels >= 1 && g->n_levels <= INVR_MAX_LEVELS, "%s: n_levels %d out of range", what, g->n_levels);
    INVR_CHECK(g->n_features >= 1 && g->n_features <= 16, "%s: n_features %d out of range", what, g->n_features);
    INVR_CHECK(g->hash != nullptr && g->bounds != nullptr, "%s: null table/bounds", what);
    INVR_CHECK(!g->separate_dense || g->dense != nullptr, "%s: separate_dense without dense table", what);
    INVR_CHECK(g->table_len > 1 && g->table_len < (1ll << 31), "%s: table_len out of range", what);
    for (int l = 0; l < g->n_levels; ++l) INVR_CHECK(g->res[l] >= 2 && g->res[l] <= 8192, "%s: level %d resolution %d out of range", what, l, g->res[l]);
    return 0;
}

static int check_mlp_deform(const InvrMlp* m) {
    INVR_CHECK(m->n_linear == 3 && m->dims[0] == 19 && m->dims[1] == 32 && m->dims[2] == 32 && m->dims[3] == 3,
               "deformer MLP must be 19-32-32-3");
    return 0;
}

// ---- per-stage HIP-event profiling -----------------------------------------------------------------
#include <vector>
struct ProfInterval { int stage; hipEvent_t t0, t1; };
static bool g_prof_on = false;
static int g_prof_renders = 0;
static std::vector<ProfInterval> g_prof_recs;
static std::vector<hipEvent_t> g_prof_pool;

static hipEvent_t prof_event() {
    hipEvent_t e = nullptr;
    if (!g_prof_pool.empty()) { e = g_prof_pool.back(); g_prof_pool.pop_back(); }
    else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
    return e;
}

extern "C" int invr_profile_enable(int32_t on) { g_prof_on = on != 0; return 0; }

extern "C" int invr_profile_read(float* ms, int32_t* n_renders) {
    INVR_CHECK(ms && n_renders, "invr_profile_read: null pointer");
    *n_renders = g_prof_renders;
    for (auto& r : g_prof_recs) {
        INVR_HIP(hipEventSynchronize(r.t1));
        float t = 0.f;
        INVR_HIP(hipEventElapsedTime(&t, r.t0, r.t1));
        ms[r.stage] += t;
        g_prof_pool.push_back(r.t0);
        g_prof_pool.push_back(r.t1);
    }
    g_prof_recs.clear();
    g_prof_renders = 0;
    return 0;
}

struct ProfStage {      // RAII: one (begin,end) event pair around a stage's launches, on the launch stream
    hipEvent_t t0 = nullptr, t1 = nullptr;
    int stage;
    hipStream_t st;
    ProfStage(int stage_, hipStream_t st_) : stage(stage_), st(st_) {
        if (g_prof_on) { t0 = prof_event(); t1 = prof_event(); if (t0 && t1) (void)hipEventRecord(t0, st); }
    }
    ~ProfStage() {
        if (t0 && t1) { (void)hipEventRecord(t1, st); g_prof_recs.push_back({stage, t0, t1}); }
    }
};

// ---- workspace carve -----------------------------------------------------------------------------
struct Carver {
    char* base;
    size_t off;
    template <class T> T* take(size_t n) {
        off = align_up(off, 256);
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

// counters + per-group pair counts, cleared by one memset per frame: a multiple of 256 bytes (an odd size splits the fill in two launches)
static size_t counters_ints(int64_t n_groups) { return align_up((size_t)CNT_ALLOC + (size_t)n_groups * INVR_NUM_PARTS, 64); }

#define KNN_MAX_PART 8192
static size_t carve(Workspace& w, void* base, int64_t N, int64_t cap) {
    Carver c{(char*)base, 0};
    int64_t nb = cdiv(N, 1024);          // cull tiles (k_cull.hip CULL_TILE)
    w.cap = cap;
    const int64_t lc = cap + 1;          // list / slot capacity incl. the far-constant entry
    w.lcap = lc;
    w.n_groups = cdiv(lc, PAIR_GROUP);
    w.counters = c.take<int32_t>(counters_ints(w.n_groups));
    w.gcount = w.counters ? w.counters + CNT_ALLOC : nullptr;
    w.knn.part_aabb = c.take<float>(INVR_NUM_PARTS * 6);
    w.knn.dfar2 = c.take<float>(1);
    w.knn.mpad = KNN_MAX_PART;
    w.knn.cpad = KNN_MAX_PART / 64;
    w.knn.sverts = c.take<float4>((size_t)INVR_NUM_PARTS * w.knn.mpad);
    w.knn.cl = c.take<float4>((size_t)INVR_NUM_PARTS * w.knn.cpad * 3);
    w.knn.sub = c.take<float4>((size_t)INVR_NUM_PARTS * w.knn.cpad * 8);
    w.knn.vmat = c.take<float4>((size_t)INVR_NUM_PARTS * w.knn.mpad * 6);
    w.knn.voxcls = c.take<uint8_t>((size_t)VOXMASK_MAX_CELLS * INVR_NUM_PARTS);
    w.knn.voxmask = c.take<unsigned long long>((size_t)VOXMASK_MAX_CELLS * INVR_NUM_PARTS);
    w.knn.voxu2 = c.take<float>((size_t)VOXMASK_MAX_CELLS * INVR_NUM_PARTS);
    w.knn.live_cells = c.take<int32_t>((size_t)VOXMASK_MAX_CELLS);
    w.mask = c.take<unsigned long long>(nb * 16);
    w.block_cnt = c.take<int32_t>(nb);
    w.block_off = c.take<int32_t>(nb);
    w.super_tot = c.take<int32_t>(cdiv(nb, 1024) + 1);
    w.active_idx = c.take<int32_t>(lc);
    w.word_off = c.take<int32_t>(nb * 16);
    w.byte_off = c.take<int32_t>(nb * 128);
    w.ord_rows = w.ord_cols = 0;
    w.pflags = c.take<uint8_t>(lc);
    w.farflags = c.take<uint8_t>(lc);
    for (int p = 0; p < INVR_NUM_PARTS; ++p) {
        w.l_slot[p] = c.take<int32_t>(lc);
        w.l_nn[p] = c.take<int32_t>(lc * 4);
        w.l_w[p] = c.take<float>(lc * 4);
        w.l_x[p] = c.take<float>(lc * 3);
        w.l_d[p] = c.take<float>(lc * 3);
        w.l_r[p] = c.take<float>(lc * 3);
    }
    for (int p = 0; p < INVR_NUM_PARTS; ++p) w.emb[p] = c.take<float>(lc * EMB_K);
    for (int p = 0; p < INVR_NUM_PARTS; ++p) {
        w.occp[p] = c.take<float>(lc);
        w.feat[p] = c.take<float4>(lc * 4);
        w.wl[p] = c.take<int32_t>(lc);
    }
    w.wcnt = c.take<int32_t>(w.n_groups * INVR_NUM_PARTS);
    w.wsel = c.take<uint8_t>(lc);
    w.rgbw = c.take<float4>(lc + 8);
    w.dslice = c.take<float2>(DF_SLICE_MAX);
    w.cullmask = c.take<uint8_t>(CULL_MASK_MAX);
    w.pdist = c.take<float>(lc * INVR_NUM_PARTS);          // (last: every older offset of invr_workspace_layout is unchanged)
    w.cullmask_d1 = c.take<uint8_t>(CULL_MASK_MAX);
    w.use_d1 = 0;
    w.knn.srow = c.take<uint16_t>((size_t)INVR_NUM_PARTS * w.knn.mpad);
    return align_up(c.off, 256);
}

extern "C" int invr_workspace_layout(int64_t n_rays, int32_t n_samples, int64_t max_active, InvrWsLayout* o) {
    INVR_CHECK(o != nullptr, "invr_workspace_layout: null output");
    Workspace w;
    int64_t N = n_rays * (int64_t)n_samples;
    if (max_active <= 0 || max_active > N) max_active = N;
    if (max_active < 1) max_active = 1;
    char* base = reinterpret_cast<char*>(uintptr_t(1) << 40);        // dummy non-null base: only offsets are used
    carve(w, base, N > 0 ? N : 1, max_active);
    auto off = [&](const void* p) { return (int64_t)(reinterpret_cast<const char*>(p) - base); };
    o->cap = w.cap; o->lcap = w.lcap;
    o->counters = off(w.counters); o->active_idx = off(w.active_idx); o->word_off = off(w.word_off); o->mask = off(w.mask);
    o->pflags = off(w.pflags); o->farflags = off(w.farflags); o->knn_dfar2 = off(w.knn.dfar2); o->byte_off = off(w.byte_off);
    for (int p = 0; p < INVR_NUM_PARTS; ++p) {
        o->l_slot[p] = off(w.l_slot[p]); o->l_nn[p] = off(w.l_nn[p]); o->l_w[p] = off(w.l_w[p]);
        o->l_x[p] = off(w.l_x[p]); o->l_d[p] = off(w.l_d[p]); o->l_r[p] = off(w.l_r[p]);
    }
    for (int p = 0; p < INVR_NUM_PARTS; ++p) o->emb[p] = off(w.emb[p]);
    for (int p = 0; p < INVR_NUM_PARTS; ++p) { o->occp[p] = off(w.occp[p]); o->wl[p] = off(w.wl[p]); }
    o->wcnt = off(w.wcnt); o->wsel = off(w.wsel); o->rgbw = off(w.rgbw); o->n_groups = w.n_groups;
    return 0;
}

extern "C" size_t invr_workspace_bytes(int64_t n_rays, int32_t n_samples, int64_t max_active) {
    Workspace w;
    int64_t N = n_rays * (int64_t)n_samples;
    if (max_active <= 0 || max_active > N) max_active = N;
    if (max_active < 1) max_active = 1;
    return carve(w, nullptr, N > 0 ? N : 1, max_active);
}

__global__ void k_export_stats(const int32_t* counters, int32_t* stats) {
    int t = threadIdx.x;
    if (t < INVR_STATS_LEN) stats[t] = t < CNT_LEN ? counters[t] : 0;
}

static PartMlpDev make_part_mlp(const InvrModel* m, int p, const int64_t* latent_index) {
    PartMlpDev pm;
    pm.occ = make_mlp_dev(&m->part[p].occ);
    pm.rgb = make_mlp_dev(&m->part[p].rgb);
    pm.rgb_latent = m->part[p].rgb_latent;
    pm.latent_index = latent_index;
    pm.latent_dim = m->part[p].latent_dim;
    pm.n_freq = m->n_dir_freq;
    pm.geo_dim = m->geo_feature_dim;
    return pm;
}

// library-owned streams for the five per-part chains of the training forward / backward (one set per device and host thread)
struct PartStreams { hipStream_t s[INVR_NUM_PARTS] = {}; hipEvent_t fork = nullptr, dfork = nullptr, djoin = nullptr, done[INVR_NUM_PARTS] = {}, done2[INVR_NUM_PARTS] = {}; };
static PartStreams* part_streams() {
    static thread_local std::vector<PartStreams> of_device;
    int dev_id = 0;
    if (hipGetDevice(&dev_id) != hipSuccess || dev_id < 0 || dev_id >= 4096) { invr_set_error("invr: bad device index"); return nullptr; }
    if ((size_t)dev_id >= of_device.size()) of_device.resize((size_t)dev_id + 1);
    PartStreams& ps = of_device[dev_id];
    if (!ps.fork) {
        if (hipEventCreateWithFlags(&ps.fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ps.dfork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ps.djoin, hipEventDisableTiming) != hipSuccess) { invr_set_error("invr: event creation failed"); return nullptr; }
        for (int p = 0; p < INVR_NUM_PARTS; ++p)
            if (hipStreamCreateWithFlags(&ps.s[p], hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&ps.done[p], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&ps.done2[p], hipEventDisableTiming) != hipSuccess) { invr_set_error("invr: stream creation failed"); return nullptr; }
    }
    return &ps;
}

static int render_impl(const InvrScene* scene, const InvrModel* model,
                       const float* ray_o, const float* ray_d, const float* near, const float* far,
                       const float* jitter, const float* wpts, const float* wdirs, int64_t n_rays, int32_t n_samples,
                       float* rgb_map, float* acc_map, float* raw, float* occ, float* weights,
                       float* z_vals, int32_t* stats,
                       void* workspace, size_t workspace_bytes, int64_t max_active, void* stream,
                       bool geometry_only = false, bool may_reorder = false) {
    hipStream_t st = (hipStream_t)stream;
    INVR_CHECK(scene && model, "invr_render_fwd: null scene/model");
    INVR_CHECK(n_rays >= 0 && (n_samples >= 2 || (wpts && n_samples == 1)), "invr_render_fwd: need n_rays >= 0 and n_samples >= 2");
    if (n_rays == 0) return 0;
    INVR_CHECK(((ray_o && ray_d && near && far) || (wpts && wdirs)) && (geometry_only || (rgb_map && acc_map)), "invr_render_fwd: null ray/output pointer");
    const int64_t N = n_rays * (int64_t)n_samples;
    INVR_CHECK(N < (1ll << 31), "invr_render_fwd: n_rays*n_samples must be < 2^31 (got %lld); split the ray list", (long long)N);
    if (max_active <= 0 || max_active > N) max_active = N;
    Workspace w;
    size_t need = carve(w, workspace, N, max_active);
    INVR_CHECK(workspace != nullptr && workspace_bytes >= need, "invr_render_fwd: workspace too small (%zu < %zu bytes)", workspace_bytes, need);
    INVR_CHECK(((uintptr_t)workspace & 255) == 0, "invr_render_fwd: workspace must be 256-byte aligned");
    if (check_grid(&model->deform_grid, "deformer grid") || check_mlp_deform(&model->deform_mlp)) return 1;
    INVR_CHECK(model->deform_grid.n_levels == 8 && model->deform_grid.n_features == 2 && !model->deform_grid.sum &&
               model->deform_grid.include_input, "deformer grid must be 8 levels x 2 features, sum=False, include_input");
    for (int p = 0; p < INVR_NUM_PARTS; ++p) if (check_grid(&model->part[p].grid, "part grid")) return 1;
    INVR_CHECK(scene->pbw_channels >= 1 && scene->part_stride >= 1, "invr_render_fwd: bad scene dims");
    INVR_CHECK(scene->tpose_viewdir, "invr_render_fwd: tpose_viewdir=False is not supported (the reference cannot run it either: "
               "TPoseHuman.forward indexes the (Na,3) view directions per part)");
    INVR_CHECK(scene->aggr >= INVR_AGGR_MAX && scene->aggr <= INVR_AGGR_MINDIST, "invr_render_fwd: InvrScene::aggr %d (0 = max-occupancy merge, 1 = mean, 2 = dist, 3 = mindist)", scene->aggr);
    INVR_CHECK(scene->part_stride <= KNN_MAX_PART, "invr_render_fwd: part_stride %d > %d vertices per part", scene->part_stride, KNN_MAX_PART);

    // survivor order of the frame: ray-major, or (eval frames with a power-of-two sample count) depth-windowed inside
    // blocks of 8192 ray-samples.  Every training forward — invr_train_fwd / invr_geometry_fwd, and the op-by-op path's
    // invr_render_fwd calls, which carry jitter or ask for the weights — keeps the reference's ray-major row order (its per-row
    // noise and its (Na*P, .) outputs are defined on it).
    static const int order_env = getenv("INVR_ORDER") ? atoi(getenv("INVR_ORDER")) : 1;      // (0 = ray-major everywhere: A/B switch, tools/ab_order.sh)
    if (order_env && may_reorder && !wpts && !jitter && !weights && n_samples >= 8 && n_samples <= 1024 && (n_samples & (n_samples - 1)) == 0) {
        w.ord_cols = n_samples / 8;
        w.ord_rows = 1024 / w.ord_cols;
    }
    RenderArgs a;
    a.scene = make_scene_dev(scene);
    a.ray_o = ray_o; a.ray_d = ray_d; a.near = near; a.far = far; a.jitter = jitter; a.z_vals = z_vals;
    a.wpts = wpts; a.wdirs = wdirs;
    a.R = n_rays; a.S = n_samples; a.N = N;

    // The front of the frame on the caller's stream (k_knn.hip "the front of a frame as two launches"): one memset, one launch for
    // everything that depends on the scene alone (KNN index, cull cell mask + live-cell list, per-vertex matrices, deformer slices),
    // one launch for the lattice-cell classes and the cull flags, then scan / compaction.  No library-owned stream: a fork / join
    // under hipGraph replay cost more than the kernels it hid.
    INVR_HIP(hipMemsetAsync(w.counters, 0, counters_ints(w.n_groups) * sizeof(int32_t), st));
    GridDev dgrid = make_grid_dev(&model->deform_grid);
    // ablation switches: the environment is read once per process, not per frame
    static const bool no_voxmask = getenv("INVR_NO_VOXMASK") != nullptr, no_voxcls = getenv("INVR_NO_VOXCLS") != nullptr,
                      no_merge = getenv("INVR_NO_MERGE") != nullptr;
    {
        ProfStage ps(INVR_STAGE_CULL, st);
        int have_cells = 0, flags_done = 0;
        if (launch_front_scene(a, w, dgrid, &have_cells, st)) return 1;
        if (no_voxmask) w.knn.voxmask = nullptr;
        if (!have_cells || no_voxcls) { w.knn.voxcls = nullptr; w.knn.voxmask = nullptr; }
        else if (launch_front_cull(a, w, &flags_done, st)) return 1;
        if (w.knn.voxcls && !flags_done && launch_knn_voxel_class(a, w, st)) return 1;      // (small calls: the cull runs unmasked)
        if (launch_cull(a, w, max_active, have_cells != 0, flags_done != 0, st)) return 1;
    }
    {
        ProfStage ps(INVR_STAGE_KNN, st);
        if (launch_knn_pairs(a, w, stats, st)) return 1;
        if (scene->aggr >= INVR_AGGR_DIST && launch_knn_pdist(a, w, st)) return 1;
    }
    {
        ProfStage ps(INVR_STAGE_WARP, st);
        MlpDev dm = make_mlp_dev(&model->deform_mlp);
        if (launch_warp_pairs(a, w, dgrid, dm, st)) return 1;
    }
    if (!geometry_only) {
        // The five parts in one encoder launch and one launch per MLP phase (stage times are booked on part 0).  Eval reads the
        // row-sum tables; the training forward / eval without row sums read the trainable 64-byte rows — a part's 1e4-5e4 pairs
        // make short, latency-bound launches, so the parts also run side by side there (the profiler and INVR_NO_MERGE keep
        // per-part encoder launches, stage times per part).
        bool row_sums = true;
        for (int p = 0; p < INVR_NUM_PARTS; ++p) row_sums = row_sums && model->part[p].grid.row_sums != nullptr;
        EncodeAllArgs ea;
        MlpAllArgs ma;
        for (int p = 0; p < INVR_NUM_PARTS; ++p) {
            ea.g[p] = make_grid_dev(&model->part[p].grid);
            ea.xs[p] = w.l_x[p]; ea.emb[p] = w.emb[p];
            ma.pm[p] = make_part_mlp(model, p, scene->latent_index);
            ma.emb[p] = w.emb[p]; ma.ds[p] = w.l_d[p]; ma.l_slot[p] = w.l_slot[p];
            ma.occp[p] = w.occp[p]; ma.feat[p] = w.feat[p]; ma.wl[p] = w.wl[p];
        }
        ea.counts = ma.counts = w.counters + CNT_PAIRS;
        ea.stride = ma.stride = w.lcap; ea.cap = ma.cap = w.lcap;
        ma.wcnt = w.wcnt; ma.gcount = w.gcount; ma.n_active = w.counters + CNT_ACTIVE; ma.rgbw = w.rgbw; ma.aggr = scene->aggr;
        if (no_merge || (g_prof_on && !row_sums)) {
            for (int p = 0; p < INVR_NUM_PARTS; ++p) {
                ProfStage ps(INVR_STAGE_ENCODE + p, st);
                if (launch_part_encode(ea.g[p], w.l_x[p], w.lcap, w.counters + CNT_PAIRS + p, w.lcap, w.emb[p], st)) return 1;
            }
        } else {
            ProfStage ps(INVR_STAGE_ENCODE, st);
            if (row_sums ? launch_part_encode_all(ea, st) : launch_part_encode_rows_all(ea, st)) return 1;
        }
        { ProfStage ps(INVR_STAGE_MLP, st); if (launch_part_mlp_all(ma, w, st)) return 1; }
    }
    if (!geometry_only) {
        ProfStage ps(INVR_STAGE_COMPOSITE, st);
        if (launch_merge_composite(a, w, rgb_map, acc_map, raw, occ, weights, st)) return 1;
    }
    if (g_prof_on) ++g_prof_renders;
    return 0;                   // (stats were exported by the KNN stage: the counters are final once the pair lists are)
}

extern "C" int invr_render_fwd(const InvrScene* scene, const InvrModel* model,
                               const float* ray_o, const float* ray_d, const float* near, const float* far,
                               const float* jitter, int64_t n_rays, int32_t n_samples,
                               float* rgb_map, float* acc_map, float* raw, float* occ, float* weights,
                               float* z_vals, int32_t* stats,
                               void* workspace, size_t workspace_bytes, int64_t max_active, void* stream) {
    return render_impl(scene, model, ray_o, ray_d, near, far, jitter, nullptr, nullptr, n_rays, n_samples, rgb_map, acc_map,
                       raw, occ, weights, z_vals, stats, workspace, workspace_bytes, max_active, stream, false, true);
}

extern "C" int invr_geometry_fwd(const InvrScene* scene, const InvrModel* model,
                                 const float* ray_o, const float* ray_d, const float* near, const float* far,
                                 const float* jitter, int64_t n_rays, int32_t n_samples, float* z_vals, int32_t* stats,
                                 void* workspace, size_t workspace_bytes, int64_t max_active, void* stream) {
    return render_impl(scene, model, ray_o, ray_d, near, far, jitter, nullptr, nullptr, n_rays, n_samples, nullptr, nullptr,
                       nullptr, nullptr, nullptr, z_vals, stats, workspace, workspace_bytes, max_active, stream, true);
}

extern "C" size_t invr_field_workspace_bytes(int64_t n_points, int64_t max_active) {
    return invr_workspace_bytes(n_points, 1, max_active) + align_up((size_t)(n_points > 0 ? n_points : 1) * 4 * sizeof(float), 256);
}

extern "C" int invr_field_fwd(const InvrScene* scene, const InvrModel* model, const float* wpts, const float* viewdir,
                              int64_t n_points, float* raw, float* occ, int32_t* stats,
                              void* workspace, size_t workspace_bytes, int64_t max_active, void* stream) {
    INVR_CHECK(n_points == 0 || (wpts && viewdir && raw), "invr_field_fwd: null pointer");
    if (n_points == 0) return 0;
    const size_t inner = invr_workspace_bytes(n_points, 1, max_active);
    INVR_CHECK(workspace && workspace_bytes >= invr_field_workspace_bytes(n_points, max_active), "invr_field_fwd: workspace too small");
    float* scratch = reinterpret_cast<float*>(static_cast<char*>(workspace) + inner);     // rgb_map (n,3) + acc_map (n)
    return render_impl(scene, model, nullptr, nullptr, nullptr, nullptr, nullptr, wpts, viewdir, n_points, 1,
                       scratch, scratch + 3 * n_points, raw, occ, nullptr, nullptr, stats, workspace, inner, max_active, stream);
}

// ---- stage-level entry points -------------------------------------------------------------------
extern "C" int invr_grid_encode_fwd(const InvrGrid* grid, const float* xyz, int64_t n, float* out, void* stream) {
    INVR_CHECK(grid && (n == 0 || (xyz && out)), "invr_grid_encode_fwd: null pointer");
    if (check_grid(grid, "grid")) return 1;
    return launch_grid_encode_generic(make_grid_dev(grid), xyz, n, out, (hipStream_t)stream);
}

extern "C" int invr_grid_encode_bwd(const InvrGrid* grid, const float* xyz, const float* g_out, int64_t n,
                                    float* g_dense, float* g_hash, float* g_xyz, void* stream) {
    INVR_CHECK(grid && (n == 0 || (xyz && g_out)) && g_hash, "invr_grid_encode_bwd: null pointer");
    if (check_grid(grid, "grid")) return 1;
    INVR_CHECK(!grid->separate_dense || g_dense, "invr_grid_encode_bwd: g_dense required for a separate dense table");
    return launch_grid_encode_bwd_generic(make_grid_dev(grid), xyz, g_out, n, g_dense, g_hash, g_xyz, (hipStream_t)stream);
}

extern "C" int invr_sample_volume(const float* vol, const int32_t dims[3], int32_t channels, int32_t c0, int32_t nc,
                                  const float* bounds, const float* pts, int64_t n, float* out, void* stream) {
    INVR_CHECK(vol && dims && bounds && (n == 0 || (pts && out)), "invr_sample_volume: null pointer");
    INVR_CHECK(c0 >= 0 && nc >= 1 && c0 + nc <= channels, "invr_sample_volume: channel range [%d,%d) outside %d", c0, c0 + nc, channels);
    VolDev v{vol, bounds, dims[0], dims[1], dims[2], channels};
    return launch_sample_volume(v, c0, nc, pts, n, out, (hipStream_t)stream);
}

extern "C" int invr_knn_blend(const InvrScene* scene, const float* pose_pts, int64_t n, float* bw, float* dist, void* stream) {
    INVR_CHECK(scene && (n == 0 || (pose_pts && bw && dist)), "invr_knn_blend: null pointer");
    return launch_knn_blend_dense(make_scene_dev(scene), pose_pts, n, bw, dist, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int invr_knn_neighbors(const InvrScene* scene, const float* pose_pts, int64_t n, int32_t* nn, float* d2, float* w,
                                  float* dist, void* stream) {
    INVR_CHECK(scene && (n == 0 || (pose_pts && nn && d2 && w && dist)), "invr_knn_neighbors: null pointer");
    return launch_knn_blend_dense(make_scene_dev(scene), pose_pts, n, nullptr, dist, nn, d2, w, (hipStream_t)stream);
}

extern "C" int invr_pose_points(const InvrScene* scene, const float* ray_o, const float* ray_d, const float* near, const float* far,
                                const float* jitter, int64_t n_rays, int32_t n_samples, const int32_t* sample_idx, int64_t n,
                                float* pose_pts, float* pose_dirs, void* stream) {
    INVR_CHECK(scene && (n == 0 || (ray_o && ray_d && near && far && pose_pts)), "invr_pose_points: null pointer");
    INVR_CHECK(n_rays >= 0 && n_samples >= 2 && n_rays * (int64_t)n_samples < (1ll << 31), "invr_pose_points: bad n_rays / n_samples");
    INVR_CHECK(sample_idx || n == n_rays * (int64_t)n_samples, "invr_pose_points: without sample_idx n must be n_rays*n_samples");
    RenderArgs a;
    memset(&a, 0, sizeof(a));
    a.scene = make_scene_dev(scene);
    a.ray_o = ray_o; a.ray_d = ray_d; a.near = near; a.far = far; a.jitter = jitter;
    a.R = n_rays; a.S = n_samples; a.N = n_rays * (int64_t)n_samples;
    return launch_pose_points(a, sample_idx, n, pose_pts, pose_dirs, (hipStream_t)stream);
}

extern "C" int invr_warp_deform(const InvrScene* scene, const InvrModel* model, const float* pose_pts,
                                const float* pose_dirs, const float* bw, const uint8_t* flag, int64_t n,
                                float* tpose, float* tdirs, float* resd, void* stream) {
    INVR_CHECK(scene && model && (n == 0 || (pose_pts && pose_dirs && bw && flag && tpose && tdirs && resd)), "invr_warp_deform: null pointer");
    if (check_grid(&model->deform_grid, "deformer grid") || check_mlp_deform(&model->deform_mlp)) return 1;
    return launch_warp_deform_dense(make_scene_dev(scene), make_grid_dev(&model->deform_grid), make_mlp_dev(&model->deform_mlp),
                                    pose_pts, pose_dirs, bw, flag, n, tpose, tdirs, resd, (hipStream_t)stream);
}

__global__ void k_aos_to_soa(const float* a, const float* b, int64_t n, float* xs, float* ds, int32_t* count) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *count = (int32_t)n;
    if (i >= n) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) { xs[c * n + i] = a[i * 3 + c]; ds[c * n + i] = b[i * 3 + c]; }
}

extern "C" size_t invr_part_field_workspace(int64_t n) {
    if (n < 1) n = 1;
    return align_up(256 + (size_t)n * (3 + 3 + EMB_K) * sizeof(float) + 3 * 256, 256);
}

extern "C" int invr_part_field_fwd(const InvrModel* model, int32_t pid, const int64_t* latent_index,
                                   const float* tpts, const float* tdirs, int64_t n, float* raw,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    INVR_CHECK(model && latent_index && pid >= 0 && pid < INVR_NUM_PARTS, "invr_part_field_fwd: bad model/pid");
    if (n == 0) return 0;
    INVR_CHECK(tpts && tdirs && raw, "invr_part_field_fwd: null pointer");
    INVR_CHECK(n < (1ll << 31), "invr_part_field_fwd: n too large");
    INVR_CHECK(workspace && workspace_bytes >= invr_part_field_workspace(n), "invr_part_field_fwd: workspace too small");
    if (check_grid(&model->part[pid].grid, "part grid")) return 1;
    Carver c{(char*)workspace, 0};
    int32_t* count = c.take<int32_t>(1);
    float* xs = c.take<float>(3 * n);
    float* ds = c.take<float>(3 * n);
    float* emb = c.take<float>(EMB_K * n);
    hipLaunchKernelGGL(k_aos_to_soa, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, tpts, tdirs, n, xs, ds, count);
    INVR_LAUNCH_CHECK();
    if (launch_part_encode(make_grid_dev(&model->part[pid].grid), xs, n, count, n, emb, st)) return 1;
    PartMlpDev pm = make_part_mlp(model, pid, latent_index);
    return launch_part_mlp(pm, emb, ds, n, count, n, reinterpret_cast<float4*>(raw), st);
}

// HashEmbedder.forward of ONE part grid through the render path's pair-list encoders (the generic any-configuration kernel is
// invr_grid_encode_fwd): kernel 0 = the XCD-partitioned row-sum kernel of eval frames (k_part_encode_rs_xcd; needs grid->row_sums),
// 1 = the 64-byte-row kernel of the training forward / eval_row_sums False (k_part_encode), 2 = the one-part row-sum kernel
// (k_part_encode_rs).  xyz (n,3) -> out (n,19) = [normalised xyz, 16 level sums].
__global__ void k_xyz_to_soa(const float* a, int64_t n, int64_t stride, float* xs, int32_t* count) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { count[0] = (int32_t)n; count[1] = count[2] = count[3] = count[4] = 0; }
    if (i >= n) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) xs[c * stride + i] = a[i * 3 + c];
}
__global__ void k_emb_to_aos(const float* emb, int64_t n, int64_t cap, float* out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 19) return;
    out[i] = emb[(i % 19) * cap + i / 19];
}

extern "C" size_t invr_part_encode_workspace(int64_t n) {
    if (n < 1) n = 1;
    return align_up(256 + (size_t)n * (3 + EMB_K) * sizeof(float) + 3 * 256, 256);
}

extern "C" int invr_part_encode_fwd(const InvrGrid* grid, const float* xyz, int64_t n, int32_t kernel, float* out,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    INVR_CHECK(grid && kernel >= 0 && kernel <= 2, "invr_part_encode_fwd: bad grid / kernel");
    if (n == 0) return 0;
    INVR_CHECK(xyz && out, "invr_part_encode_fwd: null pointer");
    INVR_CHECK(n < (1ll << 31), "invr_part_encode_fwd: n too large");
    INVR_CHECK(workspace && workspace_bytes >= invr_part_encode_workspace(n), "invr_part_encode_fwd: workspace too small");
    if (check_grid(grid, "part grid")) return 1;
    INVR_CHECK(kernel == 1 || grid->row_sums, "invr_part_encode_fwd: the row-sum kernels need grid->row_sums (invr_grid_row_sums)");
    Carver c{(char*)workspace, 0};
    int32_t* count = c.take<int32_t>(8);
    float* xs = c.take<float>(3 * n);
    float* emb = c.take<float>(EMB_K * n);
    hipLaunchKernelGGL(k_xyz_to_soa, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, xyz, n, n, xs, count);
    INVR_LAUNCH_CHECK();
    GridDev g = make_grid_dev(grid);
    if (kernel == 0) {
        EncodeAllArgs ea;
        for (int p = 0; p < INVR_NUM_PARTS; ++p) { ea.g[p] = g; ea.xs[p] = xs; ea.emb[p] = emb; }      // counts = {n, 0, 0, 0, 0}
        ea.counts = count; ea.stride = n; ea.cap = n;
        if (launch_part_encode_all(ea, st)) return 1;
    } else {
        if (kernel == 1) g.row_sums = nullptr;
        if (launch_part_encode(g, xs, n, count, n, emb, st)) return 1;
    }
    hipLaunchKernelGGL(k_emb_to_aos, dim3((unsigned)cdiv(n * 19, 256)), dim3(256), 0, st, emb, n, n, out);
    INVR_LAUNCH_CHECK();
    return 0;
}

extern "C" int invr_deform_fwd(const InvrScene* scene, const InvrModel* model, const float* pts, int64_t n, float* resd,
                               void* stream) {
    INVR_CHECK(scene && model && (n == 0 || (pts && resd)), "invr_deform_fwd: null pointer");
    if (check_grid(&model->deform_grid, "deformer grid") || check_mlp_deform(&model->deform_mlp)) return 1;
    INVR_CHECK(model->deform_grid.n_levels == 8 && model->deform_grid.n_features == 2 && !model->deform_grid.sum &&
               model->deform_grid.include_input, "deformer grid must be 8 levels x 2 features, sum=False, include_input");
    return launch_deform_points(make_scene_dev(scene), make_grid_dev(&model->deform_grid), make_mlp_dev(&model->deform_mlp),
                                pts, n, resd, (hipStream_t)stream);
}

extern "C" int invr_distortion_fwd(const float* weights, const float* z_vals, int64_t n_rays, int32_t n_samples,
                                   float* out, void* stream) {
    INVR_CHECK(n_rays == 0 || (weights && z_vals && out), "invr_distortion_fwd: null pointer");
    INVR_CHECK(n_samples >= 1, "invr_distortion_fwd: n_samples must be >= 1");
    return launch_distortion(weights, z_vals, n_rays, n_samples, out, (hipStream_t)stream);
}

extern "C" int invr_composite_fwd(const float* raw, int64_t n_rays, int32_t n_samples, float* weights,
                                  float* rgb_map, float* acc_map, void* stream) {
    INVR_CHECK(n_rays == 0 || (raw && rgb_map && acc_map), "invr_composite_fwd: null pointer");
    INVR_CHECK(n_samples >= 1, "invr_composite_fwd: n_samples must be >= 1");
    return launch_composite(raw, n_rays, n_samples, 0.0f, weights, rgb_map, acc_map, (hipStream_t)stream);      // (epsilon 0: the INB call)
}

extern "C" int invr_composite_bwd(const float* raw, const float* g_rgb_map, const float* g_acc_map, const float* g_weights,
                                  int64_t n_rays, int32_t n_samples, float* g_raw, void* stream) {
    INVR_CHECK(n_rays == 0 || (raw && g_rgb_map && g_raw), "invr_composite_bwd: null pointer");
    INVR_CHECK(n_samples >= 1, "invr_composite_bwd: n_samples must be >= 1");
    return launch_composite_bwd(raw, g_rgb_map, g_acc_map, g_weights, n_rays, n_samples, 0.0f, g_raw, (hipStream_t)stream);
}

extern "C" int invr_generate_rays(const double* k_inv, const double* R, const double* T, const double* cam_o,
                                  const float* bounds, int32_t H, int32_t W, float* ray_d, float* near, float* far,
                                  uint8_t* mask, void* stream) {
    INVR_CHECK(k_inv && R && T && cam_o && bounds, "invr_generate_rays: null camera pointer");
    INVR_CHECK(H >= 0 && W >= 0 && (H * (int64_t)W == 0 || (ray_d && near && far && mask)), "invr_generate_rays: bad size / null output");
    return launch_generate_rays(k_inv, R, T, cam_o, bounds, H, W, ray_d, near, far, mask, (hipStream_t)stream);
}

extern "C" int invr_rigid_transformation(const double* poses, const double* joints, const int32_t* parents, float* A, void* stream) {
    INVR_CHECK(poses && joints && parents && A, "invr_rigid_transformation: null pointer");
    return launch_rigid_transformation(poses, joints, parents, A, (hipStream_t)stream);
}

extern "C" int invr_pack_parts(const float* ppts, const float* weights, const int64_t* parts, const float* tpose,
                               int32_t n_verts, int32_t n_weights, int32_t stride, float bbox_overlap,
                               float* part_pts, float* part_pbw, int64_t* lengths2, float* bounds, void* stream) {
    INVR_CHECK(ppts && weights && parts && tpose && part_pts && part_pbw && lengths2 && bounds, "invr_pack_parts: null pointer");
    INVR_CHECK(n_verts >= 0 && n_weights >= 1 && stride >= n_verts, "invr_pack_parts: stride must be >= n_verts");
    return launch_pack_parts(ppts, weights, parts, tpose, n_verts, n_weights, stride, bbox_overlap, part_pts, part_pbw, lengths2,
                             bounds, (hipStream_t)stream);
}

extern "C" int64_t invr_grid_row_sums_len(const InvrGrid* grid) {
    if (!grid || grid->n_levels < 1 || grid->n_levels > INVR_MAX_LEVELS) return 0;
    GridDev g = make_grid_dev(grid);
    return g.separate_dense ? g.dense_rows + (int64_t)(g.L - g.start_hash) * g.T : (int64_t)g.L * g.T;
}

extern "C" int invr_grid_row_sums(const InvrGrid* grid, float* out, void* stream) {
    INVR_CHECK(grid && out, "invr_grid_row_sums: null pointer");
    if (check_grid(grid, "grid")) return 1;
    INVR_CHECK(grid->sum && grid->sum_over_features, "invr_grid_row_sums: only sum && sum_over_features grids have row sums");
    INVR_CHECK(grid->n_features % 4 == 0, "invr_grid_row_sums: n_features must be a multiple of 4");
    return launch_row_sums(make_grid_dev(grid), out, (hipStream_t)stream);
}

extern "C" int32_t invr_adam_chunk_elems(void) { return 16384; }

extern "C" int invr_adam_advance(InvrAdamTensor* tensors, int32_t n, double beta1, double beta2, void* stream) {
    INVR_CHECK(n == 0 || tensors, "invr_adam_advance: null pointer");
    return launch_adam_advance(tensors, n, beta1, beta2, (hipStream_t)stream);
}

extern "C" int invr_adam_step(const InvrAdamTensor* tensors, const int32_t* chunk_tensor, const int32_t* chunk_index,
                              int64_t n_chunks, double beta1, double beta2, float eps, void* stream) {
    INVR_CHECK(n_chunks == 0 || (tensors && chunk_tensor && chunk_index), "invr_adam_step: null pointer");
    INVR_CHECK(n_chunks >= 0 && n_chunks < (1ll << 31), "invr_adam_step: bad chunk count");
    return launch_adam(tensors, chunk_tensor, chunk_index, n_chunks, beta1, beta2, eps, (hipStream_t)stream);
}

extern "C" int invr_part_mlp_fwd(const InvrModel* model, int32_t pid, const int64_t* latent_index, const float* emb_soa,
                                 const float* dirs_soa, int64_t n, const int32_t* count_dev, float* raw, void* stream) {
    INVR_CHECK(model && latent_index && pid >= 0 && pid < INVR_NUM_PARTS, "invr_part_mlp_fwd: bad model/pid");
    if (n == 0) return 0;
    INVR_CHECK(emb_soa && dirs_soa && count_dev && raw, "invr_part_mlp_fwd: null pointer");
    PartMlpDev pm = make_part_mlp(model, pid, latent_index);
    return launch_part_mlp(pm, emb_soa, dirs_soa, n, count_dev, n, reinterpret_cast<float4*>(raw), (hipStream_t)stream);
}

extern "C" int invr_part_mlp_bwd(const InvrModel* model, int32_t pid, const int64_t* latent_index, const float* emb_soa,
                                 const float* dirs_soa, int64_t n, const float* g_raw, const InvrMlpBwdOut* out, void* stream) {
    INVR_CHECK(model && latent_index && pid >= 0 && pid < INVR_NUM_PARTS && out, "invr_part_mlp_bwd: bad model/pid/out");
    if (n == 0) return 0;
    PartMlpDev pm = make_part_mlp(model, pid, latent_index);
    INVR_CHECK(emb_soa && dirs_soa && g_raw && out->g_emb && out->gz && out->a && out->g_latent && out->n_pad >= n, "invr_part_mlp_bwd: null pointer / n_pad < n");
    const MlpDev& oc = pm.occ;
    const MlpDev& r = pm.rgb;
    INVR_CHECK(oc.n_linear == 2 && oc.dims[0] == 19 && oc.dims[1] == 64 && oc.dims[2] == 17 && (r.n_linear == 2 || r.n_linear == 3) &&
               r.dims[0] == 70 && r.dims[1] == 64 && r.dims[r.n_linear] == 3 && pm.n_freq == 4 && pm.latent_dim == 8 && pm.geo_dim == 16,
               "invr_part_mlp_bwd: supports occ 19-64-17 and rgb 70-64(-64)-3");
    MlpBwdOut o{out->g_emb, out->gz, out->a, out->n_pad, out->g_latent, 0};
    return launch_part_mlp_bwd(pm, emb_soa, dirs_soa, n, n, nullptr, g_raw, nullptr, pid, o, (hipStream_t)stream);
}

// ---- training iteration (k_train.hip) ----------------------------------------------------------------------------------
static size_t carve_train(TrainWs& t, void* base, size_t off0, int64_t N, int64_t lcap) {
    Carver c{(char*)base, off0};
    t.NB = lcap * INVR_NUM_PARTS;
    t.DM = lcap * INVR_NUM_PARTS + t.NB;
    t.pair_of = c.take<int32_t>(lcap * INVR_NUM_PARTS);
    t.terms = nullptr;                                   // caller-owned (the forward's output)
    t.nb_x = c.take<float>(t.NB * 3);
    t.nb_r = c.take<float>(t.NB * 3);
    t.nb_ref = c.take<int32_t>(t.NB);
    t.g_w = c.take<float>(N);
    t.g_rawfull = c.take<float4>(N);
    t.g_raws = c.take<float4>(lcap * INVR_NUM_PARTS);
    for (int p = 0; p < INVR_NUM_PARTS; ++p) t.g_emb[p] = c.take<float>(lcap * EMB_K);
    for (int p = 0; p < INVR_NUM_PARTS; ++p) t.g_x[p] = c.take<float>(lcap * 3);
    for (int p = 0; p < INVR_NUM_PARTS; ++p) { t.gz[p] = c.take<float>(lcap * 5 * 64); t.a[p] = c.take<float>(lcap * 5 * 72); }
    t.d_pts = c.take<float>(t.DM * 3);
    t.d_g = c.take<float>(t.DM * 3);
    t.d_uvt = c.take<float>(t.DM * 3);
    t.d_gfeat = c.take<float>(t.DM * 19);
    t.d_gz1 = c.take<float>(t.DM * 32);
    t.d_gz2 = c.take<float>(t.DM * 32);
    t.d_gz3 = c.take<float>(t.DM * 4);
    t.d_a0 = c.take<float>(t.DM * 20);
    t.d_a1 = c.take<float>(t.DM * 32);
    t.d_a2 = c.take<float>(t.DM * 32);
    return align_up(c.off, 256);
}

static int64_t clamp_active(int64_t N, int64_t max_active) {
    if (max_active <= 0 || max_active > N) max_active = N;
    return max_active < 1 ? 1 : max_active;
}

extern "C" size_t invr_train_workspace_bytes(int64_t n_rays, int32_t n_samples, int64_t max_active) {
    const int64_t N = n_rays * (int64_t)n_samples > 0 ? n_rays * (int64_t)n_samples : 1;
    const int64_t cap = clamp_active(N, max_active);
    Workspace w;
    TrainWs t;
    const size_t inner = carve(w, nullptr, N, cap);
    return carve_train(t, nullptr, inner, N, cap + 1);
}

extern "C" int invr_train_fwd(const InvrScene* scene, const InvrModel* model,
                              const float* ray_o, const float* ray_d, const float* near, const float* far, const float* jitter,
                              int64_t n_rays, int32_t n_samples, const float* pair_noise, int64_t pair_noise_rows,
                              float* rgb_map, float* acc_map, float* raw, float* occ, float* weights, float* z_vals,
                              float* dist_loss, float* terms, int32_t* stats,
                              void* workspace, size_t workspace_bytes, int64_t max_active, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    INVR_CHECK(scene && model && terms, "invr_train_fwd: null scene / model / terms");
    INVR_HIP(hipMemsetAsync(terms, 0, TERM_LEN * sizeof(float), st));
    if (n_rays == 0) return 0;
    INVR_CHECK(raw && weights && z_vals, "invr_train_fwd: raw, weights and z_vals are required (the backward reads them)");
    const int64_t N = n_rays * (int64_t)n_samples;
    const int64_t cap = clamp_active(N, max_active);
    INVR_CHECK(workspace && workspace_bytes >= invr_train_workspace_bytes(n_rays, n_samples, max_active), "invr_train_fwd: workspace too small");
    INVR_CHECK(!pair_noise || pair_noise_rows >= cap * INVR_NUM_PARTS, "invr_train_fwd: pair_noise needs >= %lld rows", (long long)(cap * INVR_NUM_PARTS));
    for (int p = 0; p < INVR_NUM_PARTS; ++p)
        INVR_CHECK(model->part[p].grid.row_sums == nullptr, "invr_train_fwd: training reads the trainable 64-byte rows (row_sums must be NULL)");
    const size_t inner = invr_workspace_bytes(n_rays, n_samples, max_active);
    if (render_impl(scene, model, ray_o, ray_d, near, far, jitter, nullptr, nullptr, n_rays, n_samples, rgb_map, acc_map, raw, occ, weights,
                    z_vals, nullptr, workspace, inner, max_active, stream))
        return 1;
    Workspace w;
    TrainWs t;
    carve(w, workspace, N, cap);
    carve_train(t, workspace, inner, N, cap + 1);
    t.terms = terms;
    if (dist_loss && launch_distortion(weights, z_vals, n_rays, n_samples, dist_loss, st)) return 1;
    RenderArgs a;
    memset(&a, 0, sizeof(a));
    a.scene = make_scene_dev(scene);
    a.R = n_rays; a.S = n_samples; a.N = N;
    if (launch_train_terms(a, w, t, make_grid_dev(&model->deform_grid), make_mlp_dev(&model->deform_mlp), pair_noise, st)) return 1;
    if (stats) {
        hipLaunchKernelGGL(k_export_stats, dim3(1), dim3(64), 0, st, w.counters, stats);
        INVR_LAUNCH_CHECK();
    }
    return 0;
}

__global__ void k_part_active(const int32_t* __restrict__ counters, float* __restrict__ act) {
    const int p = threadIdx.x;
    if (p < INVR_NUM_PARTS && (counters[CNT_PAIRS + p] > 1 || counters[CNT_FAR + p] > 0)) act[p] += 1.0f;    // (the list always ends with the far constant)
}

__global__ void k_add_inplace(float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

extern "C" int invr_train_bwd(const InvrScene* scene, const InvrModel* model, int64_t n_rays, int32_t n_samples,
                              const float* raw, const float* weights, const float* z_vals,
                              const float* g_rgb_map, const float* g_acc_map, const float* g_dist_loss, const float* g_raw,
                              const float* g_offset_sum, const float* g_pair_sum, const InvrTrainGrads* grads, int32_t stages,
                              void* workspace, size_t workspace_bytes, int64_t max_active, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    INVR_CHECK(scene && model && grads, "invr_train_bwd: null scene / model / grads");
    if (stages == 0) stages = INVR_BWD_ALL;
    if (n_rays == 0) return 0;
    INVR_CHECK(raw && weights && z_vals && g_rgb_map, "invr_train_bwd: raw, weights, z_vals and g_rgb_map are required");
    const int64_t N = n_rays * (int64_t)n_samples;
    const int64_t cap = clamp_active(N, max_active);
    INVR_CHECK(workspace && workspace_bytes >= invr_train_workspace_bytes(n_rays, n_samples, max_active), "invr_train_bwd: workspace too small");
    const size_t inner = invr_workspace_bytes(n_rays, n_samples, max_active);
    Workspace w;
    TrainWs t;
    carve(w, workspace, N, cap);
    carve_train(t, workspace, inner, N, cap + 1);
    const int64_t lcap = w.lcap;
    // distortion^T -> compositing^T (-> + direct gradient of raw) -> merge^T
    if (stages & INVR_BWD_HEAD) {
    if (g_dist_loss && launch_distortion_bwd(weights, z_vals, g_dist_loss, n_rays, n_samples, t.g_w, st)) return 1;
    if (launch_composite_bwd(raw, g_rgb_map, g_acc_map, g_dist_loss ? t.g_w : nullptr, n_rays, n_samples, scene->composite_eps, reinterpret_cast<float*>(t.g_rawfull), st)) return 1;
    if (g_raw) {
        hipLaunchKernelGGL(k_add_inplace, dim3((unsigned)cdiv(N * 4, 256)), dim3(256), 0, st, reinterpret_cast<float*>(t.g_rawfull), g_raw, N * 4);
        INVR_LAUNCH_CHECK();
    }
    if (launch_merge_bwd(w, scene->aggr, t.g_rawfull, t.g_raws, st)) return 1;
    if (grads->part_active) {
        hipLaunchKernelGGL(k_part_active, dim3(1), dim3(64), 0, st, w.counters, grads->part_active);
        INVR_LAUNCH_CHECK();
    }
    }
    // per part: MLPs^T -> weight gradients -> encoder^T.  The five chains are independent (own scratch, own gradient tensors) and
    // each is a handful of latency-bound launches (a part's 1e4-5e4 pairs do not fill the chip: 115 us per MLP^T launch whatever
    // the pair count), so with more than one part in the call they run on library-owned streams side by side.
    int n_parts = 0;
    for (int p = 0; p < INVR_NUM_PARTS; ++p) n_parts += (stages & INVR_BWD_PART(p)) ? 1 : 0;
    PartStreams* ps = nullptr;
    if (n_parts > 1) {
        ps = part_streams();
        if (!ps) return 1;
        INVR_HIP(hipEventRecord(ps->fork, st));
    }
    for (int p = 0; p < INVR_NUM_PARTS; ++p) {
        if (!(stages & INVR_BWD_PART(p))) continue;
        hipStream_t sp = ps ? ps->s[p] : st;
        if (ps) INVR_HIP(hipStreamWaitEvent(sp, ps->fork, 0));
        const InvrPartGrads& G = grads->part[p];
        PartMlpDev pm = make_part_mlp(model, p, scene->latent_index);
        const int n_rgb = pm.rgb.n_linear;
        INVR_CHECK(G.row_grad && G.rgb_latent && G.occ_w[0] && G.occ_b[0] && G.occ_w[1] && G.occ_b[1] && G.rgb_w[0] && G.rgb_b[0] &&
                   G.rgb_w[n_rgb - 1] && G.rgb_b[n_rgb - 1], "invr_train_bwd: null gradient pointer (part %d)", p);
        INVR_CHECK(pm.occ.n_linear == 2 && pm.occ.dims[0] == 19 && pm.occ.dims[1] == 64 && pm.occ.dims[2] == 17 && (n_rgb == 2 || n_rgb == 3) &&
                   pm.rgb.dims[0] == 70 && pm.rgb.dims[1] == 64 && pm.rgb.dims[n_rgb] == 3 && pm.n_freq == 4 && pm.latent_dim == 8 && pm.geo_dim == 16,
                   "invr_train_bwd: supports occ 19-64-17 and rgb 70-64(-64)-3");
        const int32_t* count = w.counters + CNT_PAIRS + p;
        MlpBwdOut o{t.g_emb[p], t.gz[p], t.a[p], lcap, G.rgb_latent, 1};
        if (launch_part_mlp_bwd(pm, w.emb[p], w.l_d[p], lcap, lcap, count, reinterpret_cast<const float*>(t.g_raws), w.l_slot[p], p, o, sp)) return 1;
        // encoder^T first: the deformer stage below waits for every part's g_x, while the weight gradients are only read by the
        // optimizer — they follow on the part's stream and the caller's stream joins them at the END of this call, behind the
        // deformer stage (round 6; they sat between MLP^T and encoder^T: 100 - 350 us on the backward's critical chain)
        GridDev g = make_grid_dev(&model->part[p].grid);
        if (launch_part_encode_bwd_lists(g, w.l_x[p], t.g_emb[p], t.g_x[p], lcap, lcap, count, G.row_grad, sp)) return 1;
        if (ps) {
            INVR_HIP(hipEventRecord(ps->done[p], sp));
            INVR_HIP(hipStreamWaitEvent(st, ps->done[p], 0));
        }
    }
    // (a second pass: the library's streams share the device's few hardware queues, and a queue runs what it is handed in order —
    // a part's weight gradients submitted before the next part's MLP^T / encoder^T would sit in front of them)
    for (int p = 0; p < INVR_NUM_PARTS; ++p) {
        if (!(stages & INVR_BWD_PART(p))) continue;
        hipStream_t sp = ps ? ps->s[p] : st;
        const InvrPartGrads& G = grads->part[p];
        const int n_rgb = model->part[p].rgb.n_linear;
        float* dW[5] = {G.occ_w[0], G.occ_w[1], G.rgb_w[0], n_rgb == 3 ? G.rgb_w[1] : nullptr, G.rgb_w[n_rgb - 1]};
        float* db[5] = {G.occ_b[0], G.occ_b[1], G.rgb_b[0], n_rgb == 3 ? G.rgb_b[1] : nullptr, G.rgb_b[n_rgb - 1]};
        if (launch_part_wgrad(t.gz[p], t.a[p], lcap, n_rgb, dW, db, w.counters + CNT_PAIRS + p, sp)) return 1;
        if (ps) INVR_HIP(hipEventRecord(ps->done2[p], sp));
    }
    auto join_wgrads = [&]() -> int {
        if (!ps) return 0;
        for (int p = 0; p < INVR_NUM_PARTS; ++p)
            if (stages & INVR_BWD_PART(p)) INVR_HIP(hipStreamWaitEvent(st, ps->done2[p], 0));
        return 0;
    };
    // deformer^T over the listed pairs and the pair-regulariser neighbours (needs the g_x of every part)
    if (!(stages & INVR_BWD_DEFORMER)) return join_wgrads();
    if (check_grid(&model->deform_grid, "deformer grid") || check_mlp_deform(&model->deform_mlp)) return 1;
    INVR_CHECK(grads->deform_hash && (!model->deform_grid.separate_dense || grads->deform_dense) && grads->deform_w[0] && grads->deform_w[1] &&
               grads->deform_w[2] && grads->deform_b[0] && grads->deform_b[1] && grads->deform_b[2], "invr_train_bwd: null deformer gradient pointer");
    DeformGrads DG{{grads->deform_w[0], grads->deform_w[1], grads->deform_w[2]}, {grads->deform_b[0], grads->deform_b[1], grads->deform_b[2]},
                   grads->deform_dense, grads->deform_hash};
    RenderArgs a;
    memset(&a, 0, sizeof(a));
    a.scene = make_scene_dev(scene);
    a.R = n_rays; a.S = n_samples; a.N = N;
    // (the deformer's own weight gradients on a library stream beside its grid^T: the stream after the LAST part's, which is free
    // here or busy with the smallest part's weight gradients)
    PartStreams* ds = ps ? ps : part_streams();
    if (!ds) return 1;
    if (launch_deform_bwd(a, w, t, make_grid_dev(&model->deform_grid), make_mlp_dev(&model->deform_mlp), g_offset_sum, g_pair_sum, DG, st,
                          ds->s[INVR_NUM_PARTS - 1], ds->dfork, ds->djoin)) return 1;
    return join_wgrads();
}

__global__ void k_expand_row_grad(const float* __restrict__ rg, int64_t rows, int F, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * F) out[i] = rg[i / F];
}

extern "C" int invr_train_loss_fwd(const float* rgb_map, const float* rgb_gt, const float* dist, const float* terms, int64_t n_rays,
                                   float w_pair, float w_dist, float w_off, int32_t use_pair, float* out8, float* err, void* stream) {
    INVR_CHECK(rgb_map && rgb_gt && terms && out8 && n_rays >= 0, "invr_train_loss_fwd: null pointer");
    return launch_train_loss(rgb_map, rgb_gt, dist, terms, n_rays, w_pair, w_dist, w_off, use_pair, out8, err, (hipStream_t)stream);
}

extern "C" int invr_train_loss_bwd(const float* rgb_map, const float* rgb_gt, const float* terms, int64_t n_rays, float w_pair, float w_dist,
                                   float w_off, int32_t use_pair, const float* g_loss, float* g_rgb, float* g_dist, float* g_terms,
                                   void* stream) {
    INVR_CHECK(rgb_map && rgb_gt && terms && g_loss && g_rgb && g_terms && n_rays >= 0, "invr_train_loss_bwd: null pointer");
    return launch_train_loss_bwd(rgb_map, rgb_gt, terms, n_rays, w_pair, w_dist, w_off, use_pair, g_loss, g_rgb, g_dist, g_terms,
                                 (hipStream_t)stream);
}

extern "C" int invr_expand_row_grad(const InvrGrid* grid, const float* row_grad, float* g_dense, float* g_hash, void* stream) {
    INVR_CHECK(grid && row_grad && g_hash, "invr_expand_row_grad: null pointer");
    if (check_grid(grid, "grid")) return 1;
    INVR_CHECK(grid->sum && grid->sum_over_features, "invr_expand_row_grad: only sum && sum_over_features grids have row-scalar gradients");
    GridDev g = make_grid_dev(grid);
    hipStream_t st = (hipStream_t)stream;
    const int64_t hrows = (int64_t)(g.separate_dense ? g.L - g.start_hash : g.L) * g.T;
    if (g.separate_dense && g.dense_rows) {
        INVR_CHECK(g_dense, "invr_expand_row_grad: g_dense required");
        hipLaunchKernelGGL(k_expand_row_grad, dim3((unsigned)cdiv(g.dense_rows * g.F, 256)), dim3(256), 0, st, row_grad, g.dense_rows, g.F, g_dense);
        INVR_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_expand_row_grad, dim3((unsigned)cdiv(hrows * g.F, 256)), dim3(256), 0, st, row_grad + g.dense_rows, hrows, g.F, g_hash);
    INVR_LAUNCH_CHECK();
    return 0;
}
