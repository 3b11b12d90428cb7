// Shared device helpers and internal launch declarations for libinvr (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/invr.h"

#define INVR_WAVE 64
// The wave scans (wave_incl_sum_i, wave_or_u32, k_composite.hip's products) use the GFX9 DPP controls wave_shr:1 / row_bcast:15 / row_bcast:31
// and 64-lane waves with every lane active; the MFMA / LDS tilings assume CDNA4.  There is no other target and no fallback path.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libinvr is written for gfx950 (MI355X) only: build with --offload-arch=gfx950"
#endif
#define HASH_P1 19349663ull
#define HASH_P2 83492791ull

// ---- error plumbing (host) -----------------------------------------------------------------
void invr_set_error(const char* fmt, ...);
#define INVR_CHECK(cond, ...)                \
    do {                                     \
        if (!(cond)) {                       \
            invr_set_error(__VA_ARGS__);     \
            return 1;                        \
        }                                    \
    } while (0)
#define INVR_HIP(call)                                                              \
    do {                                                                            \
        hipError_t e_ = (call);                                                     \
        if (e_ != hipSuccess) {                                                     \
            invr_set_error("%s failed: %s", #call, hipGetErrorString(e_));          \
            return 1;                                                               \
        }                                                                           \
    } while (0)
#define INVR_LAUNCH_CHECK()                                                         \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) {                                                     \
            invr_set_error("kernel launch failed (%s:%d): %s", __FILE__, __LINE__,  \
                           hipGetErrorString(e_));                                  \
            return 1;                                                               \
        }                                                                           \
    } while (0)

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- device-side parameter blocks (passed by value as kernel arguments) ----------------------
struct GridDev {
    const float* dense;
    const float* hash;
    const float* bounds;
    int32_t L, F, start_hash, separate_dense;
    int64_t T;
    double inv_T;
    // T = 2^mod_k + mod_c with a small mod_c (nextprime(2^k)): 32-bit reduction is valid (host-checked)
    int32_t mod_k, mod32;
    uint32_t mod_c;
    int32_t xdelta;             // mod24 and T > 2^14: the c1x corners' rows follow from the c0x corners' by a +-delta fold (k_encode.hip)
    int32_t mod1r;              // xdelta and c * (max hash key >> k) < 2 T for THIS grid's resolutions: one folding round + two fix-ups (hash_mod24_1r)
    int32_t mod24;              // mod32 and every multiplicand of its folding rounds < 2^24: v_mul_u32_u24 (full rate) instead of
                                // v_mul_lo_u32 (quarter rate)
    int32_t res[INVR_MAX_LEVELS];
    float cell[INVR_MAX_LEVELS];
    float rcell[INVR_MAX_LEVELS];   // RN(1 / cell[l]) when the level qualifies for the reciprocal form of x / cell (level_corners), else 0
    int64_t dense_off[INVR_MAX_LEVELS];
    int32_t sum, sum_over_features, include_input;
    const float* row_sums;      // optional inference-only (rows,) table of per-row feature sums
    int64_t dense_rows;         // rows of `dense` (0 when !separate_dense)
};

struct VolDev {          // (Dx,Dy,Dz,C) volume + (2,3) bounds
    const float* data;
    const float* bounds;
    int32_t dx, dy, dz, c;
};

struct SceneDev {
    const float* R;
    const float* Th;
    const float* A;
    const float* big_A;
    VolDev pbw;
    VolDev tuv;
    const float* part_pts;
    const float* part_pbw;
    const int64_t* lengths2;
    int32_t M;
    const float* frame_dim;
    const int64_t* latent_index;
    float thresh;
    int32_t tpose_viewdir;
    // squared nearest-vertex distances between which a (point,part) pair is provably NOT flagged
    // (k_knn.hip header); near_hi2 = +inf disables the class
    float near_hi2, band_lo2;
    float comp_eps;              // epsilon of render_weights (InvrScene::composite_eps): 0, or 1 with cfg.random_bg
    int aggr;                    // InvrScene::aggr: 0 = max-occupancy merge, 1 = mean over the parts
};

struct MlpDev {
    const float* w[INVR_MAX_LINEAR];
    const float* b[INVR_MAX_LINEAR];
    int32_t dims[INVR_MAX_LINEAR + 1];
    int32_t n_linear;
};

GridDev make_grid_dev(const InvrGrid* g);
SceneDev make_scene_dev(const InvrScene* s);
MlpDev make_mlp_dev(const InvrMlp* m);

// ---- device math ------------------------------------------------------------------------------
// Raw transcendental pipes (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp, no denormal fix-up code).
#define INVR_LOG2E 1.4426950408889634f
#define INVR_LN2 0.6931471805599453f
__device__ __forceinline__ float exp2_raw(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float log2_raw(float x) { return __builtin_amdgcn_logf(x); }

// torch.nn.Softplus(beta=1, threshold=20): x > 20 ? x : log1p(exp(x)).
// Evaluated as max(x,0) + ln2*log2(1 + exp2(-|x| log2e)): branch-free, 6 VALU ops, absolute error
// < 1.5e-7 (1+e rounds at 6e-8; the reference keeps relative accuracy for very negative x, the
// path only needs 1e-4 absolute at the pixel).  For x > 20 the correction is < 2.1e-9 < ulp(20)/2,
// so the result is x exactly, as with the reference's threshold.
__device__ __forceinline__ float softplus_f(float x) {
    const float e = exp2_raw(-fabsf(x) * INVR_LOG2E);
    return fmaf(log2_raw(1.0f + e), INVR_LN2, fmaxf(x, 0.0f));
}
// 1 - exp(-s), s >= 0
__device__ __forceinline__ float one_minus_exp_neg(float s) { return 1.0f - exp2_raw(-s * INVR_LOG2E); }
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + exp2_raw(-x * INVR_LOG2E)); }
// sigmoid(x) with ~3 ulp RELATIVE accuracy for every x (the backward kernels' softplus'(z)): e = exp(-|x|) with the rounding of
// |x| log2e carried into a first-order correction (plain exp2(x log2e) is off by |x| 1e-7 relative), then 1 / (1 + e) or e / (1 + e)
__device__ __forceinline__ float sigmoid_acc(float x) {
    const float ax = fabsf(x);
    const float hi = -ax * INVR_LOG2E;
    const float lo = fmaf(-ax, INVR_LOG2E, -hi) + -ax * 1.925963033500011e-8f;      // log2e - float(log2e) = 1.9259630335e-8
    const float e = exp2_raw(hi) * fmaf(lo, INVR_LN2, 1.0f);
    return (x >= 0.0f ? 1.0f : e) * __builtin_amdgcn_rcpf(1.0f + e);
}

// sin and cos of a moderate argument (|a| < ~100): Cody-Waite reduction to [-pi/4, pi/4] by
// multiples of pi/2 (two-term constant, FMA), Cephes sinf/cosf minimax polynomials; |error| < 2e-7.
__device__ __forceinline__ void sincos_f(float a, float* sn, float* cs) {
    const float k = rintf(a * 0.63661977236758134f);
    float r = fmaf(-k, 1.5707962512969971f, a);        // pi/2 high part
    r = fmaf(-k, 7.5497894158615964e-8f, r);           // pi/2 low part
    const float z = r * r;
    const float s = fmaf(r * z, fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
    const float c = fmaf(z * z, fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f),
                         fmaf(-0.5f, z, 1.0f));
    const int q = (int)k;
    const float ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
    *sn = (q & 2) ? -ss : ss;
    *cs = ((q + 1) & 2) ? -cc : cc;
}

// sin and cos on the hardware pipes (v_sin_f32 / v_cos_f32 take revolutions): 3 issue slots instead of ~20.  The scaling
// multiply costs up to |a| * 1e-8 revolutions = 6e-8 |a| rad, the pipes ~1e-6 absolute: |error| <= 2e-6 for the view-direction
// encoding's arguments (|a| <= 8 |d|) — used where the value only feeds an MLP input (weights O(0.1), pixel budget 1e-4).
__device__ __forceinline__ void sincos_hw(float a, float* sn, float* cs) {
    const float r = a * 0.15915494309189535f;
    *sn = __builtin_amdgcn_sinf(r);
    *cs = __builtin_amdgcn_cosf(r);
}

// torch.linspace(0,1,S)[i] in float32 (symmetric fill used by ATen's CPU/GPU kernels)
__device__ __forceinline__ float linspace01(int i, int S) {
    float step = 1.0f / (float)(S - 1);
    return (i < S / 2) ? step * (float)i : 1.0f - step * (float)(S - 1 - i);
}

// Trilinear sample of channel block [c0,c0+NC) of a (Dx,Dy,Dz,C) volume at a pose/canonical
// point; F.grid_sample(mode=bilinear, padding_mode=border, align_corners=True) semantics with the
// reference's xyz->zyx flip (blend_utils.py:501-555): axis a index = ((u_a+1)/2)*(size_a-1),
// clamped to [0,size_a-1] before corner selection.
// SMALL (the caller's host side has checked volume_is_small(): dx*dy*dz*c <= 2^24): every corner offset is formed with full-rate 24-bit
// multiplies and 32-bit adds on per-axis strides — the 64-bit form costs ~50 quarter-rate v_mul_lo_u32 / v_mad_u64_u32 per point.
// Same corners, same weights, same accumulation order: same bits.
template <int NC, bool SMALL = false>
__device__ __forceinline__ void sample_volume_dev(const VolDev& v, int c0, float px, float py, float pz, float* out) {
    const float b0x = v.bounds[0], b0y = v.bounds[1], b0z = v.bounds[2];
    const float b1x = v.bounds[3], b1y = v.bounds[4], b1z = v.bounds[5];
    float gx = (px - b0x) / (b1x - b0x) * 2.0f - 1.0f;
    float gy = (py - b0y) / (b1y - b0y) * 2.0f - 1.0f;
    float gz = (pz - b0z) / (b1z - b0z) * 2.0f - 1.0f;
    float ix = ((gx + 1.0f) * 0.5f) * (float)(v.dx - 1);          // /2 == *0.5 exactly in binary fp
    float iy = ((gy + 1.0f) * 0.5f) * (float)(v.dy - 1);
    float iz = ((gz + 1.0f) * 0.5f) * (float)(v.dz - 1);
    ix = fminf(fmaxf(ix, 0.0f), (float)(v.dx - 1));
    iy = fminf(fmaxf(iy, 0.0f), (float)(v.dy - 1));
    iz = fminf(fmaxf(iz, 0.0f), (float)(v.dz - 1));
    float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    float tx = ix - fx, ty = iy - fy, tz = iz - fz;
    int x1 = min(x0 + 1, v.dx - 1), y1 = min(y0 + 1, v.dy - 1), z1 = min(z0 + 1, v.dz - 1);
#pragma unroll
    for (int c = 0; c < NC; ++c) out[c] = 0.0f;
    if (SMALL) {
        const unsigned sz = (unsigned)v.c, sy = (unsigned)v.dz * sz, sx = (unsigned)v.dy * sy;      // wave-uniform strides
        const unsigned ox[2] = {(unsigned)__umul24((unsigned)x0, sx), (unsigned)__umul24((unsigned)x1, sx)};
        const unsigned oy[2] = {(unsigned)__umul24((unsigned)y0, sy), (unsigned)__umul24((unsigned)y1, sy)};
        const unsigned oz[2] = {(unsigned)__umul24((unsigned)z0, sz) + (unsigned)c0, (unsigned)__umul24((unsigned)z1, sz) + (unsigned)c0};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float w = ((k & 4) ? tx : 1.0f - tx) * ((k & 2) ? ty : 1.0f - ty) * ((k & 1) ? tz : 1.0f - tz);
            const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(v.data) + ((ox[k >> 2] + oy[(k >> 1) & 1] + oz[k & 1]) << 2));
#pragma unroll
            for (int c = 0; c < NC; ++c) out[c] = fmaf(w, p[c], out[c]);
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        int xx = (k & 4) ? x1 : x0, yy = (k & 2) ? y1 : y0, zz = (k & 1) ? z1 : z0;
        float w = ((k & 4) ? tx : 1.0f - tx) * ((k & 2) ? ty : 1.0f - ty) * ((k & 1) ? tz : 1.0f - tz);
        const float* p = v.data + (((int64_t)xx * v.dy + yy) * v.dz + zz) * v.c + c0;
#pragma unroll
        for (int c = 0; c < NC; ++c) out[c] = fmaf(w, p[c], out[c]);
    }
}
static inline bool volume_is_small(const VolDev& v) { return (int64_t)v.dx * v.dy * v.dz * v.c <= (int64_t)1 << 24; }

// (a ^ b*P1 ^ c*P2) mod T, exact for T < 2^31 and products < 2^52 (int64 arithmetic of
// part_base_embedder.py:132-136) using one fp64 reciprocal multiply + correction.
__device__ __forceinline__ uint32_t hash_mod64(uint64_t x, int64_t T, double inv_T) {
    double xd = (double)x;
    double q = floor(xd * inv_T);
    double r = fma(-q, (double)T, xd);
    if (r < 0.0) r += (double)T;
    if (r >= (double)T) r -= (double)T;
    return (uint32_t)r;
}
// x mod T for T = 2^k + c (c small, x < 2^41) with 32-bit integer ops only: 2^k == -c (mod T), so
// x = h*2^k + a == a - c*h; three folding rounds bring the value into (-2T, 2T).  The bounds that make
// three rounds sufficient are verified on the host (make_grid_dev); otherwise the fp64 path is used.
__device__ __forceinline__ uint32_t hash_mod32(uint64_t x, int k, uint32_t c, uint32_t T) {
    const uint32_t mask = (1u << k) - 1u, lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    const uint32_t a0 = lo & mask, h0 = (hi << (32 - k)) | (lo >> k);
    const uint32_t y1 = c * h0, a1 = y1 & mask, h1 = y1 >> k;
    const uint32_t y2 = c * h1, a2 = y2 & mask, h2 = y2 >> k;
    const uint32_t y3 = c * h2;
    int32_t v = (int32_t)(a0 + a2) - (int32_t)(a1 + y3);
    v += (v < 0) ? (int32_t)T : 0;
    v += (v < 0) ? (int32_t)T : 0;
    v -= (v >= (int32_t)T) ? (int32_t)T : 0;
    return (uint32_t)v;
}
struct GridDev;
__device__ __forceinline__ uint32_t grid_hash_mod(uint64_t x, const GridDev& g);

__device__ __forceinline__ uint32_t hash_mod(uint32_t cx, uint32_t cy, uint32_t cz, int64_t T, double inv_T) {
    uint64_t x = (uint64_t)cx ^ ((uint64_t)cy * HASH_P1) ^ ((uint64_t)cz * HASH_P2);
    double xd = (double)x;
    double q = floor(xd * inv_T);
    double r = fma(-q, (double)T, xd);
    if (r < 0.0) r += (double)T;
    if (r >= (double)T) r -= (double)T;
    return (uint32_t)r;
}

// Normalised coordinate -> per-axis clipped corners and fractional offsets of one level
// (part_base_embedder.py:115-118): f = x / cell; c0 = clip(trunc(f)), c1 = clip(trunc(f + 1));
// t = f - c0 (may leave [0,1] outside the box -> extrapolation).
// x / b through the correctly rounded reciprocal y = RN(1 / b) (host-computed): q0 = RN(x y); two Markstein steps
// r = fma(-q, b, x), q = fma(r, y, q).  The first makes q faithful, the second then yields RN(x / b) — the IEEE quotient, bit for
// bit (Markstein 1990; excluded on the host: a divisor whose mantissa is all ones; no over/underflow: |x / b| < 2^40 here and an
// |x| below 2^-60 takes the hardware division) — in 5 full-rate instructions instead of the ~11 of the v_div_scale / v_rcp /
// v_div_fmas / v_div_fixup sequence.  The encoders spend 3 divisions per level and pair on it.
__device__ __forceinline__ float div_by_rcp(float x, float b, float y) {
    float q = x * y;
    float r = fmaf(-q, b, x);
    q = fmaf(r, y, q);
    r = fmaf(-q, b, x);
    return fmaf(r, y, q);
}

// x / b, bit for bit: the reciprocal form where it is proven (y = RN(1 / b) != 0 handed in, x in the safe range), else the hardware
// division
__device__ __forceinline__ float div_exact(float x, float b, float y) {
    const bool ok = y != 0.0f && fabsf(x) > 8.7e-19f && fabsf(x) < 1.0e6f;
    // a WAVE-uniform branch: written as a select the compiler evaluates both forms for every lane (measured: slower than the
    // plain division); the hardware division only runs when some lane of the wave needs it
    if (__ballot(!ok) == 0ull) return div_by_rcp(x, b, y);
    return ok ? div_by_rcp(x, b, y) : x / b;
}
// y for div_exact from a divisor only known on the device: 1.0f / b is the correctly rounded reciprocal (IEEE division); 0 = do not
// use the reciprocal form (Markstein's exception: a mantissa of all ones; divisors outside [2^-20, 2^20])
__device__ __forceinline__ float rcp_for_div(float b) {
    const bool ok = fabsf(b) > 9.6e-7f && fabsf(b) < 1.0e6f && (__float_as_uint(b) & 0x7fffffu) != 0x7fffffu;
    return ok ? 1.0f / b : 0.0f;
}

__device__ __forceinline__ void level_corners(float x, float cell, float rcell, int res, int& c0, int& c1, float& t) {
    float f = div_exact(x, cell, rcell);
    int a = (int)f;              // v_cvt_i32_f32: truncates toward zero (torch .long())
    int b = (int)(f + 1.0f);
    c0 = min(max(a, 0), res - 1);
    c1 = min(max(b, 0), res - 1);
    t = f - (float)c0;
}

__device__ __forceinline__ void level_corners(float x, float cell, int res, int& c0, int& c1, float& t) {
    float f = x / cell;
    int a = (int)f;              // v_cvt_i32_f32: truncates toward zero (torch .long())
    int b = (int)(f + 1.0f);
    c0 = min(max(a, 0), res - 1);
    c1 = min(max(b, 0), res - 1);
    t = f - (float)c0;
}

__device__ __forceinline__ uint32_t grid_hash_mod(uint64_t x, const GridDev& g) {
    return g.mod32 ? hash_mod32(x, g.mod_k, g.mod_c, (uint32_t)g.T) : hash_mod64(x, g.T, g.inv_T);
}
// hash_mod32 with 24-bit multiplies (host-checked: c, h0, h1, h2 < 2^24): the same integer arithmetic, exact
__device__ __forceinline__ uint32_t hash_mod24(uint64_t x, int k, uint32_t c, uint32_t T) {
    const uint32_t mask = (1u << k) - 1u, lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    const uint32_t a0 = lo & mask, h0 = (hi << (32 - k)) | (lo >> k);
    const uint32_t y1 = __umul24(c, h0), a1 = y1 & mask, h1 = y1 >> k;
    const uint32_t y2 = __umul24(c, h1), a2 = y2 & mask, h2 = y2 >> k;
    const uint32_t y3 = __umul24(c, h2);
    int32_t v = (int32_t)(a0 + a2) - (int32_t)(a1 + y3);
    v += (v < 0) ? (int32_t)T : 0;
    v += (v < 0) ? (int32_t)T : 0;
    v -= (v >= (int32_t)T) ? (int32_t)T : 0;
    return (uint32_t)v;
}

// hash_mod24 when c * h1 < 2^k (host-checked: GridDev.xdelta implies it; T = 2^20 + 7: h1 <= 14): h2 = 0, the third round adds nothing —
// x == a0 - a1 + c h1 (mod T) lies in (-2^k, 2^k + c h1), one fix-up each way.  Same value as hash_mod24, 7 instructions fewer.
__device__ __forceinline__ uint32_t hash_mod24_2r(uint64_t x, int k, uint32_t c, uint32_t T) {
    const uint32_t mask = (1u << k) - 1u, lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    const uint32_t a0 = lo & mask, h0 = (hi << (32 - k)) | (lo >> k);
    const uint32_t y1 = __umul24(c, h0), a1 = y1 & mask, h1 = y1 >> k;
    uint32_t v = (a0 + __umul24(c, h1)) - a1;       // in (-2^k, 2^k + c h1) as a signed value: the two fix-ups as unsigned minima
    v = min(v, v + T);                                // (a negative value is a huge unsigned one: v + T wraps to the small result)
    v = min(v, v - T);
    return v;
}

// ONE folding round: x = h0 2^k + a0 == a0 - c h0 (mod T = 2^k + c).  Valid when c h0 < 2 T for every key the grid can form (host-checked
// from its largest resolution: GridDev.mod1r — T = 2^20 + 7 with keys < 2^38: c h0 < 1.84 M < 2 T): a0 - c h0 + 2 T lies in (0, 3 T), two
// fix-ups written as unsigned minima (v - T wraps above v when v < T).  Same value as hash_mod24 / _2r, 5 instructions fewer than _2r.
__device__ __forceinline__ uint32_t hash_mod24_1r(uint64_t x, int k, uint32_t c, uint32_t T) {
    const uint32_t mask = (1u << k) - 1u, lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    const uint32_t a0 = lo & mask, h0 = (hi << (32 - k)) | (lo >> k);
    uint32_t v = a0 + (2u * T - __umul24(c, h0));
    v = min(v, v - T);
    v = min(v, v - T);
    return v;
}

// ---- wave scans on the DPP network ------------------------------------------------------------------------------------------
// Inclusive prefix sum over the 64 lanes: row_shr 1 / 2 / 4 / 8 inside the 16-lane rows, then row_bcast:15 / row_bcast:31 across
// rows (the sequence LLVM's own wave scans use on gfx9) — six full-rate VALU instructions.  __shfl_up compiles to ds_bpermute_b32:
// an LDS round trip per step, which is what the short list kernels (dependent chains of a few dozen steps) spent their time on.
// ALL 64 lanes must be active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int identity, int x) {
    return __builtin_amdgcn_update_dpp(identity, x, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ int wave_incl_sum_i(int x) {
    x += dpp_i<0x111, 0xF>(0, x);
    x += dpp_i<0x112, 0xF>(0, x);
    x += dpp_i<0x114, 0xF>(0, x);
    x += dpp_i<0x118, 0xF>(0, x);
    x += dpp_i<0x142, 0xA>(0, x);
    x += dpp_i<0x143, 0xC>(0, x);
    return x;
}

// OR over the 64 lanes (wave-uniform result) on the same network: the partial ORs travel down the rows and across them, lane 63 holds
// the total.  ALL 64 lanes must be active.  (The __shfl_xor butterfly is six dependent ds_bpermute round trips per word.)
__device__ __forceinline__ unsigned wave_or_u32(unsigned v) {
    int x = (int)v;
    x |= dpp_i<0x111, 0xF>(0, x);
    x |= dpp_i<0x112, 0xF>(0, x);
    x |= dpp_i<0x114, 0xF>(0, x);
    x |= dpp_i<0x118, 0xF>(0, x);
    x |= dpp_i<0x142, 0xA>(0, x);
    x |= dpp_i<0x143, 0xC>(0, x);
    return (unsigned)__builtin_amdgcn_readlane(x, 63);
}

// ---- kernel launchers (defined in the .hip files) ---------------------------------------------
int launch_grid_encode_generic(const GridDev& g, const float* xyz, int64_t n, float* out, hipStream_t st);
int launch_grid_encode_bwd_generic(const GridDev& g, const float* xyz, const float* gout, int64_t n, float* g_dense,
                                   float* g_hash, float* g_xyz, hipStream_t st, const int32_t* count = nullptr);
int launch_part_encode_bwd_lists(const GridDev& g, const float* x_soa, const float* gout_soa, float* gx_soa, int64_t stride,
                                 int64_t n_max, const int32_t* count, float* rowgrad, hipStream_t st);
int launch_composite_bwd(const float* raw, const float* g_rgb, const float* g_acc, const float* g_w, int64_t n_rays, int S, float eps,
                         float* g_raw, hipStream_t st);
int launch_sample_volume(const VolDev& v, int c0, int nc, const float* pts, int64_t n, float* out, hipStream_t st);
int launch_knn_blend_dense(const SceneDev& s, const float* pose_pts, int64_t n, float* bw, float* dist, int32_t* nn, float* d2,
                           float* w, hipStream_t st);
int launch_warp_deform_dense(const SceneDev& s, const GridDev& dg, const MlpDev& dm, const float* pose_pts,
                             const float* pose_dirs, const float* bw, const uint8_t* flag, int64_t n,
                             float* tpose, float* tdirs, float* resd, hipStream_t st);
int launch_deform_points(const SceneDev& s, const GridDev& dg, const MlpDev& dm, const float* pts, int64_t n, float* out, hipStream_t st);
int launch_distortion(const float* weights, const float* z, int64_t n_rays, int S, float* out, hipStream_t st);
int launch_generate_rays(const double* kinv, const double* r, const double* t, const double* o, const float* bounds,
                         int H, int W, float* ray_d, float* near, float* far, uint8_t* mask, hipStream_t st);
int launch_adam_advance(void* tensors, int n, double b1, double b2, hipStream_t st);
int launch_adam(const void* tensors, const int32_t* chunk_tensor, const int32_t* chunk_index, int64_t n_chunks, double b1, double b2,
                float eps, hipStream_t st);
int launch_row_sums(const GridDev& g, float* out, hipStream_t st);
int launch_rigid_transformation(const double* poses, const double* joints, const int32_t* parents, float* A, hipStream_t st);
int launch_pack_parts(const float* ppts, const float* weights, const int64_t* parts, const float* tpose, int n_verts, int n_w,
                      int stride, float overlap, float* part_pts, float* part_pbw, int64_t* lengths2, float* bounds, hipStream_t st);
int launch_composite(const float* raw, int64_t n_rays, int S, float eps, float* weights, float* rgb_map, float* acc_map, hipStream_t st);
