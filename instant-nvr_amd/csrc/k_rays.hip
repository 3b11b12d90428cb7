// Row f2: ray generation on the device — get_rays + get_near_far of the reference's dataset code
// (lib/utils/if_nerf/if_nerf_data_utils.py:24-38, 92-107, 313-327), which runs in NumPy on the host
// per frame: pixel -> camera -> world in float64, normalise, cast to float32, then the ray / AABB
// slab test in float32.  One thread per pixel; full-frame outputs + the mask_at_box byte mask (the
// caller compacts).  inv(K), R, T and the camera centre are tiny host-side float64 values (the
// Python wrapper forms them with NumPy exactly as the reference does).
#include "common.h"

struct RayCam {
    double kinv[9], r[9], t[3], o[3];
    float bounds[6];
};

__global__ void k_generate_rays(RayCam c, int H, int W, float* __restrict__ ray_d, float* __restrict__ near,
                                float* __restrict__ far, uint8_t* __restrict__ mask) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)H * W) return;
    const double i = (double)(float)(idx % W), j = (double)(float)(idx / W);     // meshgrid(arange(W), arange(H)) float32
    double pc[3], pw[3], d[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) pc[a] = i * c.kinv[a * 3] + j * c.kinv[a * 3 + 1] + c.kinv[a * 3 + 2];   // xy1 @ inv(K).T
#pragma unroll
    for (int b = 0; b < 3; ++b)
        pw[b] = (pc[0] - c.t[0]) * c.r[b] + (pc[1] - c.t[1]) * c.r[3 + b] + (pc[2] - c.t[2]) * c.r[6 + b];   // (pc - T) @ R
#pragma unroll
    for (int a = 0; a < 3; ++a) d[a] = pw[a] - c.o[a];
    const double nrm = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    float rd[3], ro[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { rd[a] = (float)(d[a] / nrm); ro[a] = (float)c.o[a]; ray_d[idx * 3 + a] = rd[a]; }
    // get_near_far in float32
    const float norm_d = sqrtf(rd[0] * rd[0] + rd[1] * rd[1] + rd[2] * rd[2]);
    float tn = -__builtin_inff(), tf = __builtin_inff();
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float v = rd[a] / norm_d;
        if (v < 1e-5f && v > -1e-10f) v = 1e-5f;
        if (v > -1e-5f && v < 1e-10f) v = -1e-5f;
        const float t0 = (c.bounds[a] - ro[a]) / v, t1 = (c.bounds[3 + a] - ro[a]) / v;
        tn = fmaxf(tn, fminf(t0, t1));
        tf = fminf(tf, fmaxf(t0, t1));
    }
    const bool m = tn < tf;
    mask[idx] = m ? 1 : 0;
    near[idx] = tn / norm_d;
    far[idx] = tf / norm_d;
}

int launch_generate_rays(const double* kinv, const double* r, const double* t, const double* o, const float* bounds,
                         int H, int W, float* ray_d, float* near, float* far, uint8_t* mask, hipStream_t st) {
    RayCam c;
    for (int k = 0; k < 9; ++k) { c.kinv[k] = kinv[k]; c.r[k] = r[k]; }
    for (int k = 0; k < 3; ++k) { c.t[k] = t[k]; c.o[k] = o[k]; }
    for (int k = 0; k < 6; ++k) c.bounds[k] = bounds[k];
    const int64_t n = (int64_t)H * W;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_generate_rays, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, c, H, W, ray_d, near, far, mask);
    INVR_LAUNCH_CHECK();
    return 0;
}
