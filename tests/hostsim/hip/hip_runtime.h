// TEST INFRASTRUCTURE — not part of the product.  A stand-in for <hip/hip_runtime.h> that lets tests/hostsim/build.py compile the
// kernel sources of instant-nvr_amd/csrc for the HOST and run them, wave by wave, on the CPU at toy sizes (tests/hostsim/README.md):
// every work-item is a fiber, the wave-level operations (ballot, shuffles, DPP, readlane, MFMA) are rendezvous points of the 64 fibers
// of a wave, __syncthreads is a rendezvous of the workgroup.  The product library (libinvr.so) is built by hipcc from the same
// sources and never sees this file; nothing under instant-nvr_amd/ loads the host build.
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <utility>

#define HOSTSIM 1

// ---- qualifiers -----------------------------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __constant__
#define __shared__ static
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)

// ---- vector types ---------------------------------------------------------------------------------------------------------------
struct uint3 { unsigned x, y, z; };
struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int3 { int x, y, z; };
struct alignas(16) int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

// ---- runtime API (synchronous: a launch runs to completion before it returns) ------------------------------------------------------
typedef int hipError_t;
#define hipSuccess 0
typedef struct hostsim_stream* hipStream_t;
typedef struct hostsim_event* hipEvent_t;
#define hipStreamNonBlocking 1
#define hipEventDisableTiming 2
#define hipFuncAttributeMaxDynamicSharedMemorySize 8
#define HIP_SYMBOL(x) x
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hostsim"; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
template <class T> static inline hipError_t hipFuncSetAttribute(T, int, int) { return hipSuccess; }
template <class T> static inline hipError_t hipMemcpyFromSymbol(void* dst, const T& sym, size_t n) { memcpy(dst, &sym, n); return hipSuccess; }
template <class T> static inline hipError_t hipMemcpyToSymbol(T& sym, const void* src, size_t n) { memcpy(&sym, src, n); return hipSuccess; }

// ---- the wave machine (hostsim_rt.cpp) ---------------------------------------------------------------------------------------------
namespace hostsim {
struct Ident { uint3 tid, bid; dim3 bdim, gdim; int lane; };
extern Ident* cur;                                    // the running work-item
enum Op { OP_BALLOT, OP_SHFL, OP_FIRST, OP_DPP, OP_MFMA16X16X4F32, OP_WAVE_BARRIER, OP_MFMA16X16X32BF16 };
struct Post {                                          // what a lane hands to a wave-level operation, and what it gets back
    int op, tag;
    const void* site;
    uint64_t a, b;
    int i0, i1, i2, i3;
    float f[6];
    const float* ext;                                  // wide operands (the bf16 MFMA: 8 + 8 values as floats), valid while the lane waits
    uint64_t res;
    float fres[4];
};
void rendezvous(Post& p);                             // blocks the calling fiber until the wave operation has been resolved
void barrier();                                       // __syncthreads
void* dyn_lds();
void launch_impl(dim3 grid, dim3 block, size_t shmem, void (*fn)(void*), void* arg);
template <class F> static void launch_thunk(void* p) { (*static_cast<F*>(p))(); }
template <class F> static inline void launch(dim3 grid, dim3 block, size_t shmem, F f) { launch_impl(grid, block, shmem, &launch_thunk<F>, &f); }

template <class T> static inline uint64_t bits_of(T v) { static_assert(sizeof(T) <= 8, ""); uint64_t u = 0; memcpy(&u, &v, sizeof(T)); return u; }
template <class T> static inline T from_bits(uint64_t u) { T v; memcpy(&v, &u, sizeof(T)); return v; }

__attribute__((noinline)) static uint64_t ballot_at(bool pred, int tag) {
    Post p; p.op = OP_BALLOT; p.site = __builtin_return_address(0); p.tag = tag; p.a = pred; rendezvous(p); return p.res;
}
__attribute__((noinline)) static uint64_t shfl_at(uint64_t v, int src, int tag) {
    Post p; p.op = OP_SHFL; p.site = __builtin_return_address(0); p.tag = tag; p.a = v; p.i0 = src; rendezvous(p); return p.res;
}
__attribute__((noinline)) static uint64_t first_at(uint64_t v, int tag) {
    Post p; p.op = OP_FIRST; p.site = __builtin_return_address(0); p.tag = tag; p.a = v; rendezvous(p); return p.res;
}
__attribute__((noinline)) static uint32_t dpp_at(uint32_t old, uint32_t src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl, int tag) {
    Post p; p.op = OP_DPP; p.site = __builtin_return_address(0); p.a = src; p.b = old; p.i0 = ctrl; p.i1 = row_mask; p.i2 = bank_mask;
    p.i3 = bound_ctrl; p.tag = tag; rendezvous(p); return (uint32_t)p.res;
}
__attribute__((noinline)) static void wave_barrier_at(int tag) {
    Post p; p.op = OP_WAVE_BARRIER; p.site = __builtin_return_address(0); p.tag = tag; rendezvous(p);
}
}  // namespace hostsim

#define threadIdx (hostsim::cur->tid)
#define blockIdx (hostsim::cur->bid)
#define blockDim (hostsim::cur->bdim)
#define gridDim (hostsim::cur->gdim)
#define warpSize 64

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hostsim::launch(dim3(grid), dim3(block), (size_t)(shmem), [=]() { (void)(stream); (kernel)(__VA_ARGS__); })

// every textual occurrence gets its own tag so that two occurrences can never be merged into one call site by the host compiler
#define __syncthreads() hostsim::barrier()
#define __ballot(p) hostsim::ballot_at((bool)(p), __COUNTER__)
template <class T> static inline __attribute__((always_inline)) T hostsim_shfl(T v, int src, int tag) { return hostsim::from_bits<T>(hostsim::shfl_at(hostsim::bits_of(v), src, tag)); }
#define __shfl(v, src, ...) hostsim_shfl((v), (int)(src) & 63, __COUNTER__)
#define __shfl_xor(v, m, ...) hostsim_shfl((v), hostsim::cur->lane ^ (int)(m), __COUNTER__)
#define __shfl_up(v, d, ...) hostsim_shfl((v), hostsim::cur->lane - (int)(d) >= 0 ? hostsim::cur->lane - (int)(d) : hostsim::cur->lane, __COUNTER__)
#define __shfl_down(v, d, ...) hostsim_shfl((v), hostsim::cur->lane + (int)(d) < 64 ? hostsim::cur->lane + (int)(d) : hostsim::cur->lane, __COUNTER__)
#define __builtin_amdgcn_readlane(v, l) hostsim_shfl((v), (int)(l) & 63, __COUNTER__)
template <class T> static inline __attribute__((always_inline)) T hostsim_first(T v, int tag) { return hostsim::from_bits<T>(hostsim::first_at(hostsim::bits_of(v), tag)); }
#define __builtin_amdgcn_readfirstlane(v) hostsim_first((v), __COUNTER__)
template <class T> static inline __attribute__((always_inline)) T hostsim_dpp(T old, T src, int ctrl, int rm, int bm, bool bc, int tag) {
    static_assert(sizeof(T) == 4, "DPP moves 32-bit registers");
    return hostsim::from_bits<T>(hostsim::dpp_at((uint32_t)hostsim::bits_of(old), (uint32_t)hostsim::bits_of(src), ctrl, rm, bm, bc, tag));
}
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) hostsim_dpp((old), (src), (ctrl), (rm), (bm), (bc), __COUNTER__)
#define __builtin_amdgcn_mov_dpp(src, ctrl, rm, bm, bc) hostsim_dpp((src), (src), (ctrl), (rm), (bm), (bc), __COUNTER__)

typedef float hostsim_v4f __attribute__((ext_vector_type(4)));
__attribute__((noinline)) static hostsim_v4f hostsim_mfma16x16x4(float a, float b, hostsim_v4f c, int tag) {
    hostsim::Post p; p.op = hostsim::OP_MFMA16X16X4F32; p.site = __builtin_return_address(0); p.tag = tag;
    p.f[0] = a; p.f[1] = b; p.f[2] = c[0]; p.f[3] = c[1]; p.f[4] = c[2]; p.f[5] = c[3];
    hostsim::rendezvous(p);
    return hostsim_v4f{p.fres[0], p.fres[1], p.fres[2], p.fres[3]};
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) hostsim_mfma16x16x4((a), (b), (c), __COUNTER__)
typedef __bf16 hostsim_v8bf __attribute__((ext_vector_type(8)));
__attribute__((noinline)) static hostsim_v4f hostsim_mfma16x16x32bf16(hostsim_v8bf a, hostsim_v8bf b, hostsim_v4f c, int tag) {
    float ab[16];
    for (int k = 0; k < 8; ++k) { ab[k] = (float)a[k]; ab[8 + k] = (float)b[k]; }
    hostsim::Post p; p.op = hostsim::OP_MFMA16X16X32BF16; p.site = __builtin_return_address(0); p.tag = tag;
    p.ext = ab; p.f[2] = c[0]; p.f[3] = c[1]; p.f[4] = c[2]; p.f[5] = c[3];
    hostsim::rendezvous(p);
    return hostsim_v4f{p.fres[0], p.fres[1], p.fres[2], p.fres[3]};
}
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) hostsim_mfma16x16x32bf16((a), (b), (c), __COUNTER__)
// lanes of a wave that exchange data through memory without any other wave-level operation in between say so with a wave barrier
// (no instruction on the device); here it is where the lanes — fibers that otherwise run one after the other — meet
#define __builtin_amdgcn_wave_barrier() hostsim::wave_barrier_at(__COUNTER__)
#define __builtin_amdgcn_sched_group_barrier(...) ((void)0)
#define __builtin_amdgcn_sched_barrier(...) ((void)0)

// ---- transcendental pipes: libm values (the pipes are ~1 ulp; every consumer is tolerance-tested) ----------------------------------
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_logf(x) log2f(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_sinf(x) ((float)sin(6.283185307179586 * (double)(x)))
#define __builtin_amdgcn_cosf(x) ((float)cos(6.283185307179586 * (double)(x)))

// ---- scalar intrinsics ----------------------------------------------------------------------------------------------------------
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
static inline double __longlong_as_double(long long i) { double f; memcpy(&f, &i, 8); return f; }
static inline long long __double_as_longlong(double f) { long long i; memcpy(&i, &f, 8); return i; }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
static inline long long clock64() { static long long t = 0; return t += 64; }

#define HOSTSIM_MINMAX(T) \
    static inline T min(T a, T b) { return b < a ? b : a; } \
    static inline T max(T a, T b) { return a < b ? b : a; }
HOSTSIM_MINMAX(int) HOSTSIM_MINMAX(unsigned) HOSTSIM_MINMAX(long) HOSTSIM_MINMAX(unsigned long) HOSTSIM_MINMAX(long long)
HOSTSIM_MINMAX(unsigned long long)
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline long min(int a, long b) { return a < b ? a : b; }
static inline long min(long a, int b) { return a < b ? a : b; }
static inline long max(int a, long b) { return a > b ? a : b; }
static inline long max(long a, int b) { return a > b ? a : b; }

// ---- atomics: fibers are cooperative, a read-modify-write cannot be interrupted -------------------------------------------------------
template <class T, class U> static inline T atomicAdd(T* p, U v) { T o = *p; *p = (T)(o + (T)v); return o; }
template <class T, class U> static inline T atomicMax(T* p, U v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T, class U> static inline T atomicMin(T* p, U v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <class T, class U> static inline T atomicOr(T* p, U v) { T o = *p; *p = (T)(o | (T)v); return o; }
template <class T, class U> static inline T atomicAnd(T* p, U v) { T o = *p; *p = (T)(o & (T)v); return o; }
template <class T, class U> static inline T atomicExch(T* p, U v) { T o = *p; *p = (T)v; return o; }
template <class T, class U, class V> static inline T atomicCAS(T* p, U cmp, V v) { T o = *p; if (o == (T)cmp) *p = (T)v; return o; }
template <class T, class U> static inline T unsafeAtomicAdd(T* p, U v) { return atomicAdd(p, v); }
