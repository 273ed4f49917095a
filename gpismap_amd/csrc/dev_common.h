// gpismap_amd -- common device/host helpers for the HIP hot path (gfx950 only).
//
// Numerical contract shared by every kernel in this directory:
//  * kernel-function values follow the reference's C++ conversion rules: the
//    exponential is evaluated in double and the product rounded once to float
//    (reference cpp/src/covFnc.cpp:29-33), distances are plain float
//    mul/add/sqrt with no contraction;
//  * every Cholesky / substitution element is ONE fmaf chain over ascending k
//    (backward substitution: descending k), divisions and square roots are IEEE
//    correctly rounded.  A blocked or MFMA implementation that keeps the chain
//    order is bit-identical to the unblocked one (v_mfma_f32_32x32x2_f32 is a
//    k-ordered fmaf chain).
// The translation units are compiled with -ffp-contract=off so only explicit
// fmaf()/MFMA fuse.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define GPIS_OK 0
#define GPIS_ERR_ARG (-1)
#define GPIS_ERR_HIP (-2)
#define GPIS_ERR_STATE (-3)
#define GPIS_ERR_LIMIT (-4)

#define GPIS_HIP(call)                                                                   \
    do {                                                                                 \
        hipError_t e__ = (call);                                                         \
        if (e__ != hipSuccess) {                                                         \
            fprintf(stderr, "[gpismap_amd] HIP error %s at %s:%d (%s)\n",               \
                    hipGetErrorString(e__), __FILE__, __LINE__, #call);                  \
            return GPIS_ERR_HIP;                                                         \
        }                                                                                \
    } while (0)

namespace gpis {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---- reference-exact scalar kernels (device) --------------------------------
__device__ __forceinline__ float d_ou_k(float r, float a) { return (float)exp((double)(-a * r)); }

// Matern-3/2 pieces given e = exp((double)(-a*r)) (covFnc.cpp:29-33)
__device__ __forceinline__ float d_kf(float r, float a, double e) { return (float)((1.0 + (double)(a * r)) * e); }
__device__ __forceinline__ float d_kf1(float dx, float a, double e) { return (float)((double)(a * a * dx) * e); }
__device__ __forceinline__ float d_kf2(float r, float dx1, float dx2, float delta, float a, double e) {
    return (float)((double)(a * a * (delta - a * dx1 * dx2 / r)) * e);
}

__device__ __forceinline__ float d_dist2(float ax, float ay, float bx, float by) {
    float tx = ax - bx, ty = ay - by;
    return sqrtf(tx * tx + ty * ty);
}
__device__ __forceinline__ float d_dist3(float ax, float ay, float az, float bx, float by, float bz) {
    float tx = ax - bx, ty = ay - by, tz = az - bz;
    return sqrtf((tx * tx + ty * ty) + tz * tz);
}

// Every map object lives on ONE device: the device current in the creating thread (gpis_set_device, or the
// caller's own hipSetDevice) at construction.  Each public entry point makes that device current for its
// duration and restores the caller's afterwards, so one process can drive maps on several GPUs and a rank of
// a multi-GPU job never lands on device 0 by accident.
struct DeviceScope {
    int prev = -1, dev = -1;
    explicit DeviceScope(int d) : dev(d) {
        if (d < 0) return;
        if (hipGetDevice(&prev) != hipSuccess) { prev = -1; return; }
        if (prev != d) (void)hipSetDevice(d);
    }
    ~DeviceScope() { if (dev >= 0 && prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel): the attribute is per device, and one process may
// drive maps on several GPUs.  Thread safe.  Returns GPIS_OK or GPIS_ERR_HIP.
int ensure_dynamic_lds(const void* kernel, int bytes);

// Simple caching device allocator (size-class free lists).  Not thread safe:
// one map object = one caller thread, as in the reference.
struct DevPool;
DevPool* pool_create();
void pool_destroy(DevPool*);
void* pool_alloc(DevPool*, size_t bytes);
void pool_free(DevPool*, void* p);
size_t pool_bytes(DevPool*);
size_t pool_block_size(DevPool*, void* p);   // bytes of a live block of the pool (0: none)
size_t pool_cache_trim();   // hand the process-wide cache of pool chunks back to the driver; returns the bytes released

}  // namespace gpis
