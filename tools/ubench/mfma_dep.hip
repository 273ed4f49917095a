// Micro-benchmark: cost of DEPENDENT v_mfma_f32_32x32x2_f32 chains.  Each wave runs, per step, 16 matrix instructions on each
// of NACC accumulator tiles, either tile after tile (16 dependent instructions in a row: the order of the K3 product chains)
// or interleaved (instruction kk of every tile before kk + 1).  Operands are registers; WAVES per workgroup, one workgroup per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ab/mfma_dep tools/ubench/mfma_dep.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, bool INTER, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, int steps, long long* cyc) {
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a[16], b[NACC][16];
    for (int kk = 0; kk < 16; ++kk) { a[kk] = 1e-3f * (threadIdx.x + kk); for (int t = 0; t < NACC; ++t) b[t][kk] = 1e-3f * (kk + t + blockIdx.x); }
    __syncthreads();
    const long long t0 = clock64();
#pragma unroll 1
    for (int s = 0; s < steps; ++s) {
        if (INTER) {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
#pragma unroll
                for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[t][kk], acc[t], 0, 0, 0);
        } else {
#pragma unroll
            for (int t = 0; t < NACC; ++t)
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[t][kk], acc[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC, bool INTER, int WAVES>
static void run(float* out, long long* cyc) {
    const int steps = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, INTER, WAVES><<<256, 64 * WAVES>>>(out, steps, cyc);
    hipEventRecord(e0);
    k<NACC, INTER, WAVES><<<256, 64 * WAVES>>>(out, steps, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double nm = (double)steps * 16 * NACC;
    printf("tiles/wave %d %-11s waves/SIMD %d: %.1f shader-clock ticks per instruction per wave, %.1f ns per instruction per SIMD -> %.1f TFLOP/s\n", NACC,
           INTER ? "interleaved" : "tile-by-tile", WAVES / 4, (double)c / nm, ms * 1e6 / (nm * (WAVES / 4)), 256.0 * WAVES * nm * 4096 / (ms * 1e-3) / 1e12);
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    run<1, false, 4>(out, cyc); run<1, false, 8>(out, cyc);
    run<2, false, 4>(out, cyc); run<2, true, 4>(out, cyc);
    run<3, false, 4>(out, cyc); run<3, true, 4>(out, cyc);
    run<3, false, 8>(out, cyc); run<3, true, 8>(out, cyc);
    run<4, false, 8>(out, cyc); run<4, true, 8>(out, cyc);
    return 0;
}
