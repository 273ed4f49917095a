// Micro-benchmark: what does v_mfma_f32_32x32x2_f32 sustain on REAL data?  The gfx950 clock follows the power budget
// and matrix-pipe power depends on operand toggling: constant / zero operands run at the maximum clock, random
// operands do not.  Each wave runs chains of 16 MFMAs on 4 accumulator tiles with A/B operands held in registers
// (no memory traffic at all); operands are zeros, small constants or random normals.
// Build: hipcc --offload-arch=gfx950 -O3 -o .ab/mfma_power tools/ubench/mfma_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WAVES, int MINW>
__global__ __launch_bounds__(64 * WAVES, MINW) void k(const float* __restrict__ src, float* out, int steps) {
    const int gid = blockIdx.x * 64 * WAVES + threadIdx.x;
    float a[2][16], b[2][16];
    for (int i = 0; i < 16; ++i) {
        a[0][i] = src[(size_t)gid * 64 + i]; a[1][i] = src[(size_t)gid * 64 + 16 + i];
        b[0][i] = src[(size_t)gid * 64 + 32 + i]; b[1][i] = src[(size_t)gid * 64 + 48 + i];
    }
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll 1
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t & 1][kk], b[(t >> 1) & 1][kk], acc[t], 0, 0, 0);
        }
        // keep the accumulators bounded without touching the operand statistics much
        if ((s & 63) == 63) for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] *= 1e-3f;
    }
    float sum = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) sum += acc[t][r];
    out[gid] = sum;
}

template <int WAVES, int MINW>
static void run(const char* name, const char* data, int wgs_per_cu, const float* d, float* dout, int steps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL((k<WAVES, MINW>), dim3(grid), dim3(64 * WAVES), 0, 0, d, dout, 64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<WAVES, MINW>), dim3(grid), dim3(64 * WAVES), 0, 0, d, dout, steps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)grid * WAVES * steps * 4 * 16 * 4096.0;
    printf("%-10s %-8s waves/WG %2d WG/CU %d: %8.3f ms  %6.1f TFLOP/s (%5.1f%% of 157.3)\n", name, data, WAVES, wgs_per_cu, ms,
           flop / ms / 1e9, 100 * flop / ms / 1e9 / 157.3);
}

int main() {
    const size_t n = (size_t)256 * 4 * 64 * 16 * 64;
    std::vector<float> h(n);
    float *d, *dout;
    hipMalloc(&d, n * 4); hipMalloc(&dout, (size_t)256 * 4 * 64 * 16 * 4);
    const int steps = 20000;
    for (int mode = 0; mode < 3; ++mode) {
        const char* nm = mode == 0 ? "zeros" : (mode == 1 ? "const" : "random");
        srand(1);
        for (size_t i = 0; i < n; ++i) {
            if (mode == 0) h[i] = 0.f;
            else if (mode == 1) h[i] = 0.001f * (float)(i & 31);
            else { float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = rand() / (float)RAND_MAX; h[i] = 0.05f * sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); }
        }
        hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
        run<4, 1>("1w/SIMD", nm, 1, d, dout, steps);
        run<8, 2>("2w/SIMD", nm, 1, d, dout, steps);
        run<8, 4>("4w/SIMD", nm, 2, d, dout, steps);
    }
    return 0;
}
