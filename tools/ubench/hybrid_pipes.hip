// Micro-benchmark: can the matrix pipe (v_mfma_f32_32x32x2_f32) and the vector pipe (v_pk_fma_f32) of a SIMD run
// fp32 work at the same time, and what does the power budget leave of it?  Each workgroup has NW waves; wave w runs MFMA
// chains when (w / 4) is even and packed-FMA chains when it is odd (mode 2), or all waves run one kind (modes 0 / 1).
// Operands are random normals held in registers: no memory traffic.
// Build: hipcc --offload-arch=gfx950 -O3 -o .ab/hybrid_pipes tools/ubench/hybrid_pipes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float mfma_chain(const float* __restrict__ src, int gid, int steps) {
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
        if ((s & 63) == 63) for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] *= 1e-3f;
    }
    float sum = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) sum += acc[t][r];
    return sum;
}

// one step = 4 x 16 packed FMAs per lane = the same 4 * 16 * 4096 / ... no: 64 instr x 64 lanes x 2 x 2 flop = 16384 flop per wave and step
__device__ __forceinline__ float valu_chain(const float* __restrict__ src, int gid, int steps) {
    f32x2 a[8], b[8], c[16];
    for (int i = 0; i < 8; ++i) {
        a[i] = f32x2{src[(size_t)gid * 64 + 2 * i], src[(size_t)gid * 64 + 2 * i + 1]};
        b[i] = f32x2{src[(size_t)gid * 64 + 32 + 2 * i], src[(size_t)gid * 64 + 33 + 2 * i]};
    }
    for (int i = 0; i < 16; ++i) c[i] = f32x2{0.f, 0.f};
#pragma unroll 1
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c[i]) : "v"(a[(i + rep) & 7]), "v"(b[(i + 3 * rep) & 7]));
        if ((s & 63) == 63) for (int i = 0; i < 16; ++i) c[i] *= 1e-3f;
    }
    float sum = 0.f;
    for (int i = 0; i < 16; ++i) sum += c[i][0] + c[i][1];
    return sum;
}

template <int NW, int MINW>
__global__ __launch_bounds__(64 * NW, MINW) void k(const float* __restrict__ src, float* out, int steps_m, int steps_v, int mode) {
    const int gid = blockIdx.x * 64 * NW + threadIdx.x;
    const int wave = threadIdx.x >> 6;
    const bool valu = mode == 1 || (mode == 2 && ((wave >> 2) & 1));
    out[gid] = valu ? valu_chain(src, gid, steps_v) : mfma_chain(src, gid, steps_m);
}

template <int NW, int MINW>
static void run(const char* name, int mode, const float* d, float* dout, int steps_m, int steps_v) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256;
    hipLaunchKernelGGL((k<NW, MINW>), dim3(grid), dim3(64 * NW), 0, 0, d, dout, 64, 64, mode);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NW, MINW>), dim3(grid), dim3(64 * NW), 0, 0, d, dout, steps_m, steps_v, mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int nm = mode == 0 ? NW : (mode == 1 ? 0 : NW / 2), nv = NW - nm;
    double fm = (double)grid * nm * steps_m * 4 * 16 * 4096.0, fv = (double)grid * nv * steps_v * 64 * 256.0;
    printf("%-28s %8.3f ms   matrix %6.1f TFLOP/s   vector %6.1f TFLOP/s   sum %6.1f (%5.1f%% of 157.3)\n", name, ms, fm / ms / 1e9,
           fv / ms / 1e9, (fm + fv) / ms / 1e9, 100 * (fm + fv) / ms / 1e9 / 157.3);
}

int main() {
    const size_t n = (size_t)256 * 16 * 64 * 64;
    std::vector<float> h(n);
    float *d, *dout;
    hipMalloc(&d, n * 4); hipMalloc(&dout, (size_t)256 * 16 * 64 * 4);
    srand(1);
    for (size_t i = 0; i < n; ++i) { float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = rand() / (float)RAND_MAX; h[i] = 0.05f * sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); }
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    // steps chosen so that both kinds of wave take about the same time when they do not disturb each other:
    // MFMA step = 64 instr x 64 cycles = 4096 cycles; VALU step = 64 instr x 4 cycles = 256 cycles (x16)
    const int sm = 10000, sv = 160000;
    run<8, 2>("matrix only, 2 waves/SIMD", 0, d, dout, sm, sv);
    run<8, 2>("vector only, 2 waves/SIMD", 1, d, dout, sm, sv / 2);
    run<8, 2>("1 matrix + 1 vector /SIMD", 2, d, dout, sm, sv);
    run<16, 4>("2 matrix + 2 vector /SIMD", 2, d, dout, sm / 2, sv / 2);
    run<8, 2>("1 matrix + 1 vector(half)", 2, d, dout, sm, sv / 2);
    return 0;
}
