// Micro-benchmark: what limits K4's bulk loop?  Each wave runs dependent chains of 16 x v_mfma_f32_32x32x2_f32
// on NT accumulator tiles per step; the A operand comes from global memory (4 x 16-byte loads per tile, one
// tile ahead) or is constant; the B operand comes from LDS (one value per MFMA) or is constant.
// Build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_stream tools/ubench/mfma_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int WAVES, int MINW, bool GLOAD, bool LDSB>
__global__ __launch_bounds__(64 * WAVES, MINW) void k(const float* __restrict__ A, float* out, int steps, int ntile_bytes) {
    __shared__ float V[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 64 * WAVES) (&V[0][0])[i] = A[(i * 7 + blockIdx.x * 13) & 1048575] + 0.001f * (i & 31);
    __syncthreads();
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (unsigned)ntile_bytes, 0x00020000);
    const int voff = lane * 16;
    float av[2][16];
    auto load_a = [&](float (&a)[16], int tile) {
        if (GLOAD) {
            const int sb = (tile & 1023) * 4096;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                auto q = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, sb + g * 1024, 0);
                a[4 * g] = __uint_as_float(q[0]); a[4 * g + 1] = __uint_as_float(q[1]);
                a[4 * g + 2] = __uint_as_float(q[2]); a[4 * g + 3] = __uint_as_float(q[3]);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) a[kk] = 0.001f * (float)(kk + (tile & 3));
        }
    };
    int tile = blockIdx.x * 37 + wave * 5;
    load_a(av[0], tile);
#pragma unroll 1
    for (int s = 0; s < steps; ++s) {
        const float* Vl = &V[s & 3][0] + lane;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            load_a(av[(t + 1) & 1], tile + t + 1);
            if (LDSB) {
                float p0 = Vl[0], p1 = Vl[64];
#pragma unroll
                for (int kk = 0; kk < 16; kk += 2) {
                    float n0 = 0.f, n1 = 0.f;
                    if (kk + 2 < 16) { n0 = Vl[(kk + 2) * 64]; n1 = Vl[(kk + 3) * 64]; }
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t & 1][kk], p0, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t & 1][kk + 1], p1, acc[t], 0, 0, 0);
                    p0 = n0; p1 = n1;
                }
            } else {
                const float p = 0.01f * lane;
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t & 1][kk], p, acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        tile += 4;
    }
    float sum = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) sum += acc[t][r];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = sum;
}
template <int WAVES, int MINW, bool GLOAD, bool LDSB>
static void run(const char* name, int wgs_per_cu, const float* dA, float* dout, int steps, int bytes) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL((k<WAVES, MINW, GLOAD, LDSB>), dim3(grid), dim3(64 * WAVES), 0, 0, dA, dout, 10, bytes);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<WAVES, MINW, GLOAD, LDSB>), dim3(grid), dim3(64 * WAVES), 0, 0, dA, dout, steps, bytes);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)grid * WAVES * steps * 4 * 16 * 4096.0;
    printf("%-34s waves/WG %2d WG/CU %d: %.3f ms  %.1f TFLOP/s (%.1f%%)\n", name, WAVES, wgs_per_cu, ms, flop / ms / 1e9, 100 * flop / ms / 1e9 / 157.3);
}

// K4-like step structure: per step a wave waits on an LDS flag (already set), loads its first A tile only then,
// updates TILES accumulator tiles with B streamed from LDS, and raises its own LDS flag.
template <int WAVES, int MINW, int TILES>
__global__ __launch_bounds__(64 * WAVES, MINW) void kstep(const float* __restrict__ A, float* out, int steps, int ntile_bytes) {
    __shared__ float V[4][1024];
    __shared__ volatile int flags[32];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 4096; i += 64 * WAVES) (&V[0][0])[i] = A[(i * 7 + blockIdx.x * 13) & 1048575] + 0.001f * (i & 31);
    if (threadIdx.x < 32) flags[threadIdx.x] = 1 << 30;
    __syncthreads();
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (unsigned)ntile_bytes, 0x00020000);
    const int voff = lane * 16;
    auto load_a = [&](float (&a)[16], int tile) {
        const int sb = (tile & 1023) * 4096;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            auto q = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, sb + g * 1024, 0);
            a[4 * g] = __uint_as_float(q[0]); a[4 * g + 1] = __uint_as_float(q[1]);
            a[4 * g + 2] = __uint_as_float(q[2]); a[4 * g + 3] = __uint_as_float(q[3]);
        }
    };
    int tile = blockIdx.x * 37 + wave * 5;
#pragma unroll 1
    for (int s = 0; s < steps; ++s) {
        float av[2][16];
        load_a(av[0], tile);
        while (flags[0] < s) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        const float* Vl = &V[s & 3][0] + lane;
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            if (t + 1 < TILES) load_a(av[(t + 1) & 1], tile + t + 1);
            float p0 = Vl[0], p1 = Vl[64];
#pragma unroll
            for (int kk = 0; kk < 16; kk += 2) {
                float n0 = 0.f, n1 = 0.f;
                if (kk + 2 < 16) { n0 = Vl[(kk + 2) * 64]; n1 = Vl[(kk + 3) * 64]; }
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t & 1][kk], p0, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t & 1][kk + 1], p1, acc[t], 0, 0, 0);
                p0 = n0; p1 = n1;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        tile += TILES;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        if (lane == 0) flags[1 + wave] = s;
    }
    float sum = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) sum += acc[t][r];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = sum;
}
template <int WAVES, int MINW, int TILES>
static void runstep(int wgs_per_cu, const float* dA, float* dout, int steps, int bytes) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL((kstep<WAVES, MINW, TILES>), dim3(grid), dim3(64 * WAVES), 0, 0, dA, dout, 10, bytes);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((kstep<WAVES, MINW, TILES>), dim3(grid), dim3(64 * WAVES), 0, 0, dA, dout, steps, bytes);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)grid * WAVES * steps * TILES * 16 * 4096.0;
    printf("stepped, %d tile(s)/step            waves/WG %2d WG/CU %d: %.3f ms  %.1f TFLOP/s (%.1f%%)\n", TILES, WAVES, wgs_per_cu, ms, flop / ms / 1e9, 100 * flop / ms / 1e9 / 157.3);
}

// as kstep, but the first A tile of every step is staged into a per-wave LDS slot by LDS-DMA during the step before
typedef const void __attribute__((address_space(1))) * gvptr_t;
typedef void __attribute__((address_space(3))) * lvptr_t;
template <int WAVES, int MINW, int TILES>
__global__ __launch_bounds__(64 * WAVES, MINW) void kstep_dma(const float* __restrict__ A, float* out, int steps, int ntile_bytes) {
    __shared__ float V[4][1024];
    __shared__ __attribute__((aligned(16))) float S[WAVES][2][1024];
    __shared__ volatile int flags[32];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 4096; i += 64 * WAVES) (&V[0][0])[i] = A[(i * 7 + blockIdx.x * 13) & 1048575] + 0.001f * (i & 31);
    if (threadIdx.x < 32) flags[threadIdx.x] = 1 << 30;
    __syncthreads();
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (unsigned)ntile_bytes, 0x00020000);
    const int voff = lane * 16;
    auto load_a = [&](float (&a)[16], int tile) {
        const int sb = (tile & 1023) * 4096;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            auto q = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, sb + g * 1024, 0);
            a[4 * g] = __uint_as_float(q[0]); a[4 * g + 1] = __uint_as_float(q[1]);
            a[4 * g + 2] = __uint_as_float(q[2]); a[4 * g + 3] = __uint_as_float(q[3]);
        }
    };
    auto stage = [&](int tile, int slot) {
        const float* src = A + (size_t)(tile & 1023) * 1024 + lane * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            __builtin_amdgcn_global_load_lds((gvptr_t)(src + g * 256), (lvptr_t)(&S[wave][slot][g * 256]), 16, 0, 0);
    };
    int tile = blockIdx.x * 37 + wave * 5;
    stage(tile, 0);
#pragma unroll 1
    for (int s = 0; s < steps; ++s) {
        float av[2][16];
        while (flags[0] < s) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): the staged tile has landed
        {
            const float4* p4 = reinterpret_cast<const float4*>(&S[wave][s & 1][0]) + lane;
#pragma unroll
            for (int g = 0; g < 4; ++g) { const float4 q = p4[g * 64]; av[0][4 * g] = q.x; av[0][4 * g + 1] = q.y; av[0][4 * g + 2] = q.z; av[0][4 * g + 3] = q.w; }
        }
        stage(tile + TILES, (s + 1) & 1);     // next step's first tile
        const float* Vl = &V[s & 3][0] + lane;
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            if (t + 1 < TILES) load_a(av[(t + 1) & 1], tile + t + 1);
            float p0 = Vl[0], p1 = Vl[64];
#pragma unroll
            for (int kk = 0; kk < 16; kk += 2) {
                float n0 = 0.f, n1 = 0.f;
                if (kk + 2 < 16) { n0 = Vl[(kk + 2) * 64]; n1 = Vl[(kk + 3) * 64]; }
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t & 1][kk], p0, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t & 1][kk + 1], p1, acc[t], 0, 0, 0);
                p0 = n0; p1 = n1;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        tile += TILES;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        if (lane == 0) flags[1 + wave] = s;
    }
    float sum = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) sum += acc[t][r];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = sum;
}
template <int WAVES, int MINW, int TILES>
static void runstep_dma(int wgs_per_cu, const float* dA, float* dout, int steps, int bytes) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL((kstep_dma<WAVES, MINW, TILES>), dim3(grid), dim3(64 * WAVES), 0, 0, dA, dout, 10, bytes);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((kstep_dma<WAVES, MINW, TILES>), dim3(grid), dim3(64 * WAVES), 0, 0, dA, dout, steps, bytes);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)grid * WAVES * steps * TILES * 16 * 4096.0;
    printf("stepped + LDS-DMA first tile, %d/step waves/WG %2d WG/CU %d: %.3f ms  %.1f TFLOP/s (%.1f%%)\n", TILES, WAVES, wgs_per_cu, ms, flop / ms / 1e9, 100 * flop / ms / 1e9 / 157.3);
}
// argv[1] = "random": A tiles and the LDS operand block hold random normals instead of zeros / small constants.  The gfx950
// clock follows the power budget and matrix-pipe power follows operand toggling: the zero-data numbers are the
// scheduling ceiling, the random-data numbers the ceiling a real kernel can reach (NOTEBOOK.md, K4).
static bool g_random = false;
int main(int argc, char** argv) {
    const int bytes = 1024 * 4096;
    g_random = argc > 1 && argv[1][0] == 'r';
    float *dA, *dout; hipMalloc(&dA, bytes); hipMemset(dA, 0, bytes); hipMalloc(&dout, 4 * 1024 * 1024 * 4);
    if (g_random) {
        float* h = (float*)malloc(bytes);
        srand(7);
        for (int i = 0; i < bytes / 4; ++i) {
            float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = rand() / (float)RAND_MAX;
            h[i] = 0.05f * sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
        }
        hipMemcpy(dA, h, bytes, hipMemcpyHostToDevice);
        free(h);
    }
    printf("data: %s\n", g_random ? "random normals" : "zeros / small constants");
    const int steps = 2000;
    run<8, 4, false, false>("const A, const B", 2, dA, dout, steps, bytes);
    run<8, 4, false, true>("const A, LDS B", 2, dA, dout, steps, bytes);
    run<8, 4, true, false>("global A, const B", 2, dA, dout, steps, bytes);
    run<8, 4, true, true>("global A, LDS B", 2, dA, dout, steps, bytes);
    run<8, 4, true, true>("global A, LDS B", 1, dA, dout, steps, bytes);
    run<4, 2, true, true>("global A, LDS B (256 thr)", 2, dA, dout, steps, bytes);
    run<4, 2, false, false>("const A, const B (256 thr)", 2, dA, dout, steps, bytes);
    run<4, 1, false, false>("const A, const B (256 thr)", 1, dA, dout, steps, bytes);
    runstep<8, 4, 1>(2, dA, dout, 4000, bytes);
    runstep<8, 4, 2>(2, dA, dout, 2000, bytes);
    runstep<8, 4, 3>(2, dA, dout, 1500, bytes);
    runstep<8, 4, 4>(2, dA, dout, 1000, bytes);
    runstep_dma<8, 4, 1>(2, dA, dout, 4000, bytes);
    runstep_dma<8, 4, 2>(2, dA, dout, 2000, bytes);
    runstep_dma<8, 4, 3>(2, dA, dout, 1500, bytes);
    return 0;
}
