// Micro-benchmark (round 6): what ONE pivot step of the in-register Cholesky of a 32 x 32 diagonal tile costs
// -- the dependent chain update -> v_readlane -> sqrt -> divide of factor32_mb (csrc/tile_solve.h) -- measured with s_memtime on
// wavefronts that run alone on their SIMD, with the compiler's IEEE sqrtf and `/` and with the range-restricted sequences, and
// with 0 .. 3 partner wavefronts on the same SIMD that issue nothing but matrix instructions / nothing but vector-ALU instructions.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Igpismap_amd/csrc -Iinclude tools/ubench/pivot_chain.hip -o /tmp/pivot_chain && /tmp/pivot_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "tile_solve.h"

using namespace gpis;

// factor32_mb with the compiler's sqrtf and `/` (the round-5 routine, kept here for the comparison)
template <int M>
__device__ __forceinline__ void factor32_mb_ieee(f32x16& t, int row, int h, int lane, float* Lc) {
    float a8[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a8[i] = lo_half(t[4 * M + i]); a8[4 + i] = hi_half(t[4 * M + i]); }
    float d = sqrtf(__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(a8[0]), 8 * M)));
    float lic = a8[0] / d;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int col = 8 * M + c;
        a8[c] = (row == col) ? d : lic;
        if (lane < 32) Lc[col * 32 + lane] = (row >= col) ? a8[c] : 0.f;
        const float nl = -lic;
        float dn = 0.f, licn = 0.f;
        if (c + 1 < 8) {
            const float lk1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(a8[c]), col + 1));
            a8[c + 1] = fmaf(nl, lk1, a8[c + 1]);
            dn = sqrtf(__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(a8[c + 1]), col + 1)));
            licn = a8[c + 1] / dn;
        }
#pragma unroll
        for (int k = c + 2; k < 8; ++k) {
            const float lkc = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(a8[c]), 8 * M + k));
            a8[k] = fmaf(nl, lkc, a8[k]);
        }
        d = dn; lic = licn;
        __builtin_amdgcn_sched_barrier(0);
    }
    if (M < 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float x = h ? a8[2 * i + 1] : a8[2 * i];
            t = __builtin_amdgcn_mfma_f32_32x32x2f32(-x, x, t, 0, 0, 0);
        }
    }
}

// wave 0 of every workgroup: REPS factorisations of one SPD tile (accumulator layout), timed; waves 1.. : partners on the same SIMD
// (workgroups of 1 + 4 P waves: waves 4, 8, 12 share SIMD 0 with wave 0 -- wave w runs on SIMD w % 4), mode 1 = matrix instructions,
// mode 2 = dependent vector-ALU fmas, until wave 0 raises the flag
template <bool IEEE>
__global__ __launch_bounds__(1024) void pivot_kernel(const float* tile, int reps, int partner_mode, unsigned long long* cycles, float* sink) {
    __shared__ float Lc[32 * 32];
    __shared__ volatile int done;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    if (tid == 0) done = 0;
    __syncthreads();
    if (wave == 0) {
        f32x16 t0;
#pragma unroll
        for (int r = 0; r < 16; ++r) t0[r] = tile[l31 * 32 + rowmap_t(r, h)];       // (symmetric tile: element (row, col) = (col, row))
        float acc = 0.f;
        const unsigned long long c0 = __builtin_readcyclecounter();
        for (int i = 0; i < reps; ++i) {
            f32x16 t = t0;
            if (IEEE) { factor32_mb_ieee<0>(t, l31, h, lane, Lc); factor32_mb_ieee<1>(t, l31, h, lane, Lc); factor32_mb_ieee<2>(t, l31, h, lane, Lc); factor32_mb_ieee<3>(t, l31, h, lane, Lc); }
            else { factor32_mb<0>(t, l31, h, lane, Lc); factor32_mb<1>(t, l31, h, lane, Lc); factor32_mb<2>(t, l31, h, lane, Lc); factor32_mb<3>(t, l31, h, lane, Lc); }
            acc += t[15];
            t0[0] += 1e-7f * acc * 0.f;     // (a dependence between the repetitions)
        }
        const unsigned long long c1 = __builtin_readcyclecounter();
        if (lane == 0) { cycles[blockIdx.x] = c1 - c0; sink[blockIdx.x] = acc + Lc[33 * 7]; }
        __builtin_amdgcn_s_waitcnt(0);
        if (lane == 0) done = 1;
    } else if ((wave & 3) == 0 && partner_mode != 0) {
        f32x16 p;
#pragma unroll
        for (int r = 0; r < 16; ++r) p[r] = 0.f;
        float x = 1.0f + lane * 1e-3f, y = 0.5f;
        while (!done) {
            if (partner_mode == 1) {
#pragma unroll
                for (int i = 0; i < 16; ++i) p = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, p, 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 64; ++i) y = fmaf(y, 0.999f, x);
            }
        }
        if (lane == 0) sink[gridDim.x + blockIdx.x * 16 + wave] = p[0] + y;
    }
}

int main() {
    std::vector<float> A(32 * 32);
    // an SPD tile shaped like a Matern-3/2 kernel block of 32 points on a line, spacing 0.3 length scales, diagonal 1.005
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            const double r = 0.3 * std::abs(i - j) * std::sqrt(3.0);
            A[i * 32 + j] = (float)((1.0 + r) * std::exp(-r)) + (i == j ? 0.005f : 0.f);
        }
    float* d_tile; unsigned long long* d_cyc; float* d_sink;
    hipMalloc(&d_tile, sizeof(float) * 1024); hipMalloc(&d_cyc, sizeof(unsigned long long) * 64); hipMalloc(&d_sink, sizeof(float) * 4096);
    hipMemcpy(d_tile, A.data(), sizeof(float) * 1024, hipMemcpyHostToDevice);
    const int reps = 200;
    printf("# one pivot step of factor32_mb (32 per tile factorisation), cycles of the shader clock counter (s_memtime, 100 MHz ticks are NOT used: __builtin_readcyclecounter)\n");
    printf("# %-34s %10s %10s\n", "partners on the chain's SIMD", "IEEE", "ranged");
    const char* names[3] = {"none", "matrix instructions", "vector-ALU fma chain"};
    for (int mode = 0; mode < 3; ++mode)
        for (int P = (mode == 0 ? 0 : 1); P <= (mode == 0 ? 0 : 3); ++P) {
            double res[2];
            for (int v = 0; v < 2; ++v) {
                const int threads = 64 * (1 + 4 * P);
                if (v == 0) hipLaunchKernelGGL(pivot_kernel<true>, dim3(1), dim3(threads), 0, 0, d_tile, reps, mode, d_cyc, d_sink);
                else hipLaunchKernelGGL(pivot_kernel<false>, dim3(1), dim3(threads), 0, 0, d_tile, reps, mode, d_cyc, d_sink);
                hipDeviceSynchronize();
                unsigned long long c = 0;
                hipMemcpy(&c, d_cyc, sizeof(c), hipMemcpyDeviceToHost);
                res[v] = (double)c / reps / 32.0;
            }
            char label[64];
            snprintf(label, sizeof(label), "%d x %s", P, names[mode]);
            printf("  %-34s %10.1f %10.1f\n", mode == 0 ? "none (the wavefront alone)" : label, res[0], res[1]);
        }
    return 0;
}
