// Device self-test of the range-restricted square root and division of the factorisation chains (tile_solve.h, round 6) against
// the compiler's own correctly rounded sqrtf and `/` on the same device: counts the operand pairs whose bits differ.
// C-ABI: gpis_selftest_ranged_arith (include/gpismap_amd.h); tests/test_gpu_ongpis.py.
#include "tile_solve.h"
#include "exp_tab.h"

namespace gpis {

__device__ __forceinline__ unsigned st_rng(unsigned long long& s) {      // splitmix64, upper half
    s += 0x9E3779B97F4A7C15ull;
    unsigned long long z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (unsigned)((z ^ (z >> 31)) >> 32);
}
// a float with a uniformly random mantissa and sign (when `signed_`) and an exponent drawn from [elo, ehi] (biased)
__device__ __forceinline__ float st_float(unsigned long long& s, int elo, int ehi, bool signed_) {
    const unsigned m = st_rng(s) & 0x007FFFFFu;
    const unsigned e = (unsigned)(elo + (int)(st_rng(s) % (unsigned)(ehi - elo + 1)));
    const unsigned sg = signed_ ? (st_rng(s) & 0x80000000u) : 0u;
    return __uint_as_float(sg | (e << 23) | m);
}

// mode 0: the documented ranges (pivots 2^-40 .. 2^12, numerators 0 or 2^-90 .. 2^12 of either sign); mode 1: operands shaped like
// the factorisations' (divisor = sqrt of something in [1e-4, 2e3], numerators within 2^-30 .. 2^11 incl. exact zeros of both signs)
__global__ void selftest_ranged_kernel(unsigned long long seed, int per_thread, int mode, unsigned long long* out) {
    unsigned long long s = seed + 0x1234567ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    unsigned long long bad_sqrt = 0, bad_div = 0;
    for (int i = 0; i < per_thread; ++i) {
        const float x = mode == 0 ? st_float(s, 127 - 80, 127 + 24, false) : st_float(s, 127 - 14, 127 + 11, false);
        const float d_ieee = sqrtf(x), d_fast = sqrt_ranged(x);
        if (__float_as_uint(d_ieee) != __float_as_uint(d_fast)) ++bad_sqrt;
        float a = mode == 0 ? st_float(s, 127 - 90, 127 + 12, true) : st_float(s, 127 - 30, 127 + 11, true);
        const unsigned pick = st_rng(s) & 63u;
        if (pick == 0) a = 0.f;
        if (pick == 1) a = -0.f;
        if (pick == 2) a = d_ieee;                 // quotient exactly 1
        float dv = d_ieee;
        if (pick == 3) { a = 0.f; dv = 0.f; }      // 0 / 0: NaN either way (the prediction kernel meets it for a query ON a training point)
        if (pick == 4) dv = -dv;
        const float q_ieee = a / dv, q_fast = div_ranged(a, dv, rcp_refined(dv));
        if (__float_as_uint(q_ieee) != __float_as_uint(q_fast) && !(q_ieee != q_ieee && q_fast != q_fast)) ++bad_div;
    }
    atomicAdd(out, bad_sqrt);
    atomicAdd(out + 1, bad_div);
}

// mode 2: the table-driven exponential of the kernels' generation (exp_tab.h) against the device library's exp on arguments -a r
// in [-12, 0] (and a few far below): out[0] = results more than one ulp apart (none: each is within about half an ulp of the truth),
// out[1] = results that differ at all (last-bit disagreements, a fraction of a per cent)
__global__ void selftest_exp_kernel(unsigned long long seed, int per_thread, unsigned long long* out) {
    __shared__ f64x2 tab[64];
    if (threadIdx.x < 64) tab[threadIdx.x] = *reinterpret_cast<const f64x2*>(kExp64Tab[threadIdx.x]);
    __syncthreads();
    unsigned long long s = seed + 0x7654321ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    unsigned long long far = 0, any = 0;
    for (int i = 0; i < per_thread; ++i) {
        const unsigned u = st_rng(s);
        float x = -12.0f * (float)(u >> 8) * (1.0f / 16777216.0f);
        if ((u & 255u) == 0) x = -700.0f - (float)(st_rng(s) & 127u);     // deep underflow region and below -745
        if ((u & 255u) == 1) x = -0.0f;
        const double a = exp((double)x), b = exp_neg_tab(x, tab);
        const long long ia = __double_as_longlong(a), ib = __double_as_longlong(b);
        const long long d = ia > ib ? ia - ib : ib - ia;
        if (d > 1) ++far;
        if (d != 0) ++any;
    }
    atomicAdd(out, far);
    atomicAdd(out + 1, any);
}

// -> mismatches[0] = square roots, [1] = divisions, of blocks * 256 * per_thread operand pairs
int selftest_ranged_arith(unsigned long long seed, int blocks, int per_thread, int mode, unsigned long long* mismatches) {
    if (blocks < 1 || per_thread < 1 || !mismatches) return GPIS_ERR_ARG;
    unsigned long long* d = nullptr;
    GPIS_HIP(hipMalloc(&d, 2 * sizeof(unsigned long long)));
    hipError_t e = hipMemset(d, 0, 2 * sizeof(unsigned long long));
    if (e == hipSuccess) {
        if (mode == 2) hipLaunchKernelGGL(selftest_exp_kernel, dim3(blocks), dim3(256), 0, 0, seed, per_thread, d);
        else hipLaunchKernelGGL(selftest_ranged_kernel, dim3(blocks), dim3(256), 0, 0, seed, per_thread, mode, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(mismatches, d, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    GPIS_HIP(e);
    return GPIS_OK;
}

}  // namespace gpis
