// Shared device helpers for 32x32 tiles held in MFMA C/D layout (v_mfma_f32_32x32x2_f32):
// lane l owns column (l & 31) and the 16 rows rowmap(r, l >> 5), r = 0..15.
#pragma once
#include "dev_common.h"

namespace gpis {

__device__ __forceinline__ int rowmap_t(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- square root and division of the factorisation chains (round 6) --------------------------------------------------------
// A pivot step of the Cholesky sweeps is ONE dependent chain: update -> broadcast -> sqrtf -> divide -> update ... .  The
// compiler's correctly rounded sqrtf and `/` spend most of their ~30 dependent instructions on operand ranges these chains never
// see: sqrtf scales arguments below 2^-96 and classifies zeros / infinities, `/` runs v_div_scale twice, v_div_fmas and
// v_div_fixup around a seven-instruction core (reciprocal estimate, one Newton step on it, quotient estimate, two residual
// corrections).  Both cores are reproduced here WITHOUT the range handling -- the same instructions on the same values, hence
// the same bits, whenever the scaling would not have triggered: arguments of the square root in [2^-96, 2^128), divisors whose
// reciprocal is normal, numerators that are zero or at least 2^-100 in magnitude with quotients inside the normal range.
// Pivots of these matrices are square roots of diagonal Schur complements of kernel matrices with diagonal 1 + sigma or
// 3 / s^2 + sigma (in (0, ~1.9e3]; a matrix that has lost positive definiteness yields NaN here as there), the numerators are sums
// of products of kernel entries (|.| <= 1.9e3; the non-zero ones are no smaller than an ulp of such products: nowhere near 2^-100).
// Zero numerators keep their sign and 0 / 0 stays NaN (the fix-up instruction's job).  The divisor-only part of a division -- the refined
// reciprocal -- is computed ONCE per divisor, off the chain where the divisor is known ahead (diagonal solves, substitutions).
// tests/test_gpu_ongpis.py::test_ranged_sqrt_and_division_equal_the_ieee_ones runs gpis_selftest_ranged_arith (millions of
// operand pairs in and around the ranges above against the compiler's own sqrtf and `/` on the device); every factor, alpha
// and prediction test compares the kernels with the oracle's IEEE sqrtf and `/` bit for bit.
__device__ __forceinline__ float sqrt_ranged(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);                                   // v_sqrt_f32: within one ulp
    const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
    const float rm = fmaf(-sm, s, x), rp = fmaf(-sp, s, x);                     // residuals of the neighbours
    float t = (0.f >= rm) ? sm : s;
    t = (0.f < rp) ? sp : t;
    return t;
}
__device__ __forceinline__ float rcp_refined(float d) {                          // the divisor-only part of a / d
    const float r = __builtin_amdgcn_rcpf(d);                                    // v_rcp_f32: within one ulp
    const float e = fmaf(-d, r, 1.0f);
    return fmaf(e, r, r);
}
__device__ __forceinline__ float div_ranged(float a, float d, float r) {         // a / d with r = rcp_refined(d)
    const float q0 = a * r;
    float e = fmaf(-d, q0, a);
    float q = fmaf(e, r, q0);
    e = fmaf(-d, q, a);
    q = fmaf(e, r, q);
    return (a == 0.f) ? q0 : q;             // (+-0 / d: the zero of the right sign -- and NaN for 0 / 0, where r is NaN -- is the first product)
}
// ... for callers that only ADD the quotient to something non-zero or take it times something: a zero quotient may come out with
// either sign (0 / 0 is still NaN: every step carries r's NaN along)
__device__ __forceinline__ float div_ranged_anyzero(float a, float d, float r) {
    const float q0 = a * r;
    float e = fmaf(-d, q0, a);
    float q = fmaf(e, r, q0);
    e = fmaf(-d, q, a);
    return fmaf(e, r, q);
}

// In-register solve of the 32x32 lower-triangular system for the 32 columns of a
// tile held in MFMA C/D layout (lane = column, 16 of the 32 rows per lane half).
// Lc = diagonal block of L, column-major in LDS.  Row i is finalised by the half
// that owns it (true division), broadcast to the partner half, then every later
// row gets one fmaf: ascending chain, identical to the unblocked order.
// Column i of the block is fetched as four 16-byte LDS reads per lane half (rows
// 8g+4h .. 8g+4h+3), software-pipelined one step ahead.
struct DiagCol { float4 g[4]; float d, r; };
__device__ __forceinline__ void diag_load(DiagCol& c, const float* Lc, int i, int h) {
    const float4* p = reinterpret_cast<const float4*>(Lc + i * 32 + 4 * h);
#pragma unroll
    for (int g = 0; g < 4; ++g)
        if (8 * g + 7 > i) c.g[g] = p[2 * g];  // compile-time prune (i is a constant after unrolling)
    c.d = Lc[i * 32 + i];
    c.r = rcp_refined(c.d);        // (off the chain: the column is fetched one step ahead of its use)
}
template <bool PIPE>
__device__ __forceinline__ void diag_solve32(f32x16& v, const float* Lc, int h) {
    DiagCol cur, nxt;
    diag_load(cur, Lc, 0, h);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        if (PIPE && i + 1 < 32) diag_load(nxt, Lc, i + 1, h);
        const int hi_ = (i >> 2) & 1, ri = (i & 3) + 4 * (i >> 3);
        float cand = div_ranged(v[ri], cur.d, cur.r);
        // broadcast row i from the half that owns it: v_permlane32_swap gives {low-half copy, high-half copy}
        unsigned cu = __float_as_uint(cand);
        auto sw = __builtin_amdgcn_permlane32_swap(cu, cu, false, false);
        float vi = __uint_as_float(hi_ ? sw[1] : sw[0]);
        v[ri] = (h == hi_) ? vi : v[ri];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row0 = (r & 3) + 8 * (r >> 2);
            if (row0 + 4 > i) {
                const int row = row0 + 4 * h;
                const float4 q = cur.g[r >> 2];
                float lri = (r & 3) == 0 ? q.x : ((r & 3) == 1 ? q.y : ((r & 3) == 2 ? q.z : q.w));
                float upd = fmaf(-lri, vi, v[r]);
                v[r] = (row > i) ? upd : v[r];
            }
        }
        if (PIPE) cur = nxt;
        else if (i + 1 < 32) diag_load(cur, Lc, i + 1, h);
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ int tri_index(int b, int c) { return b * (b + 1) / 2 + c; }

// Right-looking factorisation of a full 32 x 32 diagonal block in registers: lane = row (both lane halves carry the same
// rows), a[k] = column k; pivots and column entries travel by v_readlane.  Element (i, k) takes fmaf(-l_ic, l_kc, .) for c
// ascending: order (O1).  Entries above the diagonal are scratch values nobody reads.  The square root and the division
// of the NEXT pivot column are started as soon as that column is updated, ahead of the other columns' updates of the
// current step: same operations in an order that lets the dependent sqrt/divide chain overlap the broadcasts and fmas
// (measured: no difference in kernel time on either the stress or the frame workload -- kept as the single shared copy).
__device__ __forceinline__ void factor32_inreg(float (&a)[32], int row) {
    float d = sqrt_ranged(__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(a[0]), 0)));
    float lic = div_ranged(a[0], d, rcp_refined(d));
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        a[c] = (row == c) ? d : lic;
        const float nl = -lic;
        float dn = 0.f, licn = 0.f;
        if (c + 1 < 32) {
            const float lk1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(a[c]), c + 1));
            a[c + 1] = fmaf(nl, lk1, a[c + 1]);
            dn = sqrt_ranged(__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(a[c + 1]), c + 1)));
            licn = div_ranged(a[c + 1], dn, rcp_refined(dn));
        }
#pragma unroll
        for (int k = c + 2; k < 32; ++k) {
            const float lkc = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(a[c]), k));
            a[k] = fmaf(nl, lkc, a[k]);
        }
        d = dn; lic = licn;
        __builtin_amdgcn_sched_barrier(0);   // keep the broadcast values of one column step together
    }
}

__device__ __forceinline__ float lo_half(float x) {   // the value of lane (l & 31) in every lane
    const unsigned u = __float_as_uint(x);
    return __uint_as_float(__builtin_amdgcn_permlane32_swap(u, u, false, false)[0]);
}
__device__ __forceinline__ float hi_half(float x) {   // the value of lane 32 + (l & 31) in every lane
    const unsigned u = __float_as_uint(x);
    return __uint_as_float(__builtin_amdgcn_permlane32_swap(u, u, false, false)[1]);
}


// Factorisation of a 32 x 32 diagonal tile, same operations in the same order as factor32_inreg (above), in four
// micro-blocks of eight columns.  The tile stays in ACCUMULATOR layout (t: lane = row, 16 of the 32 columns per lane
// half); per micro-block the eight columns are pulled into every lane of their row (a8, cross-half swaps), factorised
// column by column (pivot sqrt, column division, fmaf updates INSIDE the micro-block: 28 instead of ~200 broadcast /
// fmaf pairs), stored column-major into Lc (zeros above the diagonal), and then folded into the rest of the tile by
// FOUR matrix instructions: t -= l l^T over the eight columns in ascending order -- for every element the same
// fmaf(-l_ic, l_kc, .) chain over ascending c as the scalar sweep (v_mfma_f32_32x32x2_f32 is that chain).  Entries of t
// in or left of the micro-block are dead afterwards (never read again); entries above the diagonal are scratch.
template <int M>
__device__ __forceinline__ void factor32_mb(f32x16& t, int row, int h, int lane, float* Lc) {
    float a8[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a8[i] = lo_half(t[4 * M + i]); a8[4 + i] = hi_half(t[4 * M + i]); }   // columns 8M .. 8M+7 of row `row`
    float d = sqrt_ranged(__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(a8[0]), 8 * M)));
    float lic = div_ranged(a8[0], d, rcp_refined(d));
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
            dn = sqrt_ranged(__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(a8[c + 1]), col + 1)));
            licn = div_ranged(a8[c + 1], dn, rcp_refined(dn));
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
        // rank-8 update of the tile: A operand and B operand are the same register (t is symmetric in its roles: element
        // (i, k) -= l_ic l_kc); lane half h supplies the columns of parity h
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float x = h ? a8[2 * i + 1] : a8[2 * i];
            t = __builtin_amdgcn_mfma_f32_32x32x2f32(-x, x, t, 0, 0, 0);
        }
    }
}

}  // namespace gpis
