// Shared device helpers for 32x32 tiles held in MFMA C/D layout (v_mfma_f32_32x32x2_f32):
// lane l owns column (l & 31) and the 16 rows rowmap(r, l >> 5), r = 0..15.
#pragma once
#include "dev_common.h"

namespace gpis {

__device__ __forceinline__ int rowmap_t(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// In-register solve of the 32x32 lower-triangular system for the 32 columns of a
// tile held in MFMA C/D layout (lane = column, 16 of the 32 rows per lane half).
// Lc = diagonal block of L, column-major in LDS.  Row i is finalised by the half
// that owns it (true division), broadcast to the partner half, then every later
// row gets one fmaf: ascending chain, identical to the unblocked order.
// Column i of the block is fetched as four 16-byte LDS reads per lane half (rows
// 8g+4h .. 8g+4h+3), software-pipelined one step ahead.
struct DiagCol { float4 g[4]; float d; };
__device__ __forceinline__ void diag_load(DiagCol& c, const float* Lc, int i, int h) {
    const float4* p = reinterpret_cast<const float4*>(Lc + i * 32 + 4 * h);
#pragma unroll
    for (int g = 0; g < 4; ++g)
        if (8 * g + 7 > i) c.g[g] = p[2 * g];  // compile-time prune (i is a constant after unrolling)
    c.d = Lc[i * 32 + i];
}
template <bool PIPE>
__device__ __forceinline__ void diag_solve32(f32x16& v, const float* Lc, int h) {
    DiagCol cur, nxt;
    diag_load(cur, Lc, 0, h);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        if (PIPE && i + 1 < 32) diag_load(nxt, Lc, i + 1, h);
        const int hi_ = (i >> 2) & 1, ri = (i & 3) + 4 * (i >> 3);
        float cand = v[ri] / cur.d;
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
    float d = sqrtf(__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(a[0]), 0)));
    float lic = a[0] / d;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        a[c] = (row == c) ? d : lic;
        const float nl = -lic;
        float dn = 0.f, licn = 0.f;
        if (c + 1 < 32) {
            const float lk1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(a[c]), c + 1));
            a[c + 1] = fmaf(nl, lk1, a[c + 1]);
            dn = sqrtf(__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(a[c + 1]), c + 1)));
            licn = a[c + 1] / dn;
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
