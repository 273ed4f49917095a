// K4 for SMALL clusters (at most 9 block rows: K <= 287, e.g. the 64-point / K = 256 clusters of BASELINE config 5):
// the explicit inverse X stays RESIDENT in registers while a workgroup walks through consecutive 8-query tiles of
// the same cluster.
//
// Same operator as ongpis_eval_kernel (ongpis_test.hip; reference OnGPIS::testSinglePoint cpp/src/OnGPIS.cpp:177-216,
// cross covariance cpp/src/covFnc.cpp:258-314 / :404-450), same arithmetic element for element: V = X B with every element
// one fmaf chain from zero over ascending k, the mean in row K of V, the sums of squares in the (wavefront, lane half,
// row) chains of the oracle's reduce_ss -- results are bit-identical to the large-cluster kernel and to the oracle.
//
// STATUS: parity-green, OPT-IN (gpis_ongpis_set_small_kernel), measured slower than the general kernel on MI355X: stress
// configuration 16.1 ms vs 15.1 ms, K = 174 clusters 33.8 vs ~35 TFLOP/s.  With X resident only one workgroup of 8
// wavefronts fits a CU (256 VGPRs, 126 KB LDS): two wavefronts per SIMD do not hide the generation / reduction phases
// and the 180 KB prologue the way eight small workgroups of the general kernel do.  Kept as the measured experiment.
//
// Why a second kernel (the idea): a K = 256 cluster is 44 tile products per 8 queries (11 k cycles of a CU's matrix pipes) against
// ~38 k cycles of fixed cost per workgroup in the general kernel (cluster vectors, exp table, first chunk, reduction,
// every X tile streamed from L2 again for every 8 queries): 35 % of the fp32 matrix peak.  Here
//   * a workgroup of 8 wavefronts takes G consecutive tiles of the tile list (sorted by cluster): X (<= 45 tiles = 180 KB)
//     is loaded into registers once per cluster change -- block row b belongs to ONE wavefront (rows 8..5 on wavefronts
//     0..3, rows 4..1 on wavefronts 7..4, row 0 with row 1: the two wavefronts of a SIMD carry 12 / 11 / 11 / 11 products);
//   * the whole B (<= 9 tiles) of a query tile fits LDS twice: the wavefronts generate tile t+1 while they multiply tile t
//     (half of them before, half after -- vector and matrix work overlap on every SIMD), one barrier per tile;
//   * finished rows of V go through LDS to the reduction, which walks them in the order of the oracle's chains, so the row
//     -> wavefront assignment above is free of the summation order.
#include <type_traits>
#include "ongpis.h"
#include "tile_solve.h"

namespace gpis {

typedef const float __attribute__((address_space(1))) * gfptr_s;
typedef const int __attribute__((address_space(1))) * giptr_s;

namespace {
constexpr int kSW = 8;                       // wavefronts
constexpr int kSNB = ONGPIS_SMALL_NBX;       // block rows at most (9)
constexpr int kBStride = 36;                 // floats per row of a B tile in LDS (as in ongpis_test.hip)
constexpr int kBTile = 32 * kBStride;
constexpr int kSG = 8;                       // consecutive tiles per workgroup (a cluster of 64 queries is one workgroup)
__device__ __forceinline__ int chains_W(int nbx) { return nbx <= 4 ? 1 : (nbx <= 8 ? 2 : 4); }   // oracle gp.hpp OnGPIS::chains_W
}  // namespace

static size_t small_lds(int maxN, int maxLd, bool table) {
    return sizeof(int) * (size_t)maxLd + 16 * (size_t)maxN + sizeof(float) * (2 * kSNB * kBTile + kSNB * 1024 + 5 * 32) + 64 +
           sizeof(float4) * 8 * kSG + sizeof(int) * 2 * kSG + (table ? 2 * sizeof(double) * (size_t)maxN * 9 : 0);
}
size_t ongpis_eval_small_lds(int maxN, int maxLd) { return small_lds(maxN, maxLd, false); }

__global__ __launch_bounds__(64 * kSW, 2) void ongpis_eval_small_kernel(EvalArgs A, int ntiles, int G, int maxN, int maxLd, int use_tab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    const int t0 = blockIdx.x * G, t1 = min(ntiles, t0 + G);
    // LDS carve
    int* s_ri = reinterpret_cast<int*>(smem);                          // [maxLd] row -> point | component
    float4* s_x4 = reinterpret_cast<float4*>(s_ri + maxLd);            // [maxN]
    float* Bbuf = reinterpret_cast<float*>(s_x4 + maxN);               // [2][kSNB][32 * 36]
    float* Vbuf = Bbuf + 2 * kSNB * kBTile;                            // [kSNB][16][64]: finished block rows of V, accumulator layout
    float* red = Vbuf + kSNB * 1024;                                   // [4][32] sums of squares per chain wavefront, then [32] means
    float4* s_q = reinterpret_cast<float4*>(red + 5 * 32 + 16);        // [kSG][8] the query points of this workgroup's tiles
    int* s_cnt = reinterpret_cast<int*>(s_q + 8 * kSG);                // [kSG] queries per tile, [kSG] first job of the tile
    double* etab = reinterpret_cast<double*>(s_cnt + 2 * kSG);         // [2][maxN][9] exp(-a r) per (training point, query) of a tile (optional)
    // every query point and the tile table of this workgroup into LDS once: the steps below never wait for global memory
    // (two wavefronts per SIMD hide nothing)
    for (int i = tid; i < 8 * (t1 - t0); i += 64 * kSW) {
        const int tl = i >> 3, q = i & 7;
        const int jo = A.tile_off[t0 + tl], jc = A.tile_cnt[t0 + tl];
        s_q[i] = (q < jc) ? A.xq[A.job_q[jo + q]] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (q == 0) { s_cnt[tl] = jc; s_cnt[kSG + tl] = jo; }
    }

    int N = 0, K = 0, nb = 0, nbx = 0, dim = 3;
    float a = 0.f, scale = 1.f;
    int brow = -1, brow2 = -1;       // this wavefront's block rows (brow2: row 0, carried by the owner of row 1)
    float xa[kSNB - 1][16];          // X tiles (brow, c), c = 0 .. min(brow, 7), A-operand order
    float xs[16];                    // one more tile: (8, 8) on the owner of row 8, (0, 0) on the wavefront that also owns row 0
    const ClusterModel* mp = nullptr;

    // exp table of a query tile: one entry per (training point, query) -- the four kernel rows of a point share it
    auto fill_table = [&](int tl, double* tab) __attribute__((always_inline)) {
        const int jcnt = s_cnt[tl];
        for (int idx = tid; idx < N * 8; idx += 64 * kSW) {
            const int p = idx >> 3, q = idx & 7;
            double e = 0.0;
            if (q < jcnt) {
                const float4 xp = s_x4[p];
                const float4 xq = s_q[8 * tl + q];
                const float d0 = xp.x - xq.x, d1 = xp.y - xq.y, d2 = xp.z - xq.z;
                const float r = (dim == 3) ? sqrtf((d0 * d0 + d1 * d1) + d2 * d2) : sqrtf(d0 * d0 + d1 * d1);
                e = exp((double)(-a * r));
            }
            tab[p * 9 + q] = e;
        }
    };
    // B of a query tile (column blocks 0 .. cmax): task (c, r, qh) produces the 16 entries of row r of block c for the queries
    // 4 qh .. 4 qh + 3 -- the formulas and operand order of ongpis_eval_kernel's emit_rows (covFnc.cpp:258-314); the
    // 64 (cmax + 1) tasks are dealt to the 512 lanes round-robin (9 blocks: 1.125 tasks per lane)
    auto gen_B = [&](int cmaxb, int tl, float* slot, const double* tab) __attribute__((always_inline)) {
        const int jcnt = s_cnt[tl];
        for (int task = tid; task < 64 * (cmaxb + 1); task += 64 * kSW) {
            const int c = task >> 6, rr_ = task & 31, qh = (task >> 5) & 1;
            const int row = c * 32 + rr_;
            float4* trow = reinterpret_cast<float4*>(slot + c * kBTile + rr_ * kBStride + 16 * qh);
            int cr = 0, p = 0;
            float4 xp = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < K) {
                const int info = s_ri[row];
                cr = (info >> 28) & 0xF;
                p = info & 0x0FFFFFFF;
                xp = s_x4[p];
            }
#pragma unroll 2
            for (int j = 0; j < 4; ++j) {
                const int q = 4 * qh + j;
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < K && q < jcnt) {
                    const float4 xq = s_q[8 * tl + q];
                    float d[3] = {xp.x - xq.x, xp.y - xq.y, xp.z - xq.z};
                    float rr = (dim == 3) ? sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) : sqrtf(d[0] * d[0] + d[1] * d[1]);
                    double e = tab ? tab[p * 9 + q] : exp((double)(-a * rr));
                    float v0, v1, v2, v3;
                    if (cr == 0) {
                        v0 = d_kf(rr, a, e); v1 = d_kf1(d[0], a, e); v2 = d_kf1(d[1], a, e); v3 = d_kf1(d[2], a, e);
                    } else {
                        const float dr = cr == 1 ? d[0] : (cr == 2 ? d[1] : d[2]);
                        v0 = -d_kf1(dr, a, e);
                        v1 = (cr == 1) ? d_kf2(rr, d[0], d[0], 1.0f, a, e) : d_kf2(rr, d[0], dr, 0.0f, a, e);
                        v2 = (cr == 2) ? d_kf2(rr, d[1], d[1], 1.0f, a, e)
                                       : (cr == 1 ? d_kf2(rr, d[0], d[1], 0.0f, a, e) : d_kf2(rr, d[1], d[2], 0.0f, a, e));
                        v3 = (cr == 3) ? d_kf2(rr, d[2], d[2], 1.0f, a, e) : d_kf2(rr, dr, d[2], 0.0f, a, e);
                    }
                    if (dim == 2) v3 = 0.f;
                    o = make_float4(v0, v1, v2, v3);
                }
                trow[j] = o;
            }
        }
    };
    int tr = t0;
    while (tr < t1) {
        // ---- a run [ra, rb) of consecutive tiles of ONE cluster: vectors into LDS, X into the registers
        const int ra = tr, model = A.tile_model[ra];
        int rb = ra + 1;
        while (rb < t1 && A.tile_model[rb] == model) ++rb;
        tr = rb;
        __syncthreads();              // (the previous run's last generation / output is complete)
        mp = A.models + model;
        N = mp->N; K = mp->K; nb = mp->nb; dim = mp->dim; scale = mp->scale;
        const int ld = mp->ld;
        nbx = ld >> 5;
        a = (float)(sqrt(3.0) / (double)scale);
        {
            giptr_s g_ri = (giptr_s)mp->rowinfo;
            gfptr_s g_x4 = (gfptr_s)mp->x4;
            for (int i = tid; i < ld; i += 64 * kSW) s_ri[i] = g_ri[i];
            for (int i = tid; i < 4 * N; i += 64 * kSW) reinterpret_cast<float*>(s_x4)[i] = g_x4[i];
        }
        // row ownership: the i-th largest row goes to wavefront i (i < 4) or 11 - i (4 <= i < 8); row 0 of a 9-row cluster rides
        // with row 1 on wavefront 4
        {
            const int i = wave < 4 ? wave : 11 - wave;
            brow = nbx - 1 - i;
            if (brow < 0) brow = -1;
            brow2 = (wave == 4 && nbx == kSNB) ? 0 : -1;
        }
        const int cmax = min(nbx - 1, nb - 1);                // last column block a row multiplies with
        {
            const int ntl = nbx * (nbx + 1) / 2;
            const __amdgpu_buffer_rsrc_t Xrs = __builtin_amdgcn_make_buffer_rsrc((void*)mp->Xt, 0, (unsigned)ntl * 4096u, 0x00020000);
#pragma unroll
            for (int c = 0; c < kSNB - 1; ++c) {
                if (brow >= 0 && c <= min(brow, cmax)) {
                    const int sbase = (brow * (brow + 1) / 2 + c) * 4096;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        auto q = __builtin_amdgcn_raw_buffer_load_b128(Xrs, lane * 16, sbase + g * 1024, 0);
                        xa[c][4 * g] = __uint_as_float(q[0]); xa[c][4 * g + 1] = __uint_as_float(q[1]);
                        xa[c][4 * g + 2] = __uint_as_float(q[2]); xa[c][4 * g + 3] = __uint_as_float(q[3]);
                    }
                }
            }
            // the extra tile: (8, 8) of a 9-row cluster whose last row has a pivot block (K not a multiple of 32), or (0, 0)
            const int xtile = (brow2 == 0) ? 0 : ((brow == kSNB - 1 && cmax == kSNB - 1) ? (brow * (brow + 1) / 2 + kSNB - 1) : -1);
            if (xtile >= 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    auto q = __builtin_amdgcn_raw_buffer_load_b128(Xrs, lane * 16, xtile * 4096 + g * 1024, 0);
                    xs[4 * g] = __uint_as_float(q[0]); xs[4 * g + 1] = __uint_as_float(q[1]);
                    xs[4 * g + 2] = __uint_as_float(q[2]); xs[4 * g + 3] = __uint_as_float(q[3]);
                }
            }
        }
        __syncthreads();
        const bool tab_on = use_tab != 0;
        if (tab_on) {
            fill_table(ra - t0, etab + (size_t)(ra & 1) * maxN * 9);
            __syncthreads();
        }
        const bool gen_first = wave < kSW / 2;
        const int W = chains_W(nbx);
        // ---- software pipeline over the run: step s multiplies tile s (s >= ra) and generates tile s + 1 (s + 1 < rb) into
        // the other ring slot -- half of the wavefronts generate first, half multiply first, so that on every SIMD one
        // wavefront feeds the vector ALU while its partner feeds the matrix pipe.  One copy of each code path.
#pragma unroll 1
        for (int s = ra - 1; s < rb; ++s) {
            const bool do_m = s >= ra, do_g = s + 1 < rb;
#pragma unroll 1
            for (int ph = 0; ph < 2; ++ph) {
                if ((ph == 0) == gen_first) {
                    if (do_g) {     // B of tile s + 1 (its exp table was filled one step earlier), then the table of tile s + 2
                        gen_B(cmax, s + 1 - t0, Bbuf + ((s + 1) & 1) * kSNB * kBTile, tab_on ? etab + (size_t)((s + 1) & 1) * maxN * 9 : nullptr);
                        if (tab_on && s + 2 < rb) fill_table(s + 2 - t0, etab + (size_t)(s & 1) * maxN * 9);
                    }
                } else if (do_m) {
                    // V(b, :) = sum_c X(b, c) B_c, ascending c, from zero
                    const float* Bt = Bbuf + (s & 1) * kSNB * kBTile;
                    // the 16 B operands of a product (rows 2 kk + h of column l31) are fetched one product ahead: with two
                    // wavefronts per SIMD nothing else hides the LDS latency in front of every matrix instruction
                    auto loadB = [&](float (&bv)[16], int c) __attribute__((always_inline)) {
                        const float* Bl = Bt + c * kBTile + h * kBStride + l31;
#pragma unroll
                        for (int kk = 0; kk < 16; ++kk) bv[kk] = Bl[kk * 2 * kBStride];
                    };
                    if (brow >= 0) {
                        const int cend = min(brow, cmax);
                        f32x16 acc;
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                        float bv0[16], bv1[16];
                        loadB(bv0, 0);
#pragma unroll
                        for (int c = 0; c < kSNB - 1; ++c) {
                            if (c <= cend) {
                                if (c + 1 <= cend) { if (c & 1) loadB(bv0, c + 1); else loadB(bv1, c + 1); }
#pragma unroll
                                for (int kk = 0; kk < 16; ++kk)
                                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[c][kk], (c & 1) ? bv1[kk] : bv0[kk], acc, 0, 0, 0);
                            }
                        }
                        if (cend == kSNB - 1) {     // ninth product of a 9-tile row: the extra tile; its operands are in bv0 (8 is even)
#pragma unroll
                            for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[kk], bv0[kk], acc, 0, 0, 0);
                        }
                        float* vb = Vbuf + brow * 1024;
#pragma unroll
                        for (int r = 0; r < 16; ++r) vb[r * 64 + lane] = acc[r];
                    }
                    if (brow2 == 0) {
                        f32x16 acc;
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                        float bv0[16];
                        loadB(bv0, 0);
#pragma unroll
                        for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[kk], bv0[kk], acc, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 16; ++r) Vbuf[r * 64 + lane] = acc[r];
                    }
                }
            }
            __syncthreads();
            if (!do_m) continue;
            // ---- reduction in the order of the oracle's chains (gp.hpp reduce_ss): chain wavefront w < W takes the rows
            // i = tt W + (tt odd ? W-1-w : w), tt = 0..3 (i-th largest row), 16 registers per row and lane half
            if (wave < W) {
                float ssv = 0.f, mean_val = 0.f;
                const int kr = K & 31;
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const int i = tt * W + ((tt & 1) ? (W - 1 - wave) : wave);
                    const int b = nbx - 1 - i;
                    if (b >= 0) {
                        const bool has_mean = (tt == 0 && wave == 0);
                        const float* vb = Vbuf + b * 1024;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float v = vb[r * 64 + lane];
                            if (has_mean && rowmap_t(r, h) == kr) mean_val = v;
                            else ssv = fmaf(v, v, ssv);
                        }
                    }
                }
                ssv = ssv + __shfl_xor(ssv, 32);
                if (h == 0) red[wave * 32 + l31] = ssv;
                if (wave == 0 && h == ((K >> 2) & 1)) red[4 * 32 + l31] = mean_val;
            }
            __syncthreads();
            if (wave == 0) {
                const int joff = s_cnt[kSG + s - t0], jcnt = s_cnt[s - t0];
                const int col = lane;
                const int qi = col >> 2, cq = col & 3;
                if (col < 32 && qi < jcnt && cq <= dim) {
                    float vs = 0.f;
                    for (int w = 0; w < W; ++w) vs += red[w * 32 + col];
                    const float ms = red[4 * 32 + col];
                    float* o = A.out + (size_t)A.job_out[joff + qi] * 8;
                    const float tos = (float)(3.0 / (double)(scale * scale));  // OnGPIS.h:58
                    float var;
                    if (dim == 3)  // OnGPIS.cpp:208-213
                        var = (cq == 0) ? (float)(1.001 - (double)vs) : (float)((double)tos + 0.001 - (double)vs);
                    else           // OnGPIS.cpp:235-237
                        var = (cq == 0) ? (float)(1.01 - (double)vs) : (float)((double)tos + 0.1 - (double)vs);
                    o[cq] = ms;
                    o[4 + cq] = var;
                }
            }
            // (no third barrier: Vbuf is read only before the barrier above; `red` is rewritten only after the next step's
            // first barrier, which wavefront 0 reaches after this output)
        }
    }
}

int ongpis_eval_small_launch(int ntiles, int maxN, int maxLd, const EvalArgs& args, hipStream_t s) {
    if (ntiles <= 0) return GPIS_OK;
    if (maxLd / 32 > kSNB) return GPIS_ERR_ARG;
    const bool table = small_lds(maxN, maxLd, true) <= (size_t)160 * 1024;   // (its own exp table in LDS when it fits)
    const size_t lds = small_lds(maxN, maxLd, table);
    if (lds > 160 * 1024) return GPIS_ERR_LIMIT;
    if (ensure_dynamic_lds((const void*)ongpis_eval_small_kernel, 160 * 1024) != GPIS_OK) return GPIS_ERR_HIP;
    const int G = kSG;
    hipLaunchKernelGGL(ongpis_eval_small_kernel, dim3((ntiles + G - 1) / G), dim3(64 * kSW), lds, s, args, ntiles, G, maxN, maxLd, table ? 1 : 0);
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) {
        fprintf(stderr, "[gpismap_amd] small-cluster K4 launch failed: %s (%d tiles, LDS %zu B)\n", hipGetErrorString(le), ntiles, lds);
        return GPIS_ERR_HIP;
    }
    return GPIS_OK;
}

}  // namespace gpis
