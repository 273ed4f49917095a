// K4 for SMALL clusters (at most 9 block rows: K <= 287, e.g. the 64-point / K = 256 clusters of BASELINE config 5):
// the explicit inverse X stays RESIDENT in registers while a workgroup walks through consecutive 8-query tiles of
// the same cluster.
//
// Same operator as ongpis_eval_kernel (ongpis_test.hip; reference OnGPIS::testSinglePoint cpp/src/OnGPIS.cpp:177-216,
// cross covariance cpp/src/covFnc.cpp:258-314 / :404-450), same arithmetic element for element: V = X B with every element
// one fmaf chain from zero over ascending k, the mean in row K of V, the sums of squares in the (wavefront, lane half,
// row) chains of the oracle's reduce_ss -- results are bit-identical to the large-cluster kernel and to the oracle.
//
// Why a second kernel: a K = 256 cluster is 44 tile products per 8 queries (11 k cycles of a CU's matrix pipes) against
// ~38 k cycles of fixed cost per workgroup in the general kernel (cluster vectors, exp table, first chunk, reduction,
// every X tile streamed from L2 again for every 8 queries): 35 % of the fp32 matrix peak.  Here
//   * a workgroup of 8 wavefronts takes G consecutive tiles of the tile list (sorted by cluster): X (<= 45 tiles = 180 KB)
//     is loaded into registers once per cluster change -- block row b belongs to ONE wavefront (rows 8..5 on wavefronts
//     0..3, rows 4..1 on wavefronts 7..4, row 0 with row 1: the two wavefronts of a SIMD carry 12 / 11 / 11 / 11 products);
//   * the whole B (<= 9 tiles) of a query tile fits LDS twice: the wavefronts generate tile t+1 while they multiply tile t
//     (half of them before, half after -- vector and matrix work overlap on every SIMD), one barrier per tile;
//   * finished rows of V go through LDS to the reduction, which walks them in the order of the oracle's chains, so the row
//     -> wavefront assignment above is free of the summation order.
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
__device__ __forceinline__ int chains_W(int nbx) { return nbx <= 4 ? 1 : (nbx <= 8 ? 2 : 4); }   // oracle gp.hpp OnGPIS::chains_W
}  // namespace

size_t ongpis_eval_small_lds(int maxN, int maxLd) {
    return sizeof(float4) * 16 + sizeof(int) * (size_t)maxLd + 16 * (size_t)maxN + sizeof(float) * (2 * kSNB * kBTile + kSNB * 1024 + 5 * 32) + 64;
}

__global__ __launch_bounds__(64 * kSW, 2) void ongpis_eval_small_kernel(EvalArgs A, int ntiles, int G, int maxN, int maxLd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    const int t0 = blockIdx.x * G, t1 = min(ntiles, t0 + G);
    // LDS carve
    float4* s_xq = reinterpret_cast<float4*>(smem);                    // [2][8] query points of the tile being generated / multiplied
    int* s_ri = reinterpret_cast<int*>(s_xq + 16);                     // [maxLd] row -> point | component
    float4* s_x4 = reinterpret_cast<float4*>(s_ri + maxLd);            // [maxN]
    float* Bbuf = reinterpret_cast<float*>(s_x4 + maxN);               // [2][kSNB][32 * 36]
    float* Vbuf = Bbuf + 2 * kSNB * kBTile;                            // [kSNB][16][64]: finished block rows of V, accumulator layout
    float* red = Vbuf + kSNB * 1024;                                   // [4][32] sums of squares per chain wavefront, then [32] means

    int cur = -1;                    // model whose vectors are staged and whose X sits in the registers
    int N = 0, K = 0, nb = 0, nbx = 0, dim = 3, ngr = 0;
    float a = 0.f, scale = 1.f;
    int brow = -1, brow2 = -1;       // this wavefront's block rows (brow2: row 0, carried by the owner of row 1)
    float xa[kSNB][16];              // X tiles (brow, c), c = 0 .. brow, A-operand order
    float xs[16];                    // X tile (0, 0) for the wavefront that also owns row 0
    const ClusterModel* mp = nullptr;

    auto row_type = [&](int r) { return r < N ? 0 : 1 + (r - N) / (ngr > 0 ? ngr : 1); };
    // one B tile (column block c) of a query tile: lane (r, qh) produces the 16 entries of tile row r for the queries
    // 4 qh .. 4 qh + 3 -- the formulas and operand order of ongpis_eval_kernel's emit_rows (covFnc.cpp:258-314)
    auto gen_tile = [&](int c, const float4* xq8, int jcnt, float* tbuf) __attribute__((always_inline)) {
        const int rr_ = lane & 31, qh = lane >> 5;
        const int row = c * 32 + rr_;
        float4* trow = reinterpret_cast<float4*>(tbuf + rr_ * kBStride + 16 * qh);
        int cr = 0;
        float4 xp = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < K) {
            const int info = s_ri[row];
            cr = (info >> 28) & 0xF;
            xp = s_x4[info & 0x0FFFFFFF];
        }
#pragma unroll 2
        for (int j = 0; j < 4; ++j) {
            const int q = 4 * qh + j;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < K && q < jcnt) {
                const float4 xq = xq8[q];
                float d[3] = {xp.x - xq.x, xp.y - xq.y, xp.z - xq.z};
                float rr = (dim == 3) ? sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) : sqrtf(d[0] * d[0] + d[1] * d[1]);
                double e = exp((double)(-a * rr));
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
    };
    // B of tile t into ring slot t & 1: column block j is made by wavefront j % 8 (the ninth by wavefront 0)
    auto generate = [&](int t) __attribute__((always_inline)) {
        const int joff = A.tile_off[t], jcnt = A.tile_cnt[t];
        float4* xq8 = s_xq + 8 * (t & 1);
        // (the 8 query points: read straight from global by every lane that needs them would cost more than this hop)
        if (wave == 0 && lane < 8) xq8[lane] = (lane < jcnt) ? A.xq[A.job_q[joff + lane]] : make_float4(0.f, 0.f, 0.f, 0.f);
        (void)xq8;
    };

    for (int t = t0; t < t1; ++t) {
        const int model = A.tile_model[t];
        if (model != cur) {
            // ---- cluster change: vectors into LDS, X into the registers (nobody is reading the old vectors: the previous
            // tile's generation finished before the barrier that ended its iteration)
            __syncthreads();
            cur = model;
            mp = A.models + model;
            N = mp->N; K = mp->K; nb = mp->nb; dim = mp->dim; scale = mp->scale;
            const int ld = mp->ld;
            nbx = ld >> 5;
            ngr = (dim > 0) ? (K - N) / dim : 0;
            a = (float)(sqrt(3.0) / (double)scale);
            giptr_s g_ri = (giptr_s)mp->rowinfo;
            gfptr_s g_x4 = (gfptr_s)mp->x4;
            for (int i = tid; i < ld; i += 64 * kSW) s_ri[i] = g_ri[i];
            for (int i = tid; i < 4 * N; i += 64 * kSW) reinterpret_cast<float*>(s_x4)[i] = g_x4[i];
            // row ownership: the i-th largest row goes to wavefront i (i < 4) or 11 - i (4 <= i < 8); row 0 (i = 8) rides with row 1
            {
                const int i = wave < 4 ? wave : 11 - wave;
                brow = nbx - 1 - i;
                brow2 = (wave == 4 && nbx == kSNB) ? 0 : -1;
                if (brow < 0) brow = -1;
            }
            const int ntl = nbx * (nbx + 1) / 2;
            const __amdgpu_buffer_rsrc_t Xrs = __builtin_amdgcn_make_buffer_rsrc((void*)mp->Xt, 0, (unsigned)ntl * 4096u, 0x00020000);
            const int cmax = min(nbx - 1, nb - 1);            // last column block a row multiplies with
#pragma unroll
            for (int c = 0; c < kSNB; ++c) {
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
            if (brow2 == 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    auto q = __builtin_amdgcn_raw_buffer_load_b128(Xrs, lane * 16, g * 1024, 0);
                    xs[4 * g] = __uint_as_float(q[0]); xs[4 * g + 1] = __uint_as_float(q[1]);
                    xs[4 * g + 2] = __uint_as_float(q[2]); xs[4 * g + 3] = __uint_as_float(q[3]);
                }
            }
            __syncthreads();
            // first tile of this cluster: generated without overlap
            generate(t);
            __syncthreads();
            {
                const int jcnt = A.tile_cnt[t];
                for (int c = wave; c <= min(nbx - 1, nb - 1); c += kSW) gen_tile(c, s_xq + 8 * (t & 1), jcnt, Bbuf + ((t & 1) * kSNB + c) * kBTile);
            }
            __syncthreads();
        }
        const int cmax = min(nbx - 1, nb - 1);
        const bool next_same = (t + 1 < t1) && (A.tile_model[t + 1] == cur);
        const bool gen_first = wave < kSW / 2;
        auto gen_next = [&]() __attribute__((always_inline)) {
            const int jn = A.tile_cnt[t + 1];
            for (int c = wave; c <= cmax; c += kSW) gen_tile(c, s_xq + 8 * ((t + 1) & 1), jn, Bbuf + (((t + 1) & 1) * kSNB + c) * kBTile);
        };
        if (next_same) generate(t + 1);      // the next tile's query points (one barrier before they are used: see below)
        // ---- multiply tile t: V(brow, :) = sum_c X(brow, c) B_c, ascending c, from zero
        auto multiply = [&](int b, float (*xt)[16], float* xsingle) __attribute__((always_inline)) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* Bt = Bbuf + (t & 1) * kSNB * kBTile;
#pragma unroll
            for (int c = 0; c < kSNB; ++c) {
                if (c <= min(b, cmax)) {
                    const float* Bl = Bt + c * kBTile + h * kBStride + l31;
#pragma unroll
                    for (int kk = 0; kk < 16; ++kk)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xsingle ? xsingle[kk] : xt[c][kk], Bl[kk * 2 * kBStride], acc, 0, 0, 0);
                }
            }
            float* vb = Vbuf + b * 1024;
#pragma unroll
            for (int r = 0; r < 16; ++r) vb[r * 64 + lane] = acc[r];
        };
        if (next_same) {
            // the query points of tile t+1 must be visible to every generating wavefront: they were written by wavefront 0
            // above; the generating code reads them only after this barrier
            __syncthreads();
        }
        if (gen_first && next_same) gen_next();
        if (brow >= 0) multiply(brow, xa, nullptr);
        if (brow2 == 0) multiply(0, nullptr, xs);
        if (!gen_first && next_same) gen_next();
        __syncthreads();
        // ---- reduction in the order of the oracle's chains (gp.hpp reduce_ss): chain wavefront w < W takes the rows
        // i = t W + (t odd ? W-1-w : w), t = 0..3 (i-th largest row), 16 registers per row and lane half
        const int W = chains_W(nbx);
        float ssv = 0.f, mean_val = 0.f;
        if (wave < W) {
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
            const int joff = A.tile_off[t], jcnt = A.tile_cnt[t];
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
        // (the next iteration's first barrier -- or the kernel end -- separates this read of `red` and Vbuf from their next writes:
        // multiply of tile t+1 writes Vbuf only after the barrier below)
        __syncthreads();
    }
}

int ongpis_eval_small_launch(int ntiles, int maxN, int maxLd, const EvalArgs& args, hipStream_t s) {
    if (ntiles <= 0) return GPIS_OK;
    if (maxLd / 32 > kSNB) return GPIS_ERR_ARG;
    const size_t lds = ongpis_eval_small_lds(maxN, maxLd);
    if (lds > 160 * 1024) return GPIS_ERR_LIMIT;
    if (ensure_dynamic_lds((const void*)ongpis_eval_small_kernel, 160 * 1024) != GPIS_OK) return GPIS_ERR_HIP;
    const int G = 8;     // consecutive tiles per workgroup (a cluster of 64 queries is one workgroup)
    hipLaunchKernelGGL(ongpis_eval_small_kernel, dim3((ntiles + G - 1) / G), dim3(64 * kSW), lds, s, args, ntiles, G, maxN, maxLd);
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) {
        fprintf(stderr, "[gpismap_amd] small-cluster K4 launch failed: %s (%d tiles, LDS %zu B)\n", hipGetErrorString(le), ntiles, lds);
        return GPIS_ERR_HIP;
    }
    return GPIS_OK;
}

}  // namespace gpis
