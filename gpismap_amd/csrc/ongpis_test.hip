// K4: batched OnGPIS prediction -- mean (value + gradient) and the four variances
// for tiles of 8 queries against one cluster model.
//
// Replaces the reference per-point chain
//   GPisMap3::test_kernel        cpp/src/GPisMap3.cpp:794-902  (2-D: GPisMap.cpp:665-763)
//     -> OnGPIS::testSinglePoint cpp/src/OnGPIS.cpp:177-216    (2-D: test2Dpoint :218-263)
//        -> matern32_sparse_deriv1_3D (cross)  cpp/src/covFnc.cpp:258-314 (2-D: :404-450)
//        -> k*^T alpha ; L^-1 k* ; column sums of squares
//
// Work decomposition.  The (1+d) cross-covariance columns of 8 queries form a
// K x 32 right-hand-side block B.  One workgroup of W wavefronts solves
// L V = B by a right-looking 32-blocked forward substitution:
//   * block row b of B lives in the accumulator registers of wave (b mod W)
//     for the whole solve (f32x16 per 32x32 tile, MFMA C/D layout);
//   * step c: the owner of block c solves the 32x32 diagonal system in
//     registers (true divisions, ascending fmaf chain), publishes V_c to LDS;
//   * every wave then applies  B_b -= L_bc V_c  to its tiles with
//     v_mfma_f32_32x32x2_f32, streaming L_bc from L2/HBM exactly once.
// Each L element is read once per workgroup and used for 32 columns.  The
// per-element operation order is the ascending-k fmaf chain of dev_common.h.
#include "ongpis.h"

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
__device__ __forceinline__ void diag_solve32(float (&v)[16], const float* Lc, int h, int l31) {
    DiagCol cur, nxt;
    diag_load(cur, Lc, 0, h);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        if (i + 1 < 32) diag_load(nxt, Lc, i + 1, h);
        const int hi_ = (i >> 2) & 1, ri = (i & 3) + 4 * (i >> 3);
        float cand = v[ri] / cur.d;
        float vi = __shfl(cand, l31 + 32 * hi_);
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
        cur = nxt;
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int W, int NBW>
__global__ __launch_bounds__(64 * W, (W <= 4 ? 2 : 1)) void ongpis_eval_kernel(EvalArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tile = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const ClusterModel m = A.models[A.tile_model[tile]];
    const int N = m.N, K = m.K, ld = m.ld, nb = m.nb, dim = m.dim;
    const int joff = A.tile_off[tile], jcnt = A.tile_cnt[tile];

    // LDS carve (all dynamic, 16-byte aligned pieces)
    float* Vbuf = reinterpret_cast<float*>(smem);                 // [2][32*32]
    float* Lc = Vbuf + 2048;                                      // [32*32] column-major diag block
    float* red = Lc + 1024;                                       // [W][64][2]
    float* stage = red + W * 128;                                 // [W][16*64] per-lane staging strips
    double* etab = reinterpret_cast<double*>(stage + W * 1024);   // [N][8] (optional)

    const float a = (float)(sqrt(3.0) / (double)m.scale);
    const float4* x4 = reinterpret_cast<const float4*>(m.x4);

    // this lane's column: query slot qi, component cq
    const int qi = l31 >> 2, cq = l31 & 3;
    const bool qact = (qi < jcnt) && (cq <= dim);
    float4 xq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (qi < jcnt) xq = A.xq[A.job_q[joff + qi]];

    // ---- stage 1: exp table, one entry per (training point, query slot) ----
    if (A.use_table) {
        for (int idx = tid; idx < N * 8; idx += 64 * W) {
            int p = idx >> 3, s = idx & 7;
            double e = 0.0;
            if (s < jcnt) {
                float4 xp = x4[p];
                float4 q = A.xq[A.job_q[joff + s]];
                float d0 = xp.x - q.x, d1 = xp.y - q.y, d2 = xp.z - q.z;
                float r = (dim == 3) ? sqrtf((d0 * d0 + d1 * d1) + d2 * d2) : sqrtf(d0 * d0 + d1 * d1);
                e = exp((double)(-a * r));
            }
            etab[idx] = e;
        }
        __syncthreads();
    }

    // ---- stage 2: B tiles into accumulators + partial means ----
    // Entries are produced by a compact runtime loop into a per-lane LDS strip and
    // then moved to the (statically indexed) accumulator registers.
    f32x16 acc[NBW];
    float mp = 0.f;  // partial k*^T alpha over this lane's rows
    float* strip = stage + wave * 1024 + lane;  // strip[r*64]
#pragma unroll
    for (int t = 0; t < NBW; ++t) {
        const int b = wave + W * t;
        if (b < nb) {
#pragma unroll 1
            for (int r = 0; r < 16; ++r) {
                float v = 0.f;
                const int row = b * 32 + rowmap_t(r, h);
                if (qact && row < K) {
                    const int info = m.rowinfo[row];
                    const int p = info & 0x0FFFFFFF, cr = (info >> 28) & 0xF;
                    float4 xp = x4[p];
                    float d[3] = {xp.x - xq.x, xp.y - xq.y, xp.z - xq.z};
                    float rr = (dim == 3) ? sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) : sqrtf(d[0] * d[0] + d[1] * d[1]);
                    double e = A.use_table ? etab[p * 8 + qi] : exp((double)(-a * rr));
                    if (cr == 0) {
                        v = (cq == 0) ? d_kf(rr, a, e) : d_kf1(cq == 1 ? d[0] : (cq == 2 ? d[1] : d[2]), a, e);
                    } else {
                        float dr = cr == 1 ? d[0] : (cr == 2 ? d[1] : d[2]);
                        if (cq == 0) v = -d_kf1(dr, a, e);
                        else {
                            int lo = min(cr, cq), hi = max(cr, cq);
                            float dlo = lo == 1 ? d[0] : (lo == 2 ? d[1] : d[2]);
                            float dhi = hi == 1 ? d[0] : (hi == 2 ? d[1] : d[2]);
                            v = d_kf2(rr, dlo, dhi, lo == hi ? 1.0f : 0.0f, a, e);
                        }
                    }
                    mp = fmaf(v, m.alpha[row], mp);
                }
                strip[r * 64] = v;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = strip[r * 64];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- stage 3: blocked forward substitution ----
    float ss = 0.f;  // partial sum of squares of V over this lane's rows
#pragma unroll 1
    for (int c = 0; c < nb; ++c) {
        const int wc = c % W, tc = c / W;
        float* Vb = Vbuf + (c & 1) * 1024;
        if (wave == wc) {
            // diagonal block -> LDS (column-major), coalesced by column
            const float* Ld = m.L + (size_t)(c * 32 + l31) + (size_t)(c * 32) * ld;
#pragma unroll
            for (int cc = 0; cc < 16; ++cc) Lc[(2 * cc + h) * 32 + l31] = Ld[(size_t)(2 * cc + h) * ld];
            float v[16];
#pragma unroll
            for (int t = 0; t < NBW; ++t)
                if (t == tc) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = acc[t][r];
                }
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
            diag_solve32(v, Lc, h, l31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rw = rowmap_t(r, h);
                Vb[rw * 32 + l31] = v[r];
                if (c * 32 + rw < K) ss = fmaf(v[r], v[r], ss);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NBW; ++t) {
            const int b = wave + W * t;
            if (b > c && b < nb) {
                const float* Lp = m.L + (size_t)(b * 32 + l31) + (size_t)(c * 32 + h) * ld;
                float av[16];
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) av[kk] = Lp[(size_t)(2 * kk) * ld];
#pragma unroll
                for (int kk = 0; kk < 16; ++kk)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(-av[kk], Vb[(2 * kk + h) * 32 + l31], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- stage 4: reduce partials (lane halves, then waves in fixed order) ----
    mp = mp + __shfl_xor(mp, 32);
    ss = ss + __shfl_xor(ss, 32);
    if (h == 0) { red[(wave * 32 + l31) * 2] = mp; red[(wave * 32 + l31) * 2 + 1] = ss; }
    __syncthreads();
    if (wave == 0 && h == 0 && qact) {
        float ms = 0.f, vs = 0.f;
        for (int w = 0; w < W; ++w) { ms += red[(w * 32 + l31) * 2]; vs += red[(w * 32 + l31) * 2 + 1]; }
        float* o = A.out + (size_t)A.job_out[joff + qi] * 8;
        const float tos = (float)(3.0 / (double)(m.scale * m.scale));  // OnGPIS.h:58
        float var;
        if (dim == 3)  // OnGPIS.cpp:208-213
            var = (cq == 0) ? (float)(1.001 - (double)vs) : (float)((double)tos + 0.001 - (double)vs);
        else           // OnGPIS.cpp:235-237
            var = (cq == 0) ? (float)(1.01 - (double)vs) : (float)((double)tos + 0.1 - (double)vs);
        o[cq] = ms;
        o[4 + cq] = var;
    }
}

static size_t eval_lds_bytes(int W, int maxN, int use_table) {
    return sizeof(float) * (2048 + 1024 + W * 128 + W * 1024) + (use_table ? sizeof(double) * 8 * (size_t)maxN : 0);
}

// wclass: 0 -> nb <= 8 (1 wave), 1 -> nb <= 32 (4 waves), 2 -> nb <= 64 (8 waves), 3 -> nb <= 96 (16 waves)
int ongpis_eval_launch(int wclass, int ntiles, int maxN, const EvalArgs& args_in, hipStream_t s) {
    if (ntiles <= 0) return GPIS_OK;
    EvalArgs args = args_in;
    const int Ws[4] = {1, 4, 8, 8};
    int W = Ws[wclass];
    args.use_table = 1;
    size_t lds = eval_lds_bytes(W, maxN, 1);
    if (lds > 96 * 1024) { args.use_table = 0; lds = eval_lds_bytes(W, maxN, 0); }
    switch (wclass) {
        case 0: hipLaunchKernelGGL((ongpis_eval_kernel<1, 8>), dim3(ntiles), dim3(64), lds, s, args); break;
        case 1: hipLaunchKernelGGL((ongpis_eval_kernel<4, 8>), dim3(ntiles), dim3(256), lds, s, args); break;
        case 2: hipLaunchKernelGGL((ongpis_eval_kernel<8, 8>), dim3(ntiles), dim3(512), lds, s, args); break;
        case 3: hipLaunchKernelGGL((ongpis_eval_kernel<8, 12>), dim3(ntiles), dim3(512), lds, s, args); break;
        default: return GPIS_ERR_ARG;
    }
    return hipGetLastError() == hipSuccess ? GPIS_OK : GPIS_ERR_HIP;
}

int ongpis_eval_class(int nb) {
    if (nb <= 8) return 0;
    if (nb <= 32) return 1;
    if (nb <= 64) return 2;
    if (nb <= 96) return 3;
    return -1;
}

}  // namespace gpis
