// K6 + K3: batched OnGPIS training, one workgroup per cluster.
//
// Replaces the reference loop
//   GPisMap3::updateGPs_kernel   cpp/src/GPisMap3.cpp:698-718   (2-D: GPisMap.cpp:574-594)
//     -> OnGPIS::train           cpp/src/OnGPIS.cpp:91-149      (2-D: :34-89)
//        -> matern32_sparse_deriv1_3D  cpp/src/covFnc.cpp:142-256 (2-D: :317-402)
//        -> K.llt(), two triangular solves (Eigen)   OnGPIS.cpp:139-143
//
//  gather : point ids (tree order) -> contiguous per-cluster batch, gradflag rule,
//           target vector y = [f; gx; gy; gz], row table.
//  buildK : lower triangle of K (column-major, ld), y appended as row K so the
//           forward substitution L z = y falls out of the factorisation.
//  chol   : right-looking 32-blocked Cholesky in HBM/L2; the trailing update is
//           v_mfma_f32_32x32x2_f32 (a k-ordered fmaf chain => same bits as the
//           unblocked chain order), then blocked backward substitution for alpha.
#include "ongpis.h"

namespace gpis {

#define JOB_MODEL(j) d_jobs[4 * (j) + 0]
#define JOB_OFF(j) d_jobs[4 * (j) + 1]
#define JOB_N(j) d_jobs[4 * (j) + 2]
#define JOB_NG(j) d_jobs[4 * (j) + 3]

// ---------------------------------------------------------------------------
// K6 gather.  grid = jobs, block = 256.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ongpis_gather_kernel(const ClusterModel* __restrict__ models,
                                                            const int* __restrict__ d_jobs,
                                                            const int* __restrict__ ids,
                                                            const float* __restrict__ pts, int cap) {
    __shared__ int cnt[256];
    const int job = blockIdx.x, tid = threadIdx.x;
    const ClusterModel m = models[JOB_MODEL(job)];
    const int off = JOB_OFF(job), N = m.N, dim = m.dim, ng = m.ng;
    const int chunk = (N + 255) / 256;
    const int k0 = tid * chunk, k1 = min(N, k0 + chunk);
    int c = 0;
    for (int k = k0; k < k1; ++k) {
        int id = ids[off + k];
        float px = pts[0 * (size_t)cap + id], py = pts[1 * (size_t)cap + id], pz = pts[2 * (size_t)cap + id];
        float gx = pts[3 * (size_t)cap + id], gy = pts[4 * (size_t)cap + id], gz = pts[5 * (size_t)cap + id];
        float val = pts[6 * (size_t)cap + id], sx = pts[7 * (size_t)cap + id], sg = pts[8 * (size_t)cap + id];
        bool tiny = ((double)fabsf(gx) < 1e-6) && ((double)fabsf(gy) < 1e-6) && (dim == 2 || (double)fabsf(gz) < 1e-6);
        bool flag = !(((double)sg > 0.1001) || tiny);  // OnGPIS.cpp:122-125
        reinterpret_cast<float4*>(m.x4)[k] = make_float4(px, py, dim == 3 ? pz : 0.f, 0.f);
        m.sig[k] = flag ? sx : 2.0f;
        m.sig[N + k] = sg;
        m.y[k] = val;
        m.rowinfo[k] = k;
        m.gidx[k] = flag ? 1 : -1;
        c += flag;
    }
    cnt[tid] = c;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < 256; ++i) { int t = cnt[i]; cnt[i] = run; run += t; }
    }
    __syncthreads();
    int g = cnt[tid];
    for (int k = k0; k < k1; ++k) {
        if (m.gidx[k] > 0) {
            int id = ids[off + k];
            m.gidx[k] = g;
            for (int cc = 0; cc < dim; ++cc) {
                int row = N + cc * ng + g;
                m.y[row] = pts[(3 + cc) * (size_t)cap + id];
                m.rowinfo[row] = k | ((cc + 1) << 28);
            }
            ++g;
        }
    }
    for (int r = m.K + tid; r < m.ld; r += 256) { m.rowinfo[r] = 0xF << 28; m.y[r] = 0.f; m.alpha[r] = 0.f; }
}

// ---------------------------------------------------------------------------
// Kernel matrix.  grid = jobs, block = 1024.  Entry formulas: covFnc.cpp:165-253
// (3-D) / :340-399 (2-D), lower triangle only, same operand order as the
// reference (delta = x_k - x_j with k < j; mixed second derivatives computed
// once with the lower component first and mirrored).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void putL(float* L, int ld, int r, int c, float v) {
    if (r >= c) L[r + (size_t)c * ld] = v; else L[c + (size_t)r * ld] = v;
}

__global__ __launch_bounds__(1024) void ongpis_buildK_kernel(const ClusterModel* __restrict__ models,
                                                             const int* __restrict__ d_jobs) {
    const int job = blockIdx.x, tid = threadIdx.x;
    const ClusterModel m = models[JOB_MODEL(job)];
    const int N = m.N, ng = m.ng, dim = m.dim, K = m.K, ld = m.ld;
    float* L = m.L;
    const float a = (float)(sqrt(3.0) / (double)m.scale);  // covFnc.cpp:147
    const float a2 = a * a;
    const float4* x4 = reinterpret_cast<const float4*>(m.x4);

    // padding rows K+1..ld-1: identity; row K: the target vector y (augmented row)
    for (int r = K + 1 + (tid >> 5); r < ld; r += 32)
        for (int c = (tid & 31); c <= r; c += 32) L[r + (size_t)c * ld] = (r == c) ? 1.f : 0.f;
    for (int c = tid; c < K; c += 1024) L[K + (size_t)c * ld] = m.y[c];
    if (tid == 0) L[K + (size_t)K * ld] = 1.f;

    const long long NN = (long long)N * N;
    for (long long idx = tid; idx < NN; idx += 1024) {
        int k = (int)(idx / N), j = (int)(idx % N);
        if (k > j) continue;
        int kg = m.gidx[k];
        int kind[3] = {N + kg, N + kg + ng, N + kg + 2 * ng};
        if (k == j) {
            L[k + (size_t)k * ld] = (float)(1.0 + (double)m.sig[k]);
            if (kg >= 0) {
                float sg = m.sig[N + k];
                for (int c = 0; c < dim; ++c) {
                    L[kind[c] + (size_t)k * ld] = 0.f;
                    for (int c2 = 0; c2 < c; ++c2) L[kind[c] + (size_t)kind[c2] * ld] = 0.f;
                }
                if (dim == 3) {
                    for (int c = 0; c < 3; ++c) L[kind[c] + (size_t)kind[c] * ld] = a2 + sg;
                } else {
                    L[kind[0] + (size_t)kind[0] * ld] = (float)((double)a2 + sqrt((double)(m.sig[k] * sg)));  // covFnc.cpp:352
                    L[kind[1] + (size_t)kind[1] * ld] = a2 + sg;
                }
            }
            continue;
        }
        float4 xk = x4[k], xj = x4[j];
        int jg = m.gidx[j];
        int jind[3] = {N + jg, N + jg + ng, N + jg + 2 * ng};
        float d[3] = {xk.x - xj.x, xk.y - xj.y, xk.z - xj.z};
        float r = (dim == 3) ? sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) : sqrtf(d[0] * d[0] + d[1] * d[1]);
        double e = exp((double)(-a * r));
        L[j + (size_t)k * ld] = d_kf(r, a, e);
        if (kg >= 0) {
            float g1[3];
            for (int c = 0; c < dim; ++c) { g1[c] = -d_kf1(d[c], a, e); L[kind[c] + (size_t)j * ld] = g1[c]; }
            if (jg >= 0) {
                for (int c = 0; c < dim; ++c) L[jind[c] + (size_t)k * ld] = -g1[c];
                for (int c1 = 0; c1 < dim; ++c1)
                    for (int c2 = c1; c2 < dim; ++c2) {
                        float v = d_kf2(r, d[c1], d[c2], c1 == c2 ? 1.0f : 0.0f, a, e);
                        putL(L, ld, kind[c1], jind[c2], v);
                        if (c2 != c1) putL(L, ld, kind[c2], jind[c1], v);
                    }
            }
        } else if (jg >= 0) {
            for (int c = 0; c < dim; ++c) L[jind[c] + (size_t)k * ld] = d_kf1(d[c], a, e);
        }
    }
}

// ---------------------------------------------------------------------------
// K3.  grid = jobs, block = 1024 (16 waves).  Factorises rows 0..K (row K = y).
// ---------------------------------------------------------------------------
__device__ __forceinline__ int rowmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__global__ __launch_bounds__(1024) void ongpis_chol_kernel(const ClusterModel* __restrict__ models,
                                                           const int* __restrict__ d_jobs) {
    __shared__ float D[32 * 33];
    __shared__ float av[32];
    const int job = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = 16;
    const ClusterModel m = models[JOB_MODEL(job)];
    const int K = m.K, ld = m.ld;
    float* L = m.L;
    const int nrows = K + 1;
    const int npan = (K + 31) / 32;
    const int nbr = ld / 32;           // block rows (ld = 32*ceil((K+1)/32))
    const int bjmax = (K - 1) / 32;    // last block column that holds real columns

    for (int p = 0; p < npan; ++p) {
        const int pr = 32 * p;
        const int pw = min(32, K - pr);
        // (a) diagonal block, wave 0, lane = row within block
        if (wave == 0) {
            if (lane < 32)
                for (int c = 0; c < 32; ++c) D[lane * 33 + c] = L[(pr + lane) + (size_t)(pr + c) * ld];
            __builtin_amdgcn_s_waitcnt(0);  // LDS writes of this wave done
            for (int j = 0; j < pw; ++j) {
                float d = sqrtf(D[j * 33 + j]);
                float lij = 0.f;
                bool below = (lane > j && lane < 32);
                if (below) lij = D[lane * 33 + j] / d;
                if (lane == j) D[j * 33 + j] = d;
                if (below) D[lane * 33 + j] = lij;
                if (below) {
                    float nl = -lij;
                    int kend = min(lane, pw - 1);
                    for (int k = j + 1; k <= kend; ++k) D[lane * 33 + k] = fmaf(nl, D[k * 33 + j], D[lane * 33 + k]);
                }
            }
            if (lane < 32)
                for (int c = 0; c < pw; ++c)
                    if (c <= lane) L[(pr + lane) + (size_t)(pr + c) * ld] = D[lane * 33 + c];
        }
        __syncthreads();
        // (b) panel solve: one thread per row below the diagonal block
        for (int i = pr + 32 + tid; i < nrows; i += 1024) {
            float x[32];
#pragma unroll
            for (int c = 0; c < 32; ++c) x[c] = (c < pw) ? L[i + (size_t)(pr + c) * ld] : 0.f;
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                if (c < pw) {
                    float s = x[c];
#pragma unroll
                    for (int k = 0; k < c; ++k) s = fmaf(-x[k], D[c * 33 + k], s);
                    x[c] = s / D[c * 33 + c];
                }
            }
#pragma unroll
            for (int c = 0; c < 32; ++c) if (c < pw) L[i + (size_t)(pr + c) * ld] = x[c];
        }
        __syncthreads();
        // (c) trailing update, 32x32 tiles, computed transposed so that lanes map to
        //     consecutive rows (coalesced C traffic): D'[i'][j'] = C[bi*32+j'][bj*32+i'].
        if (pw == 32) {
            int cntr = 0;
            const int h = lane >> 5, l31 = lane & 31;
            for (int bi = p + 1; bi < nbr; ++bi) {
                int bje = min(bi, bjmax);
                for (int bj = p + 1; bj <= bje; ++bj, ++cntr) {
                    if ((cntr % nwaves) != wave) continue;
                    f32x16 acc;
                    float* Cb = L + (size_t)(bi * 32 + l31) + (size_t)(bj * 32) * ld;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = Cb[(size_t)rowmap(r, h) * ld];
                    const float* Pa = L + (size_t)(bj * 32 + l31) + (size_t)(pr + h) * ld;
                    const float* Pb = L + (size_t)(bi * 32 + l31) + (size_t)(pr + h) * ld;
                    float av_[16], bv_[16];
#pragma unroll
                    for (int kk = 0; kk < 16; ++kk) { av_[kk] = Pa[(size_t)(2 * kk) * ld]; bv_[kk] = Pb[(size_t)(2 * kk) * ld]; }
#pragma unroll
                    for (int kk = 0; kk < 16; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(-av_[kk], bv_[kk], acc, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) Cb[(size_t)rowmap(r, h) * ld] = acc[r];
                }
            }
        }
        __syncthreads();
    }

    // z = row K of the factor -> y ; then alpha = L^-T z, blocked, chain order (O2)
    for (int j = tid; j < K; j += 1024) m.y[j] = L[K + (size_t)j * ld];
    __syncthreads();
    const int nb = m.nb;
    for (int c = nb - 1; c >= 0; --c) {
        const int cr = 32 * c;
        if (wave == 0) {
            if (lane < 32)
                for (int cc = 0; cc < 32; ++cc) D[lane * 33 + cc] = L[(cr + lane) + (size_t)(cr + cc) * ld];
            __builtin_amdgcn_s_waitcnt(0);
            float b = (lane < 32 && cr + lane < K) ? m.y[cr + lane] : 0.f;
            for (int k = 31; k >= 0; --k) {
                if (cr + k >= K) continue;
                float t = b / D[(lane & 31) * 33 + (lane & 31)];
                float ak = __shfl(t, k);
                if (lane == k) b = ak;
                if (lane < k) b = fmaf(-D[k * 33 + lane], ak, b);
            }
            if (lane < 32) {
                av[lane] = (cr + lane < K) ? b : 0.f;
                if (cr + lane < K) m.alpha[cr + lane] = b;
            }
        }
        __syncthreads();
        for (int j = tid; j < cr; j += 1024) {
            float s = m.y[j];
            const float* col = L + (size_t)cr + (size_t)j * ld;
            for (int k = 31; k >= 0; --k)
                if (cr + k < K) s = fmaf(-col[k], av[k], s);
            m.y[j] = s;
        }
        __syncthreads();
    }
    // restore the identity in row K so the padded square is a valid triangular factor
    for (int j = tid; j < K; j += 1024) L[K + (size_t)j * ld] = 0.f;
    if (tid == 0) L[K + (size_t)K * ld] = 1.f;
}

// ---------------------------------------------------------------------------
// Re-tile the factor for K4: block (b, c), b >= c, is stored as 1024 consecutive floats in the
// order the MFMA A operand consumes it -- [g][lane][j] holds L[32b + (lane&31)][32c + 2(4g+j) + (lane>>5)]
// -- so a lane fetches its 16 operands of a tile with four 16-byte loads.  grid = jobs, block = 256.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ongpis_tile_kernel(const ClusterModel* __restrict__ models,
                                                          const int* __restrict__ d_jobs) {
    const int job = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const ClusterModel m = models[JOB_MODEL(job)];
    const int nb = m.nb, ld = m.ld;
    const int ntiles = nb * (nb + 1) / 2;
    const int h = lane >> 5, l31 = lane & 31;
    for (int t = wave; t < ntiles; t += 4) {
        // t = b(b+1)/2 + c
        int b = (int)((sqrtf(8.f * t + 1.f) - 1.f) * 0.5f);
        while ((b + 1) * (b + 2) / 2 <= t) ++b;
        while (b * (b + 1) / 2 > t) --b;
        int c = t - b * (b + 1) / 2;
        const float* src = m.L + (size_t)(b * 32 + l31) + (size_t)(c * 32 + h) * ld;
        float4* dst = reinterpret_cast<float4*>(m.Lt + (size_t)t * 1024);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 q;
            q.x = src[(size_t)(2 * (4 * g + 0)) * ld];
            q.y = src[(size_t)(2 * (4 * g + 1)) * ld];
            q.z = src[(size_t)(2 * (4 * g + 2)) * ld];
            q.w = src[(size_t)(2 * (4 * g + 3)) * ld];
            dst[g * 64 + lane] = q;
        }
    }
}

void ongpis_launch_gather(const ClusterModel* d_models, const int* d_jobs, int njobs, const int* d_ids,
                          const float* d_pts, int pts_cap, hipStream_t s) {
    hipLaunchKernelGGL(ongpis_gather_kernel, dim3(njobs), dim3(256), 0, s, d_models, d_jobs, d_ids, d_pts, pts_cap);
}
void ongpis_launch_buildK(const ClusterModel* d_models, const int* d_jobs, int njobs, hipStream_t s) {
    hipLaunchKernelGGL(ongpis_buildK_kernel, dim3(njobs), dim3(1024), 0, s, d_models, d_jobs);
}
void ongpis_launch_tile(const ClusterModel* d_models, const int* d_jobs, int njobs, hipStream_t s) {
    hipLaunchKernelGGL(ongpis_tile_kernel, dim3(njobs), dim3(256), 0, s, d_models, d_jobs);
}
void ongpis_launch_chol(const ClusterModel* d_models, const int* d_jobs, int njobs, hipStream_t s) {
    hipLaunchKernelGGL(ongpis_chol_kernel, dim3(njobs), dim3(1024), 0, s, d_models, d_jobs);
}

}  // namespace gpis
