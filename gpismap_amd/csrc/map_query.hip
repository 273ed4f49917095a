// K5 and the test() pipeline on the device.
//
//   lookup : per query, candidate cluster cells whose box intersects the search box
//            (inclusive test, reference octree.h:128-135), squared centre distances
//            (octree.cpp:24-31), the three nearest in (distance, traversal order).
//            Replaces OcTree::QueryNonEmptyLevelC(range, quads, sqdst) octree.cpp:861-893
//            + the sort at GPisMap3.cpp:826-829 (2-D: GPisMap.cpp:695-698).
//   bin    : counting sort of evaluation jobs by cluster model (histogram, scan,
//            scatter) and tile list construction -- all on the device.
//   blend  : GPisMap3.cpp:816-898 / GPisMap.cpp:685-757 (nearest, fallback to the next two
//            when var > thre, pick or variance-weighted blend).
#include <algorithm>
#include <cstring>
#include "map_query.h"
#include "stdsort_emul.h"

namespace gpis {

// ------------------------------------------------------------------ lookup ----

template <int DIM>
__device__ __forceinline__ bool cell_hit(const ClusterTableView& T, int ci, const float* qlo, const float* qhi) {
    float4 lo = T.lo[ci], hi = T.hi[ci];
    bool hit = !((qhi[0] < lo.x) || (qlo[0] > hi.x) || (qhi[1] < lo.y) || (qlo[1] > hi.y));
    if (DIM == 3) hit = hit && !((qhi[2] < lo.z) || (qlo[2] > hi.z));
    // every ancestor must intersect too (top-down pruning of the reference's tree walk)
    for (int a = T.parent[ci]; hit && a >= 0; a = T.anc_parent[a]) {
        float4 alo = T.anc_lo[a], ahi = T.anc_hi[a];
        hit = !((qhi[0] < alo.x) || (qlo[0] > ahi.x) || (qhi[1] < alo.y) || (qlo[1] > ahi.y));
        if (DIM == 3) hit = hit && !((qhi[2] < alo.z) || (qlo[2] > ahi.z));
    }
    return hit;
}

template <int DIM>
__global__ __launch_bounds__(256) void lookup_kernel(ClusterTableView T, const float* __restrict__ x, int n,
                                                     float half, float4* __restrict__ xq4,
                                                     int* __restrict__ cand, int* __restrict__ ncand, int cap,
                                                     int* __restrict__ tie_count, int* __restrict__ tie_list) {
    int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    float p[3] = {x[(size_t)DIM * q], x[(size_t)DIM * q + 1], DIM == 3 ? x[(size_t)DIM * q + 2] : 0.f};
    xq4[q] = make_float4(p[0], p[1], p[2], 0.f);
    float qlo[3], qhi[3];
    int i0[3], i1[3];
    const double org[3] = {T.ox, T.oy, T.oz};
    const int gdim[3] = {T.gx, T.gy, T.gz};
    for (int d = 0; d < 3; ++d) {
        qlo[d] = p[d] - half; qhi[d] = p[d] + half;  // AABB3 ctor, octree.h:64-69
        double a = floor(((double)qlo[d] - org[d]) / T.pitch) - 1.0;
        double b = floor(((double)qhi[d] - org[d]) / T.pitch) + 1.0;
        a = fmax(a, 0.0); b = fmin(b, (double)(gdim[d] - 1));
        i0[d] = (int)a; i1[d] = (int)b;
        if (d >= DIM) { i0[d] = 0; i1[d] = 0; }
    }
    float bd[4] = {3.4e38f, 3.4e38f, 3.4e38f, 3.4e38f};
    int bi[4] = {-1, -1, -1, -1};
    int count = 0;
    for (int iz = i0[2]; iz <= i1[2]; ++iz)
        for (int iy = i0[1]; iy <= i1[1]; ++iy)
            for (int ix = i0[0]; ix <= i1[0]; ++ix) {
                int ci = T.grid[((size_t)iz * T.gy + iy) * T.gx + ix];
                if (ci < 0) continue;
                if (!cell_hit<DIM>(T, ci, qlo, qhi)) continue;
                float4 c = T.c[ci];
                float dx = c.x - p[0], dy = c.y - p[1], dz = c.z - p[2];
                float sd = (DIM == 3) ? (dx * dx + dy * dy) + dz * dz : dx * dx + dy * dy;
                ++count;
                // sorted top-4 by (sd, ci): ci = traversal rank
                int cj = ci; float sj = sd;
#pragma unroll
                for (int k = 0; k < 4; ++k) {     // (branch-free and fully unrolled: bd / bi stay in registers)
                    const bool better = (cj >= 0) && ((bi[k] < 0) || (sj < bd[k]) || (sj == bd[k] && cj < bi[k]));
                    const float ts = bd[k]; const int ti = bi[k];
                    bd[k] = better ? sj : ts; bi[k] = better ? cj : ti;
                    sj = better ? ts : sj; cj = better ? ti : cj;
                }
            }
    // A distance tie that can change the first three entries: reproduce std::sort exactly.
    const bool tie = (count > 1 && bd[1] == bd[0]) || (count > 2 && bd[2] == bd[1]) || (count > 3 && bd[3] == bd[2]);
    // (resolved by lookup_tie_kernel, which rewrites the three candidates of the listed queries: the emulation needs
    // ~2 KB of arrays per query, which would sit in scratch memory here)
    if (tie && count > 1 && count <= 128) tie_list[atomicAdd(tie_count, 1)] = q;
    int nc = count > 3 ? 3 : count;
    ncand[q] = nc;
#pragma unroll
    for (int k = 0; k < 3; ++k) cand[(size_t)k * cap + q] = (k < nc) ? T.model[bi[k]] : -1;
}

// Exact distance ties among the nearest candidates (lattice-aligned queries): the reference sorts the candidate cells
// with std::sort (GPisMap3.cpp:826-829), whose result on equal keys depends on the algorithm -- reproduced operation by
// operation (stdsort_emul.h).  One lane per listed query; the work arrays (candidates in traversal order, keys, the
// permutation, the explicit recursion stack: 480 words) live in LDS, lane-interleaved.  grid = any, the list is strided.
constexpr int kTieLanes = 32;
template <int DIM>
__global__ __launch_bounds__(kTieLanes) void lookup_tie_kernel(ClusterTableView T, const float* __restrict__ x, float half,
                                                               int* __restrict__ cand, int cap,
                                                               const int* __restrict__ tie_count, const int* __restrict__ tie_list) {
    __shared__ float s_key[128 * kTieLanes];
    __shared__ int s_rk[128 * kTieLanes], s_v[128 * kTieLanes], s_st[96 * kTieLanes];
    const int lane = threadIdx.x;
    const int ntie = *tie_count;
    for (int e = blockIdx.x * kTieLanes + lane; e < ntie; e += gridDim.x * kTieLanes) {
        const int q = tie_list[e];
        float p[3] = {x[(size_t)DIM * q], x[(size_t)DIM * q + 1], DIM == 3 ? x[(size_t)DIM * q + 2] : 0.f};
        float qlo[3], qhi[3];
        int i0[3], i1[3];
        const double org[3] = {T.ox, T.oy, T.oz};
        const int gdim[3] = {T.gx, T.gy, T.gz};
        for (int d = 0; d < 3; ++d) {
            qlo[d] = p[d] - half; qhi[d] = p[d] + half;
            double a = floor(((double)qlo[d] - org[d]) / T.pitch) - 1.0;
            double b = floor(((double)qhi[d] - org[d]) / T.pitch) + 1.0;
            a = fmax(a, 0.0); b = fmin(b, (double)(gdim[d] - 1));
            i0[d] = (int)a; i1[d] = (int)b;
            if (d >= DIM) { i0[d] = 0; i1[d] = 0; }
        }
        Strided<float> key{s_key + lane, kTieLanes};
        Strided<int> rk{s_rk + lane, kTieLanes}, v{s_v + lane, kTieLanes};
        Strided<int> stF{s_st + lane, kTieLanes}, stL{s_st + 32 * kTieLanes + lane, kTieLanes}, stD{s_st + 64 * kTieLanes + lane, kTieLanes};
        int n2 = 0;
        for (int iz = i0[2]; iz <= i1[2]; ++iz)
            for (int iy = i0[1]; iy <= i1[1]; ++iy)
                for (int ix = i0[0]; ix <= i1[0]; ++ix) {
                    int ci = T.grid[((size_t)iz * T.gy + iy) * T.gx + ix];
                    if (ci < 0) continue;
                    if (n2 >= 128 || !cell_hit<DIM>(T, ci, qlo, qhi)) continue;
                    float4 c = T.c[ci];
                    float dx = c.x - p[0], dy = c.y - p[1], dz = c.z - p[2];
                    float sd = (DIM == 3) ? (dx * dx + dy * dy) + dz * dz : dx * dx + dy * dy;
                    // insert by traversal rank (the order QueryNonEmptyLevelC returns the cells in)
                    int k = n2;
                    while (k > 0 && rk[k - 1] > ci) { rk[k] = rk[k - 1]; key[k] = key[k - 1]; --k; }
                    rk[k] = ci; key[k] = sd;
                    ++n2;
                }
        for (int k = 0; k < n2; ++k) v[k] = k;
        if (stdsort_emulate(key, v, n2, stF, stL, stD)) {
            for (int k = 0; k < 3 && k < n2; ++k) cand[(size_t)k * cap + q] = T.model[rk[v[k]]];
        }
    }
}

// pass-1 jobs: job j = query j, model = nearest candidate
__global__ void jobs_pass1_kernel(const int* __restrict__ cand, const int* __restrict__ ncand, int n, int* __restrict__ jm) {
    int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    jm[q] = (ncand[q] >= 1) ? cand[q] : -1;
}

// pass-2 jobs: job j = 2q+s (s = 0,1) -> candidate s+1 when the first variance exceeds thre
__global__ void jobs_pass2_kernel(const int* __restrict__ cand, const int* __restrict__ ncand, int n, int cap,
                                  const float* __restrict__ out, float var_thre, float prior_var, int vidx,
                                  int* __restrict__ jm) {
    int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    int nc = ncand[q];
    bool more = false;
    if (nc > 1) {
        float v0 = (cand[q] >= 0) ? out[(size_t)q * 8 + 4] : prior_var;
        more = v0 > var_thre;
    }
    jm[2 * q] = (more && nc >= 2) ? cand[(size_t)cap + q] : -1;
    jm[2 * q + 1] = (more && nc >= 3) ? cand[(size_t)2 * cap + q] : -1;
}

// ------------------------------------------------------------------ binning ----
__global__ void hist_kernel(const int* __restrict__ jm, int njobs, int* __restrict__ cnt) {
    int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= njobs) return;
    int m = jm[j];
    if (m >= 0) atomicAdd(&cnt[m], 1);
}

// Block-aggregated variants (model table fits in LDS): neighbouring queries mostly hit the same few
// clusters, so counting in LDS first turns millions of same-address global atomics into one per
// (block, cluster).  A wave whose active lanes all carry the same model adds its popcount once.
#define BIN_CHUNK 4096   // jobs per block
__device__ __forceinline__ void lds_count(int* sh, int m) {
    const bool act = m >= 0;
    const unsigned long long am = __builtin_amdgcn_ballot_w64(act);
    if (am == 0) return;
    const int lead = __builtin_ctzll(am);
    const int m0 = __builtin_amdgcn_readlane(m, lead);
    if (__builtin_amdgcn_ballot_w64(act && m == m0) == am) {
        if ((int)(threadIdx.x & 63) == lead) atomicAdd(&sh[m0], __builtin_popcountll(am));
    } else if (act) atomicAdd(&sh[m], 1);
}
__global__ __launch_bounds__(256) void hist_lds_kernel(const int* __restrict__ jm, int njobs, int nmodels,
                                                       int* __restrict__ cnt) {
    extern __shared__ int sh[];
    for (int i = threadIdx.x; i < nmodels; i += 256) sh[i] = 0;
    __syncthreads();
    const int j0 = blockIdx.x * BIN_CHUNK;
    for (int k = 0; k < BIN_CHUNK; k += 256) {
        const int j = j0 + k + threadIdx.x;
        lds_count(sh, j < njobs ? jm[j] : -1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nmodels; i += 256) { const int c = sh[i]; if (c) atomicAdd(&cnt[i], c); }
}
__global__ __launch_bounds__(256) void scatter_lds_kernel(const int* __restrict__ jm, int njobs, int nmodels, int shift,
                                                          int rec_base, const int* __restrict__ base,
                                                          int* __restrict__ cursor, int* __restrict__ jq,
                                                          int* __restrict__ jo) {
    extern __shared__ int sh[];          // [nmodels] running count, [nmodels] first slot of this block
    int* sb = sh + nmodels;
    for (int i = threadIdx.x; i < nmodels; i += 256) sh[i] = 0;
    __syncthreads();
    const int j0 = blockIdx.x * BIN_CHUNK;
    for (int k = 0; k < BIN_CHUNK; k += 256) {
        const int j = j0 + k + threadIdx.x;
        lds_count(sh, j < njobs ? jm[j] : -1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nmodels; i += 256) {
        const int c = sh[i];
        if (c) { sb[i] = base[i] + atomicAdd(&cursor[i], c); sh[i] = 0; }
    }
    __syncthreads();
    for (int k = 0; k < BIN_CHUNK; k += 256) {
        const int j = j0 + k + threadIdx.x;
        const int m = j < njobs ? jm[j] : -1;
        const bool act = m >= 0;
        const unsigned long long am = __builtin_amdgcn_ballot_w64(act);
        if (am == 0) continue;
        const int lead = __builtin_ctzll(am);
        const int m0 = __builtin_amdgcn_readlane(m, lead);
        int pos = -1;
        if (__builtin_amdgcn_ballot_w64(act && m == m0) == am) {   // one model in this wave: rank by lane
            int first = 0;
            if ((int)(threadIdx.x & 63) == lead) first = atomicAdd(&sh[m0], __builtin_popcountll(am));
            first = __builtin_amdgcn_readlane(first, lead);
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(am >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)am, 0));
            if (act) pos = sb[m0] + first + rank;
        } else if (act) pos = sb[m] + atomicAdd(&sh[m], 1);
        if (act) { jq[pos] = j >> shift; jo[pos] = rec_base + j; }
    }
}

// single block: exclusive scan of job counts; per-class tile bases.  tot[0..3] = tiles per
// class (tot[0..6]), tot[15] = total jobs, tot[8..14] = class tile offsets.
__global__ __launch_bounds__(1024) void scan_kernel(const ClusterModel* __restrict__ models, int nmodels,
                                                    const int* __restrict__ cnt, int* __restrict__ base,
                                                    int* __restrict__ tbase, int* __restrict__ cursor,
                                                    int* __restrict__ tot, unsigned long long* __restrict__ flops) {
    __shared__ int sj[1024];
    __shared__ unsigned long long sf[1024];
    __shared__ int st[ONGPIS_NCLASS][1024];
    const int tid = threadIdx.x;
    const int chunk = (nmodels + 1023) / 1024;
    const int m0 = tid * chunk, m1 = min(nmodels, m0 + chunk);
    int aj = 0, at[ONGPIS_NCLASS] = {0};
    unsigned long long af = 0;
    for (int m = m0; m < m1; ++m) {
        int c = cnt[m];
        if (c > 0) {
            int cls = ongpis_class_of_nbx(models[m].ld >> 5);
            aj += c; at[cls] += (c + ONGPIS_TILE_Q - 1) / ONGPIS_TILE_Q;
            // algorithmic flops of one evaluation: (1+d) K^2 + 2 (1+d) K + 25 N
            unsigned long long K = models[m].K, N = models[m].N, d1 = 1 + models[m].dim;
            af += (unsigned long long)c * (d1 * K * K + 2 * d1 * K + 25 * N);
        }
    }
    sj[tid] = aj; sf[tid] = af;
    for (int k = 0; k < ONGPIS_NCLASS; ++k) st[k][tid] = at[k];
    __syncthreads();
    if (tid == 0) {
        int rj = 0, rt[ONGPIS_NCLASS] = {0};
        for (int i = 0; i < 1024; ++i) {
            int t = sj[i]; sj[i] = rj; rj += t;
            for (int k = 0; k < ONGPIS_NCLASS; ++k) { int u = st[k][i]; st[k][i] = rt[k]; rt[k] += u; }
        }
        tot[15] = rj;
        unsigned long long fs = 0;
        for (int i = 0; i < 1024; ++i) fs += sf[i];
        flops[0] = fs;
        int off = 0;
        for (int k = 0; k < ONGPIS_NCLASS; ++k) { tot[k] = rt[k]; tot[8 + k] = off; off += rt[k]; }
    }
    __syncthreads();
    int rj = sj[tid], rt[ONGPIS_NCLASS];
    for (int k = 0; k < ONGPIS_NCLASS; ++k) rt[k] = st[k][tid];
    for (int m = m0; m < m1; ++m) {
        int c = cnt[m];
        base[m] = rj; cursor[m] = 0;
        if (c > 0) {
            int cls = ongpis_class_of_nbx(models[m].ld >> 5);
            tbase[m] = rt[cls];
            rt[cls] += (c + ONGPIS_TILE_Q - 1) / ONGPIS_TILE_Q;
            rj += c;
        } else tbase[m] = 0;
    }
}

__global__ void scatter_kernel(const int* __restrict__ jm, int njobs, int shift, int rec_base,
                               const int* __restrict__ base, int* __restrict__ cursor, int* __restrict__ jq,
                               int* __restrict__ jo) {
    int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= njobs) return;
    int m = jm[j];
    if (m < 0) return;
    int pos = base[m] + atomicAdd(&cursor[m], 1);
    jq[pos] = j >> shift;
    jo[pos] = rec_base + j;
}

// one block per model: emit its tiles into the class segment
__global__ void tiles_kernel(const ClusterModel* __restrict__ models, int nmodels, const int* __restrict__ cnt,
                             const int* __restrict__ base, const int* __restrict__ tbase,
                             const int* __restrict__ tot, int* __restrict__ tile_model, int* __restrict__ tile_off,
                             int* __restrict__ tile_cnt) {
    int m = blockIdx.x;
    int c = cnt[m];
    if (c <= 0) return;
    int cls = ongpis_class_of_nbx(models[m].ld >> 5);
    int t0 = tot[8 + cls] + tbase[m];
    int nt = (c + ONGPIS_TILE_Q - 1) / ONGPIS_TILE_Q;
    for (int i = threadIdx.x; i < nt; i += blockDim.x) {
        tile_model[t0 + i] = m;
        tile_off[t0 + i] = base[m] + ONGPIS_TILE_Q * i;
        tile_cnt[t0 + i] = min(ONGPIS_TILE_Q, c - ONGPIS_TILE_Q * i);
    }
}

// -------------------------------------------------------------------- blend ----
template <int DIM>
__global__ __launch_bounds__(256) void blend_kernel(const int* __restrict__ cand, const int* __restrict__ ncand, int n,
                                                    int cap, const float* __restrict__ out, float var_thre,
                                                    float prior_var, float* __restrict__ res) {
    constexpr int NC = 1 + DIM;
    int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    float* r = res + (size_t)2 * NC * q;
    const int nc = ncand[q];
    r[NC] = prior_var;  // GPisMap3.cpp:816
    if (nc == 0) return;
    const bool has0 = cand[q] >= 0;
    const float* o0 = out + (size_t)q * 8;
    if (has0) {
        for (int c = 0; c < NC; ++c) { r[c] = o0[c]; r[NC + c] = o0[4 + c]; }
    }
    if (nc == 1) return;
    if (!(r[NC] > var_thre)) return;
    // candidates 1.. : records n + 2q + s
    float f2[3][NC], v2[3][NC];
    for (int c = 0; c < NC; ++c) { f2[0][c] = r[c]; v2[0][c] = r[NC + c]; }
    for (int s = 1; s < nc; ++s) {
        const float* o = out + ((size_t)cap + 2 * (size_t)q + (s - 1)) * 8;
        // A cell without a trained model (point stored through the set-less root-growth insert, or a failed
        // training) got no pass-2 job: its record is stale scratch.  The reference dereferences a null GP there;
        // here the candidate contributes the prior (mean 0, prior variance) instead of garbage.
        const bool has = cand[(size_t)s * cap + q] >= 0;
        for (int c = 0; c < NC; ++c) { f2[s][c] = has ? o[c] : 0.f; v2[s][c] = has ? o[4 + c] : (c == 0 ? prior_var : 0.f); }
    }
    // stable insertion sort of <= 3 indices by value variance (std::sort on <= 16 elements)
    int id[3] = {0, 1, 2};
    for (int i = 1; i < nc; ++i) {
        int k = id[i]; int j = i - 1;
        while (j >= 0 && v2[k][0] < v2[id[j]][0]) { id[j + 1] = id[j]; --j; }
        id[j + 1] = k;
    }
    const int b0 = id[0];
    if (v2[b0][0] < var_thre) {
        for (int c = 0; c < NC; ++c) { r[c] = f2[b0][c]; r[NC + c] = v2[b0][c]; }
    } else {
        const int b1 = id[1];
        float w1 = v2[b0][0] - var_thre, w2 = v2[b1][0] - var_thre, w12 = w1 + w2;
        for (int c = 0; c < NC; ++c) {
            r[c] = (w2 * f2[b0][c] + w1 * f2[b1][c]) / w12;
            r[NC + c] = (w2 * v2[b0][c] + w1 * v2[b1][c]) / w12;
        }
    }
}

__global__ void prior_only_kernel(int n, int nc2, int vidx, float prior_var, float* __restrict__ res) {
    int q = blockIdx.x * 256 + threadIdx.x;
    if (q < n) res[(size_t)nc2 * q + vidx] = prior_var;
}

// --------------------------------------------------------------- host side ----
MapQuery::MapQuery(int dim, float search_half, float var_thre, float prior_var)
    : dim_(dim), search_half_(search_half), var_thre_(var_thre), prior_var_(prior_var), h_maxN_(ONGPIS_NCLASS, 0), h_maxLd_(ONGPIS_NCLASS, 0) {
    std::memset(&tv_, 0, sizeof(tv_));
}

MapQuery::~MapQuery() {
    (void)hipFree(d_tab_); (void)hipFree(d_grid_); (void)hipFree(d_xq_); (void)hipFree(d_cand_); (void)hipFree(d_ncand_); (void)hipFree(d_tie_);
    (void)hipFree(d_jm_); (void)hipFree(d_jq_); (void)hipFree(d_jo_); (void)hipFree(d_out_); (void)hipFree(d_cnt_);
    (void)hipFree(d_base_); (void)hipFree(d_cursor_); (void)hipFree(d_tbase_); (void)hipFree(d_tile_); (void)hipFree(d_tot_);
    if (ev0_) (void)hipEventDestroy(ev0_);
    if (ev1_) (void)hipEventDestroy(ev1_);
    for (int i = 0; i < kSide; ++i) { if (side_[i]) (void)hipStreamDestroy(side_[i]); if (evjoin_[i]) (void)hipEventDestroy(evjoin_[i]); }
    if (evfork_) (void)hipEventDestroy(evfork_);
}

int MapQuery::set_clusters(const std::vector<ClusterEntry>& cl, const std::vector<AncestorEntry>& anc, double pitch, hipStream_t s) {
    ncl_ = (int)cl.size();
    tv_.n = ncl_;
    if (ncl_ == 0) return GPIS_OK;
    // lattice bounds from cell centres
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (auto& e : cl)
        for (int d = 0; d < 3; ++d) { lo[d] = std::min(lo[d], (double)e.c[d]); hi[d] = std::max(hi[d], (double)e.c[d]); }
    int g[3];
    double org[3];
    for (int d = 0; d < 3; ++d) {
        org[d] = lo[d] - 0.5 * pitch;
        g[d] = (int)llround((hi[d] - lo[d]) / pitch) + 1;
        if (d >= dim_) { g[d] = 1; org[d] = -0.5 * pitch; }
    }
    size_t ncell = (size_t)g[0] * g[1] * g[2];
    if (ncell > ((size_t)1 << 31)) return GPIS_ERR_LIMIT;
    std::vector<int> grid(ncell, -1);
    const size_t na = anc.size();
    std::vector<float4> tab((size_t)3 * ncl_ + 2 * na);
    std::vector<int> mdl((size_t)2 * ncl_ + na);
    for (int i = 0; i < ncl_; ++i) {
        const ClusterEntry& e = cl[i];
        int ix[3];
        for (int d = 0; d < 3; ++d) {
            ix[d] = (d < dim_) ? (int)floor(((double)e.c[d] - org[d]) / pitch) : 0;
            ix[d] = std::min(std::max(ix[d], 0), g[d] - 1);
        }
        size_t cell = ((size_t)ix[2] * g[1] + ix[1]) * g[0] + ix[0];
        if (grid[cell] >= 0) {
            fprintf(stderr, "[gpismap_amd] lattice collision between cluster cells %d and %d\n", grid[cell], i);
            return GPIS_ERR_STATE;
        }
        grid[cell] = i;
        tab[i] = make_float4(e.c[0], e.c[1], e.c[2], 0.f);
        tab[(size_t)ncl_ + i] = make_float4(e.lo[0], e.lo[1], e.lo[2], 0.f);
        tab[(size_t)2 * ncl_ + i] = make_float4(e.hi[0], e.hi[1], e.hi[2], 0.f);
        mdl[i] = e.model;
        mdl[(size_t)ncl_ + i] = e.parent;
    }
    for (size_t a = 0; a < na; ++a) {
        tab[(size_t)3 * ncl_ + a] = make_float4(anc[a].lo[0], anc[a].lo[1], anc[a].lo[2], 0.f);
        tab[(size_t)3 * ncl_ + na + a] = make_float4(anc[a].hi[0], anc[a].hi[1], anc[a].hi[2], 0.f);
        mdl[(size_t)2 * ncl_ + a] = anc[a].parent;
    }
    size_t tbytes = sizeof(float4) * tab.size() + sizeof(int) * mdl.size();
    if (tbytes > cap_tab_) { (void)hipFree(d_tab_); d_tab_ = nullptr; GPIS_HIP(hipMalloc(&d_tab_, tbytes * 2)); cap_tab_ = tbytes * 2; }
    if (ncell > cap_grid_) { (void)hipFree(d_grid_); d_grid_ = nullptr; GPIS_HIP(hipMalloc(&d_grid_, sizeof(int) * ncell * 2)); cap_grid_ = ncell * 2; }
    GPIS_HIP(hipMemcpyAsync(d_tab_, tab.data(), sizeof(float4) * tab.size(), hipMemcpyHostToDevice, s));
    int* d_model = reinterpret_cast<int*>(reinterpret_cast<char*>(d_tab_) + sizeof(float4) * tab.size());
    GPIS_HIP(hipMemcpyAsync(d_model, mdl.data(), sizeof(int) * mdl.size(), hipMemcpyHostToDevice, s));
    GPIS_HIP(hipMemcpyAsync(d_grid_, grid.data(), sizeof(int) * ncell, hipMemcpyHostToDevice, s));
    GPIS_HIP(hipStreamSynchronize(s));
    tv_.c = reinterpret_cast<const float4*>(d_tab_);
    tv_.lo = tv_.c + ncl_; tv_.hi = tv_.c + 2 * (size_t)ncl_;
    tv_.model = d_model; tv_.grid = d_grid_;
    tv_.parent = d_model + ncl_;
    tv_.anc_lo = tv_.c + 3 * (size_t)ncl_; tv_.anc_hi = tv_.anc_lo + na;
    tv_.anc_parent = d_model + 2 * (size_t)ncl_;
    tv_.gx = g[0]; tv_.gy = g[1]; tv_.gz = g[2];
    tv_.ox = org[0]; tv_.oy = org[1]; tv_.oz = org[2]; tv_.pitch = pitch;
    return GPIS_OK;
}

int MapQuery::ensure_scratch(int n, int nmodels) {
    if (n > cap_n_) {
        (void)hipFree(d_xq_); (void)hipFree(d_cand_); (void)hipFree(d_ncand_); (void)hipFree(d_jm_); (void)hipFree(d_jq_);
        (void)hipFree(d_jo_); (void)hipFree(d_out_); (void)hipFree(d_tile_);
        d_xq_ = nullptr; d_cand_ = d_ncand_ = d_jm_ = d_jq_ = d_jo_ = d_tile_ = nullptr; d_out_ = nullptr; cap_n_ = 0;
        tile_cap_ = 0;   // d_tile_ is gone: the size check below must reallocate it even when need_tiles did not grow
        size_t c = (size_t)n;
        GPIS_HIP(hipMalloc(&d_xq_, sizeof(float4) * c));
        GPIS_HIP(hipMalloc(&d_cand_, sizeof(int) * 3 * c));
        GPIS_HIP(hipMalloc(&d_ncand_, sizeof(int) * c));
        if (!d_tie_) GPIS_HIP(hipMalloc(&d_tie_, sizeof(int) * 4));
        GPIS_HIP(hipMalloc(&d_jm_, sizeof(int) * 2 * c));
        GPIS_HIP(hipMalloc(&d_jq_, sizeof(int) * 2 * c));
        GPIS_HIP(hipMalloc(&d_jo_, sizeof(int) * 2 * c));
        GPIS_HIP(hipMalloc(&d_out_, sizeof(float) * 8 * 3 * c));
        cap_n_ = n;
    }
    int need_tiles = 2 * (cap_n_ / ONGPIS_TILE_Q + 1) + nmodels + 64;
    if (need_tiles > tile_cap_) {
        (void)hipFree(d_tile_); d_tile_ = nullptr;
        GPIS_HIP(hipMalloc(&d_tile_, sizeof(int) * 3 * (size_t)need_tiles));
        tile_cap_ = need_tiles;
    }
    if (nmodels > cap_models_) {
        (void)hipFree(d_cnt_); (void)hipFree(d_base_); (void)hipFree(d_cursor_); (void)hipFree(d_tbase_);
        d_cnt_ = d_base_ = d_cursor_ = d_tbase_ = nullptr;
        int c = nmodels * 2 + 256;
        GPIS_HIP(hipMalloc(&d_cnt_, sizeof(int) * c));
        GPIS_HIP(hipMalloc(&d_base_, sizeof(int) * c));
        GPIS_HIP(hipMalloc(&d_cursor_, sizeof(int) * c));
        GPIS_HIP(hipMalloc(&d_tbase_, sizeof(int) * c));
        cap_models_ = c;
    }
    if (!d_tot_) GPIS_HIP(hipMalloc(&d_tot_, sizeof(int) * 32));
    return GPIS_OK;
}

// Bin the jobs in d_jm_ (njobs entries) by model and run K4 over the tiles.
int MapQuery::eval_pass(OnGPISStore& store, int njobs, int shift, int rec_base, int nmodels, hipStream_t s) {
    GPIS_HIP(hipMemsetAsync(d_cnt_, 0, sizeof(int) * (size_t)nmodels, s));
    const bool lds_bins = nmodels <= 8192;   // 2 x 32 KB of LDS at most
    const int nbin_blocks = (njobs + BIN_CHUNK - 1) / BIN_CHUNK;
    if (lds_bins)
        hipLaunchKernelGGL(hist_lds_kernel, dim3(nbin_blocks), dim3(256), sizeof(int) * (size_t)nmodels, s, d_jm_, njobs, nmodels, d_cnt_);
    else
        hipLaunchKernelGGL(hist_kernel, dim3((njobs + 255) / 256), dim3(256), 0, s, d_jm_, njobs, d_cnt_);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, store.d_models(), nmodels, d_cnt_, d_base_, d_tbase_,
                       d_cursor_, d_tot_, reinterpret_cast<unsigned long long*>(d_tot_ + 16));
    if (lds_bins)
        hipLaunchKernelGGL(scatter_lds_kernel, dim3(nbin_blocks), dim3(256), sizeof(int) * 2 * (size_t)nmodels, s, d_jm_, njobs,
                           nmodels, shift, rec_base, d_base_, d_cursor_, d_jq_, d_jo_);
    else
        hipLaunchKernelGGL(scatter_kernel, dim3((njobs + 255) / 256), dim3(256), 0, s, d_jm_, njobs, shift, rec_base, d_base_,
                           d_cursor_, d_jq_, d_jo_);
    int* t_model = d_tile_;
    int* t_off = d_tile_ + tile_cap_;
    int* t_cnt = d_tile_ + 2 * (size_t)tile_cap_;
    hipLaunchKernelGGL(tiles_kernel, dim3(nmodels), dim3(64), 0, s, store.d_models(), nmodels, d_cnt_, d_base_, d_tbase_,
                       d_tot_, t_model, t_off, t_cnt);
    int tot[20];
    GPIS_HIP(hipMemcpyAsync(tot, d_tot_, sizeof(int) * 20, hipMemcpyDeviceToHost, s));
    GPIS_HIP(hipStreamSynchronize(s));
    last_evals += tot[15];
    { unsigned long long f; std::memcpy(&f, tot + 16, sizeof(f)); last_flops += (long long)f; }
    for (int c = 0; c < ONGPIS_NCLASS; ++c) if (tot[c] > 0) ++last_launches;
    if (profile) {
        if (!ev0_) { GPIS_HIP(hipEventCreate(&ev0_)); GPIS_HIP(hipEventCreate(&ev1_)); }
        GPIS_HIP(hipEventRecord(ev0_, s));
    }
    // The launches of the size classes are independent (disjoint tiles, disjoint outputs): the largest class goes on the caller's
    // stream, the others on side streams forked from it and joined back -- a demo-sized grid spends its time in the critical
    // path of each launch (one tile of the largest clusters: 0.3-0.4 ms) six times in a row otherwise: test() of 13 824 points
    // 2.8 -> 2.0 ms.  GPIS_K4_SERIAL=1: one stream always.
    // (only while the pass is too small to fill the chip: on the 256^3 grid -- half a million tiles per launch -- launches that
    // overlap take each other's LDS and run 2.3 % slower than one after the other)
    int ntiles_pass = 0, nlaunch_pass = 0;
    for (int c = 0; c < ONGPIS_NCLASS; ++c) { ntiles_pass += std::max(0, tot[c]); nlaunch_pass += tot[c] > 0; }
    bool fork = !side_off_ && nlaunch_pass > 1 && ntiles_pass <= 8192;
    if (fork && !evfork_) {
        // (created by the first pass that forks, not before: every stream is a hardware queue)
        if (const char* e = getenv("GPIS_K4_SERIAL")) side_off_ = atoi(e) != 0;
        bool okc = !side_off_ && hipEventCreateWithFlags(&evfork_, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < kSide && okc; ++i)
            okc = hipStreamCreateWithFlags(&side_[i], hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&evjoin_[i], hipEventDisableTiming) == hipSuccess;
        if (!okc) { side_off_ = true; (void)hipGetLastError(); }
        fork = !side_off_;
    }
    if (fork) GPIS_HIP(hipEventRecord(evfork_, s));
    int nlaunched = 0;
    bool used[kSide] = {false, false, false};
    // (an error return inside the loop must not leave forked side streams un-joined to s: the caller's next work on s would
    // race the launches already made)
    auto join_sides = [&]() {
        for (int i = 0; i < kSide; ++i)
            if (used[i]) { (void)hipEventRecord(evjoin_[i], side_[i]); (void)hipStreamWaitEvent(s, evjoin_[i], 0); used[i] = false; }
    };
    for (int c = ONGPIS_NCLASS - 1; c >= 0; --c) {
        int nt = tot[c];
        if (nt <= 0) continue;
        if (tot[8 + c] + nt > tile_cap_) { join_sides(); return GPIS_ERR_STATE; }
        EvalArgs a;
        a.models = store.d_models(); a.xq = d_xq_;
        a.tile_model = t_model + tot[8 + c]; a.tile_off = t_off + tot[8 + c]; a.tile_cnt = t_cnt + tot[8 + c];
        a.job_q = d_jq_; a.job_out = d_jo_; a.out = d_out_; a.cb = 0; a.debug = store.debug_inject; a.err = store.eval_err(); a.trace = nullptr;
        hipStream_t st = s;
        if (fork && nlaunched > 0) {
            const int i = (nlaunched - 1) % kSide;
            st = side_[i];
            if (!used[i]) { GPIS_HIP(hipStreamWaitEvent(st, evfork_, 0)); used[i] = true; }
        }
        int rc = ongpis_eval_launch(c, nt, h_maxN_[c], h_maxLd_[c], a, st);
        if (rc) { join_sides(); return rc; }
        ++nlaunched;
    }
    for (int i = 0; i < kSide; ++i)
        if (used[i]) { GPIS_HIP(hipEventRecord(evjoin_[i], side_[i])); GPIS_HIP(hipStreamWaitEvent(s, evjoin_[i], 0)); }
    if (profile) {
        GPIS_HIP(hipEventRecord(ev1_, s));
        GPIS_HIP(hipStreamSynchronize(s));
        float ms = 0.f;
        GPIS_HIP(hipEventElapsedTime(&ms, ev0_, ev1_));
        last_eval_ms += ms;
    }
    return GPIS_OK;
}

int MapQuery::run_chunk(OnGPISStore& store, const float* d_x, int n, float* d_res, hipStream_t s) {
    const int nmodels = store.num_slots();
    int rc = ensure_scratch(std::max(n, 1), std::max(nmodels, 1));
    if (rc) return rc;
    const int nblk = (n + 255) / 256;
    GPIS_HIP(hipMemsetAsync(d_tie_, 0, sizeof(int), s));
    if (dim_ == 3)
        hipLaunchKernelGGL((lookup_kernel<3>), dim3(nblk), dim3(256), 0, s, tv_, d_x, n, search_half_, d_xq_, d_cand_, d_ncand_, cap_n_, d_tie_, d_jq_);
    else
        hipLaunchKernelGGL((lookup_kernel<2>), dim3(nblk), dim3(256), 0, s, tv_, d_x, n, search_half_, d_xq_, d_cand_, d_ncand_, cap_n_, d_tie_, d_jq_);
    // exact distance ties (if any): std::sort emulation, one lane per listed query, 512 workgroups striding the list
    if (dim_ == 3)
        hipLaunchKernelGGL((lookup_tie_kernel<3>), dim3(512), dim3(kTieLanes), 0, s, tv_, d_x, search_half_, d_cand_, cap_n_, d_tie_, d_jq_);
    else
        hipLaunchKernelGGL((lookup_tie_kernel<2>), dim3(512), dim3(kTieLanes), 0, s, tv_, d_x, search_half_, d_cand_, cap_n_, d_tie_, d_jq_);
    if (nmodels > 0) {
        hipLaunchKernelGGL(jobs_pass1_kernel, dim3(nblk), dim3(256), 0, s, d_cand_, d_ncand_, n, d_jm_);
        rc = eval_pass(store, n, 0, 0, nmodels, s);
        if (rc) return rc;
        hipLaunchKernelGGL(jobs_pass2_kernel, dim3(nblk), dim3(256), 0, s, d_cand_, d_ncand_, n, cap_n_, d_out_, var_thre_,
                           prior_var_, 1 + dim_, d_jm_);
        rc = eval_pass(store, 2 * n, 1, cap_n_, nmodels, s);
        if (rc) return rc;
    }
    if (dim_ == 3)
        hipLaunchKernelGGL((blend_kernel<3>), dim3(nblk), dim3(256), 0, s, d_cand_, d_ncand_, n, cap_n_, d_out_, var_thre_, prior_var_, d_res);
    else
        hipLaunchKernelGGL((blend_kernel<2>), dim3(nblk), dim3(256), 0, s, d_cand_, d_ncand_, n, cap_n_, d_out_, var_thre_, prior_var_, d_res);
    GPIS_HIP(hipGetLastError());
    return GPIS_OK;
}

int MapQuery::run(OnGPISStore& store, const float* d_x, int n, float* d_res, hipStream_t s) {
    last_evals = 0; last_eval_ms = 0.f; last_flops = 0; last_launches = 0;
    if (n <= 0) return GPIS_OK;
    int rc = store.ensure_inverses(s);     // (lazy inverse: the models retrained since the last prediction get their X now)
    if (rc) return rc;
    rc = store.sync_models(s);
    if (rc) return rc;
    // per-class max N (LDS sizing of K4)
    std::fill(h_maxN_.begin(), h_maxN_.end(), 0);
    std::fill(h_maxLd_.begin(), h_maxLd_.end(), 0);
    for (int i = 0; i < store.num_slots(); ++i) {
        const ClusterModel* m = store.model(i);
        if (!m || !m->base) continue;
        int c = ongpis_eval_class(m->ld / 32);
        h_maxN_[c] = std::max(h_maxN_[c], m->N);
        h_maxLd_[c] = std::max(h_maxLd_[c], m->ld);
    }
    const int nc = 2 * (1 + dim_);
    if (ncl_ == 0) {  // no cluster anywhere: only the prior variance is written
        hipLaunchKernelGGL(prior_only_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, nc, 1 + dim_, prior_var_, d_res);
        GPIS_HIP(hipGetLastError());
        GPIS_HIP(hipStreamSynchronize(s));
        return GPIS_OK;
    }
    for (int off = 0; off < n; off += chunk) {
        int len = std::min(chunk, n - off);
        rc = run_chunk(store, d_x + (size_t)dim_ * off, len, d_res + (size_t)nc * off, s);
        if (rc) return rc;
    }
    GPIS_HIP(hipStreamSynchronize(s));
    if (const int ew = store.take_eval_err()) {
        // (a ring wait of K4 expired: a protocol error.  The affected queries carry NaN; the call fails instead of handing them out.)
        fprintf(stderr, "[gpismap_amd] prediction kernels reported error word 0x%x (a ring wait expired): the affected results are NaN\n", ew);
        return GPIS_ERR_STATE;
    }
    return GPIS_OK;
}

}  // namespace gpis
