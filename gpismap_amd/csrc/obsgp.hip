// K1 / K2: observation-space GP (Ornstein-Uhlenbeck, <= 64 points per group).
//
// Replaces the reference loops
//   K1  ObsGP2D::trainValidPoints  cpp/src/ObsGP.cpp:280-329  -> GPou::train :32-48
//       ObsGP1D::train             cpp/src/ObsGP.cpp:85-143
//       ornstein_uhlenbeck         cpp/src/covFnc.cpp:47-68
//   K2  ObsGP2D::test_kernel       cpp/src/ObsGP.cpp:352-408  -> GPou::test :50-62
//       ObsGP1D::test              cpp/src/ObsGP.cpp:145-187
//       ornstein_uhlenbeck (cross) cpp/src/covFnc.cpp:93-109
//
// One 64-lane wavefront owns one group (K1) or one query (K2): lane i owns
// point i.  K (n x n) is built and factored in LDS; the query path does a
// column-oriented forward substitution with lane broadcasts.  Summation orders
// are the fixed chains of dev_common.h, so results are bit-identical to the
// unblocked CPU order (given equal exp()).
#include "obsgp.h"
#include "tile_solve.h"
#include "exp_tab.h"

namespace gpis {

static constexpr float OU_SCALE = 0.5f;   // params.h:99  DEFAULT_OBSGP_SCALE_PARAM
static constexpr float OU_NOISE = 0.01f;  // params.h:100 DEFAULT_OBSGP_NOISE_PARAM
static constexpr int LDA = 65;            // padded LDS row (conflict-free column walks)

// ---------------------------------------------------------------------------
// K1.  grid = number of groups, block = 64.
//   mode 2: group g = (m, n) tile of the (ni x nj) grid; pixels with f > 0 are
//           compacted in j-outer / i-inner order (ObsGP.cpp:301-310).
//   mode 1: group g = contiguous range [ga[g], ga[g]+glen[g]) of a 1-D scan.
// Output per group: n, x[64][2], alpha[64], L[64x64] column-major (ld 64).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void obsgp_train_kernel(ObsGPView v) {
    __shared__ float A[64 * LDA];
    __shared__ float xs[64][2];
    __shared__ float ys[64];

    const int g = blockIdx.x;
    const int lane = threadIdx.x;
    int ind = -1;
    if (v.mode == 2) {
        int n_ = g % v.ng0, m_ = g / v.ng0;
        int i0 = v.i0[n_], i1 = v.i1[n_], j0 = v.j0[m_], j1 = v.j1[m_];
        int wi = i1 - i0 + 1, wj = j1 - j0 + 1;
        if (lane < wi * wj) ind = (j0 + lane / wi) * v.ni + (i0 + lane % wi);
    } else {
        if (lane < v.glen[g]) ind = v.ga[g] + lane;
    }
    float f = -1.f, x0 = 0.f, x1 = 0.f;
    if (ind >= 0) {
        f = v.f[ind];
        if (v.mode == 2) { x0 = v.x[2 * ind]; x1 = v.x[2 * ind + 1]; }
        else x0 = v.x[ind];
    }
    bool valid = (ind >= 0) && (v.mode == 1 || f > 0.f);
    unsigned long long mask = __ballot(valid);
    const int n = __popcll(mask);
    int rank = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) v.tn[g] = n;
    if (n == 0) return;
    if (valid) { xs[rank][0] = x0; xs[rank][1] = x1; ys[rank] = f; }
    __syncthreads();

    const float a = 1 / OU_SCALE;
    // lane i builds row i (lower part).  covFnc.cpp:55-64
    if (lane < n) {
        float xi0 = xs[lane][0], xi1 = xs[lane][1];
        for (int j = 0; j < lane; ++j) A[lane * LDA + j] = d_ou_k(d_dist2(xs[j][0], xs[j][1], xi0, xi1), a);
        A[lane * LDA + lane] = (float)(1.0 + (double)OU_NOISE);
    }
    __syncthreads();

    // right-looking Cholesky, one column per step; chain order (O1).
    for (int j = 0; j < n; ++j) {
        float d = sqrtf(A[j * LDA + j]);
        float lij = 0.f;
        if (lane > j && lane < n) lij = A[lane * LDA + j] / d;
        __syncthreads();
        if (lane == j) A[j * LDA + j] = d;
        if (lane > j && lane < n) A[lane * LDA + j] = lij;
        __syncthreads();
        if (lane > j && lane < n) {
            float nl = -lij;
            for (int k = j + 1; k <= lane; ++k) A[lane * LDA + k] = fmaf(nl, A[k * LDA + j], A[lane * LDA + k]);
        }
        __syncthreads();
    }

    // alpha = L^-T (L^-1 y).  forward: column oriented, ascending chain.
    float b = (lane < n) ? ys[lane] : 0.f;
    for (int k = 0; k < n; ++k) {
        float t = b / A[lane < n ? lane * LDA + lane : 0];
        float zk = __shfl(t, k);
        if (lane == k) b = zk;
        if (lane > k && lane < n) b = fmaf(A[lane * LDA + k], -zk, b);
    }
    // backward: alpha_j = (z_j - sum_{k>j, descending} L_kj alpha_k) / L_jj
    for (int k = n - 1; k >= 0; --k) {
        float t = b / A[lane < n ? lane * LDA + lane : 0];
        float ak = __shfl(t, k);
        if (lane == k) b = ak;
        if (lane < k) b = fmaf(-A[k * LDA + lane], ak, b);
    }

    float* tx = v.tx + (size_t)g * 128;
    tx[2 * lane] = (lane < n) ? xs[lane][0] : 0.f;
    tx[2 * lane + 1] = (lane < n) ? xs[lane][1] : 0.f;
    v.talpha[(size_t)g * 64 + lane] = (lane < n) ? b : 0.f;
    float* tL = v.tL + (size_t)g * 4096;
    for (int c = 0; c < 64; ++c) {  // column-major, coalesced per column
        float val = 0.f;
        if (lane < n && c <= lane && c < n) val = A[lane * LDA + c];
        else if (lane == c) val = 1.f;
        tL[c * 64 + lane] = val;
    }
}

// ---------------------------------------------------------------------------
// K2.  Group lookup reproduces ObsGP.cpp:359-406 (2-D) / :154-183 (1-D) exactly.
// Outputs val (untouched when no group answers) and var (1e6 then).
// ---------------------------------------------------------------------------
__device__ __forceinline__ int obsgp_lookup2(const ObsGPView& v, float q0, float q1) {
    const float margin = 0.005f;  // params.h:107
    if (q0 < v.vali[0] + margin) return -1;
    if (q0 > v.vali[v.ng0] - margin) return -1;
    if (q1 < v.valj[0] + margin) return -1;
    if (q1 > v.valj[v.ng1] - margin) return -1;
    int n = 0;
    for (int k = 1; k <= v.ng0; ++k, ++n) if (q0 < v.vali[k]) break;
    int m = 0;
    for (int k = 1; k <= v.ng1; ++k, ++m) if (q1 < v.valj[k]) break;
    int gi = m * v.ng0 + n;
    if (gi < v.ngroups && v.tn[gi] > 0) return gi;
    return -1;
}
__device__ __forceinline__ int obsgp_lookup1(const ObsGPView& v, float q0) {
    const float margin = 0.0175f;  // params.h:105
    int nr = v.ngroups + 1;        // range table has ngroups+1 entries
    float liml = v.vali[0] + margin, limr = v.vali[nr - 1] - margin;
    if (q0 < liml || q0 > limr) return -1;
    for (int k = 1; k < nr; ++k)
        if (q0 > v.vali[k - 1] && q0 < v.vali[k]) return (v.tn[k - 1] > 0) ? k - 1 : -1;
    return -1;
}

// One group's share of a wavefront's queries: the lanes with `mine` hold a query (q0, q1) of group gt; its mean and variance
// go to val[qi] / var[qi].  sbuf is used twice: first the k* vectors of the lanes (sbuf[i * 64 + lane]), then -- once those are in
// registers -- the factor, column-major (sbuf[j * 64 + i] = L(i, j)).  All 64 lanes call it (barriers inside).
__device__ __forceinline__ void obsgp_query_group(const ObsGPView& v, int gt, bool mine, float q0, float q1, int qi, int lane,
                                                  float* __restrict__ val, float* __restrict__ var, float* sbuf, float* sx, float* sa, const f64x2* sexp) {
    const float a = 1 / OU_SCALE;
    {
        const int n = v.tn[gt];
        sx[lane] = v.tx[(size_t)gt * 128 + lane];
        sx[64 + lane] = v.tx[(size_t)gt * 128 + 64 + lane];
        sa[lane] = v.talpha[(size_t)gt * 64 + lane];
        __syncthreads();
        // cross-covariances in a rolled loop (the exp sequence stays compact), parked in LDS, then into registers
        if (mine) {
#pragma unroll 1
            for (int i = 0; i < n; ++i) {
                // d_ou_k(d_dist2(...)) with the table-driven exponential (exp_tab.h) and the range-restricted square root (tile_solve.h):
                // (half of this kernel's vector instructions; its run time did not move -- it waits on the LDS reads of the substitution)
                const float tx_ = sx[2 * i] - q0, ty_ = sx[2 * i + 1] - q1;
                sbuf[i * 64 + lane] = (float)exp_neg_tab(-a * sqrt_ranged(tx_ * tx_ + ty_ * ty_), sexp);
            }
        }
        float k[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) k[i] = (i < n) ? sbuf[i * 64 + lane] : 0.f;   // (lanes that are not `mine` read stale values: unused)
        __syncthreads();
        {   // the group's factor: n columns, 16-byte pieces
            const float4* src = reinterpret_cast<const float4*>(v.tL + (size_t)gt * 4096);
            float4* dst = reinterpret_cast<float4*>(sbuf);
            for (int c = lane; c < n * 16; c += 64) dst[c] = src[c];
        }
        __syncthreads();
        if (mine) {
            // mean: the xor butterfly of the 64 products as a register tree (order O4)
            float s[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const float pa = k[i] * sa[i], pb = k[i + 32] * sa[i + 32];
                s[i] = pa + pb;
            }
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1)
#pragma unroll
                for (int i = 0; i < off; ++i) s[i] = s[i] + s[i + off];
            // variance: forward substitution, chain (O1); accumulation chain (O5)
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 64; ++j) {
                if (j < n) {   // wave-uniform
                    const float vj = k[j] / sbuf[j * 64 + j];
#pragma unroll
                    for (int i = j + 1; i < 64; ++i) k[i] = fmaf(sbuf[j * 64 + i], -vj, k[i]);
                    acc = fmaf(vj, vj, acc);
                }
                __builtin_amdgcn_sched_barrier(0);   // one column's operands at a time
            }
            val[qi] = s[0];
            var[qi] = (1 + OU_NOISE) - acc;  // ObsGP.cpp:61
        }
    }
}

// K2, lanes = queries.  A wavefront takes 64 consecutive queries, looks their groups up (lane = query) and then serves
// one DISTINCT group at a time: the group's factor (n columns, <= 16 KB), inputs and alpha are staged in LDS once, and every
// lane whose query belongs to the group runs the whole prediction for its own query with the k* vector in registers --
//   k_i = OU(x_i, q),  mean = tree sum of k_i alpha_i,  forward substitution  v_j = k_j / L_jj ; k_i -= L_ij v_j (i > j),
//   var = (1 + noise) - sum v_j^2
// The operations per query are exactly the ones of the one-wavefront-per-query formulation (chains (O1), (O5), the 64-slot
// butterfly (O4) written out as the same pairwise tree), so the results are bit-identical to it; but the factor is read
// once per (wavefront, group) instead of once per query, and the substitution costs ~45 instructions per query instead of
// ~1200.  update()'s batches are coherent (the 7 queries of a pixel and its neighbours share a group), so a wavefront
// usually sees one to three groups.  Worst case (64 different groups) it degenerates to one group per pass.
// PERM: the wave takes the queries perm[64 b .. 64 b + 63] (queries sorted by group on the device: obsgp_bin_* below), so that a
// wave meets one or two groups instead of every group its 64 consecutive queries happen to fall into; lanes are independent,
// so the order does not touch any result.  ngq = number of queries that have a group (the sorted list's length).
template <bool PERM>
__global__ __launch_bounds__(64) void obsgp_query_kernel(ObsGPView v, const float* __restrict__ q, int nq,
                                                         float* __restrict__ val, float* __restrict__ var,
                                                         const int* __restrict__ perm, const int* __restrict__ gq, const int* __restrict__ ngq) {
    // one 16 KB buffer, used twice per group: first the k* vectors of the lanes (sbuf[i * 64 + lane]), then -- once those
    // are in registers -- the factor, column-major (sbuf[j * 64 + i] = L(i, j))
    __shared__ __attribute__((aligned(16))) float sbuf[64 * 64];
    __shared__ __attribute__((aligned(16))) float sx[128];
    __shared__ __attribute__((aligned(16))) float sa[64];
    __shared__ f64x2 sexp[64];
    const int lane = threadIdx.x;
    sexp[lane] = *reinterpret_cast<const f64x2*>(kExp64Tab[lane]);      // (read behind the first barrier of obsgp_query_group)
    const int slot = blockIdx.x * 64 + lane;
    const bool have = PERM ? (slot < *ngq) : (slot < nq);
    const int qi = PERM ? (have ? perm[slot] : 0) : slot;
    float q0 = 0.f, q1 = 0.f;
    if (have) {
        if (v.mode == 2) { q0 = q[2 * qi]; q1 = q[2 * qi + 1]; }
        else q0 = q[qi];
    }
    int g = -1;
    if (have) g = PERM ? gq[qi] : ((v.mode == 2) ? obsgp_lookup2(v, q0, q1) : obsgp_lookup1(v, q0));
    if (!PERM && have && g < 0) var[qi] = 1e6f;
    bool pending = have && g >= 0;
    for (;;) {
        const unsigned long long todo = __ballot(pending);
        if (todo == 0ull) break;
        const int leader = __ffsll((long long)todo) - 1;
        const int gt = __builtin_amdgcn_readlane(g, leader);
        const bool mine = pending && g == gt;
        obsgp_query_group(v, gt, mine, q0, q1, qi, lane, val, var, sbuf, sx, sa, sexp);
        pending = pending && !mine;
        __syncthreads();
    }
}

// The same for a batch that was sorted by group: workgroup (g, c) of kQueryChunks per group (1 ... 4 by the batch's mean
// queries per group: empty workgroups cost dispatch time, most of all while the training of the previous frame holds the CUs)
// takes the chunks c, c + kQueryChunks,
// ... of 64 queries of group g (base / count: the counting sort's prefix and fill arrays).  A wavefront meets exactly ONE group
// per chunk; with 64 consecutive entries of the sorted list per wavefront (the first binned version) a batch with ten queries per
// group -- the centre queries of a re-evaluation -- made every wavefront walk through six or seven groups one after the other
// (0.42 ms for 30 000 queries; the 537 000 pixel queries took 0.28).
__global__ __launch_bounds__(64) void obsgp_query_grouped_kernel(ObsGPView v, const float* __restrict__ q, float* __restrict__ val, float* __restrict__ var,
                                                                 const int* __restrict__ perm, const int* __restrict__ base, const int* __restrict__ count,
                                                                 int kQueryChunks) {
    __shared__ __attribute__((aligned(16))) float sbuf[64 * 64];
    __shared__ __attribute__((aligned(16))) float sx[128];
    __shared__ __attribute__((aligned(16))) float sa[64];
    __shared__ f64x2 sexp[64];
    const int lane = threadIdx.x;
    sexp[lane] = *reinterpret_cast<const f64x2*>(kExp64Tab[lane]);      // (read behind the first barrier of obsgp_query_group)
    const int g = blockIdx.x / kQueryChunks, c0 = blockIdx.x % kQueryChunks;
    const int b = base[g], cnt = count[g];
    for (int chunk = c0; chunk * 64 < cnt; chunk += kQueryChunks) {
        const int idx = chunk * 64 + lane;
        const bool mine = idx < cnt;
        const int qi = mine ? perm[b + idx] : 0;
        float q0 = 0.f, q1 = 0.f;
        if (mine) {
            if (v.mode == 2) { q0 = q[2 * qi]; q1 = q[2 * qi + 1]; }
            else q0 = q[qi];
        }
        obsgp_query_group(v, g, mine, q0, q1, qi, lane, val, var, sbuf, sx, sa, sexp);
        __syncthreads();
    }
}

void obsgp_launch_train(const ObsGPView& v, hipStream_t s) {
    hipLaunchKernelGGL(obsgp_train_kernel, dim3(v.ngroups), dim3(64), 0, s, v);
}

// ---- queries sorted by group on the device (counting sort: group of every query, histogram, scan, scatter) ----
// cnt: [ngroups + 1] zeroed; gq: [nq] group of the query or -1 (those get var = 1e6 here: ObsGP.cpp:363)
__global__ __launch_bounds__(256) void obsgp_bin_count_kernel(ObsGPView v, const float* __restrict__ q, int nq, int* __restrict__ gq,
                                                              int* __restrict__ cnt, float* __restrict__ var) {
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    const int g = (v.mode == 2) ? obsgp_lookup2(v, q[2 * qi], q[2 * qi + 1]) : obsgp_lookup1(v, q[qi]);
    gq[qi] = g;
    if (g < 0) var[qi] = 1e6f;
    else atomicAdd(&cnt[g], 1);
}
// exclusive scan of cnt[0 .. n) in place (one workgroup), total -> cnt[n]; fill[] zeroed for the scatter
__global__ __launch_bounds__(1024) void obsgp_bin_scan_kernel(int* __restrict__ cnt, int n) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int b = tid * per, e = min(n, b + per);
    int sum = 0;
    for (int i = b; i < e; ++i) sum += cnt[i];
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) { int run = 0; for (int i = 0; i < 1024; ++i) { const int t = part[i]; part[i] = run; run += t; } cnt[n] = run; }
    __syncthreads();
    int run = part[tid];
    for (int i = b; i < e; ++i) { const int t = cnt[i]; cnt[i] = run; run += t; }
}
__global__ __launch_bounds__(256) void obsgp_bin_scatter_kernel(const int* __restrict__ gq, int nq, const int* __restrict__ base, int* __restrict__ fill,
                                                                int* __restrict__ perm) {
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    const int g = gq[qi];
    if (g >= 0) perm[base[g] + atomicAdd(&fill[g], 1)] = qi;
}

void obsgp_launch_query(const ObsGPView& v, const float* d_q, int nq, float* d_val, float* d_var, hipStream_t s) {
    hipLaunchKernelGGL(obsgp_query_kernel<false>, dim3((nq + 63) / 64), dim3(64), 0, s, v, d_q, nq, d_val, d_var, (const int*)nullptr, (const int*)nullptr, (const int*)nullptr);
}
// scratch: 2 nq + 2 (ngroups + 1) ints
void obsgp_launch_query_binned(const ObsGPView& v, const float* d_q, int nq, float* d_val, float* d_var, int* scratch, hipStream_t s) {
    int* gq = scratch; int* perm = gq + nq; int* cnt = perm + nq; int* fill = cnt + (v.ngroups + 1);
    (void)hipMemsetAsync(cnt, 0, sizeof(int) * 2 * (size_t)(v.ngroups + 1), s);
    hipLaunchKernelGGL(obsgp_bin_count_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, v, d_q, nq, gq, cnt, d_var);
    hipLaunchKernelGGL(obsgp_bin_scan_kernel, dim3(1), dim3(1024), 0, s, cnt, v.ngroups);
    hipLaunchKernelGGL(obsgp_bin_scatter_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, gq, nq, cnt, fill, perm);
    // (cnt: start of every group in the sorted list, fill: its length)
    const int chunks = std::max(1, std::min(4, (int)((3 * (long long)nq / std::max(1, v.ngroups) + 127) / 128)));   // ~1.5 x the mean chunks per group
    hipLaunchKernelGGL(obsgp_query_grouped_kernel, dim3(v.ngroups * chunks), dim3(64), 0, s, v, d_q, d_val, d_var, perm, cnt, fill, chunks);
}

}  // namespace gpis
