// K6 + K3 + K3b fused for clusters of at most 8 block rows (K <= 256): ONE launch, one 8-wavefront workgroup per
// cluster, the whole factorisation on chip.
//
// Replaces, for these clusters, the chain gather -> buildK -> chol -> inverse of ongpis_train.hip, i.e. the reference loop
//   GPisMap3::updateGPs_kernel   cpp/src/GPisMap3.cpp:698-718   (2-D: GPisMap.cpp:574-594)
//     -> OnGPIS::train           cpp/src/OnGPIS.cpp:91-149      (2-D: :34-89)
//        -> matern32_sparse_deriv1_3D  cpp/src/covFnc.cpp:142-256 (2-D: :317-402)
//        -> K.llt(), two triangular solves (Eigen)   OnGPIS.cpp:139-143
//
// The separate kernels moved 12-17 x the algorithmic bytes through HBM (K written by buildK and re-read, L written twice,
// every tile product re-fetching its operands, Zt) -- the stress configuration (50 000 clusters x K = 256) ran at 3-4 TB/s
// of HBM-side traffic.  Here the kernel matrix is built INTO LDS (32 x 32 tiles in MFMA operand order), the trailing
// tiles live in MFMA accumulators (right-looking sweep), finished tiles of L replace the K tiles in LDS and feed the
// matrix cores from there, and the only global traffic is 36 bytes per training point in and Xt / x4 / rowinfo out
// (L, alpha, gidx too when the model carries those pointers: parity tests and gpis_ongpis_get_model).
//
// Arithmetic: unchanged, element for element.
//   * kernel entries: the formulas of ongpis_buildK_kernel (one exp per point pair, evaluated in double);
//   * Cholesky: tile(bi, c) -= L(bi, j) L(c, j)^T for j ascending (v_mfma_f32_32x32x2_f32 = ascending-k fmaf chain),
//     diagonal tiles by factor32_inreg, panel tiles by diag_solve32 -- order (O1);
//   * z = L^-1 y: the chain the y row of the augmented matrix takes in the separate kernels, here on the vector ALU
//     (fmaf(-l, z, s) over ascending k, true divisions);
//   * alpha = L^-T z: blocked back substitution, descending chains -- order (O2);
//   * X = L^-1 by block columns: X_bc = inv(L_bb) sum_p (-L_bp) X_pc, ascending (p, k) from zero, product with the
//     inverted diagonal block in the order (O6).
// Results are bit-identical to the separate kernels and to the oracle's tiled mode (tests/test_gpu_ongpis.py,
// tests/test_gpu_stress.py).
//
// Schedule per block column j (two workgroup barriers):
//   P1  the owner of tile (j, j) factorises it in registers and publishes the factor column group by column group (LDS
//       progress word); the other wavefronts first apply column j-1 to their tiles of the columns > j (the matrix pipes
//       run under the serial factorisation), then solve their panel tile (bi, j) -- or z_j -- TRAILING the factorisation;
//   P3  panel owners fold z_j into the rows below; column j is applied to the tiles of column j+1 (the next diagonal
//       and its panel are then complete).
// Then: the eight diagonal blocks are inverted in parallel, four wavefronts run two block columns of X each (transposed
// tiles kept in registers as the B operands of the later rows), the other four run the back substitution for alpha as
// a pipeline over row blocks and write the mean rows of Xt.
#include <cstdlib>
#include <type_traits>
#include "ongpis.h"
#include "tile_solve.h"

namespace gpis {

namespace {

constexpr int kFW = 8;             // wavefronts per workgroup
constexpr int kFT = 64 * kFW;
constexpr int kStage = 16 * 33;    // floats of one half-tile staging buffer (inverse phase, one per wavefront of the four)
constexpr int kRegion = 4 * kStage;

// element (row, k) of a tile in MFMA A-operand order: [g][lane][j] = T[lane & 31][2 (4g + j) + (lane >> 5)]
__device__ __forceinline__ int a_addr(int row, int k) { return ((((k >> 3) << 6) + ((k & 1) << 5) + row) << 2) + ((k >> 1) & 3); }
// element (row, col) of a diagonal slot in the k order of an accumulator tile (O6): k(kk, h) = (kk & 3) + 8 (kk >> 2) + 4 h
__device__ __forceinline__ int d_addr(int row, int col) { return ((((col >> 3) << 6) + (((col >> 2) & 1) << 5) + row) << 2) + (col & 3); }

typedef volatile int __attribute__((address_space(3))) * lds_flag_t;

// diag_solve32 (tile_solve.h: same operations, same order), steps I0 .. I0+7.  The row select is only emitted for the
// registers whose row can still be <= i in one of the lane halves.
template <int I0>
__device__ __forceinline__ void diag_solve8(f32x16& v, const float* Lc, int h) {
    DiagCol cur, nxt;
    diag_load(cur, Lc, I0, h);
#pragma unroll
    for (int i = I0; i < I0 + 8; ++i) {
        if (i + 1 < I0 + 8) diag_load(nxt, Lc, i + 1, h);
        const int hi_ = (i >> 2) & 1, ri = (i & 3) + 4 * (i >> 3);
        float cand = div_ranged(v[ri], cur.d, cur.r);
        unsigned cu = __float_as_uint(cand);
        auto sw = __builtin_amdgcn_permlane32_swap(cu, cu, false, false);
        float vi = __uint_as_float(hi_ ? sw[1] : sw[0]);
        v[ri] = (h == hi_) ? vi : v[ri];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row0 = (r & 3) + 8 * (r >> 2);
            if (row0 + 4 > i) {
                const float4 q = cur.g[r >> 2];
                float lri = (r & 3) == 0 ? q.x : ((r & 3) == 1 ? q.y : ((r & 3) == 2 ? q.z : q.w));
                float upd = fmaf(-lri, vi, v[r]);
                if (row0 > i) v[r] = upd;                           // both halves' rows are below row i
                else v[r] = (row0 + 4 * h > i) ? upd : v[r];
            }
        }
        if (i + 1 < I0 + 8) cur = nxt;
        __builtin_amdgcn_sched_barrier(0);
    }
}
// LDS traffic only: drain this wavefront's LDS queue, then the workgroup barrier.  Used INSIDE wave-uniform branches:
// the two roles of a block column (factorise / solve) are separate instruction streams with the same barrier count.
__device__ __forceinline__ void wg_sync() {
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
}

#ifdef GPIS_INSTRUMENT
#include "ongpis_fused_instr.inc"
#else
#define FSTAMP(i) do {} while (0)
#define FSTAMP_AT(i) do {} while (0)
#define FSTAMP_W(i) do {} while (0)
#endif

}  // namespace

size_t ongpis_fused_lds_bytes(int nb) {
    return sizeof(float) * ((size_t)nb * (nb + 1) / 2 * 1024 + 256 + 256 + kRegion) + 256;   // + the flag words
}

// K6 gather: the cluster's points into the phase-local region R (x4s, sig, gidx), the targets into yv; x4 / rowinfo (/ gidx) of the
// model are written to global memory.  FT threads per workgroup (FT >= N).  Contains one workgroup barrier; the caller
// synchronises before it reads R or yv.
template <int FT>
__device__ __forceinline__ void fused_gather(const FusedTrainArgs& A, const ClusterModel* __restrict__ mp, float* yv, float* R,
                                             int job, int tid, int lane, int wave, int N, int ng, int K, int ld, int dim) {
    float4* x4s = reinterpret_cast<float4*>(R);              // [N]
    float* sig = R + 1024;                                   // [2 N] sigx' (after the 2.0 override), sigg
    int* gidx = reinterpret_cast<int*>(R + 1536);            // [N]
    int* wcnt = reinterpret_cast<int*>(R + 1792);            // [8]
    const int off = A.jobs[4 * job + 1];
    const float* __restrict__ pts = A.pts;
    const size_t cap = (size_t)A.cap;
    bool flag = false;
    float gv[3] = {0.f, 0.f, 0.f};
    if (tid < N) {
        const int id = A.ids[off + tid];
        const float px = pts[id], py = pts[cap + id], pz = pts[2 * cap + id];
        gv[0] = pts[3 * cap + id]; gv[1] = pts[4 * cap + id]; gv[2] = pts[5 * cap + id];
        const float val = pts[6 * cap + id], sx = pts[7 * cap + id], sg = pts[8 * cap + id];
        const bool tiny = ((double)fabsf(gv[0]) < 1e-6) && ((double)fabsf(gv[1]) < 1e-6) && (dim == 2 || (double)fabsf(gv[2]) < 1e-6);
        flag = !(((double)sg > 0.1001) || tiny);   // OnGPIS.cpp:122-125
        const float4 xp = make_float4(px, py, dim == 3 ? pz : 0.f, 0.f);
        x4s[tid] = xp;
        reinterpret_cast<float4*>(mp->x4)[tid] = xp;
        sig[tid] = flag ? sx : 2.0f;
        sig[N + tid] = sg;
        yv[tid] = val;
        mp->rowinfo[tid] = tid;
    }
    {
        const unsigned long long bal = __ballot(flag);
        if (lane == 0) wcnt[wave] = __popcll(bal);
        __syncthreads();
        int g = __popcll(bal & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) g += wcnt[w];
        if (tid < N) {
            gidx[tid] = flag ? g : -1;
            if (mp->gidx) mp->gidx[tid] = flag ? g : -1;
            if (flag)
                for (int cc = 0; cc < dim; ++cc) {
                    const int row = N + cc * ng + g;
                    yv[row] = gv[cc];
                    mp->rowinfo[row] = tid | ((cc + 1) << 28);
                }
        }
        for (int r = K + tid; r < 256; r += FT) yv[r] = 0.f;
        for (int r = K + tid; r < ld; r += FT) mp->rowinfo[r] = 0xF << 28;
    }
}

// The kernel matrix of the gathered cluster, entry by entry of its lower triangle: put(row, col, value) (the pair is swapped
// into the lower triangle first), every entry exactly once; the caller has zeroed its destination (structural zeros are not
// all put).  One exp per point pair, evaluated in double: the formulas of ongpis_buildK_kernel.
template <int FT, class Put>
__device__ __forceinline__ void fused_build_entries(const ClusterModel* __restrict__ mp, const float* R, int tid, int N, int ng, int K, int nb, int dim, Put putl) {
    const float4* x4s = reinterpret_cast<const float4*>(R);
    const float* sig = R + 1024;
    const int* gidx = reinterpret_cast<const int*>(R + 1536);
    auto put = [&](int r, int c, float v) { putl(r >= c ? r : c, r >= c ? c : r, v); };
    {
        const float a = (float)(sqrt(3.0) / (double)mp->scale);  // covFnc.cpp:147
        const float a2 = a * a;
        for (int r = K + tid; r < 32 * nb; r += FT) put(r, r, 1.f);   // identity padding of the last block
        // diagonal pairs (k == k): cheap, one thread per point
        for (int k = tid; k < N; k += FT) {
            const int kg = gidx[k];
            const int kind[3] = {N + kg, N + kg + ng, N + kg + 2 * ng};
            put(k, k, (float)(1.0 + (double)sig[k]));
            if (kg >= 0) {
                const float sg = sig[N + k];
                for (int c = 0; c < dim; ++c) {
                    put(kind[c], k, 0.f);
                    for (int c2 = 0; c2 < c; ++c2) put(kind[c], kind[c2], 0.f);
                }
                if (dim == 3) {
                    for (int c = 0; c < 3; ++c) put(kind[c], kind[c], a2 + sg);
                } else {
                    put(kind[0], kind[0], (float)((double)a2 + sqrt((double)(sig[k] * sg))));  // covFnc.cpp:352
                    put(kind[1], kind[1], a2 + sg);
                }
            }
        }
        // off-diagonal pairs k < j (one double-precision exp each): N (N - 1) / 2 of them, dealt evenly -- with the diagonal
        // pairs in the same list 64 points gave 2080 = 4 x 512 + 32 pairs: a fifth trip for everybody because of 32 lanes
        const int P = N * (N - 1) / 2;
        for (int p = tid; p < P; p += FT) {
            int j = (int)((sqrtf(8.f * (float)p + 1.f) + 1.f) * 0.5f);      // p = j (j - 1) / 2 + k, k < j
            while (j * (j - 1) / 2 > p) --j;
            while ((j + 1) * j / 2 <= p) ++j;
            const int k = p - j * (j - 1) / 2;
            const int kg = gidx[k];
            const int kind[3] = {N + kg, N + kg + ng, N + kg + 2 * ng};
            const float4 xk = x4s[k], xj = x4s[j];
            const int jg = gidx[j];
            const int jind[3] = {N + jg, N + jg + ng, N + jg + 2 * ng};
            const float d[3] = {xk.x - xj.x, xk.y - xj.y, xk.z - xj.z};
            const float r = (dim == 3) ? sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) : sqrtf(d[0] * d[0] + d[1] * d[1]);
            const double e = exp((double)(-a * r));
            put(j, k, d_kf(r, a, e));
            if (kg >= 0) {
                float g1[3];
                for (int c = 0; c < dim; ++c) { g1[c] = -d_kf1(d[c], a, e); put(kind[c], j, g1[c]); }
                if (jg >= 0) {
                    for (int c = 0; c < dim; ++c) put(jind[c], k, -g1[c]);
                    for (int c1 = 0; c1 < dim; ++c1)
                        for (int c2 = c1; c2 < dim; ++c2) {
                            const float v = d_kf2(r, d[c1], d[c2], c1 == c2 ? 1.0f : 0.0f, a, e);
                            put(kind[c1], jind[c2], v);
                            if (c2 != c1) put(kind[c2], jind[c1], v);
                        }
                }
            } else if (jg >= 0) {
                for (int c = 0; c < dim; ++c) put(jind[c], k, d_kf1(d[c], a, e));
            }
        }
    }
}

// gather + kernel matrix into the LDS tile slots (accumulator order, d_addr) of the one-cluster-per-CU kernels; ends WITHOUT the
// closing barrier.
__device__ __forceinline__ void fused_gather_build(const FusedTrainArgs& A, const ClusterModel* __restrict__ mp, float* slots, float* yv, float* R,
                                                   int job, int tid, int lane, int wave, int N, int ng, int K, int ld, int nb, int dim, int ntl) {
    fused_gather<kFT>(A, mp, yv, R, job, tid, lane, wave, N, ng, K, ld, dim);
    FSTAMP(1);
    {
        float4* z4 = reinterpret_cast<float4*>(slots);
        for (int i = tid; i < ntl * 256; i += kFT) z4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    fused_build_entries<kFT>(mp, R, tid, N, ng, K, nb, dim, [&](int rr, int cc, float v) {
        slots[tri_index(rr >> 5, cc >> 5) * 1024 + d_addr(rr & 31, cc & 31)] = v;   // accumulator order (see the Cholesky below)
    });
}

#ifndef V1_PRIO_DIAG
#define V1_PRIO_DIAG 0     // issue priority of the wavefront factorising the diagonal tile (the serial chain of a block column)
#endif
#ifndef V1_PRIO_ALPHA
#define V1_PRIO_ALPHA 3    // ... and of the four wavefronts of the back substitution: they share their SIMDs with the X wavefronts and
                           // are the tail of the kernel (stress config 24.1 -> 23.1 ms; the diagonal wavefront's priority: no effect)
#endif
// NT = largest number of block rows, ZR = transposed X tiles a column of the inverse keeps in registers (NT - 1)
template <int NT, int ZR, int MINW>
__global__ __launch_bounds__(kFT, MINW) void ongpis_train_fused_kernel(FusedTrainArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int job = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    const ClusterModel* __restrict__ mp = A.models + A.jobs[4 * job];
    const int N = mp->N, ng = mp->ng, K = mp->K, ld = mp->ld, nb = mp->nb, dim = mp->dim;
    const int ntl = nb * (nb + 1) / 2;
    if (nb > ZR + 1 || nb > NT || N > 256) {   // host routing error: refuse loudly, touch nothing
        if (tid == 0) atomicOr(A.err, 1);
        return;
    }
    FSTAMP(0);
    float* slots = smem;                        // [ntl][1024]: K tiles, then L tiles (A-operand order); diagonal slots: see below
    float* yv = slots + (size_t)ntl * 1024;     // [256] y -> z -> alpha
    float* Ldiag = yv + 256;                    // [256] diagonal of L
    float* R = Ldiag + 256;                     // phase-local region
    fused_gather_build(A, mp, slots, yv, R, job, tid, lane, wave, N, ng, K, ld, nb, dim, ntl);
    __syncthreads();
    FSTAMP(2);

    // ---------------------------------------------------------------- Cholesky, right-looking, every tile in LDS
    // A tile that still takes updates sits in its slot in ACCUMULATOR order (d_addr: what a wavefront's 16 accumulator
    // registers of the transposed tile hold, four 16-byte pieces per lane -- conflict-free both ways), so "apply column j
    // to tile (bi, c)" is: 4 loads, 16 matrix instructions, 4 stores, by ANY wavefront.  The finished tile of L is written
    // back in A-operand order.  Nothing but the tile in flight lives in registers.
    auto apply_column = [&](int bi, int c, int j) __attribute__((always_inline)) {   // tile(bi, c) -= L(bi, j) L(c, j)^T
        float4* tt = reinterpret_cast<float4*>(slots + tri_index(bi, c) * 1024);
        const float4* ta = reinterpret_cast<const float4*>(slots + tri_index(c, j) * 1024);
        const float4* tbp = reinterpret_cast<const float4*>(slots + tri_index(bi, j) * 1024);
        f32x16 d;
        float av[16], bv[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 qa = ta[g * 64 + lane], qb = tbp[g * 64 + lane], qd = tt[g * 64 + lane];
            av[4 * g] = qa.x; av[4 * g + 1] = qa.y; av[4 * g + 2] = qa.z; av[4 * g + 3] = qa.w;
            bv[4 * g] = qb.x; bv[4 * g + 1] = qb.y; bv[4 * g + 2] = qb.z; bv[4 * g + 3] = qb.w;
            d[4 * g] = qd.x; d[4 * g + 1] = qd.y; d[4 * g + 2] = qd.z; d[4 * g + 3] = qd.w;
        }
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) d = __builtin_amdgcn_mfma_f32_32x32x2f32(-av[kk], bv[kk], d, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) tt[g * 64 + lane] = make_float4(d[4 * g], d[4 * g + 1], d[4 * g + 2], d[4 * g + 3]);
    };
    lds_flag_t aflag = (lds_flag_t)(R + kRegion);           // alpha blocks published (from the last block up)
    if (tid == 0) *aflag = 0;
    FSTAMP(3);
    for (int j = 0; j < nb; ++j) {
        const int ncol = nb - j;                         // tiles in column j
        const int dw = j & (kFW - 1);                    // wavefront that factorises tile (j, j)
        const int pidx = (wave - dw) & (kFW - 1);        // 1 .. ncol-1: solves panel tile (j + pidx, j)
        const bool has_panel = pidx >= 1 && pidx < ncol;
        const int pbi = j + pidx;
        // z_j: the first wavefront without a panel tile; with eight tiles in the column, the first panel owner does both
        const bool does_z = wave == ((ncol <= kFW - 1) ? ((dw + ncol) & (kFW - 1)) : ((dw + 1) & (kFW - 1)));
        float* Lc = slots + tri_index(j, j) * 1024;      // in: the diagonal tile (accumulator order); out: its factor, column-major
        f32x16 v;
        float zs = 0.f;
        // z_j: the triangle of the forward substitution, eight steps (lane = row; true division, one fmaf per later row)
        auto z_seg = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int k = k0 + kk;
                const float dk = Lc[k * 32 + k];
                const float cand = div_ranged(zs, dk, rcp_refined(dk));      // (the diagonal entry and its reciprocal are off the chain)
                const float zk = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(cand), k));
                const float lik = Lc[k * 32 + l31];
                zs = (l31 == k) ? zk : ((l31 > k) ? fmaf(-lik, zk, zs) : zs);
            }
        };
        // ---- P1.  Two roles, separate instruction streams, four workgroup barriers each: stage q of the owner factorises
        // columns 8q .. 8q+7 of the diagonal tile while the others solve their panel tile against columns 8(q-1) ..
        if (wave == dw) {
            if (V1_PRIO_DIAG) __builtin_amdgcn_s_setprio(V1_PRIO_DIAG);
            f32x16 t;
            {
                const float4* tt = reinterpret_cast<const float4*>(Lc);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 q = tt[g * 64 + lane];
                    t[4 * g] = q.x; t[4 * g + 1] = q.y; t[4 * g + 2] = q.z; t[4 * g + 3] = q.w;
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();                                // (the factor overwrites the slot)
            factor32_mb<0>(t, l31, h, lane, Lc);
            wg_sync();
            factor32_mb<1>(t, l31, h, lane, Lc);
            wg_sync();
            factor32_mb<2>(t, l31, h, lane, Lc);
            wg_sync();
            factor32_mb<3>(t, l31, h, lane, Lc);
            if (V1_PRIO_DIAG) __builtin_amdgcn_s_setprio(0);
            wg_sync();
            if (lane < 32) Ldiag[32 * j + lane] = Lc[lane * 32 + lane];
        } else {
            // column j-1 to the tiles right of column j: dealt to the seven wavefronts, one product per stage, so that the
            // matrix work runs under the serial factorisation and never delays a stage barrier by more than one product
            const int widx = pidx - 1;
            const int nrest = (j > 0) ? (nb - j - 1) * (nb - j) / 2 : 0;
            auto rest_item = [&](int stage) __attribute__((always_inline)) {
                const int m = widx + (kFW - 1) * stage;
                if (m < nrest) {
                    int c = j + 1, rem = m;
                    while (rem >= nb - c) { rem -= nb - c; ++c; }
                    apply_column(c + rem, c, j - 1);
                }
            };
            if (has_panel) {
                const float4* tt = reinterpret_cast<const float4*>(slots + tri_index(pbi, j) * 1024);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 q = tt[g * 64 + lane];
                    v[4 * g] = q.x; v[4 * g + 1] = q.y; v[4 * g + 2] = q.z; v[4 * g + 3] = q.w;
                }
            }
            if (does_z) zs = yv[32 * j + l31];
            rest_item(0);
            wg_sync();
            if (has_panel) diag_solve8<0>(v, Lc, h);
            if (does_z) z_seg(0);
            rest_item(1);
            wg_sync();
            if (has_panel) diag_solve8<8>(v, Lc, h);
            if (does_z) z_seg(8);
            rest_item(2);
            wg_sync();
            if (has_panel) diag_solve8<16>(v, Lc, h);
            if (does_z) z_seg(16);
            rest_item(3);
            wg_sync();
            if (has_panel) diag_solve8<24>(v, Lc, h);
            if (does_z) z_seg(24);
        }
        if (has_panel) {
            float* tile = slots + tri_index(pbi, j) * 1024;
#pragma unroll
            for (int r = 0; r < 16; ++r) tile[a_addr(l31, rowmap_t(r, h))] = v[r];
        }
        if (does_z && lane < 32) yv[32 * j + lane] = zs;
        __syncthreads();
        FSTAMP(4 + 3 * j);
        // ---- P3: z_j into the rows below; column j to the tiles of column j+1
        if (has_panel) {
            // rows of block pbi of the right-hand side: s -= L(pbi, j) z_j, ascending k (the columns of a tile row alternate
            // between the lane halves in groups of four)
            float zl[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) zl[r] = yv[32 * j + rowmap_t(r, h)];
            float s0 = yv[32 * pbi + l31];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float t0 = s0;
#pragma unroll
                for (int i = 0; i < 4; ++i) t0 = fmaf(-v[4 * g + i], zl[4 * g + i], t0);
                float t1 = lo_half(t0);
#pragma unroll
                for (int i = 0; i < 4; ++i) t1 = fmaf(-v[4 * g + i], zl[4 * g + i], t1);
                s0 = hi_half(t1);
            }
            if (lane < 32) yv[32 * pbi + lane] = s0;
        }
        if (j + 1 < nb) {
            const int bi = j + 1 + wave;                 // tiles (bi, j + 1): at most 7
            if (bi < nb) apply_column(bi, j + 1, j);
        }
        __syncthreads();
        FSTAMP(6 + 3 * j);
    }

    // ---------------------------------------------------------------- diagonal slots: inv(L_cc) below the diagonal (order O6),
    // the strictly lower part of L_cc transposed above it (the back substitution still needs it)
    if (wave < nb) {
        const int c = wave;
        float* Lc = slots + tri_index(c, c) * 1024;
        float a[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) a[k] = Lc[k * 32 + l31];      // row l31 of L_cc
        f32x16 x;
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = (rowmap_t(r, h) == l31) ? 1.f : 0.f;
        diag_solve32<true>(x, Lc, h);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rowmap_t(r, h);
            if (row >= l31) Lc[d_addr(row, l31)] = x[r];             // inv[row][col = l31]
        }
        if (lane < 32) {
#pragma unroll
            for (int k = 0; k < 32; ++k)
                if (k < l31) Lc[d_addr(k, l31)] = a[k];              // position (row k, col l31), k < l31, holds L_cc[l31][k]
        }
    }
    __syncthreads();
    FSTAMP(28);

    const int bK = K >> 5, rK = K & 31;     // block row / row of the mean row of Xt
    if (wave < 4) {
        // ------------------------------------------------------------ X = L^-1, two block columns per wavefront
        float* T = R + wave * kStage;
        auto times_inverse = [&](const f32x16& sacc, int b) {
            const float4* td = reinterpret_cast<const float4*>(slots + tri_index(b, b) * 1024);
            float ai[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 q = td[g * 64 + lane];
                const int k0 = 8 * g + 4 * h;                       // k of register 4g + j is k0 + j
                ai[4 * g] = (k0 <= l31) ? q.x : 0.f; ai[4 * g + 1] = (k0 + 1 <= l31) ? q.y : 0.f;
                ai[4 * g + 2] = (k0 + 2 <= l31) ? q.z : 0.f; ai[4 * g + 3] = (k0 + 3 <= l31) ? q.w : 0.f;
            }
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) o = __builtin_amdgcn_mfma_f32_32x32x2f32(ai[kk], sacc[kk], o, 0, 0, 0);
            return o;
        };
        // x = X_bc in C/D layout (lane = column, 16 rows per lane half) -> Xt(b, c) in A-operand order (global) and the
        // transposed tile (X_bc)^T in A-operand order (registers zk: the B operand of the later rows); two passes of 16 rows
        auto emit_keep = [&](const f32x16& x, int b, int c, float (&zk)[16], auto keep_tag) {
            constexpr bool KEEP = decltype(keep_tag)::value;
            float4* xt = reinterpret_cast<float4*>(mp->Xt + (size_t)tri_index(b, c) * 1024);
            const bool skip = (b == bK) && (l31 == rK);            // the mean row is written by the alpha wavefront
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) {
#pragma unroll
                for (int r = 8 * hp; r < 8 * hp + 8; ++r) T[(rowmap_t(r, h) - 16 * hp) * 33 + l31] = x[r];
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                if ((l31 >> 4) == hp && !skip) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float4 q;
                        q.x = T[(l31 - 16 * hp) * 33 + 2 * (4 * g + 0) + h];
                        q.y = T[(l31 - 16 * hp) * 33 + 2 * (4 * g + 1) + h];
                        q.z = T[(l31 - 16 * hp) * 33 + 2 * (4 * g + 2) + h];
                        q.w = T[(l31 - 16 * hp) * 33 + 2 * (4 * g + 3) + h];
                        xt[g * 64 + lane] = q;
                    }
                }
#pragma unroll
                for (int kk = 8 * hp; kk < 8 * hp + 8; ++kk)
                    if (KEEP) zk[kk] = T[(2 * kk + h - 16 * hp) * 33 + l31];
                __builtin_amdgcn_wave_barrier();
            }
        };
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            const int c = pass == 0 ? wave : nb - 1 - wave;
            if (2 * wave > nb - 1 || (pass == 1 && c <= wave)) continue;   // wavefront w: columns w and nb-1-w (w in the lower half)
            const int rows = nb - c;
            float zr[ZR][16];
            {
                f32x16 e;
#pragma unroll
                for (int r = 0; r < 16; ++r) e[r] = (rowmap_t(r, h) == l31) ? 1.f : 0.f;
                emit_keep(times_inverse(e, c), c, c, zr[0], std::true_type());
            }
#pragma unroll
            for (int i = 1; i <= ZR; ++i) {
                if (i < rows) {   // wave-uniform
                    const int b = c + i;
                    f32x16 sacc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
                    for (int pi = 0; pi < i; ++pi) {
                        const float4* tl = reinterpret_cast<const float4*>(slots + tri_index(b, c + pi) * 1024);
                        float av[16];
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const float4 q = tl[g * 64 + lane];
                            av[4 * g] = q.x; av[4 * g + 1] = q.y; av[4 * g + 2] = q.z; av[4 * g + 3] = q.w;
                        }
#pragma unroll
                        for (int kk = 0; kk < 16; ++kk) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(-av[kk], zr[pi][kk], sacc, 0, 0, 0);
                    }
                    // (the last possible row is never an operand: its transposed tile is not kept)
                    if (i < ZR) emit_keep(times_inverse(sacc, b), b, c, zr[i < ZR ? i : 0], std::true_type());
                    else emit_keep(times_inverse(sacc, b), b, c, zr[0], std::false_type());
                }
            }
        }
        FSTAMP(29);
    } else {
        if (V1_PRIO_ALPHA) __builtin_amdgcn_s_setprio(V1_PRIO_ALPHA);
        // ------------------------------------------------------------ alpha = L^-T z, blocked, descending chains (O2).
        // Row block r belongs to wavefront 4 + (r & 3): it folds the blocks below into its rows as they are published
        // (aflag = blocks done, from the last one up), then solves its own 32 x 32 triangle and publishes.  The chain of
        // a row is unchanged: blocks descending, k descending inside a block.
        for (int r = nb - 1 - ((nb - 1 - (wave - 4)) & 3); r >= 0; r -= 4) {
            const int rr = 32 * r;
            float s = yv[rr + l31];
            for (int c = nb - 1; c > r; --c) {
                while (*aflag < nb - c) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const int cr = 32 * c;
                const float* tile = slots + tri_index(c, r) * 1024;
                // the operands of four steps are requested together, ahead of the steps' branches: a step that is its own basic
                // block waits for its two LDS operands in full (measured on the register-resident variant, NOTEBOOK R5.9: 400
                // cycles per step against 130).  Rows >= K - cr are padding (the last row block only; yv holds zeros there).
                const int kv = K - cr;
#pragma unroll
                for (int kg = 7; kg >= 0; --kg) {
                    float tl[4], ak[4];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) { tl[jj] = tile[a_addr(4 * kg + jj, l31)]; ak[jj] = yv[cr + 4 * kg + jj]; }
#pragma unroll
                    for (int jj = 3; jj >= 0; --jj)
                        if (4 * kg + jj < kv) s = fmaf(-tl[jj], ak[jj], s);
                }
            }
            const float* Dc = slots + tri_index(r, r) * 1024;
            float dcol[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) dcol[k] = Dc[d_addr(l31, k)];     // L_rr[k][l31] for k > l31 (other entries unused)
            float b = (rr + l31 < K) ? s : 0.f;
            const float dd = (rr + l31 < K) ? Ldiag[rr + l31] : 1.f;
            const float ddr = rcp_refined(dd);
#pragma unroll
            for (int k = 31; k >= 0; --k) {
                if (rr + k >= K) continue;
                const float t = div_ranged(b, dd, ddr);
                const float ak = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(t), k));
                if (l31 == k) b = ak;
                if (l31 < k) b = fmaf(-dcol[k], ak, b);
            }
            if (lane < 32) yv[rr + lane] = (rr + lane < K) ? b : 0.f;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) *aflag = nb - r;
        }
    }
    if (wave == 4) {
        while (*aflag < nb) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // the mean row of Xt: row K of X carries alpha (K4 reads k*^T alpha off row K of V = X k*)
        if (rK != 0) {
            // row rK of the tiles (bK, c), c <= bK (bK = nb - 1): 16 floats per lane half and tile
            const int c = lane >> 1, hh = lane & 1;
            if (c <= bK) {
                float4* xt = reinterpret_cast<float4*>(mp->Xt + (size_t)tri_index(bK, c) * 1024);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 q;
                    float* qa = reinterpret_cast<float*>(&q);
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx) {
                        const int col = 32 * c + 2 * (4 * g + jx) + hh;
                        qa[jx] = (col < K) ? yv[col] : 0.f;
                    }
                    xt[g * 64 + hh * 32 + rK] = q;
                }
            }
        } else {
            // K a multiple of 32: row K sits alone in an extra block row whose tiles hold only alpha (row 0)
            for (int c = 0; c < nb; ++c) {
                float4* xt = reinterpret_cast<float4*>(mp->Xt + (size_t)tri_index(nb, c) * 1024);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (l31 == 0) {
                        float* qa = reinterpret_cast<float*>(&q);
#pragma unroll
                        for (int jx = 0; jx < 4; ++jx) qa[jx] = yv[32 * c + 2 * (4 * g + jx) + h];
                    }
                    xt[g * 64 + lane] = q;
                }
            }
        }
        if (mp->alpha) {
            for (int r = lane; r < ld; r += 64) mp->alpha[r] = (r < K) ? yv[r] : 0.f;
        }
        if (lane == 0 && blockIdx.x == 0) FSTAMP_AT(30);
    } else if (wave > 4 && mp->L) {
        // ------------------------------------------------------------ the factor itself, column-major with identity padding
        // (models that carry the pointer: parity tests, gpis_ongpis_get_model)
        float* L = mp->L;
        const int t3 = tid - 5 * 64;
        for (int c = 0; c < ld; ++c) {
            for (int r = c + t3; r < ld; r += 3 * 64) {
                float v;
                if (r >= K || c >= K) v = (r == c) ? 1.f : 0.f;
                else if (r == c) v = Ldiag[r];
                else if ((r >> 5) == (c >> 5)) v = slots[tri_index(r >> 5, r >> 5) * 1024 + d_addr(c & 31, r & 31)];
                else v = slots[tri_index(r >> 5, c >> 5) * 1024 + a_addr(r & 31, c & 31)];
                L[r + (size_t)c * ld] = v;
            }
        }
    }
}


#ifdef GPIS_EXPERIMENTS
#include "../../tools/experiments/ongpis_fused_df.inc"   // data-flow schedule of the same factorisation (measured slower)
#include "../../tools/experiments/ongpis_fused_rp.inc"   // register-resident tiles, four wavefronts per cluster, two clusters per CU (measured equal)
#endif

int ongpis_launch_train_fused(const FusedTrainArgs& a, int njobs, int max_nb, hipStream_t s) {
    if (njobs <= 0) return GPIS_OK;
    if (max_nb < 1 || max_nb > 8) return GPIS_ERR_ARG;
    typedef void (*kern_t)(FusedTrainArgs);
    const bool small = max_nb <= 5;   // 15 tiles: two accumulator tiles per wavefront, two workgroups per CU
    kern_t kern = small ? (kern_t)ongpis_train_fused_kernel<5, 4, 3> : (kern_t)ongpis_train_fused_kernel<8, 7, 2>;
#ifdef GPIS_EXPERIMENTS
    if (!small && getenv("GPIS_FUSED_DF") && atoi(getenv("GPIS_FUSED_DF"))) kern = (kern_t)ongpis_train_fused_df_kernel<2>;
#endif
    size_t lds = ongpis_fused_lds_bytes(max_nb);
    int threads = kFT;
#ifdef GPIS_EXPERIMENTS
    const bool rp = !small && getenv("GPIS_FUSED_RP") && atoi(getenv("GPIS_FUSED_RP"));
    if (rp) { kern = (kern_t)ongpis_train_fused_rp_kernel<2>; lds = ongpis_fused_rp_lds_bytes(); threads = kRT; }
#endif
    if (ensure_dynamic_lds((const void*)kern, 160 * 1024) != GPIS_OK) return GPIS_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(njobs), dim3(threads), lds, s, a);
    GPIS_HIP(hipGetLastError());
#ifdef GPIS_INSTRUMENT
#ifdef GPIS_EXPERIMENTS
    if (rp) rp_trace_dump(s); else
#endif
    fused_trace_dump(s);
#endif
    return GPIS_OK;
}

}  // namespace gpis
