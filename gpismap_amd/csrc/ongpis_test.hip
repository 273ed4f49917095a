// K4: batched OnGPIS prediction -- mean (value + gradient) and the four variances
// for tiles of 8 queries against one cluster model.
//
// Replaces the reference per-point chain
//   GPisMap3::test_kernel        cpp/src/GPisMap3.cpp:794-902  (2-D: GPisMap.cpp:665-763)
//     -> OnGPIS::testSinglePoint cpp/src/OnGPIS.cpp:177-216    (2-D: test2Dpoint :218-263)
//        -> matern32_sparse_deriv1_3D (cross)  cpp/src/covFnc.cpp:258-314 (2-D: :404-450)
//        -> k*^T alpha ; L^-1 k* ; column sums of squares
//
// Work decomposition.  The (1+d) cross-covariance columns of 8 queries form a K x 32 right-hand-side
// block B.  The reference solves L V = B by substitution -- a chain of K dependent steps.  Here the
// factor's explicit inverse X = L^-1 comes out of training (K3b, ongpis_train.hip), so
//     V = X B      (lower-triangular matrix product, 32 x 32 tiles, v_mfma_f32_32x32x2_f32)
// has NO dependency between its block rows: every wavefront of the workgroup owns a few block rows
// (accumulators in registers), streams the X tiles of those rows from L2/HBM exactly once and reads the
// B tiles all wavefronts share from LDS.  alpha rides along as row K of X, so row K of V is the mean
// k*^T alpha (one ascending fmaf chain) and needs no separate reduction.
//   * B is generated in chunks of CB column blocks into a THREE-slot LDS ring handed over through LDS counters (round 5): a
//     wavefront multiplies chunk i as soon as its tiles are counted in, then generates its tile of chunk i+2 (the duty goes
//     round the wavefronts) once every wavefront is through with the chunk that slot held before.  No workgroup barrier in
//     the loop: a wavefront may run a chunk ahead of the slowest one, which absorbs the per-chunk imbalance of the
//     triangular product -- with a barrier per chunk (rounds 2-4: two slots, half of the wavefronts generating before
//     multiplying, half after) 15 % of the wave-cycles waited there.  -DK4_RING=0 builds the barrier variant.
//   * block rows are dealt to the wavefronts from the largest down, snake-wise (row b costs b+1 tile
//     products); clusters with more than 4 W block rows run several row groups (B chunks are regenerated
//     for the later, cheaper groups), so there is no upper limit on K.
// Per element V[r][j] is ONE fmaf chain from zero over ascending k -- the order of the oracle's tiled mode.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "ongpis.h"
#include "tile_solve.h"

namespace gpis {

// Pointers read out of a ClusterModel live in global memory; say so, otherwise the compiler must
// emit flat_load (LDS-or-global at run time), which counts on both wait counters.
typedef const float __attribute__((address_space(1))) * gfptr;
typedef const int __attribute__((address_space(1))) * giptr;

// Instrumented builds only (make EXTRA=-DGPIS_INSTRUMENT): per-workgroup cycle stamps, ongpis_test_instr.inc.  No
// wrong-result timing knobs live in this file (the round-2 / round-4 ablations are recorded in NOTEBOOK.md R4.2).
#ifdef GPIS_INSTRUMENT
#include "ongpis_test_instr.inc"
#else
#define K4_TRACE_DECL(A)
#define K4_STAMP() do {} while (0)
#define K4_LAP_START() do {} while (0)
#define K4_LAP(i) do {} while (0)
#define K4_LAP_COUNT(i) do {} while (0)
#define K4_LAP_FLUSH() do {} while (0)
#endif
#ifndef K4_MINW
#define K4_MINW (K4_QS == 1 ? 4 : 2)   // wavefronts per SIMD the register budget is cut for (128 / 256 VGPRs)
#endif

// ---- build-time shape of the kernels (defaults = the measured best; tools/k4_ablate.sh overrides them) ----
#ifndef K4_W3
#define K4_W3 8            // wavefronts per workgroup of the widest class
#endif
#ifndef K4_NBW
#define K4_NBW 4
#endif
#ifndef K4_AVBUF
#define K4_AVBUF 1           // X tile buffers per wave.  2 (next tile prefetched during the current product) pushes the kernel over
                            // 128 VGPRs: one accumulator tile then lives in scratch (2.8 TB of spill traffic per 256^3 pass) --
                            // F = 5 bench 1379 ms/step with 2 buffers, 1233 ms/step with 1 (107 VGPRs, no scratch)
#endif
#ifndef K4_GEN_UNROLL
#define K4_GEN_UNROLL 2     // queries of a lane generated per trip of the B-tile loop (2: two independent chains hide the double-precision latencies; 4 measured slower)
#endif
#ifndef K4_PRIO
#define K4_PRIO 0            // s_setprio level of a wave while it multiplies (0: none)
#endif
#ifndef K4_WP3
#define K4_WP3 0            // generating-only wavefronts of the widest classes (0: unified).  Measured on the F = 5 bench: 2 of 8
                            // waves generating (6 x 4 rows per group) 1497 ms/step, unified 1360 ms/step
#endif

#ifndef K4_RING
#define K4_RING 1           // 1: the B chunks in a THREE-slot ring handed over through LDS counters instead of a workgroup barrier per chunk --
                            // a wavefront may run one chunk ahead of the slowest one, so the per-chunk imbalance of the triangular product
                            // (profiles/r05_k4_stamps.txt: 15 % of the wave-cycles wait at that barrier) is absorbed instead of waited for
#endif
#ifndef K4_RING_SLOTS
#define K4_RING_SLOTS 3
#endif
#ifndef K4_XCD_REMAP
#define K4_XCD_REMAP 1      // eight consecutive tiles of the list on one XCD (0: tile = workgroup id)
#endif
#ifndef K4_BT
#define K4_BT 1             // B tiles in LDS k-contiguous per column: Bt[n][h][kk] = B[2 kk + h][n], so that the 16 operand values of a lane are
                            // FOUR 16-byte reads instead of sixteen 4-byte reads (the operand reads of 16 wavefronts took half of the LDS cycles)
#endif
constexpr int kTileStride = 36;                 // floats per row (K4_BT: per column) of a B tile in LDS: conflict-free writes AND operand reads
constexpr int kTileFloats = 32 * kTileStride;   // 1152

// K4_RING hand-overs through LDS counters (cumulative, never reset).  The counter is read through readfirstlane: a per-lane loop
// condition would make everything after the loop divergent to the compiler (waterfall loops around the buffer loads, wave-uniform
// values demoted to VGPRs).  The wait is bounded: a protocol error must not hang the queue (the results are then wrong and the
// parity tests say so).
typedef volatile int __attribute__((address_space(3))) * lds_cnt_t;
__device__ __forceinline__ void k4_ring_wait(lds_cnt_t p, int need, lds_cnt_t expired) {
    int spins = 0;
    for (; spins < (1 << 22); ++spins) {
        if (__builtin_amdgcn_readfirstlane(*p) >= need) break;
        if ((spins & 1023) == 1023 && __builtin_amdgcn_readfirstlane(*expired) != 0) break;   // (a partner gave up already: one bounded wait per workgroup, not one per chunk)
        __builtin_amdgcn_s_sleep(1);
    }
    if (spins == (1 << 22)) *expired = 1;      // (a protocol error: the tile's results are written as NaN, never as plausible numbers)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void k4_ring_signal(lds_cnt_t p, int lane) {
    __builtin_amdgcn_s_waitcnt(0xc07f);                       // this wavefront's LDS traffic on the slot is complete
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add((int __attribute__((address_space(3)))*)p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// W wavefronts of which WP only generate B tiles (0: every wavefront generates and multiplies), NBW block rows per
// multiplying wavefront and row group, QS query sets of 8 per workgroup (every X tile feeds QS tile products)
template <int W, bool TABLE, int QS, int NBW, int WP>
__global__ __launch_bounds__(64 * W, K4_MINW) void ongpis_eval_kernel(EvalArgs A) {
    constexpr int WC = W - WP;    // wavefronts that own block rows (consumers)
    constexpr int RG = NBW * WC;  // block rows per row group
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // Workgroup ids go round the eight XCDs (id b runs on XCD b % 8), and the tiles of ONE cluster are consecutive in the tile list:
    // taken as they come, eight consecutive 8-query tiles land on eight different L2s and every one of them fetches the cluster's X
    // from HBM for itself -- the stress configuration (64 queries = 8 tiles per cluster) read every model eight times.  Inside
    // every block of 64 workgroup ids the (id / 8, id % 8) grid is transposed: XCD x takes the tiles 8x .. 8x + 7 of the block, so
    // eight consecutive tiles share one L2 and run at about the same time; every XCD still takes every eighth group of eight
    // (cutting the LIST into eight contiguous ranges instead was measured: stress +6 %, but the 256^3 bench 0.75 -> 0.63 of peak,
    // the ranges of a launch differ by the K^2 of their clusters).  Which workgroup evaluates a tile does not enter any result.
    int tile = blockIdx.x;
    if (K4_XCD_REMAP) {
        const int n64 = (int)gridDim.x & ~63;
        if (tile < n64) tile = (tile & ~63) | ((tile & 7) << 3) | ((tile >> 3) & 7);
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps the row ownership logic in SGPRs
    const int h = lane >> 5, l31 = lane & 31;
    const ClusterModel* __restrict__ mp = A.models + A.tile_model[tile];   // only the fields needed are read (scalar loads)
    const int N = mp->N, K = mp->K, ld = mp->ld, nb = mp->nb, dim = mp->dim;
    const int nbx = ld >> 5;      // block rows of V: ceil((K+1)/32), row K = mean
    const int joff = A.tile_off[tile], jcnt = A.tile_cnt[tile];   // 1..16 queries
    const int nset = (QS == 2 && jcnt > 8) ? 2 : 1;
    const int CB = A.cb, NSLOT = A.nslot;
    K4_TRACE_DECL(A)
    K4_STAMP();

    // LDS carve (all dynamic, 16-byte aligned pieces)
    constexpr int NC = 32 * QS;   // result columns of the workgroup
    constexpr int ES = 8 * QS + 1;  // doubles per point in the exp table: odd stride -> conflict-free 8-byte reads across points
    float* red = reinterpret_cast<float*>(smem);                       // [W][NC] sums of squares, then [NC] means
    float4* s_xq = reinterpret_cast<float4*>(red + W * NC + NC);       // [16] the tile's query points
    int* s_ri = reinterpret_cast<int*>(s_xq + 16) + 32;                // [ld] row -> point | component
    lds_cnt_t ring_cnt = (lds_cnt_t)(reinterpret_cast<int*>(s_xq + 16));    // K4_RING: [0..3] tiles generated per slot, [8..11] wavefronts done per slot (cumulative), [15] a wait expired
    if (K4_RING && tid < 16) ring_cnt[tid] = 0;
    float4* s_x4 = reinterpret_cast<float4*>(s_ri + ld);               // [N]   (ld is a multiple of 32 -> 16-B aligned)
    double* etab = reinterpret_cast<double*>(s_x4 + N);                // [N][16] exp table (optional)
    float* Bbuf = reinterpret_cast<float*>(etab + (TABLE ? (((size_t)N * ES + 1) & ~(size_t)1) : 0));   // [NSLOT][CB][QS][32*36]

    const float scale = mp->scale;
    const float a = (float)(sqrt(3.0) / (double)scale);

    {   // stage 0: per-cluster vectors and the queries into LDS with coalesced loads
        giptr g_ri = (giptr)mp->rowinfo;
        gfptr g_x4 = (gfptr)mp->x4;
        for (int i = tid; i < ld; i += 64 * W) s_ri[i] = g_ri[i];
        for (int i = tid; i < 4 * N; i += 64 * W) reinterpret_cast<float*>(s_x4)[i] = g_x4[i];
        if (tid < 16) s_xq[tid] = (tid < jcnt) ? A.xq[A.job_q[joff + tid]] : make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
    }
    K4_STAMP();
    const float4* x4 = s_x4;
    // X tiles (A-operand order, 4 x 16-byte loads per tile) through a buffer resource: one VGPR byte offset per
    // lane + scalar tile offsets
    const int ntl = nbx * (nbx + 1) / 2;
    const __amdgpu_buffer_rsrc_t Xrs = __builtin_amdgcn_make_buffer_rsrc((void*)mp->Xt, 0, (unsigned)ntl * 4096u, 0x00020000);
    const int Tvoff = lane * 16;

    // ---- stage 1: exp table, one entry per (training point, query slot) ----
    if (TABLE) {
        for (int idx = tid; idx < N * 8 * QS; idx += 64 * W) {
            const int p = idx / (8 * QS), s = idx % (8 * QS);
            double e = 0.0;
            if (s < jcnt) {
                const float4 xp = x4[p];
                const float4 q = s_xq[s];
                const float d0 = xp.x - q.x, d1 = xp.y - q.y, d2 = xp.z - q.z;
                const float r = (dim == 3) ? sqrtf((d0 * d0 + d1 * d1) + d2 * d2) : sqrtf(d0 * d0 + d1 * d1);
                e = exp((double)(-a * r));
            }
            etab[p * ES + s] = e;
        }
        __syncthreads();
    }

    K4_STAMP();
    // ---- generation of one B tile (column block c, query set qs) into an LDS tile: lane (r, qh) produces the 16
    // entries of tile row r for the queries 4qh..4qh+3 of the set (distance, exp-table lookup and coefficient set-up
    // shared by the 4 components) and writes them as four 16-byte stores.  KIND = row type of the whole tile when it
    // is uniform (0: value rows, 1..3: d/dx_c rows -- the rows are ordered by type, so almost every tile is uniform
    // and the type selects of covFnc.cpp:292-308 fold away), -1: mixed tile, per-row type.
    const int ngr = (dim > 0) ? (K - N) / dim : 0;   // rows per derivative component
    auto row_type = [&](int r) { return r < N ? 0 : 1 + (r - N) / (ngr > 0 ? ngr : 1); };
    auto emit_rows = [&](auto kind_tag, int c, int qs, float* tbuf) {
        constexpr int KIND = decltype(kind_tag)::value;
        const int rr_ = lane & 31, qh = lane >> 5;
        const int row = c * 32 + rr_;
        float4* trow = reinterpret_cast<float4*>(tbuf + rr_ * kTileStride + 16 * qh);
        int p = 0, cr = 0;
        float4 xp = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < K) {
            const int info = s_ri[row];
            p = info & 0x0FFFFFFF;
            cr = (KIND >= 0) ? KIND : ((info >> 28) & 0xF);
            xp = x4[p];
        }
#pragma unroll K4_GEN_UNROLL
        for (int j = 0; j < 4; ++j) {
            const int q = 8 * qs + 4 * qh + j;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < K && q < jcnt) {
                const float4 xq = s_xq[q];
                float d[3] = {xp.x - xq.x, xp.y - xq.y, xp.z - xq.z};
                float rr = (dim == 3) ? sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) : sqrtf(d[0] * d[0] + d[1] * d[1]);
                double e = TABLE ? etab[p * ES + q] : exp((double)(-a * rr));
                float v0, v1, v2, v3;
                if (cr == 0) {
                    v0 = d_kf(rr, a, e); v1 = d_kf1(d[0], a, e); v2 = d_kf1(d[1], a, e); v3 = d_kf1(d[2], a, e);
                } else {
                    const float dr = cr == 1 ? d[0] : (cr == 2 ? d[1] : d[2]);
                    v0 = -d_kf1(dr, a, e);
                    // mixed second derivatives: lower component first (covFnc.cpp:300-308)
                    v1 = (cr == 1) ? d_kf2(rr, d[0], d[0], 1.0f, a, e) : d_kf2(rr, d[0], dr, 0.0f, a, e);
                    v2 = (cr == 2) ? d_kf2(rr, d[1], d[1], 1.0f, a, e)
                                   : (cr == 1 ? d_kf2(rr, d[0], d[1], 0.0f, a, e) : d_kf2(rr, d[1], d[2], 0.0f, a, e));
                    v3 = (cr == 3) ? d_kf2(rr, d[2], d[2], 1.0f, a, e) : d_kf2(rr, dr, d[2], 0.0f, a, e);
                }
                if (dim == 2) v3 = 0.f;
                o = make_float4(v0, v1, v2, v3);
            }
            if (K4_BT) {
                // element (row, n = 16 qh + 4 j + comp) -> tbuf[n * 36 + (row & 1) * 16 + (row >> 1)]: the 32 rows of a wavefront's
                // store instruction fall into 32 different banks
                float* tcol = tbuf + (16 * qh + 4 * j) * kTileStride + (rr_ & 1) * 16 + (rr_ >> 1);
                tcol[0] = o.x; tcol[kTileStride] = o.y; tcol[2 * kTileStride] = o.z; tcol[3 * kTileStride] = o.w;
            } else {
                trow[j] = o;
            }
        }
    };
    auto gen_tile = [&](int c, int qs, float* tbuf) {
        const int r0 = c * 32, r1 = min(K, r0 + 32) - 1;
        const int k0 = row_type(r0), k1 = row_type(r1);
        if (k0 != k1) emit_rows(std::integral_constant<int, -1>(), c, qs, tbuf);
        else if (k0 == 0) emit_rows(std::integral_constant<int, 0>(), c, qs, tbuf);
        else if (k0 == 1) emit_rows(std::integral_constant<int, 1>(), c, qs, tbuf);
        else if (k0 == 2) emit_rows(std::integral_constant<int, 2>(), c, qs, tbuf);
        else emit_rows(std::integral_constant<int, 3>(), c, qs, tbuf);
    };

    // ---- B chunks.  The chunks of all row groups form one sequence gci = 0, 1, ...; chunk gci lives in ring slot
    // gci % NSLOT (ring mode: three slots and LDS counters; barrier mode: two slots, one workgroup barrier per chunk).
    // WP = 0: tile j of a chunk is made by wave j % W, every wave generates AND multiplies.  WP > 0 (the widest classes):
    // the last WP waves only generate, the first WC only multiply -- the accumulators (64 VGPRs) and the generation code
    // (double-precision kernel entries) are then never live in the same wave, which is what keeps the kernel inside
    // 128 VGPRs without spilling accumulator tiles to scratch, and generation never sits in a multiplying wave's
    // instruction stream.
    const int ngroups = (nbx + RG - 1) / RG;
    auto group_cmax = [&](int g) { return min(nbx - 1 - g * RG, nb - 1); };   // last column block a row of group g multiplies with
    int pg = 0, pci = 0, pgci = 0;   // producer cursor: group, chunk in group, chunk in sequence
    // K4_RING: expected cumulative tile count per slot (what the slot's counter shows once every chunk produced into it so far is
    // complete); a wait is bounded (a protocol error would otherwise hang the queue: the results are then wrong and the parity
    // tests say so)
    int exp_gen0 = 0, exp_gen1 = 0, exp_gen2 = 0, exp_gen3 = 0;      // (up to four slots: K4_RING_SLOTS)
    auto produce_next = [&]() {
        if (pg >= ngroups) return;
        const int cmax = group_cmax(pg);
        const int pslot = pgci % NSLOT;
        float* slot = Bbuf + (size_t)pslot * CB * QS * kTileFloats;
        const int c0 = pci * CB;
        if (K4_RING) {
            // tile t of chunk p is made by wavefront (t + p CB) mod W: the duty goes round, every wavefront generates the same
            // number of tiles over a few chunks.  The slot is free once every wavefront has multiplied the chunk it held before.
            const int nt = min(CB, cmax - c0 + 1);
            const int my_t = (((wave - pgci * CB) % W) + W) % W;
            if (my_t < nt) {
                k4_ring_wait(ring_cnt + 8 + pslot, W * (pgci / NSLOT), ring_cnt + 15);
                for (int t = my_t; t < nt; t += W) {      // (chunks wider than the workgroup has wavefronts: several tiles per wavefront)
                    for (int qs = 0; qs < nset; ++qs) gen_tile(c0 + t, qs, slot + (size_t)(t * QS + qs) * kTileFloats);
                    k4_ring_signal(ring_cnt + pslot, lane);
                }
            }
            // (unconditional adds: an if / else chain over the three becomes a SELECT OF POINTERS to captured variables, and with it every
            // capture of the kernel's lambdas stays in scratch memory -- 240 allocas survived)
            exp_gen0 += (pslot == 0) ? nt : 0; exp_gen1 += (pslot == 1) ? nt : 0; exp_gen2 += (pslot == 2) ? nt : 0; exp_gen3 += (pslot == 3) ? nt : 0;
        } else {
            const int j0 = (WP > 0) ? wave - WC : wave, jstep = (WP > 0) ? WP : W;
            for (int j = j0; j < CB && c0 + j <= cmax; j += jstep)
                for (int qs = 0; qs < nset; ++qs) gen_tile(c0 + j, qs, slot + (size_t)(j * QS + qs) * kTileFloats);
        }
        ++pgci;
        if (++pci > cmax / CB) { pci = 0; ++pg; }
    };

    auto load_a = [&](float (&av)[16], int b, int c) {
        const int sbase = (b * (b + 1) / 2 + c) * 4096;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            auto q = __builtin_amdgcn_raw_buffer_load_b128(Xrs, Tvoff, sbase + g * 1024, 0);
            av[4 * g + 0] = __uint_as_float(q[0]); av[4 * g + 1] = __uint_as_float(q[1]);
            av[4 * g + 2] = __uint_as_float(q[2]); av[4 * g + 3] = __uint_as_float(q[3]);
        }
    };
    // acc_q += X(b, c) B^q_c for the query sets q ; B operand kk of lane (h, n) = B^q_c[2 kk + h][n] from LDS, the
    // independent accumulator chains of the sets interleaved
    const bool two = (QS == 2) && (nset == 2);
    auto mfma_tile = [&](f32x16& acc0, f32x16& acc1, const float (&av)[16], const float* Bt) {
        if (K4_BT) {
            const float4* Bq = reinterpret_cast<const float4*>(Bt + l31 * kTileStride + h * 16);
            float bv[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) { const float4 q = Bq[g]; bv[4 * g] = q.x; bv[4 * g + 1] = q.y; bv[4 * g + 2] = q.z; bv[4 * g + 3] = q.w; }
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[kk], acc0, 0, 0, 0);
            if (QS == 2) {
                if (two) {
                    const float4* Bq1 = reinterpret_cast<const float4*>(Bt + kTileFloats + l31 * kTileStride + h * 16);
#pragma unroll
                    for (int g = 0; g < 4; ++g) { const float4 q = Bq1[g]; bv[4 * g] = q.x; bv[4 * g + 1] = q.y; bv[4 * g + 2] = q.z; bv[4 * g + 3] = q.w; }
#pragma unroll
                    for (int kk = 0; kk < 16; ++kk) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[kk], acc1, 0, 0, 0);
                }
            }
            return;
        }
        const float* Bl = Bt + h * kTileStride + l31;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], Bl[kk * 2 * kTileStride], acc0, 0, 0, 0);
            if (QS == 2) { if (two) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], Bl[kTileFloats + kk * 2 * kTileStride], acc1, 0, 0, 0); }
        }
    };

    float ss[2] = {0.f, 0.f};         // partial sums of squares of V over this lane's rows (order O3, oracle reduce_ss)
    float mean_val[2] = {0.f, 0.f};   // row K of V (the lane that owns it)
    const int LA = NSLOT - 1;         // chunks generated ahead of the multiplication
    int total_chunks = 0;
    for (int g = 0; g < ngroups; ++g) total_chunks += group_cmax(g) / CB + 1;
    if (WP > 0 && wave >= WC) {
        // ---- generating wavefronts
        for (int i = 0; i < LA; ++i) produce_next();
        __syncthreads();
        for (int gci = 0; gci < total_chunks; ++gci) {
            produce_next();          // chunk gci + LA, while the others multiply chunk gci
            __syncthreads();
        }
    } else {
        // ---- multiplying wavefronts (and, with WP = 0, generating ones)
        // barrier mode: half of the waves generate before multiplying, half after; ring mode: every wave multiplies first (a wave that
        // generates first waits for the slowest wave of the chunk before: measured 0.4 % slower)
        const bool gen_first = !K4_RING && ((W < 2) || (wave < W / 2));
        if (WP == 0) for (int i = 0; i < LA; ++i) produce_next();
        if (!K4_RING) __syncthreads();
        K4_STAMP();
        int gci = 0;
        for (int g = 0; g < ngroups; ++g) {
            // this wave's block rows in group g: slot t holds the (g RG + t WC + q)-th largest row, q snaking with t
            // (Round 5: pairing complementary rows q / 7 - q on the two wavefronts of a SIMD -- every SIMD then carries the same number
            // of products in every chunk -- measured 0.7 % SLOWER on the bench and 4 points slower at K = 1598: a wavefront left alone on
            // its SIMD with seven products does not hide its own X-tile loads.  The row table also fixes the order of the variance sums
            // (oracle reduce_ss), so it is not free to change.  NOTEBOOK R5.2.)
            int brow[NBW];
#pragma unroll
            for (int t = 0; t < NBW; ++t) {
                const int i = g * RG + t * WC + ((t & 1) ? (WC - 1 - wave) : wave);
                brow[t] = nbx - 1 - i;      // < 0: no row
            }
            const int cmax = group_cmax(g);
            const int nch = cmax / CB + 1;
            f32x16 acc[NBW][QS];
#pragma unroll
            for (int t = 0; t < NBW; ++t)
#pragma unroll
                for (int q = 0; q < QS; ++q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][q][r] = 0.f;

            K4_LAP_START();
            for (int ci = 0; ci < nch; ++ci, ++gci) {
                if (WP == 0 && gen_first) produce_next();
                K4_LAP(0);
                if (K4_RING) {      // chunk gci complete in its slot?  (its tiles were generated one or two chunks ago)
                    const int sl = gci % NSLOT;
                    k4_ring_wait(ring_cnt + sl, sl == 0 ? exp_gen0 : (sl == 1 ? exp_gen1 : (sl == 2 ? exp_gen2 : exp_gen3)), ring_cnt + 15);
                    K4_LAP(2);
                }
                {
                    if (K4_PRIO) __builtin_amdgcn_s_setprio(K4_PRIO);
                    const float* buf = Bbuf + (size_t)(gci % NSLOT) * CB * QS * kTileFloats;
                    const int c0 = ci * CB;
#pragma unroll
                    for (int t = 0; t < NBW; ++t) {
                        const int b = brow[t];
                        if (b >= c0) {
                            const int cend = min(min(c0 + CB - 1, b), cmax);
#if K4_AVBUF == 1
                            if constexpr (TABLE) {
                            float av1[16];
#pragma unroll 1
                            for (int c = c0; c <= cend; ++c) {
                                load_a(av1, b, c);
                                mfma_tile(acc[t][0], acc[t][QS - 1], av1, buf + (size_t)(c - c0) * QS * kTileFloats);
                            }
                            } else {
                            // (clusters too large for the exp table in LDS: generation is heavier there, fewer wavefronts compete for
                            // the matrix pipe at a time, and a product that waits for its X tile shows -- measured +4.8 % at
                            // K = 2040 / 2380, -1 % on the table kernels, so only here)
                            // ONE X-tile buffer used as a ring of four 16-byte pieces: as soon as the four matrix instructions
                            // that read piece g have been issued, piece g of the NEXT tile of this block row is requested into
                            // the same registers -- the next product's operands arrive under the current product (3/4 of a
                            // product = 768 cycles ahead, about one L2 round trip) without a second buffer.  The last tile of
                            // the row re-requests itself (clamped address: branch-free, harmless).
                            float av1[16];
                            load_a(av1, b, c0);
#pragma unroll 1
                            for (int c = c0; c <= cend; ++c) {
                                const float* Bl = buf + (size_t)(c - c0) * QS * kTileFloats + h * kTileStride + l31;
                                const float4* Bq = reinterpret_cast<const float4*>(buf + (size_t)(c - c0) * QS * kTileFloats + l31 * kTileStride + h * 16);
                                const int cn = min(c + 1, cend);
                                const int nbase = (b * (b + 1) / 2 + cn) * 4096;
#pragma unroll
                                for (int g = 0; g < 4; ++g) {
                                    float4 bq = make_float4(0.f, 0.f, 0.f, 0.f), bq1 = make_float4(0.f, 0.f, 0.f, 0.f);
                                    if (K4_BT) bq = Bq[g];
                                    if (K4_BT && QS == 2) { if (two) bq1 = Bq[g + kTileFloats / 4]; }       // (the second query set's tile follows the first)
#pragma unroll
                                    for (int j = 0; j < 4; ++j) {
                                        acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[4 * g + j], K4_BT ? (j == 0 ? bq.x : (j == 1 ? bq.y : (j == 2 ? bq.z : bq.w))) : Bl[(4 * g + j) * 2 * kTileStride], acc[t][0], 0, 0, 0);
                                        if (QS == 2) { if (two) acc[t][QS - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[4 * g + j], K4_BT ? (j == 0 ? bq1.x : (j == 1 ? bq1.y : (j == 2 ? bq1.z : bq1.w))) : Bl[kTileFloats + (4 * g + j) * 2 * kTileStride], acc[t][QS - 1], 0, 0, 0); }
                                    }
                                    auto q = __builtin_amdgcn_raw_buffer_load_b128(Xrs, Tvoff, nbase + g * 1024, 0);
                                    av1[4 * g + 0] = __uint_as_float(q[0]); av1[4 * g + 1] = __uint_as_float(q[1]);
                                    av1[4 * g + 2] = __uint_as_float(q[2]); av1[4 * g + 3] = __uint_as_float(q[3]);
                                    __builtin_amdgcn_sched_barrier(0);     // keep the request where it is: right behind its four instructions
                                }
                            }
                            }
#else
                            float av[2][16];
                            load_a(av[0], b, c0);
#pragma unroll 1
                            for (int c = c0; c <= cend; c += 2) {
                                if (c + 1 <= cend) load_a(av[1], b, c + 1);
                                mfma_tile(acc[t][0], acc[t][QS - 1], av[0], buf + (size_t)(c - c0) * QS * kTileFloats);
                                if (c + 1 <= cend) {
                                    if (c + 2 <= cend) load_a(av[0], b, c + 2);
                                    mfma_tile(acc[t][0], acc[t][QS - 1], av[1], buf + (size_t)(c + 1 - c0) * QS * kTileFloats);
                                }
                            }
#endif
                        }
                    }
                }
                if (K4_PRIO) __builtin_amdgcn_s_setprio(0);
                K4_LAP(1);
                if (K4_RING) k4_ring_signal(ring_cnt + 8 + gci % NSLOT, lane);       // this wavefront is through with chunk gci
                if (WP == 0 && !gen_first) produce_next();
                K4_LAP(0);
                if (!K4_RING) __syncthreads();   // chunk gci multiplied by every wave, chunk gci + LA generated
                K4_LAP(2);
                K4_LAP_COUNT(4);
            }
            // sums of squares of the finished rows; row K (block nbx-1: group 0, slot 0, wave 0) is the mean
#pragma unroll
            for (int t = 0; t < NBW; ++t) {
                if (brow[t] < 0) continue;
                const bool has_mean = (g == 0 && t == 0 && wave == 0);
                const int kr = K & 31;
#pragma unroll
                for (int q = 0; q < QS; ++q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[t][q][r];
                        if (has_mean && rowmap_t(r, h) == kr) mean_val[q] = v;
                        else ss[q] = fmaf(v, v, ss[q]);
                    }
            }
            K4_LAP(3);
        }
    }
    K4_LAP_FLUSH();

    K4_STAMP();
    // ---- reduce partials (lane halves, then waves in fixed order) ----
#pragma unroll
    for (int q = 0; q < QS; ++q) {
        ss[q] = ss[q] + __shfl_xor(ss[q], 32);
        if (h == 0) red[wave * NC + q * 32 + l31] = ss[q];
        if (wave == 0 && h == ((K >> 2) & 1)) red[W * NC + q * 32 + l31] = mean_val[q];   // the half that owns row K & 31
    }
    __syncthreads();
    {
        const int col = tid & 63;                       // wave 0: one lane per (query, component) column
        const int qi = col >> 2, cq = col & 3;
        if (wave == 0 && col < NC && qi < jcnt && cq <= dim) {
            float vs = 0.f;
            for (int w = 0; w < W; ++w) vs += red[w * NC + col];
            const float ms = red[W * NC + col];
            float* o = A.out + (size_t)A.job_out[joff + qi] * 8;
            const float tos = (float)(3.0 / (double)(scale * scale));  // OnGPIS.h:58
            float var;
            if (dim == 3)  // OnGPIS.cpp:208-213
                var = (cq == 0) ? (float)(1.001 - (double)vs) : (float)((double)tos + 0.001 - (double)vs);
            else           // OnGPIS.cpp:235-237
                var = (cq == 0) ? (float)(1.01 - (double)vs) : (float)((double)tos + 0.1 - (double)vs);
            const bool ring_failed = K4_RING && ring_cnt[15] != 0;      // a bounded ring wait ran out (protocol error): poison, loudly
            o[cq] = ring_failed ? __uint_as_float(0x7fc00000u) : ms;
            o[4 + cq] = ring_failed ? __uint_as_float(0x7fc00000u) : var;
        }
    }
    K4_STAMP();
}


#ifdef GPIS_EXPERIMENTS
#include "../../tools/experiments/ongpis_eval_seg.inc"   // several consecutive tiles per workgroup, pipelined across the tiles (measured no faster)
#endif

// Size classes by nbx = ceil((K+1)/32): W wavefronts per workgroup (ongpis.h, ongpis_class_of_nbx).
static const int kClassW[ONGPIS_NCLASS] = {1, 2, 4, 4, K4_W3, K4_W3, K4_W3};
constexpr int kQS = ONGPIS_TILE_Q / 8;
constexpr int kWavesPerCU = 4 * K4_MINW;   // resident wavefronts per CU the register budget of the kernels admits

static size_t eval_lds_fixed(int W, int maxN, int maxLd, int use_table) {
    return sizeof(float) * (W * 32 * kQS + 32 * kQS) + 16 * sizeof(float4) + 32 * sizeof(int) + sizeof(int) * (size_t)maxLd + 16 * (size_t)maxN +
           (use_table ? sizeof(double) * (((size_t)maxN * (8 * kQS + 1) + 1) & ~(size_t)1) : 0);
}

int ongpis_eval_launch(int wclass, int ntiles, int maxN, int maxLd, const EvalArgs& args_in, hipStream_t s) {
    if (ntiles <= 0) return GPIS_OK;
    if (wclass < 0 || wclass >= ONGPIS_NCLASS) return GPIS_ERR_ARG;
    EvalArgs args = args_in;
#ifdef GPIS_EXPERIMENTS   // the archived resident-X kernel for clusters of at most ONGPIS_SMALL_NBX block rows (tools/experiments/)
    if (getenv("GPIS_SMALL_KERNEL") && atoi(getenv("GPIS_SMALL_KERNEL")) && wclass <= 2 && maxLd / 32 <= ONGPIS_SMALL_NBX && ongpis_eval_small_lds(maxN, maxLd) <= (size_t)160 * 1024)
        return ongpis_eval_small_launch(ntiles, maxN, maxLd, args_in, s);
#endif
    const int W = kClassW[wclass];
    // LDS budget: the register file admits kWavesPerCU wavefronts per CU, i.e. kWavesPerCU / W workgroups; give each an
    // equal share of the 160 KB and spend what the per-cluster tables leave on the ring of B chunks: NSLOT slots
    // of cb column blocks x kQS query sets (three slots of >= 2 blocks when they fit, else two; cb <= 8).
    const size_t hard = 158 * 1024;
    const size_t share = std::min(hard, hard * W / kWavesPerCU);
    const size_t blk = kQS * sizeof(float) * kTileFloats;    // one column block, all query sets
    int use_table = args_in.use_table ? 1 : 0;
#if K4_RING || defined(K4_NO_TABLE)
    // Ring mode: never an exp table.  The table of a K ~ 1000 cluster takes 22 KB of the workgroup's 79 KB, and what the ring wants is
    // WIDE chunks: three slots of five column blocks without the table 0.772-0.775 of peak, three slots of three with it 0.751,
    // two slots of five with it and a barrier per chunk (rounds 2-4) 0.748-0.753.  Every entry evaluates its own exponential, the
    // path the largest clusters always took (test_predict_without_exp_table_is_identical).
    use_table = 0;
#endif
#if K4_RING
    const int nslot = K4_RING_SLOTS;
#elif defined(K4_NSLOT)
    const int nslot = K4_NSLOT;
#else
    const int nslot = 2;   // (barrier mode) two large chunks beat three smaller ones: 811 vs 851 ms on the 256^3 bench
#endif
    size_t fixed = eval_lds_fixed(W, maxN, maxLd, use_table);
    if (use_table && fixed + 4 * blk > share) {              // the exp table does not fit beside a useful ring
        const size_t f0 = eval_lds_fixed(W, maxN, maxLd, 0);
        if (f0 + 4 * blk <= share || fixed + 2 * blk > hard) { use_table = 0; fixed = f0; }
    }
    if (fixed + nslot * blk > hard) return GPIS_ERR_LIMIT;
    // (the small classes keep their occupancy: chunks of at least two / three column blocks at the price of fewer workgroups per CU
    // measured 35 -> 30 / 24 % at K = 204, 17 -> 13 / 9 % at K = 102)
    const size_t budget = std::min(hard, std::max(share, fixed + nslot * blk));
    const int nblk = (int)((budget - fixed) / blk);           // column blocks the ring can hold
    int cb = std::max(1, std::min(nblk / nslot, 8));
    cb = std::min(cb, std::max(1, maxLd / 32));
    args.use_table = use_table;
    args.cb = cb;
    args.nslot = nslot;
    const size_t lds = fixed + (size_t)nslot * cb * blk;
    typedef void (*kern_t)(EvalArgs);
    static const kern_t kern[2][4] = {
        {ongpis_eval_kernel<1, false, kQS, K4_NBW, 0>, ongpis_eval_kernel<2, false, kQS, K4_NBW, 0>, ongpis_eval_kernel<4, false, kQS, K4_NBW, 0>,
         ongpis_eval_kernel<K4_W3, false, kQS, K4_NBW, K4_WP3>},
#if K4_RING || defined(K4_NO_TABLE)      // (the table kernels are not part of such a build)
        {nullptr, nullptr, nullptr, nullptr}};
#else
        {ongpis_eval_kernel<1, true, kQS, K4_NBW, 0>, ongpis_eval_kernel<2, true, kQS, K4_NBW, 0>, ongpis_eval_kernel<4, true, kQS, K4_NBW, 0>,
         ongpis_eval_kernel<K4_W3, true, kQS, K4_NBW, K4_WP3>}};
#endif
    const int kidx = wclass < 3 ? wclass : (wclass == 3 ? 2 : 3);
#ifdef GPIS_EXPERIMENTS
    // several consecutive tiles per workgroup, software-pipelined across the tiles (K4_SEG tiles; 0: one tile per workgroup, the kernel above)
    static const int seg_env = getenv("GPIS_K4_SEG") ? atoi(getenv("GPIS_K4_SEG")) : 0;
    if (seg_env > 0 && kQS == 1) {
        typedef void (*skern_t)(EvalArgs, int, int);
        static const skern_t skern[2][4] = {
            {ongpis_eval_seg_kernel<1, false, K4_NBW>, ongpis_eval_seg_kernel<2, false, K4_NBW>, ongpis_eval_seg_kernel<4, false, K4_NBW>, ongpis_eval_seg_kernel<K4_W3, false, K4_NBW>},
            {ongpis_eval_seg_kernel<1, true, K4_NBW>, ongpis_eval_seg_kernel<2, true, K4_NBW>, ongpis_eval_seg_kernel<4, true, K4_NBW>, ongpis_eval_seg_kernel<K4_W3, true, K4_NBW>}};
        if (ensure_dynamic_lds((const void*)skern[use_table][kidx], 160 * 1024) != GPIS_OK) return GPIS_ERR_HIP;
        args.nslot = 2;
        const size_t lds2 = fixed + (size_t)2 * cb * blk;
        hipLaunchKernelGGL(skern[use_table][kidx], dim3((ntiles + seg_env - 1) / seg_env), dim3(64 * W), lds2, s, args, ntiles, seg_env);
        const hipError_t le2 = hipGetLastError();
        if (le2 != hipSuccess) { fprintf(stderr, "[gpismap_amd] K4 launch failed: %s (class %d, %d waves, %d tiles, segment %d)\n", hipGetErrorString(le2), wclass, W, ntiles, seg_env); return GPIS_ERR_HIP; }
        return GPIS_OK;
    }
#endif
    if (ensure_dynamic_lds((const void*)kern[use_table][kidx], 160 * 1024) != GPIS_OK) return GPIS_ERR_HIP;
#ifdef GPIS_INSTRUMENT
    k4_trace_arm(args, s);
#endif
    hipLaunchKernelGGL(kern[use_table][kidx], dim3(ntiles), dim3(64 * W), lds, s, args);
#ifdef GPIS_INSTRUMENT
    k4_trace_dump(s, W, use_table, ntiles, maxN, maxLd, cb);
#endif
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) {
        fprintf(stderr, "[gpismap_amd] K4 launch failed: %s (class %d, %d waves, %d tiles, LDS %zu B, cb %d, slots %d, table %d)\n",
                hipGetErrorString(le), wclass, W, ntiles, lds, cb, nslot, use_table);
        return GPIS_ERR_HIP;
    }
    return GPIS_OK;
}

int ongpis_eval_class(int nbx) { return ongpis_class_of_nbx(nbx); }


// Can K4 hold a cluster of N points / leading dimension ld?  It stages the row table and the points in LDS beside a
// two-slot ring of at least one column block each (the exp table is optional).  Asked at TRAINING time: a cluster that
// could be factorised but never evaluated is refused there (GPIS_ERR_LIMIT, the previous model is kept).
bool ongpis_eval_fits(int N, int ld) {
    const int W = kClassW[ongpis_class_of_nbx(ld / 32)];
    const size_t blk = kQS * sizeof(float) * kTileFloats;
    return eval_lds_fixed(W, N, ld, 0) + (K4_RING ? K4_RING_SLOTS : 2) * blk <= (size_t)158 * 1024;
}

}  // namespace gpis
