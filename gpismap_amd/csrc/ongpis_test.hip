// K4: batched OnGPIS prediction -- mean (value + gradient) and the four variances for tiles of 8 queries against one cluster model.
//
// Replaces the reference per-point chain
//   GPisMap3::test_kernel        cpp/src/GPisMap3.cpp:794-902  (2-D: GPisMap.cpp:665-763)
//     -> OnGPIS::testSinglePoint cpp/src/OnGPIS.cpp:177-216    (2-D: test2Dpoint :218-263)
//        -> matern32_sparse_deriv1_3D (cross)  cpp/src/covFnc.cpp:258-314 (2-D: :404-450)
//        -> k*^T alpha ; L^-1 k* ; column sums of squares
//
// The (1+d) cross-covariance columns of 8 queries form a K x 32 block B.  The reference solves L V = B by substitution (K dependent
// steps); here the factor's explicit inverse X = L^-1 comes out of training (K3b), so V = X B (32 x 32 tiles,
// v_mfma_f32_32x32x2_f32) has NO dependency between its block rows: every wavefront owns a few block rows (accumulators in
// registers), streams their X tiles from L2 once and reads the B tiles all wavefronts share from LDS.  alpha rides as row K of X.
//   * B is generated in chunks of CB column blocks into a THREE-slot LDS ring handed over through LDS counters: a wavefront multiplies
//     chunk i once its tiles are counted in, then generates its tile of chunk i+2 once every wavefront is through with the slot's
//     previous chunk.  No workgroup barrier in the loop.  (Two and four slots: measured slower, NOTEBOOK R6.2.)
//   * block rows are dealt from the largest down, snake-wise; clusters with more than 4 W block rows run several row groups (B
//     regenerated for the later, cheaper groups): no upper limit on K.
//   * every vector instruction of the generation stalls the SIMD's matrix pipe (profiles/r06_pivot_step.txt): the exponential is
//     table-driven (exp_tab.h), sqrt and the three divisions of a gradient row are the range-restricted ones (tile_solve.h).
//   * what lost and left this file (git history, NOTEBOOK R2-R6): the barrier kernel, exp tables in LDS and in global memory,
//     generating-only wavefronts, two X-tile buffers, row-major B tiles, two query sets per workgroup.
// Per element V[r][j] is ONE fmaf chain from zero over ascending k -- the order of the oracle's tiled mode.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "ongpis.h"
#include "tile_solve.h"
#include "exp_tab.h"

namespace gpis {

// Pointers read out of a ClusterModel live in global memory; say so (else: flat_load, which counts on both wait counters).
typedef const float __attribute__((address_space(1))) * gfptr;
typedef const int __attribute__((address_space(1))) * giptr;

// Instrumented builds only (make EXTRA=-DGPIS_INSTRUMENT): per-workgroup cycle stamps, ongpis_test_instr.inc.
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

// ---- shape of the kernel (the measured best of rounds 2-5) ----
constexpr int kW3 = 8;              // wavefronts per workgroup of the widest classes
constexpr int kNBW = 4;             // block rows per wavefront and row group (4 accumulator tiles = 64 VGPRs)
constexpr int kMinW = 4;            // wavefronts per SIMD the register budget is cut for (128 VGPRs)
constexpr int kSlots = 3;           // chunks in the LDS ring
constexpr int kTileStride = 36;     // floats per column of a B tile in LDS: conflict-free writes AND operand reads
constexpr int kTileFloats = 32 * kTileStride;   // 1152
static_assert(ONGPIS_TILE_Q == 8, "one set of 8 queries per workgroup (two sets at 256 VGPRs measured slower: NOTEBOOK R2, R5.8)");

// Hand-overs of the ring through LDS counters (cumulative, never reset), read through readfirstlane (a per-lane loop condition makes
// everything behind the loop divergent to the compiler).  Bounded: a protocol error must not hang the queue -- the tile's results
// are then NaN and the launch's error word is raised (MapQuery::run / eval_jobs return GPIS_ERR_STATE).
typedef volatile int __attribute__((address_space(3))) * lds_cnt_t;
__device__ __forceinline__ void k4_ring_wait(lds_cnt_t p, int need, lds_cnt_t expired, int limit) {
    int spins = 0;
    for (; spins < limit; ++spins) {
        if (__builtin_amdgcn_readfirstlane(*p) >= need) break;
        if ((spins & 1023) == 1023 && __builtin_amdgcn_readfirstlane(*expired) != 0) break;   // (a partner gave up already: one bounded wait per workgroup, not one per chunk)
        __builtin_amdgcn_s_sleep(1);
    }
    if (spins == limit) *expired = 1;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void k4_ring_signal(lds_cnt_t p, int lane) {
    __builtin_amdgcn_s_waitcnt(0xc07f);                       // this wavefront's LDS traffic on the slot is complete
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add((int __attribute__((address_space(3)))*)p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// W wavefronts per workgroup, every one generating B tiles and multiplying; kNBW block rows per wavefront and row group
template <int W>
__global__ __launch_bounds__(64 * W, kMinW) void ongpis_eval_kernel(EvalArgs A) {
    constexpr int NBW = kNBW;
    constexpr int RG = NBW * W;   // block rows per row group
    constexpr int NSLOT = kSlots;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // Workgroup ids go round the eight XCDs (id b on XCD b % 8) and the tiles of ONE cluster are consecutive in the list: taken as they
    // come, eight consecutive tiles land on eight L2s and each fetches the cluster's X from HBM for itself.  Inside every block of 64
    // ids the (id / 8, id % 8) grid is transposed: XCD x takes the tiles 8x .. 8x + 7, so eight consecutive tiles share one L2 (round 5:
    // stress predict +5 %; contiguous ranges per XCD instead: bench 0.75 -> 0.63).  Which workgroup evaluates a tile enters no result.
    int tile = blockIdx.x;
    {
        const int n64 = (int)gridDim.x & ~63;
        if (tile < n64) tile = (tile & ~63) | ((tile & 7) << 3) | ((tile >> 3) & 7);
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps the row ownership logic in SGPRs
    const int h = lane >> 5, l31 = lane & 31;
    const ClusterModel* __restrict__ mp = A.models + A.tile_model[tile];   // only the fields needed are read (scalar loads)
    const int N = mp->N, K = mp->K, ld = mp->ld, nb = mp->nb, dim = mp->dim;
    const int nbx = ld >> 5;      // block rows of the stored inverse: ceil((K+1)/32), row K of it = alpha (the mean)
    // K a multiple of 32 puts alpha ALONE into the last block row: as a block row of V it would multiply 31 zero rows in every
    // column block (8 of the 44 tile products of a query tile at K = 256, the stress configuration).  Then only the nb = K / 32
    // block rows of V go to the matrix cores and the mean is one fmaf chain per column on the vector ALU of the last wavefront,
    // fed from the same ring slots -- the same chain (ascending k from zero) the matrix instruction would run.
    const bool mean_alone = (K & 31) == 0;
    const int nbv = mean_alone ? nbx - 1 : nbx;       // block rows of V dealt to the wavefronts (= nb, the column blocks of B)
    const int joff = A.tile_off[tile], jcnt = A.tile_cnt[tile];   // 1..8 queries
    const int CB = A.cb;
    // test hook (gpis_ongpis_set_debug, inject bit 4): in the launch's first workgroup wavefront 0 withholds its first ring signal
    const bool withhold = (A.debug & 16) != 0 && blockIdx.x == 0;
    const int wait_limit = (A.debug & 16) ? (1 << 14) : (1 << 22);
    K4_TRACE_DECL(A)
    K4_STAMP();
    // LDS carve (all dynamic, 16-byte aligned pieces)
    constexpr int NC = 32;        // result columns of the workgroup
    float* red = reinterpret_cast<float*>(smem);                       // [W][NC] sums of squares, then [NC] means
    float4* s_xq = reinterpret_cast<float4*>(red + W * NC + NC);       // [16] the tile's query points
    int* s_ri = reinterpret_cast<int*>(s_xq + 16) + 32;                // [ld] row -> point | component
    lds_cnt_t ring_cnt = (lds_cnt_t)(reinterpret_cast<int*>(s_xq + 16));    // [0..2] tiles generated per slot, [8..10] wavefronts done per slot (cumulative), [15] a wait expired
    if (tid < 16) ring_cnt[tid] = 0;
    float4* s_x4 = reinterpret_cast<float4*>(s_ri + ld);               // [N]   (ld is a multiple of 32 -> 16-B aligned)
    f64x2* s_exp = reinterpret_cast<f64x2*>(s_x4 + N);                 // [64] 2^(j/64) as (hi, lo): exp_tab.h
    float* Bbuf = reinterpret_cast<float*>(s_exp + 64);                // [NSLOT][CB][32*36]
    const float scale = mp->scale;
    const float a = (float)(sqrt(3.0) / (double)scale);
    {   // stage 0: per-cluster vectors and the queries into LDS with coalesced loads
        giptr g_ri = (giptr)mp->rowinfo;
        gfptr g_x4 = (gfptr)mp->x4;
        for (int i = tid; i < ld; i += 64 * W) s_ri[i] = g_ri[i];
        for (int i = tid; i < 4 * N; i += 64 * W) reinterpret_cast<float*>(s_x4)[i] = g_x4[i];
        if (tid < 16) s_xq[tid] = (tid < jcnt) ? A.xq[A.job_q[joff + tid]] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < 64) s_exp[tid] = *reinterpret_cast<const f64x2*>(kExp64Tab[tid]);
        __syncthreads();
    }
    K4_STAMP();
    const float4* x4 = s_x4;
    // X tiles (A-operand order, 4 x 16-byte loads per tile) through a buffer resource: one VGPR byte offset per
    // lane + scalar tile offsets
    const int ntl = nbx * (nbx + 1) / 2;
    const __amdgpu_buffer_rsrc_t Xrs = __builtin_amdgcn_make_buffer_rsrc((void*)mp->Xt, 0, (unsigned)ntl * 4096u, 0x00020000);
    const int Tvoff = lane * 16;
    K4_STAMP();
    // ---- generation of one B tile (column block c) into an LDS tile: lane (r, qh) produces the 16 entries of tile row r for the
    // queries 4qh..4qh+3.  KIND = row type of the whole tile when it is uniform (0: value rows, 1..3: d/dx_c rows -- rows are ordered
    // by type, so almost every tile is uniform and the type selects of covFnc.cpp:292-308 fold away), -1: mixed tile.
    // Stored k-contiguous per column: element (row, n = 16 qh + 4 j + comp) -> tbuf[n * 36 + (row & 1) * 16 + (row >> 1)], i.e.
    // Bt[n][h][kk] = B[2 kk + h][n]: a lane's 16 operand values are FOUR 16-byte reads, a wavefront's stores hit 32 different banks.
    const int ngr = (dim > 0) ? (K - N) / dim : 0;   // rows per derivative component
    auto row_type = [&](int r) { return r < N ? 0 : 1 + (r - N) / (ngr > 0 ? ngr : 1); };
    auto emit_rows = [&](auto kind_tag, int c, float* tbuf) {
        constexpr int KIND = decltype(kind_tag)::value;
        const int rr_ = lane & 31, qh = lane >> 5;
        const int row = c * 32 + rr_;
        int cr = 0;
        float4 xp = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < K) {
            const int info = s_ri[row];
            cr = (KIND >= 0) ? KIND : ((info >> 28) & 0xF);
            xp = x4[info & 0x0FFFFFFF];
        }
#pragma unroll 2      // (two independent chains hide the double-precision latencies; 4 measured slower)
        for (int j = 0; j < 4; ++j) {
            const int q = 4 * qh + j;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < K && q < jcnt) {
                const float4 xq = s_xq[q];
                float d[3] = {xp.x - xq.x, xp.y - xq.y, xp.z - xq.z};
                float rr = sqrt_ranged((dim == 3) ? (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2] : d[0] * d[0] + d[1] * d[1]);   // (= sqrtf: squared distances are 0 or far above 2^-96)
                const double e = exp_neg_tab(-a * rr, s_exp);
                float v0, v1, v2, v3;
                if (cr == 0) {
                    v0 = d_kf(rr, a, e); v1 = d_kf1(d[0], a, e); v2 = d_kf1(d[1], a, e); v3 = d_kf1(d[2], a, e);
                } else {
                    const float dr = cr == 1 ? d[0] : (cr == 2 ? d[1] : d[2]);
                    v0 = -d_kf1(dr, a, e);
                    // mixed second derivatives: lower component first (covFnc.cpp:300-308).  The three divisions by r share the divisor's
                    // refined reciprocal (tile_solve.h div_ranged: the bits of `/`; r = 0 -- a query ON a training point -- gives the
                    // reference's NaN)
                    const float ri = rcp_refined(rr);
                    auto kf2 = [&](float dx1, float dx2, float delta) { return (float)((double)(a * a * (delta - div_ranged_anyzero(a * dx1 * dx2, rr, ri))) * e); };   // (delta - (+-0) is delta either way)
                    v1 = (cr == 1) ? kf2(d[0], d[0], 1.0f) : kf2(d[0], dr, 0.0f);
                    v2 = (cr == 2) ? kf2(d[1], d[1], 1.0f) : (cr == 1 ? kf2(d[0], d[1], 0.0f) : kf2(d[1], d[2], 0.0f));
                    v3 = (cr == 3) ? kf2(d[2], d[2], 1.0f) : kf2(dr, d[2], 0.0f);
                }
                if (dim == 2) v3 = 0.f;
                o = make_float4(v0, v1, v2, v3);
            }
            float* tcol = tbuf + (16 * qh + 4 * j) * kTileStride + (rr_ & 1) * 16 + (rr_ >> 1);
            tcol[0] = o.x; tcol[kTileStride] = o.y; tcol[2 * kTileStride] = o.z; tcol[3 * kTileStride] = o.w;
        }
    };
    auto gen_tile = [&](int c, float* tbuf) {
        const int r0 = c * 32, r1 = min(K, r0 + 32) - 1;
        const int k0 = row_type(r0), k1 = row_type(r1);
        if (k0 != k1) emit_rows(std::integral_constant<int, -1>(), c, tbuf);
        else if (k0 == 0) emit_rows(std::integral_constant<int, 0>(), c, tbuf);
        else if (k0 == 1) emit_rows(std::integral_constant<int, 1>(), c, tbuf);
        else if (k0 == 2) emit_rows(std::integral_constant<int, 2>(), c, tbuf);
        else emit_rows(std::integral_constant<int, 3>(), c, tbuf);
    };
    // ---- B chunks.  The chunks of all row groups form one sequence gci = 0, 1, ...; chunk gci lives in ring slot gci % 3.
    // Tile t of chunk p is made by wavefront (t + p CB) mod W: the duty goes round, every wavefront generates the same number of
    // tiles over a few chunks.  A slot is free once every wavefront has multiplied the chunk it held before.
    const int ngroups = (nbv + RG - 1) / RG;
    auto group_cmax = [&](int g) { return min(nbv - 1 - g * RG, nb - 1); };   // last column block a row of group g multiplies with
    int pg = 0, pci = 0, pgci = 0;   // producer cursor: group, chunk in group, chunk in sequence
    // expected cumulative tile count per slot (what the slot's counter shows once every chunk produced into it so far is complete)
    int exp_gen0 = 0, exp_gen1 = 0, exp_gen2 = 0;
    bool withheld = false;
    auto produce_next = [&]() {
        if (pg >= ngroups) return;
        const int cmax = group_cmax(pg);
        const int pslot = pgci % NSLOT;
        float* slot = Bbuf + (size_t)pslot * CB * kTileFloats;
        const int c0 = pci * CB;
        const int nt = min(CB, cmax - c0 + 1);
        const int my_t = (((wave - pgci * CB) % W) + W) % W;
        if (my_t < nt) {
            k4_ring_wait(ring_cnt + 8 + pslot, W * (pgci / NSLOT), ring_cnt + 15, wait_limit);
            for (int t = my_t; t < nt; t += W) {      // (chunks wider than the workgroup has wavefronts: several tiles per wavefront)
                gen_tile(c0 + t, slot + (size_t)t * kTileFloats);
                if (withhold && wave == 0 && !withheld) { withheld = true; continue; }     // (test hook: this tile is never counted in)
                k4_ring_signal(ring_cnt + pslot, lane);
            }
        }
        // (unconditional adds: an if / else chain over the three becomes a SELECT OF POINTERS to captured variables, and with it every
        // capture of the kernel's lambdas stays in scratch memory -- 240 allocas survived)
        exp_gen0 += (pslot == 0) ? nt : 0; exp_gen1 += (pslot == 1) ? nt : 0; exp_gen2 += (pslot == 2) ? nt : 0;
        ++pgci;
        if (++pci > cmax / CB) { pci = 0; ++pg; }
    };
    float ss = 0.f;               // partial sum of squares of V over this lane's rows (order O3, oracle reduce_ss)
    float mean_val = 0.f;         // row K of V (the lane that owns it)
    for (int i = 0; i < NSLOT - 1; ++i) produce_next();     // two chunks generated ahead of the multiplication
    K4_STAMP();
    int gci = 0;
    for (int g = 0; g < ngroups; ++g) {
        // this wave's block rows in group g: slot t holds the (g RG + t W + q)-th largest row, q snaking with t.  (The row table also
        // fixes the order of the variance sums -- oracle reduce_ss.  Pairing complementary rows on a SIMD's wavefronts: slower, R5.2.)
        int brow[NBW];
#pragma unroll
        for (int t = 0; t < NBW; ++t) {
            const int i = g * RG + t * W + ((t & 1) ? (W - 1 - wave) : wave);
            brow[t] = nbv - 1 - i;      // < 0: no row
        }
        const int cmax = group_cmax(g);
        const int nch = cmax / CB + 1;
        f32x16 acc[NBW];
#pragma unroll
        for (int t = 0; t < NBW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

        K4_LAP_START();
        for (int ci = 0; ci < nch; ++ci, ++gci) {
            K4_LAP(0);
            {   // chunk gci complete in its slot?  (its tiles were generated one or two chunks ago)
                const int sl = gci % NSLOT;
                k4_ring_wait(ring_cnt + sl, sl == 0 ? exp_gen0 : (sl == 1 ? exp_gen1 : exp_gen2), ring_cnt + 15, wait_limit);
                K4_LAP(2);
            }
            {
                const float* buf = Bbuf + (size_t)(gci % NSLOT) * CB * kTileFloats;
                const int c0 = ci * CB;
#pragma unroll
                for (int t = 0; t < NBW; ++t) {
                    const int b = brow[t];
                    if (b >= c0) {
                        const int cend = min(min(c0 + CB - 1, b), cmax);
                        // ONE X-tile buffer used as a ring of four 16-byte pieces: once the four matrix instructions that read piece g
                        // are issued, piece g of the NEXT tile of the row is requested into the same registers (768 cycles ahead, about
                        // one L2 round trip) -- no second buffer.  The last tile of the row re-requests itself (clamped: branch-free).
                        float av1[16];
                        {
                            const int sbase = (b * (b + 1) / 2 + c0) * 4096;
#pragma unroll
                            for (int g4 = 0; g4 < 4; ++g4) {
                                auto q = __builtin_amdgcn_raw_buffer_load_b128(Xrs, Tvoff, sbase + g4 * 1024, 0);
                                av1[4 * g4 + 0] = __uint_as_float(q[0]); av1[4 * g4 + 1] = __uint_as_float(q[1]);
                                av1[4 * g4 + 2] = __uint_as_float(q[2]); av1[4 * g4 + 3] = __uint_as_float(q[3]);
                                // (the four requests in THIS order: the loop below waits for piece g with three requests still in flight; issued
                                // in another order -- the scheduler once reversed them -- every wait of the loop becomes a wait for all four)
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
#pragma unroll 1
                        for (int c = c0; c <= cend; ++c) {
                            // B operand kk of lane (h, n) = B_c[2 kk + h][n]
                            const float4* Bq = reinterpret_cast<const float4*>(buf + (size_t)(c - c0) * kTileFloats + l31 * kTileStride + h * 16);
                            const int cn = min(c + 1, cend);
                            const int nbase = (b * (b + 1) / 2 + cn) * 4096;
#pragma unroll
                            for (int g4 = 0; g4 < 4; ++g4) {
                                const float4 bq = Bq[g4];
                                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[4 * g4 + 0], bq.x, acc[t], 0, 0, 0);
                                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[4 * g4 + 1], bq.y, acc[t], 0, 0, 0);
                                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[4 * g4 + 2], bq.z, acc[t], 0, 0, 0);
                                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[4 * g4 + 3], bq.w, acc[t], 0, 0, 0);
                                auto q = __builtin_amdgcn_raw_buffer_load_b128(Xrs, Tvoff, nbase + g4 * 1024, 0);
                                av1[4 * g4 + 0] = __uint_as_float(q[0]); av1[4 * g4 + 1] = __uint_as_float(q[1]);
                                av1[4 * g4 + 2] = __uint_as_float(q[2]); av1[4 * g4 + 3] = __uint_as_float(q[3]);
                                __builtin_amdgcn_sched_barrier(0);     // keep the request where it is: right behind its four instructions
                            }
                        }
                    }
                }
            }
            if (mean_alone && g == 0 && wave == W - 1) {
                // the mean of a cluster whose alpha sits alone in its block row: mean[n] = fmaf(alpha[k], B[k][n], mean[n]), k ascending
                // through this chunk's column blocks (group 0 meets every one once, in order).  alpha[32 c + k] = element (row 0, column k)
                // of tile (nbx - 1, c) in A-operand order: float (k >> 3) * 256 + (k & 1) * 128 + ((k >> 1) & 3).  ONE gather load per pair
                // of column blocks (lanes 0..31 block c, 32..63 block c + 1), operands by v_readlane (scalar loads: four dependent round
                // trips per block made this wavefront the workgroup's tail).  Lane n reads column n of the B tile, k-contiguous.
                const float* buf = Bbuf + (size_t)(gci % NSLOT) * CB * kTileFloats;
                const int c0 = ci * CB, cl = min(c0 + CB - 1, cmax);
                const int avoff = (lane >> 5) * 4096 + ((l31 >> 3) * 1024 + (l31 & 1) * 512 + ((l31 >> 1) & 3) * 4);
#pragma unroll 1
                for (int cq = c0; cq <= cl; cq += 4) {      // four column blocks at a time (two gather registers: four cost a spill)
                    float al[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        al[i] = 0.f;
                        if (cq + 2 * i <= cl)       // (an odd last pair reads the diagonal tile behind the row's last column block: in range, unused)
                            al[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(Xrs, avoff, ((nbx - 1) * nbx / 2 + cq + 2 * i) * 4096, 0));
                    }
#pragma unroll
                    for (int t4 = 0; t4 < 4; ++t4) {
                        if (cq + t4 <= cl) {
                            const float4* Bc = reinterpret_cast<const float4*>(buf + (size_t)(cq - c0 + t4) * kTileFloats + l31 * kTileStride);
#pragma unroll
                            for (int g4 = 0; g4 < 4; ++g4) {
                                const float4 b0 = Bc[g4], b1 = Bc[4 + g4];                // B[32 c + 8 g4 + 2 j][n] , [.. + 2 j + 1][n]
                                const float bb[8] = {b0.x, b1.x, b0.y, b1.y, b0.z, b1.z, b0.w, b1.w};
#pragma unroll
                                for (int j = 0; j < 8; ++j) {
                                    const float ak = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(al[t4 >> 1]), (t4 & 1) * 32 + 8 * g4 + j));
                                    mean_val = fmaf(ak, bb[j], mean_val);
                                }
                            }
                        }
                    }
                }
            }
            K4_LAP(1);
            k4_ring_signal(ring_cnt + 8 + gci % NSLOT, lane);       // this wavefront is through with chunk gci
            produce_next();        // every wave multiplies first (a wave that generates first waits for the slowest wave of the chunk before: 0.4 % slower)
            K4_LAP(0);
            K4_LAP_COUNT(4);
        }
        // sums of squares of the finished rows; row K (block nbx-1: group 0, slot 0, wave 0) is the mean unless alpha sits alone
#pragma unroll
        for (int t = 0; t < NBW; ++t) {
            if (brow[t] < 0) continue;
            const bool has_mean = (g == 0 && t == 0 && wave == 0 && !mean_alone);
            const int kr = K & 31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[t][r];
                if (has_mean && rowmap_t(r, h) == kr) mean_val = v;
                else ss = fmaf(v, v, ss);
            }
        }
        K4_LAP(3);
    }
    K4_LAP_FLUSH();

    K4_STAMP();
    // ---- reduce partials (lane halves, then waves in fixed order) ----
    ss = ss + __shfl_xor(ss, 32);
    if (h == 0) red[wave * NC + l31] = ss;
    if (mean_alone ? (wave == W - 1 && h == 0) : (wave == 0 && h == ((K >> 2) & 1))) red[W * NC + l31] = mean_val;   // the half that owns row K & 31
    __syncthreads();
    {
        const int col = tid & 63;                       // wave 0: one lane per (query, component) column
        const int qi = col >> 2, cq = col & 3;
        const bool ring_failed = ring_cnt[15] != 0;     // a bounded ring wait ran out (protocol error): poison, loudly
        if (ring_failed && tid == 0 && A.err) __hip_atomic_fetch_or(A.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
            o[cq] = ring_failed ? __uint_as_float(0x7fc00000u) : ms;
            o[4 + cq] = ring_failed ? __uint_as_float(0x7fc00000u) : var;
        }
    }
    K4_STAMP();
}

// Size classes by nbx = ceil((K+1)/32): W wavefronts per workgroup (ongpis.h, ongpis_class_of_nbx).
static const int kClassW[ONGPIS_NCLASS] = {1, 2, 4, 4, kW3, kW3, kW3};
constexpr int kWavesPerCU = 4 * kMinW;   // resident wavefronts per CU the register budget of the kernels admits

static size_t eval_lds_fixed(int W, int maxN, int maxLd) {
    return sizeof(float) * (W * 32 + 32) + 16 * sizeof(float4) + 32 * sizeof(int) + sizeof(int) * (size_t)maxLd + 16 * (size_t)maxN + 64 * 16;
}

int ongpis_eval_launch(int wclass, int ntiles, int maxN, int maxLd, const EvalArgs& args_in, hipStream_t s) {
    if (ntiles <= 0) return GPIS_OK;
    if (wclass < 0 || wclass >= ONGPIS_NCLASS) return GPIS_ERR_ARG;
    EvalArgs args = args_in;
#ifdef GPIS_EXPERIMENTS   // the archived resident-X kernel for clusters of at most ONGPIS_SMALL_NBX block rows (tools/experiments/)
    if (getenv("GPIS_SMALL_KERNEL") && atoi(getenv("GPIS_SMALL_KERNEL")) && wclass <= 2 && maxLd / 32 <= ONGPIS_SMALL_NBX && ongpis_eval_small_lds(maxN, maxLd) <= (size_t)160 * 1024) return ongpis_eval_small_launch(ntiles, maxN, maxLd, args_in, s);
#endif
    const int W = kClassW[wclass];
    // LDS budget: the register file admits kWavesPerCU / W workgroups per CU; each gets an equal share of the 160 KB and spends what
    // the per-cluster tables leave on the ring: three slots of cb column blocks (cb <= 8; wide chunks beat more slots, NOTEBOOK R6.2).
    const size_t hard = 158 * 1024;
    const size_t share = std::min(hard, hard * W / kWavesPerCU);
    const size_t blk = sizeof(float) * kTileFloats;    // one column block
    const size_t fixed = eval_lds_fixed(W, maxN, maxLd);
    if (fixed + kSlots * blk > hard) return GPIS_ERR_LIMIT;
    // (the small classes keep their occupancy: wider chunks at the price of fewer workgroups per CU measured 35 -> 30 % at K = 204)
    const size_t budget = std::min(hard, std::max(share, fixed + kSlots * blk));
    const int nblk = (int)((budget - fixed) / blk);           // column blocks the ring can hold
    int cb = std::max(1, std::min(nblk / kSlots, 8));
    cb = std::min(cb, std::max(1, maxLd / 32));
    args.cb = cb;
    const size_t lds = fixed + (size_t)kSlots * cb * blk;
    typedef void (*kern_t)(EvalArgs);
    static const kern_t kern[4] = {ongpis_eval_kernel<1>, ongpis_eval_kernel<2>, ongpis_eval_kernel<4>, ongpis_eval_kernel<kW3>};
    const int kidx = wclass < 3 ? wclass : (wclass == 3 ? 2 : 3);
    if (ensure_dynamic_lds((const void*)kern[kidx], 160 * 1024) != GPIS_OK) return GPIS_ERR_HIP;
#ifdef GPIS_INSTRUMENT
    k4_trace_arm(args, s);
#endif
    hipLaunchKernelGGL(kern[kidx], dim3(ntiles), dim3(64 * W), lds, s, args);
#ifdef GPIS_INSTRUMENT
    k4_trace_dump(s, W, 0, ntiles, maxN, maxLd, cb);
#endif
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) { fprintf(stderr, "[gpismap_amd] K4 launch failed: %s (class %d, %d waves, %d tiles, LDS %zu B, cb %d)\n", hipGetErrorString(le), wclass, W, ntiles, lds, cb); return GPIS_ERR_HIP; }
    return GPIS_OK;
}

int ongpis_eval_class(int nbx) { return ongpis_class_of_nbx(nbx); }

// Can K4 hold a cluster (row table + points in LDS beside a three-slot ring of one column block each)?  Asked at TRAINING time: a
// cluster that could be factorised but never evaluated is refused there (GPIS_ERR_LIMIT, the previous model is kept).
bool ongpis_eval_fits(int N, int ld) {
    const int W = kClassW[ongpis_class_of_nbx(ld / 32)];
    return eval_lds_fixed(W, N, ld) + kSlots * sizeof(float) * kTileFloats <= (size_t)158 * 1024;
}

}  // namespace gpis
