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
//   * step c: the owner of block c multiplies its tile by the inverted diagonal block
//     (V_c = inv(L_cc) U_c, 16 matrix instructions; the inverse comes from K3) and publishes V_c to LDS;
//   * every wave then applies  B_b -= L_bc V_c  to its tiles with
//     v_mfma_f32_32x32x2_f32, streaming L_bc from L2/HBM exactly once.
// Each L element is read once per workgroup and used for 32 columns.  The
// per-element operation order is the ascending-k fmaf chain of dev_common.h.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "ongpis.h"
#include "tile_solve.h"

namespace gpis {

// Timing ablations and the per-wave cycle trace exist only in instrumented builds
// (make EXTRA=-DGPIS_K4_INSTRUMENT; tools/k4_bench.py): the shipped library has no debug branches in the
// kernel, reads no environment variables at launch and writes no files.
#ifdef GPIS_K4_INSTRUMENT
#define K4_DBG(bits) (A.dbg & (bits))
#else
#define K4_DBG(bits) 0
#endif

// Pointers read out of a ClusterModel live in global memory; say so, otherwise the compiler must
// emit flat_load (LDS-or-global at run time), which counts on both wait counters.
typedef const float __attribute__((address_space(1))) * gfptr;
typedef const int __attribute__((address_space(1))) * giptr;
typedef const float4 __attribute__((address_space(1))) * gf4ptr;
typedef const void __attribute__((address_space(1))) * gvptr;
typedef float __attribute__((address_space(3))) * lds_fptr;
typedef void __attribute__((address_space(3))) * lds_vptr;
typedef volatile int __attribute__((address_space(3))) * lds_flag_ptr;   // explicit LDS: volatile generic pointers become flat loads


// Size classes (block rows nb = ceil(K/32)): W waves x NBW tiles per wave.
//   nb <= 4: 1x4   <= 8: 2x4   <= 16: 4x4   <= 32: 8x4 (128 VGPRs)   <= 64: 16x4 (128 VGPRs)   <= 96: 8x12
// VREG: V_c is read into registers once per step (256-VGPR classes) or streamed from LDS per MFMA pair
// (128-VGPR classes, four waves per SIMD).  TR: cycle-trace build of the kernel (GPIS_K4_TRACE).
template <int W, int NBW, int MINW, bool VREG, bool TR>
__global__ __launch_bounds__(64 * W, MINW) void ongpis_eval_kernel(EvalArgs A) {
    constexpr int RING = 4;   // published V blocks / diagonal blocks kept in LDS
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tile = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps the ownership logic in SGPRs
    const int h = lane >> 5, l31 = lane & 31;
    const ClusterModel m = A.models[A.tile_model[tile]];
    const int N = m.N, K = m.K, ld = m.ld, nb = m.nb, dim = m.dim;
    const int joff = A.tile_off[tile], jcnt = A.tile_cnt[tile];

    // LDS carve (all dynamic, 16-byte aligned pieces).  Region U is time-shared: stage 1/2 keep the
    // per-lane staging strips and the exp table there, stage 3 the ring of published V blocks.
    float* red = reinterpret_cast<float*>(smem);                  // [W][64][2]
    lds_flag_ptr flags = (lds_flag_ptr)(red + W * 128);           // [0] = pub, [1..W] = done[w]; 32 ints reserved
    float* s_alpha = red + W * 128 + 32;                          // [ld]
    int* s_ri = reinterpret_cast<int*>(s_alpha + ld);             // [ld]
    float4* s_x4 = reinterpret_cast<float4*>(s_ri + ld);          // [N]   (ld is a multiple of 32 -> 16-B aligned)
    float* U = reinterpret_cast<float*>(s_x4 + N);
    float* stage = U;                                             // stage 2: [W][32*36] per-wave padded tile
    double* etab = reinterpret_cast<double*>(U + W * 1152);       // stage 1/2: [N][8] exp table (optional)
    float* Vbuf = U;                                              // stage 3: [RING][32*32] published V blocks

    const float a = (float)(sqrt(3.0) / (double)m.scale);

    // this lane's column: query slot qi, component cq
    const int qi = l31 >> 2, cq = l31 & 3;
    const bool qact = (qi < jcnt) && (cq <= dim);

    // optional cycle trace of one workgroup (A.trace != nullptr): [wave][slot] timestamps
    unsigned long long* trc = (TR && A.trace && (int)blockIdx.x == A.trace_block) ? A.trace + wave * 512 : nullptr;
    int tri = 0;
#define TRACE() do { if constexpr (TR) { if (trc && lane == 0 && tri < 512) trc[tri++] = __builtin_readcyclecounter(); } } while (0)
    // owner-path events (traced build): [8 + wave][slot]
    int tro = 0;
#define TRACE_OWN() do { if constexpr (TR) { if (trc && lane == 0 && tro < 511) trc[8 * 512 + tro++] = __builtin_readcyclecounter(); } } while (0)
    TRACE();
    if (tid < 32) flags[tid] = -1;
    if (K4_DBG(512) && (blockIdx.x & 1)) {  // experiment: stagger the two workgroups sharing a CU
        for (int i = 0; i < (A.dbg >> 10); ++i) __builtin_amdgcn_s_sleep(127);
    }
    {   // stage 0: per-cluster vectors into LDS with coalesced loads (always fits for K <= 3072)
        gfptr g_alpha = (gfptr)m.alpha;
        giptr g_ri = (giptr)m.rowinfo;
        gfptr g_x4 = (gfptr)m.x4;
        for (int i = tid; i < ld; i += 64 * W) { s_alpha[i] = g_alpha[i]; s_ri[i] = g_ri[i]; }
        for (int i = tid; i < 4 * N; i += 64 * W) reinterpret_cast<float*>(s_x4)[i] = g_x4[i];
        __syncthreads();
    }
    const float4* x4 = s_x4;
    // off-diagonal tiles come from the re-tiled copy Lt (-L, MFMA A-operand order) through a buffer resource:
    // one VGPR byte offset per lane + scalar offsets, 4 x 16-byte loads per tile
    const int ntl = nb * (nb + 1) / 2;
    const __amdgpu_buffer_rsrc_t Trs = __builtin_amdgcn_make_buffer_rsrc((void*)m.Lt, 0, (unsigned)ntl * 4096u, 0x00020000);
    const int Tvoff = lane * 16;

    TRACE();
    // ---- stage 1: exp table, one entry per (training point, query slot) ----
    if (A.use_table && !K4_DBG(8)) {
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
    }
    __syncthreads();

    TRACE();
    // ---- stage 2: B tiles into accumulators + partial means ----
    // Cooperative generation: lane (r, qh) produces the 16 entries of tile row r for the queries
    // 4qh..4qh+3 (distance, exp-table lookup and coefficient set-up shared by the 4 components),
    // writes them to a padded per-wave LDS tile, which is then read back in MFMA C/D layout.
    f32x16 acc[NBW];
    float mp = 0.f;  // partial k*^T alpha over this lane's rows (order O3)
    float* tbuf = stage + wave * (32 * 36);   // [32 rows][36] (stride 36 floats: conflict-free 16-byte writes)
    {
        const int rr_ = lane & 31, qh = lane >> 5;
        float4 xqs[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = 4 * qh + j;
            xqs[j] = (q < jcnt) ? A.xq[A.job_q[joff + q]] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // The 16 entries of one (row, 4 queries) strip.  KIND = row type of the whole tile when it is uniform
        // (0: value rows, 1..3: d/dx_c rows -- the rows are ordered by type, so almost every tile is uniform and
        // the compiler folds the type selects of covFnc.cpp:292-308 away), -1: mixed tile, per-row type.
        auto emit_rows = [&](auto kind_tag, int b) {
            constexpr int KIND = decltype(kind_tag)::value;
            const int row = b * 32 + rr_;
            float4 out[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) out[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < K) {
                const int info = s_ri[row];
                const int p = info & 0x0FFFFFFF;
                const int cr = (KIND >= 0) ? KIND : ((info >> 28) & 0xF);
                const float4 xp = x4[p];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int q = 4 * qh + j;
                    if (q < jcnt) {
                        float d[3] = {xp.x - xqs[j].x, xp.y - xqs[j].y, xp.z - xqs[j].z};
                        float rr = (dim == 3) ? sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) : sqrtf(d[0] * d[0] + d[1] * d[1]);
                        double e = A.use_table ? etab[p * 8 + q] : exp((double)(-a * rr));
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
                        out[j] = make_float4(v0, v1, v2, v3);
                    }
                }
            }
            float4* trow = reinterpret_cast<float4*>(tbuf + rr_ * 36 + 16 * qh);
#pragma unroll
            for (int j = 0; j < 4; ++j) trow[j] = out[j];
        };
        // padded LDS strip -> accumulator tile t (static register index) + partial means
        auto take_tile = [&](f32x16& tl, int b) {
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes have landed
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rw = rowmap_t(r, h);
                const float v = tbuf[rw * 36 + l31];
                tl[r] = v;
                mp = fmaf(v, s_alpha[b * 32 + rw], mp);   // rows >= K: B = 0 and alpha = 0 (K3 pads)
            }
            __builtin_amdgcn_wave_barrier();
        };
        const int ngr = (dim > 0) ? (K - N) / dim : 0;   // rows per derivative component
        auto row_type = [&](int r) { return r < N ? 0 : 1 + (r - N) / (ngr > 0 ? ngr : 1); };
#pragma unroll
        for (int t = 0; t < NBW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll 1
        for (int t = 0; t < NBW; ++t) {
            // The evaluation code exists once: every tile is generated into the LAST register tile after a rotation
            // by one place; after NBW trips each tile sits in its home position.
            {
                const f32x16 t0 = acc[0];
#pragma unroll
                for (int u = 0; u + 1 < NBW; ++u) acc[u] = acc[u + 1];
                acc[NBW - 1] = t0;
            }
            const int b = wave + W * t;
            if (b < nb && !K4_DBG(1)) {
                const int r0 = b * 32, r1 = min(K, r0 + 32) - 1;
                const int k0 = row_type(r0), k1 = row_type(r1);
                if (k0 != k1) emit_rows(std::integral_constant<int, -1>(), b);
                else if (k0 == 0) emit_rows(std::integral_constant<int, 0>(), b);
                else if (k0 == 1) emit_rows(std::integral_constant<int, 1>(), b);
                else if (k0 == 2) emit_rows(std::integral_constant<int, 2>(), b);
                else emit_rows(std::integral_constant<int, 3>(), b);
                take_tile(acc[NBW - 1], b);
            }
        }
        if (K4_DBG(1024)) {   // ablation (with dbg & 1): skip the generation but keep non-trivial operand data
#pragma unroll
            for (int t = 0; t < NBW; ++t)
                if (wave + W * t < nb) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        unsigned hsh = (unsigned)(lane * 2654435761u) ^ (unsigned)(((wave + W * t) * 16 + r) * 40503u + blockIdx.x * 97u);
                        hsh ^= hsh >> 13; hsh *= 0x5bd1e995u; hsh ^= hsh >> 15;
                        acc[t][r] = (float)(int)(hsh & 0xffff) * (1.0f / 65536.0f) - 0.5f;
                    }
                }
        }
    }

    TRACE();
    // ---- stage 3: blocked forward substitution ----
    __syncthreads();  // staging strips alias the rings
    // Dataflow synchronisation instead of a barrier per step: `pub` = last published block,
    // done[w] = last step whose updates wave w finished.  The owner of block c+1 updates that
    // tile first, turns it into V_{c+1} and publishes it while the other waves are still busy with
    // step c.  The owner chain (update -> V = inv(L_cc) U -> publish) is the critical path of the
    // workgroup; it is two dependent runs of 16 matrix instructions -- the accumulator tile itself is the
    // B operand of the second run (Lt's diagonal tiles are stored in the matching k order) -- and
    // executes at raised wave priority.
    float ss = 0.f;  // partial sum of squares of V over this lane's rows
    lds_flag_ptr pub = flags;
    lds_flag_ptr done = flags + 1;
    // u = U_c (C/D layout), ai = inv(L_cc) operands; returns with V_c published and its squares summed
    auto invert_publish = [&](const f32x16& u, const float (&ai)[16], int c) {
        f32x16 v;
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = 0.f;
        if (!K4_DBG(2)) {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) v = __builtin_amdgcn_mfma_f32_32x32x2f32(ai[kk], u[kk], v, 0, 0, 0);
        }
        TRACE_OWN();
        // ring slot free once every wave has finished step c - RING
        if (c >= RING) {   // one LDS round trip: lane w looks at done[w]
            while (__builtin_amdgcn_ballot_w64(lane < W && done[lane < W ? lane : 0] < c - RING)) __builtin_amdgcn_s_sleep(1);
        }
        // one lane pointer + compile-time row offsets; padding rows >= K hold exact zeros (ongpis_train.hip),
        // so the sum of squares needs no row predicate
        float* Vw = Vbuf + (c % RING) * 1024 + (4 * h * 32 + l31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            Vw[((r & 3) + 8 * (r >> 2)) * 32] = v[r];
            ss = fmaf(v[r], v[r], ss);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");   // LDS only: global loads in flight need not drain
        if (lane == 0) *pub = c;
        TRACE_OWN();   // published
    };
    // av = -L tile operands (Lt holds the negated factor), vb = V_c in MFMA B-operand order
    auto update_tile = [&](f32x16& a_, const float (&vb)[16], const float (&av)[16]) {
        if (K4_DBG(4)) return;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
            a_ = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], vb[kk], a_, 0, 0, 0);
    };
    // same, V_c streamed from its LDS ring slot one MFMA pair ahead
    auto update_tile_lds = [&](f32x16& a_, const float* Vl, const float (&av)[16]) {
        if (K4_DBG(4)) return;
        float p0 = Vl[0], p1 = Vl[64];
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            float n0 = 0.f, n1 = 0.f;
            if (kk + 2 < 16) { n0 = Vl[(kk + 2) * 64]; n1 = Vl[(kk + 3) * 64]; }
            a_ = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], p0, a_, 0, 0, 0);
            a_ = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk + 1], p1, a_, 0, 0, 0);
            p0 = n0; p1 = n1;
        }
    };
    auto load_a = [&](float (&av)[16], int b, int c) {
        const int sbase = (b * (b + 1) / 2 + c) * 4096;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            auto q = __builtin_amdgcn_raw_buffer_load_b128(Trs, Tvoff, sbase + g * 1024, 0);
            av[4 * g + 0] = __uint_as_float(q[0]); av[4 * g + 1] = __uint_as_float(q[1]);
            av[4 * g + 2] = __uint_as_float(q[2]); av[4 * g + 3] = __uint_as_float(q[3]);
        }
    };

    if (wave == 0) {  // block 0 has no dependency
        float ai[16];
        load_a(ai, 0, 0);
        invert_publish(acc[0], ai, 0);
    }
    // Steps c = tc*W + wc.  Block c+1 belongs to wave (wc+1) mod W; its tile index is tc (same
    // group) or tc+1 (wave 0 at the group boundary): static after unrolling tc, so the solve
    // works in place on the accumulator registers.
#pragma clang loop unroll(full)
    for (int tc = 0; tc < NBW; ++tc) {
#pragma unroll 1
        for (int wc = 0; wc < W; ++wc) {
            const int c = tc * W + wc;
            if (c >= nb) break;
            const bool has_next = (c + 1 < nb);
            const bool own_same = has_next && (wc + 1 < W) && (wave == wc + 1);
            const bool own_next = has_next && (wc + 1 == W) && (wave == 0) && (tc + 1 < NBW);
            const bool owner = own_same || own_next;
            auto active = [&](int t_) { const int b_ = wave + W * t_; return b_ > c && b_ < nb && b_ != c + 1; };
            float avp[2][16];   // A operands: [0] doubles as the buffer of the look-ahead tile
            float vb[VREG ? 16 : 1];
            // issue the loads this step needs before waiting for V_c
            if (owner) { load_a(avp[0], c + 1, c); load_a(avp[1], c + 1, c + 1); }   // [1]: inv(L_{c+1,c+1})
            else if (active(tc)) load_a(avp[tc & 1], wave + W * tc, c);
            TRACE();
            while (*pub < c) __builtin_amdgcn_s_sleep(1);
            TRACE();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            const float* Vb = Vbuf + (c % RING) * 1024 + (h * 32 + l31);   // B operand kk: Vb[64 kk]
            if (owner) {
                __builtin_amdgcn_s_setprio(3);
                auto own_tile = [&](f32x16& a_) {
                    if constexpr (TR) { TRACE_OWN(); __builtin_amdgcn_s_waitcnt(0x0f70); TRACE_OWN(); }   // vmcnt(0): A operands arrived
                    if constexpr (VREG) {
                        float vbo[16];
#pragma unroll
                        for (int kk = 0; kk < 16; ++kk) vbo[kk] = Vb[64 * kk];
                        update_tile(a_, vbo, avp[0]);
                    } else {
                        update_tile_lds(a_, Vb, avp[0]);
                    }
                    TRACE_OWN();
                    invert_publish(a_, avp[1], c + 1);
                };
                if (own_same) own_tile(acc[tc]);
                if (tc + 1 < NBW) {
                    if (own_next) own_tile(acc[tc + 1 < NBW ? tc + 1 : tc]);
                }
                __builtin_amdgcn_s_setprio(0);
                if (active(tc)) load_a(avp[tc & 1], wave + W * tc, c);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (VREG) {
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) vb[kk] = Vb[64 * kk];
            }
            TRACE();
            __builtin_amdgcn_sched_barrier(0);
            // remaining tiles of this wave (t >= tc), A operands prefetched one tile ahead
#pragma unroll
            for (int t = tc; t < NBW; ++t) {
                if (t + 1 < NBW) { if (active(t + 1)) load_a(avp[(t + 1) & 1], wave + W * (t + 1), c); }
                if (active(t)) {
                    if constexpr (VREG) update_tile(acc[t], vb, avp[t & 1]);
                    else update_tile_lds(acc[t], Vb, avp[t & 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            TRACE();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");   // LDS only: global loads in flight need not drain
            if (lane == 0) done[wave] = c;
        }
    }
    __syncthreads();

    TRACE();
    if constexpr (TR) { if (trc && lane == 0) { trc[511] = tri; trc[8 * 512 + 511] = tro; } }
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

// Size classes by nb = ceil(K/32): W waves x NBW tiles per wave (ongpis.h, ongpis_class_of_nb).
static const int kClassW[6] = {1, 2, 4, 8, 16, 8};
static const int kClassNb[6] = {4, 8, 16, 32, 64, 96};

static size_t eval_lds_bytes(int W, int maxN, int maxLd, int use_table) {
    size_t fixed = sizeof(float) * (W * 128 + 32) + sizeof(float) * 2 * (size_t)maxLd + 16 * (size_t)maxN;
    size_t s2 = sizeof(float) * (size_t)W * 1152 + (use_table ? sizeof(double) * 8 * (size_t)maxN : 0);
    size_t s3 = sizeof(float) * 4 * 1024;   // RING = 4 published V blocks
    return fixed + std::max(s2, s3);
}

int ongpis_eval_launch(int wclass, int ntiles, int maxN, const EvalArgs& args_in, hipStream_t s) {
    if (ntiles <= 0) return GPIS_OK;
    if (wclass < 0 || wclass >= ONGPIS_NCLASS) return GPIS_ERR_ARG;
    EvalArgs args = args_in;
    const int W = kClassW[wclass];
    const int maxLd = kClassNb[wclass] * 32 + 32;
    const size_t budget = 150 * 1024;     // one workgroup must fit; two per CU when <= 80 KB
    args.use_table = 1; args.lds_model = 1;
    args.dbg = 0; args.trace = nullptr; args.trace_block = 0;
#ifdef GPIS_K4_INSTRUMENT
    { const char* e = getenv("GPIS_K4_DBG"); args.dbg = e ? atoi(e) : 0; }
    static unsigned long long* d_trace = nullptr;
    if (const char* e = getenv("GPIS_K4_TRACE")) {
        if (!d_trace) { (void)hipMalloc(&d_trace, sizeof(unsigned long long) * 512 * 16); }
        (void)hipMemsetAsync(d_trace, 0, sizeof(unsigned long long) * 512 * 16, s);
        args.trace = d_trace; args.trace_block = atoi(e);
    }
#endif
    size_t lds = eval_lds_bytes(W, maxN, maxLd, 1);
    if (lds > budget) { args.use_table = 0; lds = eval_lds_bytes(W, maxN, maxLd, 0); }
    if (lds > budget) return GPIS_ERR_LIMIT;
    if (args.use_table && args_in.use_table == 0) { args.use_table = 0; lds = eval_lds_bytes(W, maxN, maxLd, 0); }   // caller asked for the table-free path (large-cluster path; tests force it)
    typedef void (*kern_t)(EvalArgs);
    static const kern_t kern[ONGPIS_NCLASS] = {
        ongpis_eval_kernel<1, 4, 2, true, false>, ongpis_eval_kernel<2, 4, 2, true, false>,
        ongpis_eval_kernel<4, 4, 2, true, false>, ongpis_eval_kernel<8, 4, 4, false, false>,
        ongpis_eval_kernel<16, 4, 4, false, false>, ongpis_eval_kernel<8, 12, 2, true, false>};
    static bool attr_set = false;
    if (!attr_set) {
        attr_set = true;
        for (int i = 0; i < ONGPIS_NCLASS; ++i)
            (void)hipFuncSetAttribute((const void*)kern[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
#ifdef GPIS_K4_INSTRUMENT
    static const kern_t kern_tr = ongpis_eval_kernel<8, 4, 4, false, true>;   // traced build of class 3
    if (args.trace && wclass == 3) {
        (void)hipFuncSetAttribute((const void*)kern_tr, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(kern_tr, dim3(ntiles), dim3(64 * W), lds, s, args);
        (void)hipStreamSynchronize(s);
        static unsigned long long h[512 * 16];
        (void)hipMemcpy(h, d_trace, sizeof(h), hipMemcpyDeviceToHost);
        FILE* f = fopen("gpurun_out/k4_trace.txt", "w");
        if (f) {
            for (int w = 0; w < 2 * W; ++w) {   // rows W..2W-1: owner-path events
                int n = (int)h[w * 512 + 511];
                fprintf(f, "wave %d n %d\n", w, n);
                for (int i = 0; i < n && i < 511; ++i) fprintf(f, "%llu\n", h[w * 512 + i] - h[0]);
            }
            fclose(f);
        }
        return hipGetLastError() == hipSuccess ? GPIS_OK : GPIS_ERR_HIP;
    }
#endif
    hipLaunchKernelGGL(kern[wclass], dim3(ntiles), dim3(64 * W), lds, s, args);
    return hipGetLastError() == hipSuccess ? GPIS_OK : GPIS_ERR_HIP;
}

int ongpis_eval_class(int nb) {
    for (int c = 0; c < ONGPIS_NCLASS; ++c) if (nb <= kClassNb[c]) return c;
    return -1;
}

}  // namespace gpis
