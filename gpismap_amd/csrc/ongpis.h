// Device-resident OnGPIS local regressors (reference cpp/include/OnGPIS.h:38-73):
// batched training (K6 gather + kernel-matrix build + K3 Cholesky/solve) and
// batched prediction (K4) over per-cluster models that live in HBM.
#pragma once
#include <vector>
#include "dev_common.h"

namespace gpis {

// One trained cluster model.  All pointers are device pointers into one pooled
// allocation (`base`).  L is column-major with leading dimension ld (multiple of
// 32, >= K+1); rows/cols >= K of the padded square carry the identity, so a
// 32-blocked triangular solve can run over nb = ceil(K/32) full blocks.
struct ClusterModel {
    int dim;    // 2 or 3
    int N;      // training points
    int ng;     // points with a usable normal
    int K;      // N + dim*ng
    int ld;     // leading dimension of L
    int nb;     // ceil(K / 32)
    float scale;            // GP length scale
    float* L;               // [ld*ld] lower Cholesky factor
    float* Lt;              // [nbr(nbr+1)/2][1024], nbr = ld/32: 32x32 tiles in MFMA A-operand order, 4 x 16 B per lane:
                            // [g][lane][j] = T[lane & 31][k(4g + j, lane >> 5)].  Off-diagonal tile (b, c): T = -L_bc,
                            // k(kk, h) = 2 kk + h.  Diagonal tile (c, c): T = inv(L_cc) (identity-padded),
                            // k(kk, h) = (kk & 3) + 8 (kk >> 2) + 4 h, the row of an accumulator tile register.
    float* Xt;              // explicit inverse X = L^-1 re-tiled for K4 (written by K3b): same tile indexing as Lt, every tile
                            // (b, c), c <= b, in NATURAL A-operand order [g][lane][j] = X_bc[lane & 31][2 (4g + j) + (lane >> 5)];
                            // row K of X (block row ld/32 - 1) carries alpha, so that row K of V = X k* is the mean k*^T alpha
    float* Zt;              // the transposed tiles (X_bc)^T in the same order at the same index: what K3b's own recurrence
                            // feeds to the matrix cores as the B operand
    float* alpha;           // [ld]
    float* x4;              // [N][4]  (x, y, z|0, 0)
    int* rowinfo;           // [ld] row -> point | comp<<28 (comp 0 = value row, 1..dim = d/dx_c)
    // training scratch (kept: tiny)
    float* y;               // [ld]  targets, then z = L^-1 y
    float* sig;             // [2*N] sigx' (after the 2.0 override), sigg
    int* gidx;              // [N]   running gradient index or -1
    void* base;
    void* scratch;          // second allocation of a fully trained model: factor, re-tiled factor, inverse scratch, training vectors
                            // (host bookkeeping; freed after the inverse when the store trims: OnGPISStore::trim_scratch)
};

// Map-point mirror in HBM: structure of arrays, index = stable point id.
struct PointsSoA {
    int n = 0, cap = 0;
    float* d = nullptr;  // [9][cap]: px py pz gx gy gz val sigx sigg (2-D: pz = gz = 0)
};

struct TrainJob {       // host description of one cluster to (re)train
    int model;          // slot in the store
    int off, n;         // range in the concatenated point-id list
    int ng;             // number of gradient-bearing points (host-computed, OnGPIS.cpp:122-125)
};

class OnGPISStore {
public:
    OnGPISStore(int dim, float scale);
    ~OnGPISStore();
    int dim() const { return dim_; }
    // allocate an empty slot / release it (cluster node destroyed)
    int new_slot();
    void release_slot(int slot);
    int drop_models(const std::vector<int>& slots, hipStream_t s);    // mark untrained (memory back to the pool)
    std::vector<int> take_dropped() { std::vector<int> v; v.swap(dropped_); return v; }   // models dropped by failed training / inverse passes since the last call
    void canonical_free_slots();    // free ids handed out in ascending order from now on (stores mirrored across devices re-align after a refused load)
    void clear();
    // Upload the point mirror (host SoA rows of length n, 9 rows).
    int upload_points(const float* soa9, int n, hipStream_t s);
    // K6 + K3 for a batch of clusters.  ids = concatenated point ids (tree order).
    int train_batch(const std::vector<TrainJob>& jobs, const std::vector<int>& ids, hipStream_t s);
    // K6 range part on the device (ongpis_range_gather_kernel): the caller lists the points of every touched cell once
    // (cell_pts), names per cluster its cell entries (cranges: begin / end into cell_pts) and gives 8 ints per cluster
    // (desc: first cell entry, cells, offset of the cluster's id list, centre x y z and range^2 as float bits, 0); the id
    // lists are written into the store's device id buffer (total_ids = sum of the clusters' capacities) and counts
    // receives (points, gradient-bearing points) per cluster.  train_batch_dev() then trains jobs whose `off` are those offsets.
    int gather_ranges(const int* cell_pts, int npts, const int* cranges, int nentries, const int* desc, int nclusters, int total_ids,
                      int* counts, hipStream_t s);
    int train_batch_dev(const std::vector<TrainJob>& jobs, hipStream_t s);
    // Predict: jobs (query index, model slot) evaluated against xq (device, [nq][4]); out
    // (device) receives 2*(1+dim) floats per job: mean(1+dim), var(1+dim).
    int eval_jobs(const float* d_xq4, const int* h_job_q, const int* h_job_model, int njobs, float* d_out,
                  hipStream_t s);
    // ---- multi-GPU exchange (model_pack.hip): packed records of what K4 needs from a trained model ----
    size_t packed_bytes(const int* slots, int n) const;                 // largest record among the listed models
    // record i at byte offset offs[i] of d_buf (offs == nullptr: i * stride; otherwise stride is ignored and the records
    // sit back to back at their own sizes: offs has n + 1 entries, the last one the end of the buffer)
    // factors = true: a model whose explicit inverse is still pending (lazy inverse) travels as its FACTOR (Lt + alpha, a kind-1
    // record of the same size) instead of forcing the inverse here; models that have their X travel as before.
    int pack_models(const int* slots, int n, void* d_buf, size_t stride, hipStream_t s, const size_t* offs = nullptr, bool factors = false);
    int last_unpack_factors = 0;    // factor records among the models of the last unpack_models() (their inverse is deferred)
    // records -> models.  slots_inout[i] < 0: a new slot is created and returned there; else that slot is (re)used.
    // The models are predict-only (no factor, no training scratch).
    int unpack_models(const void* d_buf, int n, size_t stride, int* slots_inout, hipStream_t s, const size_t* offs = nullptr);
    // Kernel matrix only (the separate build kernel on caller-given arrays, no gather rule): x [N][dim], gidx [N] running
    // gradient index or -1, sigx / sigg [N]; K_out receives the K x K lower triangle, column-major (ld = K).  Parity of
    // covFnc.cpp:142-256 / :317-402 against committed fixtures.
    int kernel_matrix(const float* x, const int* gidx, const float* sigx, const float* sigg, int N, float* K_out, hipStream_t s);
    const ClusterModel* model(int slot) const { return (slot >= 0 && slot < (int)models_.size() && live_[slot]) ? &models_[slot] : nullptr; }
    ClusterModel* d_models() { return d_models_; }   // device array mirroring models_
    int sync_models(hipStream_t s);                  // re-upload descriptor table if dirty
    int num_slots() const { return (int)models_.size(); }
    std::vector<int> live_slots() const { std::vector<int> v; for (int i = 0; i < (int)models_.size(); ++i) if (live_[i]) v.push_back(i); return v; }
    size_t device_bytes() const;
    DevPool* pool() { return pool_; }
    // timing of the dominant kernels (hipEvents on the launch stream), ms of the last call
    float last_train_ms = 0.f, last_eval_ms = 0.f;
    double last_train_flops = 0.0, last_train_bytes = 0.0;   // algorithmic: sum K^3/3 + 2K^2 ; 36 N + 4 [K(K+1)/2 + K]
    int last_train_jobs = 0, last_train_maxK = 0;
    long long last_eval_flops = 0;
    bool profile = false;
    bool use_exp_table = true;   // accepted and ignored: K4 evaluates the exponential per entry since round 5 (an exp table in LDS cost ring width,
                                 // one in global memory -- round 6 -- a memory round trip per generated tile: both measured slower)
    bool keep_factor = false;    // models of at most ONGPIS_FUSED_MAX_K rows are trained on chip and keep only what K4 reads
                                 // (rowinfo, x4, Xt); true: they also receive L, alpha, gidx (parity tests, gpis_ongpis_get_model)
    int debug_inject = 0;        // test-only fault injection: bits 0..3 the cooperative kernel (ongpis_train.hip, ctl[1]), bit 4 the ring of K4
    // Error word of the prediction kernels: page-locked host memory the kernels raise bits in (nothing is written in a healthy
    // launch); the caller that synchronises the launch stream reads and clears it (take_eval_err).
    int* eval_err();
    int take_eval_err() { if (!h_eval_err_) return 0; const int v = *(volatile int*)h_eval_err_; *(volatile int*)h_eval_err_ = 0; return v; }
    int wait_limit_ticks = 0;    // bound of the in-kernel waits in 100 MHz ticks (0: 2 s)
    bool use_fused = true;       // false: every cluster takes the separate gather / build / factorise / invert kernels
    // Pipelined training: train_batch() returns once the kernels are enqueued (on the caller's stream and the side streams)
    // and train_finish() joins them -- waits, reads the error word, drops the batch on error.  The map object sets this so
    // that the host work of the NEXT update() runs beside the factorisations of this one; every other entry point of the
    // store that touches models, points or the training buffers joins first.  false: train_batch() joins before it returns.
    bool defer_finish = false;
    // CUs the training streams leave alone (hipExtStreamCreateWithCUMask on the side streams; the caller masks its own stream
    // with ongpis_make_train_stream).  The pipelined map update sets it: the ObsGP queries of the next frame then find free CUs
    // at once instead of waiting for a factorisation workgroup (milliseconds each) to end.  Changing it joins the training.
    int set_cu_reserve(int n);
    int cu_reserve() const { return cu_reserve_; }
    // Lazy inverse (default): training of the K > 256 clusters stops at the factor and alpha; the explicit inverse X = L^-1
    // that prediction multiplies with (K3b) is computed by ensure_inverses() at the FIRST use of the models after a
    // training -- prediction (eval_jobs, the map's test()), packing for the exchange.  A cluster retrained in several
    // consecutive updates is inverted once, when somebody asks (the reference's train() does not pay for prediction
    // either: OnGPIS.cpp:139-143 stops at L and alpha).  false: K3b runs behind K3 in every training batch.
    bool lazy_inverse = true;
    // Models of more than 256 rows live in TWO allocations: what prediction reads (rowinfo, x4, Xt) and the training side (L,
    // alpha, y, sig, gidx, Lt, Zt = 8 of their 10 K^2 bytes).  With trim_scratch the training side goes back to the pool as soon
    // as the inverse exists (the maps set it: a cluster then holds 2 K^2 bytes between its retrainings, half of what the
    // reference's dense L costs; gpis_ongpis_get_model needs the factor and leaves it off).
    bool trim_scratch = false;
    // Lazy inverse: bytes of training-side memory that models waiting for their inverse may hold before a training batch
    // inverts and trims them (a caller that never predicts would otherwise keep 10 K^2 bytes per cluster for ever)
    size_t stale_bytes_limit = (size_t)4 << 30;
    int ensure_inverses(hipStream_t s);
    float last_inverse_ms = 0.f;     // K3b pass of the last ensure_inverses() that had work (profiling on)
    int last_inverse_jobs = 0;
    int train_finish();
    bool train_pending() const { return pend_active_; }

private:
    enum AllocKind { kAllocFull = 0, kAllocPredictOnly = 1, kAllocLeanFactor = 2, kAllocFactorImport = 3 };
    int alloc_model(int slot, int N, int ng, int kind = kAllocFull);
    int train_batch_impl(const std::vector<TrainJob>& jobs, const std::vector<int>* ids, hipStream_t s);
    void build_inverse_work(const std::vector<int>& tab, int jbeg, int jend, int kLongCol, std::vector<int>& work,
                            int& off, int& nlong, int& nmid, int& nshort) const;
    void free_model_mem(ClusterModel& m);
    void trim_models(const std::vector<int>& slots);
    void mark_stale(int slot) { if ((int)xstale_.size() < (int)models_.size()) xstale_.resize(models_.size(), 0); xstale_[slot] = 1; stale_list_.push_back(slot); }
    std::vector<char> xstale_;                   // per slot: the factor is newer than Xt (lazy inverse)
    std::vector<int> stale_list_;                // slots marked since the last ensure_inverses()
    int* h_eval_err_ = nullptr;
    std::vector<int> dropped_;                   // slots whose models an error word made this store drop (take_dropped)
    int train_allocated(const std::vector<TrainJob>& jobs, const std::vector<int>* ids, hipStream_t s, int deferred_rc);
    int train_enqueue(const std::vector<TrainJob>& jobs, const std::vector<int>* ids, hipStream_t s, int deferred_rc);
    int coop_dev_ = 0, coop_held_ = 0;           // cooperative workgroups this store holds of its device's budget (batch in flight)
    std::vector<int> coop_hdr_;                  // host copy of the cooperative clusters' header words (kept: the upload is asynchronous)
    bool enqueued_ = false;                      // train_enqueue() got as far as handing the batch to train_finish()
    int* d_rg_ = nullptr; int cap_rg_ = 0;       // range gather: cell point lists, cell entries, cluster descriptors, counts
    int dev_ids_ = 0;                            // ids the last gather_ranges() left in d_ids_
    int dim_;
    float scale_;
    DevPool* pool_;
    std::vector<ClusterModel> models_;
    std::vector<char> live_;
    std::vector<int> free_slots_;
    ClusterModel* d_models_ = nullptr;
    int d_models_cap_ = 0;
    bool dirty_ = true;
    PointsSoA pts_;
    int* d_ids_ = nullptr; int cap_ids_ = 0;
    int* d_jobs_ = nullptr; int cap_jobs_ = 0;   // train job table (4 ints per job)
    int* d_work_ = nullptr; int cap_work_ = 0;   // K3b work list (job, block column)
    int* d_cwork_ = nullptr; int cap_cwork_ = 0; // cooperative K3: (job, g, G) per workgroup, then 2 sync ints per job
    int* d_ej_ = nullptr; int cap_ej_ = 0;       // eval job arrays
    int* d_slots_ = nullptr; int cap_slots_ = 0; // pack / unpack slot list (grown on demand: no hipMalloc / hipFree per call)
    int* d_err_ = nullptr;                       // device error word of the training kernels (zeroed per batch)
    int* h_err_ = nullptr;                       // its page-locked host copy (written by the batch's last copy)
    bool pend_active_ = false, pend_profile_ = false, pend_lazy_ = false;   // a training batch is in flight (train_finish joins it)
    hipStream_t pend_stream_ = nullptr;
    std::vector<int> pend_models_;               // slots of the batch in flight
    hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
    int cu_reserve_ = 0;
    hipStream_t s2_ = nullptr, s3_ = nullptr;    // side streams: the three size groups of a training batch run beside each other
    hipEvent_t evf_ = nullptr, evj_ = nullptr, evj3_ = nullptr;   // fork / joins
};

// K6 + K3 + K3b in one launch for clusters of at most 8 block rows (K <= 256), everything on chip: ongpis_fused.hip
struct FusedTrainArgs {
    const ClusterModel* models;
    const int* jobs;        // 4 ints per job: model slot, offset into ids, N, ng
    const int* ids;         // concatenated point ids
    const float* pts;       // point mirror [9][cap]
    int cap;
    int* err;               // device error word (bit 0: a job the kernel cannot hold was routed to it)
};
#define ONGPIS_FUSED_MAX_K 256
size_t ongpis_fused_lds_bytes(int nb);
int ongpis_launch_train_fused(const FusedTrainArgs& a, int njobs, int max_nb, hipStream_t s);

// kernels (ongpis_train.hip / ongpis_test.hip)
size_t packed_model_bytes(int ld, int N);
void model_pack_launch(bool pack, const ClusterModel* d_models, const int* d_slots, int n, char* d_buf, const unsigned long long* d_offs, hipStream_t s);
void model_headers_launch(const char* d_buf, const unsigned long long* d_offs, int n, int* d_out, hipStream_t s);
void ongpis_launch_gather(const ClusterModel* d_models, const int* d_jobs, int njobs, const int* d_ids,
                          const float* d_pts, int pts_cap, hipStream_t s);
void ongpis_launch_buildK(const ClusterModel* d_models, const int* d_jobs, int njobs, hipStream_t s);
void ongpis_launch_range_gather(const int* d_desc, const int* d_cranges, const int* d_cell_pts, int nclusters, const float* d_pts, int pts_cap,
                                int dim, int* d_ids, int* d_counts, hipStream_t s);
void ongpis_launch_chol(const ClusterModel* d_models, const int* d_jobs, int njobs, int tier, hipStream_t s);
#ifdef GPIS_EXPERIMENTS   // tools/experiments/: archived kernels, built only with EXTRA=-DGPIS_EXPERIMENTS, selected by GPIS_ASYNC_CHOL / GPIS_SMALL_KERNEL
void ongpis_launch_chol_async(const ClusterModel* d_models, const int* d_jobs, int njobs, int* d_ctl, hipStream_t s);
#endif
// K3 for the largest clusters: G cooperating workgroups each; cwork = (job, g, G) per workgroup (job < 0: padding), sync = 3 ints per job (zeroed)
// d_ctl: 4 ints -- [0] error word of the batch (bit 0 fused kernel refused a job, bit 1 cooperative wait expired, bit 2 K3b row
// wait expired), [1] test-only fault injection, [2] wait bound in ticks of the 100 MHz device clock (0: 2 s)
// the cooperative clusters (data flow over per-row progress words, ongpis_train.hip): d_sync[3 j + 2] = offset of cluster j's 2 x rows flag words
void ongpis_launch_chol_flow(const ClusterModel* d_models, const int* d_jobs, const int* d_cwork, int nwg, int* d_sync, int* d_ctl, hipStream_t s);
int ongpis_coop_capacity();
// A stream for training kernels: reserve_cus == 0: non-blocking, lowest priority; reserve_cus > 0: restricted to the first
// (CUs - reserve_cus) bits of the CU mask (such a stream is a DEFAULT-flag, normal-priority stream: HIP offers no masked creator
// with flags; it is kept off the reserved CUs instead of being de-prioritised) -- on gfx950 bit i is CU i / 8 of XCD i % 8, so every XCD keeps the same number of CUs and workgroup ids still go round
// the eight XCDs (tools/ubench/cumask_probe.hip).
int ongpis_make_train_stream(hipStream_t* s, int reserve_cus);
// K3b: explicit inverse of every factor of the batch, one wavefront per (job, block column); work = (job, column) pairs
void ongpis_launch_inverse(const ClusterModel* d_models, const int* d_jobs, const int* d_work, int nlong, int nmid, int nshort, int* d_ctl, hipStream_t s);
int ongpis_inverse_short_rows();
int ongpis_inverse_short_waves();
int ongpis_inverse_mid_waves();

struct EvalArgs {
    const ClusterModel* models;
    const float4* xq;        // [nq] query points (x, y, z|0, 0)
    const int* tile_model;   // [ntiles]
    const int* tile_off;     // [ntiles] first job of the tile in job_q / job_out
    const int* tile_cnt;     // [ntiles] 1..ONGPIS_TILE_Q
    const int* job_q;        // query index per job (sorted by model)
    const int* job_out;      // output record per job
    float* out;              // [records][8]: mean(4) var(4)  (2-D uses 3+3, slots 3 and 7 unused)
    int cb;                  // column blocks per B chunk (set by ongpis_eval_launch from the LDS budget)
    int debug;               // test hooks (gpis_ongpis_set_debug): bit 4 = one ring signal of the launch's first workgroup is withheld
    int* err;                // error word of the store (page-locked host memory, OnGPISStore::eval_err): bit 0 = a ring wait of K4 expired
    unsigned long long* trace;   // instrumented builds only (tools/k4_ablate.sh); nullptr otherwise
};
// K4 size classes by nbx = ld / 32 = ceil((K+1)/32) block rows: W = 1, 2, 4 wavefronts per workgroup for nbx <= 4, 8, 16
// and 8 above (no upper limit: large clusters run several row groups).  The widest class is cut in three by size
// (nbx <= 32, <= 48, more) only because the LDS of a launch is sized by its largest cluster: one giant cluster must not
// push the exp table of every other cluster out of LDS.
#define ONGPIS_NCLASS 7
#define ONGPIS_SMALL_NBX 9    // clusters of at most this many block rows (K <= 287) take the resident-X kernel (ongpis_test_small.hip)
#ifndef K4_QS
#define K4_QS 1   // measured on the 256^3 bench: 1 set / 128 VGPRs / 2 workgroups per CU 811 ms, 2 sets / 256 VGPRs / 1 per CU 870 ms
#endif
#define ONGPIS_TILE_Q (8 * K4_QS)   // queries per K4 workgroup (K4_QS sets of 8 sharing every X tile)
#define ONGPIS_MAX_K 16384   // allocation sanity bound (10 K^2 bytes per model); the binding limit is K4's LDS: ongpis_eval_fits
__host__ __device__ inline int ongpis_class_of_nbx(int nbx) {
    return nbx <= 4 ? 0 : (nbx <= 8 ? 1 : (nbx <= ONGPIS_SMALL_NBX ? 2 : (nbx <= 16 ? 3 : (nbx <= 32 ? 4 : (nbx <= 48 ? 5 : 6)))));
}
#ifdef GPIS_EXPERIMENTS
size_t ongpis_eval_small_lds(int maxN, int maxLd);
int ongpis_eval_small_launch(int ntiles, int maxN, int maxLd, const EvalArgs& args, hipStream_t s);
#endif
int ongpis_eval_class(int nbx);
bool ongpis_eval_fits(int N, int ld);
int ongpis_eval_launch(int wclass, int ntiles, int maxN, int maxLd, const EvalArgs& args, hipStream_t s);

}  // namespace gpis
