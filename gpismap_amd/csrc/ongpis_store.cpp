// Host side of the device OnGPIS model store: pooled per-cluster allocations,
// the map-point mirror, batched train launches (K6 + K3) and a job-level predict
// entry (K4) used by the kernel-level C-ABI and the parity tests.
#include <algorithm>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <numeric>
#include <set>
#include <utility>
#include "ongpis.h"

namespace gpis {

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int ensure_dynamic_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return GPIS_ERR_HIP;
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({dev, kernel})) return GPIS_OK;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) { fprintf(stderr, "[gpismap_amd] hipFuncSetAttribute(device %d): %s\n", dev, hipGetErrorString(e)); return GPIS_ERR_HIP; }
    done.insert({dev, kernel});
    return GPIS_OK;
}

// The cooperative factorisation waits on its partner workgroups, so every workgroup of a launch must be resident.  Several
// stores may drive ONE device from their own host threads (GPIS_DEVICES=0,0,0; ranks sharing a GPU inside one process): they
// share one per-device budget of cooperative workgroups.  A store takes what is free when it enqueues a batch and gives it
// back when the batch is joined; clusters that do not fit the share take the one-workgroup kernel (same chains, same bits).
namespace {
std::mutex g_coop_mu;
std::map<int, std::pair<int, int>> g_coop;   // device -> (capacity, in use)
int coop_take_all(int dev) {
    std::lock_guard<std::mutex> lk(g_coop_mu);
    auto it = g_coop.find(dev);
    if (it == g_coop.end()) it = g_coop.emplace(dev, std::make_pair(std::max(2, ongpis_coop_capacity()), 0)).first;
    const int free_wg = std::max(0, it->second.first - it->second.second);
    it->second.second += free_wg;
    return free_wg;
}
void coop_give_back(int dev, int n) {
    if (n <= 0) return;
    std::lock_guard<std::mutex> lk(g_coop_mu);
    auto it = g_coop.find(dev);
    if (it != g_coop.end()) it->second.second = std::max(0, it->second.second - n);
}
}  // namespace

OnGPISStore::OnGPISStore(int dim, float scale) : dim_(dim), scale_(scale), pool_(pool_create()) {
}

int ongpis_make_train_stream(hipStream_t* s, int reserve_cus) {
    int pr_least = 0, pr_greatest = 0, dev = 0, ncu = 0;
    GPIS_HIP(hipGetDevice(&dev));
    GPIS_HIP(hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest));
    GPIS_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    if (reserve_cus <= 0 || reserve_cus >= ncu) { GPIS_HIP(hipStreamCreateWithPriority(s, hipStreamNonBlocking, pr_least)); return GPIS_OK; }
    std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
    for (int i = 0; i < ncu - reserve_cus; ++i) mask[(size_t)i / 32] |= 1u << (i % 32);
    // (HIP has no creator that takes a CU mask AND flags / a priority: as far as the clr sources go, a masked stream is a
    // hipStreamDefault stream -- it synchronises implicitly with the legacy NULL stream -- at normal priority.  The library itself
    // never uses the NULL stream; a host process that does should set GPIS_PIPELINE_RESERVE_CUS=0 (ordinary non-blocking,
    // lowest-priority training streams) or GPIS_PIPELINE_UPDATE=0: INTEGRATION.md "Pipelined update and the NULL stream".)
    if (hipExtStreamCreateWithCUMask(s, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
        (void)hipGetLastError();      // (a device or runtime without CU masks: an ordinary lowest-priority stream -- the pipelined update still works, its ObsGP batches just wait for CUs)
        GPIS_HIP(hipStreamCreateWithPriority(s, hipStreamNonBlocking, pr_least));
    }
    return GPIS_OK;
}

int OnGPISStore::set_cu_reserve(int n) {
    n = std::max(0, n);
    if (n == cu_reserve_) return GPIS_OK;
    const int rc = train_finish();
    if (s2_) { (void)hipStreamSynchronize(s2_); (void)hipStreamDestroy(s2_); s2_ = nullptr; }
    if (s3_) { (void)hipStreamSynchronize(s3_); (void)hipStreamDestroy(s3_); s3_ = nullptr; }
    cu_reserve_ = n;
    return rc;
}

OnGPISStore::~OnGPISStore() {
    clear();
    if (h_err_) (void)hipHostFree(h_err_);
    if (h_eval_err_) (void)hipHostFree(h_eval_err_);
    (void)hipFree(d_models_); (void)hipFree(pts_.d); (void)hipFree(d_ids_); (void)hipFree(d_jobs_); (void)hipFree(d_work_); (void)hipFree(d_cwork_); (void)hipFree(d_ej_); (void)hipFree(d_err_); (void)hipFree(d_slots_); (void)hipFree(d_rg_);
    if (ev0_) (void)hipEventDestroy(ev0_);
    if (ev1_) (void)hipEventDestroy(ev1_);
    if (evf_) (void)hipEventDestroy(evf_);
    if (evj_) (void)hipEventDestroy(evj_);
    if (s2_) (void)hipStreamDestroy(s2_);
    if (s3_) (void)hipStreamDestroy(s3_);
    if (evj3_) (void)hipEventDestroy(evj3_);
    pool_destroy(pool_);
}

void OnGPISStore::clear() {
    (void)train_finish();
    for (size_t i = 0; i < models_.size(); ++i)
        if (live_[i]) free_model_mem(models_[i]);
    models_.clear(); live_.clear(); free_slots_.clear();
    xstale_.clear(); stale_list_.clear(); dropped_.clear();
    dirty_ = true;
}

int OnGPISStore::new_slot() {
    int s;
    if (!free_slots_.empty()) { s = free_slots_.back(); free_slots_.pop_back(); }
    else { s = (int)models_.size(); models_.emplace_back(); live_.push_back(0); }
    std::memset(&models_[s], 0, sizeof(ClusterModel));
    live_[s] = 1;
    dirty_ = true;
    return s;
}

void OnGPISStore::release_slot(int s) {
    if (s < 0 || s >= (int)models_.size() || !live_[s]) return;
    (void)train_finish();   // (the memory of a model of the batch in flight must not return to the pool under the kernels)
    if (s < (int)xstale_.size()) xstale_[s] = 0;
    free_model_mem(models_[s]);
    std::memset(&models_[s], 0, sizeof(ClusterModel));
    live_[s] = 0;
    free_slots_.push_back(s);
    dirty_ = true;
}

// Mark models untrained (their cells answer test() with the prior): used to bring the stores of several devices behind one
// map back in step after ONE of them had to drop models (an inverse pass of imported factors that reported an error word).
int OnGPISStore::drop_models(const std::vector<int>& slots, hipStream_t s) {
    (void)train_finish();
    bool any = false;
    for (int slot : slots) {
        if (slot < 0 || slot >= (int)models_.size() || !live_[slot] || !models_[slot].base) continue;
        ClusterModel& m = models_[slot];
        free_model_mem(m);
        std::memset(&m, 0, sizeof(ClusterModel));
        if (slot < (int)xstale_.size()) xstale_[slot] = 0;
        any = true;
    }
    if (!any) return GPIS_OK;
    dirty_ = true;
    return sync_models(s);
}

int* OnGPISStore::eval_err() {
    if (!h_eval_err_) {
        if (hipHostMalloc((void**)&h_eval_err_, 64, hipHostMallocMapped) != hipSuccess) { h_eval_err_ = nullptr; return nullptr; }
        *h_eval_err_ = 0;
    }
    return h_eval_err_;
}

void OnGPISStore::canonical_free_slots() { std::sort(free_slots_.begin(), free_slots_.end(), std::greater<int>()); }

size_t OnGPISStore::device_bytes() const { return pool_bytes(pool_); }

void OnGPISStore::free_model_mem(ClusterModel& m) {
    if (m.base) pool_free(pool_, m.base);
    if (m.scratch) pool_free(pool_, m.scratch);
    m.base = nullptr; m.scratch = nullptr;
}

// the training side of models whose inverse exists goes back to the pool (trim_scratch)
void OnGPISStore::trim_models(const std::vector<int>& slots) {
    for (int slot : slots) {
        if (slot < 0 || slot >= (int)models_.size() || !live_[slot]) continue;
        ClusterModel& m = models_[slot];
        if (!m.scratch) continue;
        if (slot < (int)xstale_.size() && xstale_[slot]) continue;   // its X is still to be computed FROM this factor
        pool_free(pool_, m.scratch);
        m.scratch = nullptr;
        m.L = nullptr; m.alpha = nullptr; m.y = nullptr; m.sig = nullptr; m.gidx = nullptr; m.Lt = nullptr; m.Zt = nullptr;
        dirty_ = true;
    }
}

int OnGPISStore::alloc_model(int slot, int N, int ng, int kind) {
    ClusterModel& m = models_[slot];
    int K = N + dim_ * ng;
    int ld = (int)align_up((size_t)K + 1, 32);
    int nbk = ld / 32;   // block rows incl. the one holding the y row (the factorisation uses those tiles as operands)
    const size_t szT = sizeof(float) * 1024 * (size_t)nbk * (nbk + 1) / 2;
    // Round 6: a model that is retrained at (about) its old size keeps its blocks -- a frame retrains ~500 models and most of them
    // grew by a few points or not at all; every one used to go through two frees and two best-fit searches of the pool.  A block is
    // kept when it holds the new size with at most a quarter to spare (everything in it is rewritten by the training kernels).
    char* keep_base = nullptr; char* keep_sc = nullptr;
    auto take = [&](void*& blk, size_t need) -> char* {
        const size_t have = pool_block_size(pool_, blk);
        if (have < need || have > need + need / 4 + (64u << 10)) return nullptr;
        char* b = (char*)blk; blk = nullptr; return b;
    };
    auto alloc_or = [&](char* kept, size_t need) -> char* { return kept ? kept : (char*)pool_alloc(pool_, need); };
    if (kind == kAllocPredictOnly || kind == kAllocLeanFactor) {
        // what K4 reads (rowinfo, x4, Xt): imported models and models trained by the fused on-chip kernel;
        // kAllocLeanFactor adds the factor, alpha and the gradient index for parity tests / gpis_ongpis_get_model
        size_t oR = 0, szR = sizeof(int) * ld;
        size_t oX = align_up(oR + szR, 256), szX = sizeof(float) * 4 * (size_t)N;
        size_t oXt = align_up(oX + szX, 256);
        size_t total = align_up(oXt + szT, 256);
        size_t oL = 0, oA = 0, oG = 0;
        if (kind == kAllocLeanFactor) {
            oL = total; oA = align_up(oL + sizeof(float) * (size_t)ld * ld, 256); oG = align_up(oA + sizeof(float) * ld, 256);
            total = align_up(oG + sizeof(int) * (size_t)N, 256);
        }
        { void* b = m.base; keep_base = take(b, total); m.base = (char*)b; }
        free_model_mem(m);
        char* base = alloc_or(keep_base, total);
        if (!base) return GPIS_ERR_HIP;
        std::memset(&m, 0, sizeof(ClusterModel));
        m.dim = dim_; m.N = N; m.ng = ng; m.K = K; m.ld = ld; m.nb = (K + 31) / 32; m.scale = scale_;
        m.rowinfo = (int*)(base + oR); m.x4 = (float*)(base + oX); m.Xt = (float*)(base + oXt);
        if (kind == kAllocLeanFactor) { m.L = (float*)(base + oL); m.alpha = (float*)(base + oA); m.gidx = (int*)(base + oG); }
        m.base = base;
        dirty_ = true;
        return GPIS_OK;
    }
    if (kind == kAllocFactorImport) {
        // a factor received from another rank: the prediction side plus what K3b needs to produce it -- the re-tiled factor, the
        // transposed-tile scratch and alpha (6 K^2 bytes until the inverse exists, 2 K^2 after the trim)
        size_t oR = 0, szR = sizeof(int) * ld;
        size_t oX = align_up(oR + szR, 256), szX = sizeof(float) * 4 * (size_t)N;
        size_t oXt = align_up(oX + szX, 256);
        const size_t total_p = align_up(oXt + szT, 256);
        size_t oA = 0, szA = sizeof(float) * ld;
        size_t oT = align_up(oA + szA, 256);
        size_t oZt = align_up(oT + szT, 256);
        const size_t total_s = align_up(oZt + szT, 256);
        free_model_mem(m);
        char* base = (char*)pool_alloc(pool_, total_p);
        if (!base) return GPIS_ERR_HIP;
        char* sc = (char*)pool_alloc(pool_, total_s);
        if (!sc) { pool_free(pool_, base); return GPIS_ERR_HIP; }
        std::memset(&m, 0, sizeof(ClusterModel));
        m.dim = dim_; m.N = N; m.ng = ng; m.K = K; m.ld = ld; m.nb = (K + 31) / 32; m.scale = scale_;
        m.rowinfo = (int*)(base + oR); m.x4 = (float*)(base + oX); m.Xt = (float*)(base + oXt);
        m.alpha = (float*)(sc + oA); m.Lt = (float*)(sc + oT); m.Zt = (float*)(sc + oZt);
        m.base = base; m.scratch = sc;
        dirty_ = true;
        return GPIS_OK;
    }
    // prediction side (what K4 reads) and training side in two allocations: the second one can go back to the pool once the
    // inverse exists (trim_scratch)
    size_t oR = 0, szR = sizeof(int) * ld;
    size_t oX = align_up(oR + szR, 256), szX = sizeof(float) * 4 * (size_t)N;
    size_t oXt = align_up(oX + szX, 256);
    const size_t total_p = align_up(oXt + szT, 256);
    size_t oL = 0, szL = sizeof(float) * (size_t)ld * ld;
    size_t oA = align_up(oL + szL, 256), szA = sizeof(float) * ld;
    size_t oY = align_up(oA + szA, 256), szY = sizeof(float) * ld;
    size_t oS = align_up(oY + szY, 256), szS = sizeof(float) * 2 * (size_t)N;
    size_t oG = align_up(oS + szS, 256), szG = sizeof(int) * (size_t)N;
    size_t oT = align_up(oG + szG, 256);
    size_t oZt = align_up(oT + szT, 256);                   // transposed tiles of the inverse (K3b's B operands)
    const size_t total_s = align_up(oZt + szT, 256);
    { void* b = m.base; keep_base = take(b, total_p); m.base = (char*)b; }
    { void* b = m.scratch; keep_sc = take(b, total_s); m.scratch = (char*)b; }
    free_model_mem(m);
    char* base = alloc_or(keep_base, total_p);
    if (!base) { if (keep_sc) pool_free(pool_, keep_sc); return GPIS_ERR_HIP; }
    char* sc = alloc_or(keep_sc, total_s);
    if (!sc) { pool_free(pool_, base); return GPIS_ERR_HIP; }
    std::memset(&m, 0, sizeof(ClusterModel));
    m.dim = dim_; m.N = N; m.ng = ng; m.K = K; m.ld = ld; m.nb = (K + 31) / 32; m.scale = scale_;
    m.rowinfo = (int*)(base + oR); m.x4 = (float*)(base + oX); m.Xt = (float*)(base + oXt);
    m.L = (float*)(sc + oL); m.alpha = (float*)(sc + oA); m.y = (float*)(sc + oY); m.sig = (float*)(sc + oS); m.gidx = (int*)(sc + oG);
    m.Lt = (float*)(sc + oT); m.Zt = (float*)(sc + oZt);
    m.base = base; m.scratch = sc;
    dirty_ = true;
    return GPIS_OK;
}

int OnGPISStore::sync_models(hipStream_t s) {
    if (!dirty_) return GPIS_OK;
    int n = (int)models_.size();
    if (n > d_models_cap_) {
        (void)hipFree(d_models_); d_models_ = nullptr;
        int cap = n + n / 2 + 64;
        GPIS_HIP(hipMalloc(&d_models_, sizeof(ClusterModel) * (size_t)cap));
        d_models_cap_ = cap;
    }
    if (n > 0) {
        GPIS_HIP(hipMemcpyAsync(d_models_, models_.data(), sizeof(ClusterModel) * (size_t)n, hipMemcpyHostToDevice, s));
        GPIS_HIP(hipStreamSynchronize(s));
    }
    dirty_ = false;
    return GPIS_OK;
}

int OnGPISStore::upload_points(const float* soa9, int n, hipStream_t s) {
    (void)train_finish();
    if (n > pts_.cap) {
        (void)hipFree(pts_.d); pts_.d = nullptr;
        int cap = n + n / 2 + 4096;
        GPIS_HIP(hipMalloc(&pts_.d, sizeof(float) * 9 * (size_t)cap));
        pts_.cap = cap;
    }
    pts_.n = n;
    for (int r = 0; r < 9 && n > 0; ++r)
        GPIS_HIP(hipMemcpyAsync(pts_.d + (size_t)r * pts_.cap, soa9 + (size_t)r * n, sizeof(float) * (size_t)n,
                                hipMemcpyHostToDevice, s));
    GPIS_HIP(hipStreamSynchronize(s));
    return GPIS_OK;
}

int OnGPISStore::gather_ranges(const int* cell_pts, int npts, const int* cranges, int nentries, const int* desc, int nclusters, int total_ids,
                               int* counts, hipStream_t s) {
    if (nclusters <= 0) return GPIS_OK;
    if (!cell_pts || !cranges || !desc || !counts || npts < 0 || nentries < 0 || total_ids < 0) return GPIS_ERR_ARG;
    (void)train_finish();
    const size_t need = (size_t)npts + 2 * (size_t)nentries + 10 * (size_t)nclusters;
    if ((int)need > cap_rg_) {
        (void)hipFree(d_rg_); d_rg_ = nullptr; cap_rg_ = 0;
        const int cap = (int)need * 3 / 2 + 4096;
        GPIS_HIP(hipMalloc(&d_rg_, sizeof(int) * (size_t)cap));
        cap_rg_ = cap;
    }
    if (total_ids > cap_ids_) {
        (void)hipFree(d_ids_); d_ids_ = nullptr; cap_ids_ = 0;
        const int cap = total_ids * 3 / 2 + 1024;
        GPIS_HIP(hipMalloc(&d_ids_, sizeof(int) * (size_t)cap));
        cap_ids_ = cap;
    }
    int* d_cp = d_rg_; int* d_cr = d_cp + npts; int* d_desc = d_cr + 2 * (size_t)nentries; int* d_cnt = d_desc + 8 * (size_t)nclusters;
    if (npts) GPIS_HIP(hipMemcpyAsync(d_cp, cell_pts, sizeof(int) * (size_t)npts, hipMemcpyHostToDevice, s));
    if (nentries) GPIS_HIP(hipMemcpyAsync(d_cr, cranges, sizeof(int) * 2 * (size_t)nentries, hipMemcpyHostToDevice, s));
    GPIS_HIP(hipMemcpyAsync(d_desc, desc, sizeof(int) * 8 * (size_t)nclusters, hipMemcpyHostToDevice, s));
    ongpis_launch_range_gather(d_desc, d_cr, d_cp, nclusters, pts_.d, pts_.cap, dim_, d_ids_, d_cnt, s);
    GPIS_HIP(hipGetLastError());
    GPIS_HIP(hipMemcpyAsync(counts, d_cnt, sizeof(int) * 2 * (size_t)nclusters, hipMemcpyDeviceToHost, s));
    GPIS_HIP(hipStreamSynchronize(s));
    dev_ids_ = total_ids;
    return GPIS_OK;
}

int OnGPISStore::train_batch(const std::vector<TrainJob>& jobs, const std::vector<int>& ids, hipStream_t s) { return train_batch_impl(jobs, &ids, s); }
int OnGPISStore::train_batch_dev(const std::vector<TrainJob>& jobs, hipStream_t s) { return train_batch_impl(jobs, nullptr, s); }

int OnGPISStore::train_batch_impl(const std::vector<TrainJob>& jobs, const std::vector<int>* ids, hipStream_t s) {
    int nj = (int)jobs.size();
    if (nj == 0) return GPIS_OK;
    (void)train_finish();
    const size_t nids = ids ? ids->size() : (size_t)dev_ids_;
    // Pass 1: validate every job before any model is touched (a refusal half-way through must not leave earlier
    // jobs pointing at recycled, untrained memory).
    for (int j = 0; j < nj; ++j) {
        const TrainJob& tj = jobs[j];
        if (tj.model < 0 || tj.model >= (int)models_.size() || !live_[tj.model] || tj.n <= 0 || tj.ng < 0 || tj.ng > tj.n ||
            tj.off < 0 || (size_t)tj.off + (size_t)tj.n > nids)
            return GPIS_ERR_ARG;
    }
    // Lazy inverse: the training side of a stale model (8 K^2 of its 10 K^2 bytes) is held until its inverse has been
    // computed.  A session that only ever calls update() would keep it for every cluster it has trained; bound that: the
    // models of THIS batch are about to be retrained (their old factor is moot), and when what the others hold exceeds
    // stale_bytes_limit they are inverted and trimmed now.
    if (!stale_list_.empty()) {
        for (int j = 0; j < nj; ++j) {
            const int Kj = jobs[j].n + dim_ * jobs[j].ng;
            const bool refused = Kj > ONGPIS_MAX_K || !ongpis_eval_fits(jobs[j].n, (int)align_up((size_t)Kj + 1, 32));   // (keeps its previous model, pass 2)
            if (!refused && jobs[j].model < (int)xstale_.size()) xstale_[jobs[j].model] = 0;
        }
        size_t held = 0;
        for (int slot : stale_list_)
            if (slot >= 0 && slot < (int)xstale_.size() && xstale_[slot] && live_[slot] && models_[slot].scratch)
                held += (size_t)8 * models_[slot].ld * models_[slot].ld;
        if (held > stale_bytes_limit) {
            // (duplicates in the list would be counted twice: the bound errs on the early side)
            const int erc = ensure_inverses(s);
            if (erc) return erc;
        }
    }
    // Pass 2: allocate.  A cluster this build cannot hold keeps its previous model (the rest of the batch is still
    // trained and the error reported); an allocation failure leaves that model UNTRAINED (base = nullptr: test()
    // treats the cell as having no GP) instead of half-initialised.
    int deferred_rc = GPIS_OK;
    std::vector<TrainJob> ok_jobs;
    ok_jobs.reserve(nj);
    for (int j = 0; j < nj; ++j) {
        const TrainJob& tj = jobs[j];
        int K = tj.n + dim_ * tj.ng;
        if (K > ONGPIS_MAX_K || !ongpis_eval_fits(tj.n, (int)align_up((size_t)K + 1, 32))) {
            fprintf(stderr, "[gpismap_amd] cluster with N=%d, K=%d exceeds what the prediction kernel can stage in LDS (4 ld + 16 N bytes + three B blocks <= 158 KB; K <= %d): previous model kept\n", tj.n, K, ONGPIS_MAX_K);
            if (!deferred_rc) deferred_rc = GPIS_ERR_LIMIT;
            continue;
        }
        const bool fused = use_fused && K <= ONGPIS_FUSED_MAX_K && tj.n <= 256;
        int rc = alloc_model(tj.model, tj.n, tj.ng, fused ? (keep_factor ? kAllocLeanFactor : kAllocPredictOnly) : kAllocFull);
        if (rc) {
            ClusterModel& m = models_[tj.model];
            std::memset(&m, 0, sizeof(ClusterModel));   // untrained; slot stays live
            dirty_ = true;
            if (!deferred_rc) deferred_rc = rc;
            continue;
        }
        ok_jobs.push_back(tj);
    }
    if (ok_jobs.empty()) { int rc0 = sync_models(s); return deferred_rc ? deferred_rc : rc0; }
    return train_allocated(ok_jobs, ids, s, deferred_rc);
}

// K3b work lists for the jobs [jbeg, jend) of a job table (4 ints per job: model, offset, N, ng): one entry per (job, block
// column); columns longer than kLongCol rows first (a workgroup of 8 pipelined wavefronts each), then the first of every
// kMidWaves adjacent one-wavefront columns, then the short columns of the clusters whose columns are all short.
// XCD-aware order of the long and the one-wavefront lists: workgroup ids are dealt round-robin to the 8 XCDs, every XCD
// has its own L2, and the block columns of ONE cluster read the same Lt tiles -- so all columns of a cluster go to one
// XCD (clusters dealt to the XCDs by accumulated work, the list interleaved so that entry 8 k + x belongs to XCD x,
// short lists padded with (-1, 0) entries whose workgroups exit at once).  Measured on the synthetic frames: L2 misses
// of the long-column kernel -62 %.
void OnGPISStore::build_inverse_work(const std::vector<int>& tab, int jbeg, int jend, int kLongCol, std::vector<int>& work,
                                     int& off, int& nlong, int& nmid, int& nshort) const {
    const int kRegCol = ongpis_inverse_short_rows();   // columns this short keep their transposed tiles in registers
    const int kRegWaves = ongpis_inverse_short_waves(), kMidWaves = ongpis_inverse_mid_waves();
    auto interleave8 = [](std::vector<int> (&sub)[8], std::vector<int>& out) {
        size_t mx = 0;
        for (int x = 0; x < 8; ++x) mx = std::max(mx, sub[x].size() / 2);
        for (size_t k = 0; k < mx; ++k)
            for (int x = 0; x < 8; ++x) {
                if (2 * k < sub[x].size()) { out.push_back(sub[x][2 * k]); out.push_back(sub[x][2 * k + 1]); }
                else { out.push_back(-1); out.push_back(0); }
            }
    };
    std::vector<int> wlong, wmid, wshort;
    std::vector<int> slong[8], smid[8];
    double wl[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = jbeg; j < jend; ++j) {
        const int nbj = (tab[4 * j + 2] + dim_ * tab[4 * j + 3] + 31) / 32;
        // register path only for clusters whose columns are ALL short (a third launch per group would serialise
        // behind the other two for nothing: the short columns of a large cluster are a few percent of its work)
        const bool reg_cluster = nbj <= kRegCol;
        std::vector<int> el, em;
        double cl = 0.0, cm = 0.0;
        for (int c = 0; c < nbj; ++c) {
            const double cw = (double)(nbj - c) * (nbj - c);
            if (!reg_cluster) {
                if (nbj - c > kLongCol) { el.push_back(j); el.push_back(c); cl += cw; }
                else {
                    if ((c - std::max(0, nbj - kLongCol)) % kMidWaves == 0) { em.push_back(j); em.push_back(c); }   // first of kMidWaves columns
                    cm += cw;
                }
            } else if ((c - std::max(0, nbj - kRegCol)) % kRegWaves == 0) {   // one entry per kRegWaves adjacent short columns
                wshort.push_back(j); wshort.push_back(c);
            }
        }
        if (!el.empty()) { int x = 0; for (int i = 1; i < 8; ++i) if (wl[i] < wl[x]) x = i; slong[x].insert(slong[x].end(), el.begin(), el.end()); wl[x] += cl; }
        if (!em.empty()) { int x = 0; for (int i = 1; i < 8; ++i) if (wm[i] < wm[x]) x = i; smid[x].insert(smid[x].end(), em.begin(), em.end()); wm[x] += cm; }
    }
    interleave8(slong, wlong);
    interleave8(smid, wmid);
    off = (int)work.size(); nlong = (int)wlong.size() / 2; nmid = (int)wmid.size() / 2; nshort = (int)wshort.size() / 2;
    work.insert(work.end(), wlong.begin(), wlong.end());
    work.insert(work.end(), wmid.begin(), wmid.end());
    work.insert(work.end(), wshort.begin(), wshort.end());
}

// The explicit inverses (K3b) of every model whose factor is newer than its X: with lazy_inverse, training leaves them to
// the first use of the models -- prediction, packing -- so that a cluster retrained in several consecutive updates is
// inverted ONCE, when somebody asks.  No-op when nothing is stale.
int OnGPISStore::ensure_inverses(hipStream_t s) {
    int frc = train_finish();
    if (frc) return frc;
    if (stale_list_.empty()) return GPIS_OK;
    std::vector<int> slots;
    for (int slot : stale_list_) {
        if (slot < 0 || slot >= (int)xstale_.size() || slot >= (int)models_.size() || !xstale_[slot]) continue;   // (released, or listed twice)
        xstale_[slot] = 0;
        if (live_[slot] && models_[slot].base && models_[slot].Zt) slots.push_back(slot);    // (retrained on chip since: its X is there)
    }
    stale_list_.clear();
    const int nj = (int)slots.size();
    if (nj == 0) return GPIS_OK;
    std::stable_sort(slots.begin(), slots.end(), [&](int a, int b) { return models_[a].K > models_[b].K; });   // largest first
    std::vector<int> tab((size_t)4 * nj);
    for (int j = 0; j < nj; ++j) { const ClusterModel& m = models_[slots[j]]; tab[4 * j] = slots[j]; tab[4 * j + 1] = 0; tab[4 * j + 2] = m.N; tab[4 * j + 3] = m.ng; }
    int rc = sync_models(s);
    if (rc) return rc;
    std::vector<int> work;
    int off = 0, nlong = 0, nmid = 0, nshort = 0;
    build_inverse_work(tab, 0, nj, 24, work, off, nlong, nmid, nshort);
    if (4 * nj > cap_jobs_) {
        (void)hipFree(d_jobs_); d_jobs_ = nullptr;
        int cap = 4 * nj * 3 / 2 + 1024;
        GPIS_HIP(hipMalloc(&d_jobs_, sizeof(int) * (size_t)cap));
        cap_jobs_ = cap;
    }
    if ((int)work.size() > cap_work_) {
        (void)hipFree(d_work_); d_work_ = nullptr;
        int cap = (int)work.size() * 3 / 2 + 1024;
        GPIS_HIP(hipMalloc(&d_work_, sizeof(int) * (size_t)cap));
        cap_work_ = cap;
    }
    if (!d_err_) GPIS_HIP(hipMalloc(&d_err_, sizeof(int) * 4));
    if (!h_err_) GPIS_HIP(hipHostMalloc((void**)&h_err_, sizeof(int) * 4));
    const int ctl[4] = {0, 0, wait_limit_ticks, 0};
    GPIS_HIP(hipMemcpyAsync(d_err_, ctl, sizeof(ctl), hipMemcpyHostToDevice, s));
    GPIS_HIP(hipMemcpyAsync(d_jobs_, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice, s));
    GPIS_HIP(hipMemcpyAsync(d_work_, work.data(), sizeof(int) * work.size(), hipMemcpyHostToDevice, s));
    if (profile) {
        if (!ev0_) { GPIS_HIP(hipEventCreate(&ev0_)); GPIS_HIP(hipEventCreate(&ev1_)); }
        GPIS_HIP(hipEventRecord(ev0_, s));
    }
    // The three column classes are independent launches (a block column of X depends on no other): side by side on two forked
    // streams instead of one after the other -- at the first test() after an update of a few dozen clusters each of them is a
    // handful of latency-bound workgroups (data/3D: 0.32 + 0.32 ms per frame in a row).
    if ((nlong > 0) + (nmid > 0) + (nshort > 0) > 1) {
        if (!s2_) {
            if (int src = ongpis_make_train_stream(&s2_, cu_reserve_)) return src;
            if (int src = ongpis_make_train_stream(&s3_, cu_reserve_)) return src;
        }
        if (!evf_) {
            GPIS_HIP(hipEventCreateWithFlags(&evf_, hipEventDisableTiming));
            GPIS_HIP(hipEventCreateWithFlags(&evj_, hipEventDisableTiming));
            GPIS_HIP(hipEventCreateWithFlags(&evj3_, hipEventDisableTiming));
        }
        GPIS_HIP(hipEventRecord(evf_, s));
        ongpis_launch_inverse(d_models_, d_jobs_, d_work_ + off, nlong, 0, 0, d_err_, s);
        if (nmid > 0) {
            GPIS_HIP(hipStreamWaitEvent(s2_, evf_, 0));
            ongpis_launch_inverse(d_models_, d_jobs_, d_work_ + off + 2 * nlong, 0, nmid, 0, d_err_, s2_);
            GPIS_HIP(hipEventRecord(evj_, s2_)); GPIS_HIP(hipStreamWaitEvent(s, evj_, 0));
        }
        if (nshort > 0) {
            hipStream_t ss = (nlong > 0 || nmid == 0) ? s3_ : s;      // (two classes only: one of them stays on the caller's stream)
            if (ss != s) GPIS_HIP(hipStreamWaitEvent(ss, evf_, 0));
            ongpis_launch_inverse(d_models_, d_jobs_, d_work_ + off + 2 * (nlong + nmid), 0, 0, nshort, d_err_, ss);
            if (ss != s) { GPIS_HIP(hipEventRecord(evj3_, ss)); GPIS_HIP(hipStreamWaitEvent(s, evj3_, 0)); }
        }
    } else ongpis_launch_inverse(d_models_, d_jobs_, d_work_ + off, nlong, nmid, nshort, d_err_, s);
    GPIS_HIP(hipGetLastError());
    if (profile) GPIS_HIP(hipEventRecord(ev1_, s));
    for (int i = 0; i < 4; ++i) h_err_[i] = 0;
    GPIS_HIP(hipMemcpyAsync(h_err_, d_err_, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
    GPIS_HIP(hipStreamSynchronize(s));
    if (profile) GPIS_HIP(hipEventElapsedTime(&last_inverse_ms, ev0_, ev1_));
    last_inverse_jobs = nj;
    if (h_err_[0]) {
        fprintf(stderr, "[gpismap_amd] inverse kernels reported error word 0x%x: the %d models of this pass are dropped\n", h_err_[0], nj);
        for (int slot : slots) {
            ClusterModel& m = models_[slot];
            free_model_mem(m);
            std::memset(&m, 0, sizeof(ClusterModel));
            dropped_.push_back(slot);
        }
        dirty_ = true;
        (void)sync_models(s);
        return GPIS_ERR_STATE;
    }
    if (trim_scratch) trim_models(slots);      // the factors have served: their memory goes back to the pool
    return GPIS_OK;
}

// K6 + kernel build + K3 for jobs whose models are allocated.  An error exit BEFORE the batch is completely enqueued (a HIP
// call, a refused fused launch) would leave models that are allocated -- base != nullptr, so a valid GP to the cluster table --
// but never trained: whatever was launched is drained and the batch's models are marked untrained, like a dropped batch.
int OnGPISStore::train_allocated(const std::vector<TrainJob>& jobs, const std::vector<int>* ids, hipStream_t s, int deferred_rc) {
    enqueued_ = false;
    const int rc = train_enqueue(jobs, ids, s, deferred_rc);
    if (enqueued_) return rc;
    (void)hipStreamSynchronize(s);
    if (s2_) (void)hipStreamSynchronize(s2_);
    if (s3_) (void)hipStreamSynchronize(s3_);
    coop_give_back(coop_dev_, coop_held_); coop_held_ = 0;
    fprintf(stderr, "[gpismap_amd] training batch not enqueued (%d): its %d models are dropped\n", rc, (int)jobs.size());
    for (const TrainJob& tj : jobs) {
        ClusterModel& m = models_[tj.model];
        free_model_mem(m);
        std::memset(&m, 0, sizeof(ClusterModel));
        if (tj.model < (int)xstale_.size()) xstale_[tj.model] = 0;
    }
    dirty_ = true;
    (void)sync_models(s);
    return rc ? rc : GPIS_ERR_STATE;
}

int OnGPISStore::train_enqueue(const std::vector<TrainJob>& jobs, const std::vector<int>* ids, hipStream_t s, int deferred_rc) {
    const int nj = (int)jobs.size();
    std::vector<int> tab((size_t)4 * nj);
    // largest clusters first: one workgroup per cluster and K^3 work, so the big factorisations must not start last
    std::vector<int> ord(nj);
    std::iota(ord.begin(), ord.end(), 0);
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) {
        return jobs[a].n + dim_ * jobs[a].ng > jobs[b].n + dim_ * jobs[b].ng; });
    for (int j = 0; j < nj; ++j) {
        const TrainJob& tj = jobs[ord[j]];
        tab[4 * j] = tj.model; tab[4 * j + 1] = tj.off; tab[4 * j + 2] = tj.n; tab[4 * j + 3] = tj.ng;
    }
    int rc = sync_models(s);
    if (rc) return rc;
    if (ids && (int)ids->size() > cap_ids_) {
        (void)hipFree(d_ids_); d_ids_ = nullptr;
        int cap = (int)ids->size() * 3 / 2 + 1024;
        GPIS_HIP(hipMalloc(&d_ids_, sizeof(int) * (size_t)cap));
        cap_ids_ = cap;
    }
    if (4 * nj > cap_jobs_) {
        (void)hipFree(d_jobs_); d_jobs_ = nullptr;
        int cap = 4 * nj * 3 / 2 + 1024;
        GPIS_HIP(hipMalloc(&d_jobs_, sizeof(int) * (size_t)cap));
        cap_jobs_ = cap;
    }
    last_train_flops = 0.0; last_train_bytes = 0.0; last_train_jobs = nj; last_train_maxK = 0;
    for (int j = 0; j < nj; ++j) {
        const double N = tab[4 * j + 2], K = N + (double)dim_ * tab[4 * j + 3];
        last_train_flops += K * K * K / 3.0 + 2.0 * K * K;
        last_train_bytes += 36.0 * N + 4.0 * (K * (K + 1) / 2.0 + K);
        last_train_maxK = std::max(last_train_maxK, (int)K);
    }
    // Three size groups, each a chain "factorise -> invert" on its own stream so that the chains overlap:
    //   group 0: the largest clusters, several cooperating workgroups each (jobs [0, ncoop))
    //   group 1: one 8-wave workgroup per cluster (jobs [ncoop, n0))
    //   group 2: K <= 256, one wavefront per cluster (jobs [n0, nj))
    int n0 = 0;
    while (n0 < nj && tab[4 * n0 + 2] + dim_ * tab[4 * n0 + 3] > 256) ++n0;
    // cooperative group: G workgroups ~ nb^2 (work nb^3 over a critical path of nb steps), one workgroup per CU, at most
    // kCoopMaxWG in the launch so that all of them are resident at once
    // Measured on the synthetic frames (instrumented builds read these from the environment: ongpis_store_instr.inc): below ~32 block rows one
    // workgroup per cluster is faster; few workgroups per cluster (so that MANY clusters fit the launch) beat many workgroups
    // for few clusters -- a cooperative cluster is bound by its serial path, and every large cluster left to the
    // one-workgroup kernel costs more than a small G costs the largest ones.
    int coop_dev = 0;
    (void)hipGetDevice(&coop_dev);
    const int coop_share = coop_take_all(coop_dev);   // free cooperative workgroups of this device (per-device budget, see above)
    int kCoopMinNb = 32, kCoopMaxWG = coop_share;
    if (cu_reserve_ > 0) {      // masked streams: fewer CUs can hold cooperative workgroups at once
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, coop_dev) == hipSuccess && ncu > cu_reserve_)
            kCoopMaxWG = std::min(kCoopMaxWG, std::max(2, (ncu - cu_reserve_) - (ncu - cu_reserve_) / 16));
    }
    int kCoopGDiv = 900, kCoopGMax = 6;
    // Round 6: the numbers above are the THROUGHPUT schedule (a frame of several hundred large clusters fills the device with
    // one workgroup each; tuned on the synthetic frames).  A batch that cannot fill the device is bound by the LATENCY of its
    // largest clusters instead -- the reference's own sequence trains 20-60 clusters per frame -- and there more cooperating
    // workgroups per cluster win: one K = 1190 cluster 1.73 ms with 2 workgroups, 1.37 with 3, 1.09 with 6, 1.02 with 13; eight
    // K = 680 clusters 0.71 ms alone, 0.58 as pairs; on data/3D the K3 chain per frame 3.39 -> 2.27 ms (median of 39 frames,
    // sum 134 -> 99 ms; tools/ab sweep of K3_MINNB / K3_GDIV / K3_GMAX in an instrumented build, profiles/r06_k3_coop_sweep.txt).
    // G ~ nb^2 / 112 was faster still on single clusters but produced 13-17 ms outlier frames (sixteen workgroups of one cluster
    // waiting for residency beside the one-workgroup kernel's clusters).
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, coop_dev) != hipSuccess) ncu = 256;
        const int avail = std::max(8, ncu - std::max(0, cu_reserve_));
        if (n0 <= avail / 4) { kCoopMinNb = 16; kCoopGDiv = 225; kCoopGMax = 16; }
        else if (n0 <= avail / 2) { kCoopGDiv = 450; kCoopGMax = 8; }
    }
    int kLongCol = 24;   // K3b: columns with more block rows than this take a workgroup of 8 pipelined wavefronts
#ifdef GPIS_INSTRUMENT
#include "ongpis_store_instr.inc"   // schedule knobs from the environment (tuning sweeps only)
#endif
    int ncoop = 0;
    std::vector<int> cwork;
    {
        // the G workgroups of a cluster on ONE XCD (they read each other's rows through that XCD's L2): clusters dealt to
        // the XCD with the fewest workgroups so far, the list interleaved so that entry 8 k + x runs on XCD x
        // (workgroup ids go round-robin over the XCDs), padded with job = -1 entries whose workgroups exit at once
        std::vector<int> sub[8];
        int total = 0;
        // first the schedule's G per cluster, as far as the budget of resident workgroups goes ...
        std::vector<int> Gs;
        for (int j = 0; j < n0; ++j) {
            const int nbj = (tab[4 * j + 2] + dim_ * tab[4 * j + 3] + 31) / 32;
            if (nbj < kCoopMinNb) break;
            const int G = std::min(kCoopGMax, std::max(2, (nbj * nbj + kCoopGDiv / 2) / kCoopGDiv));
            if (total + G > kCoopMaxWG) break;
            Gs.push_back(G);
            total += G;
        }
        // ... then, in a batch that leaves the device room (the latency regimes above), what is left of the budget goes to the
        // largest clusters until each block row has up to four wavefronts (8 G >= 4 rows: tools/k3_bench.py, one K = 1190 cluster
        // 0.84 ms with G = 6, 0.78 with 10; K = 680 0.44 / 0.36)
        if (kCoopGDiv < 900) {
            // (a cluster's workgroups sit on ONE XCD and must all be resident: 32 CUs, or what the training streams' CU mask leaves of
            // them -- bit i of the mask is CU i / 8 of XCD i % 8, so a reserve of 64 leaves 24 per XCD)
            int ncu = 0;
            if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, coop_dev) != hipSuccess) ncu = 256;
            const int per_xcd = std::max(2, (ncu - std::max(0, cu_reserve_)) / 8);
            for (size_t j = 0; j < Gs.size() && total < kCoopMaxWG; ++j) {
                const int nbj = (tab[4 * j + 2] + dim_ * tab[4 * j + 3] + 31) / 32;
                const int want = std::min(per_xcd, std::max(Gs[j], (nbj + 1) / 2));      // (up to four wavefronts per block row)
                const int add = std::max(0, std::min(want - Gs[j], kCoopMaxWG - total));
                Gs[j] += add; total += add;
            }
        }
        for (size_t j = 0; j < Gs.size(); ++j) {
            const int G = Gs[j];
            int x = 0;
            for (int i = 1; i < 8; ++i) if (sub[i].size() < sub[x].size()) x = i;
            for (int g = 0; g < G; ++g) { sub[x].push_back((int)j); sub[x].push_back(g); sub[x].push_back(G); }
            ncoop = (int)j + 1;
        }
        coop_give_back(coop_dev, coop_share - total);   // keep what the launch needs until the batch is joined
        coop_dev_ = coop_dev; coop_held_ = total;
        size_t mx = 0;
        for (int x = 0; x < 8; ++x) mx = std::max(mx, sub[x].size() / 3);
        for (size_t k = 0; k < mx && ncoop > 0; ++k)
            for (int x = 0; x < 8; ++x) {
                if (3 * k < sub[x].size()) { cwork.push_back(sub[x][3 * k]); cwork.push_back(sub[x][3 * k + 1]); cwork.push_back(sub[x][3 * k + 2]); }
                else { cwork.push_back(-1); cwork.push_back(0); cwork.push_back(1); }
            }
    }
#ifdef GPIS_INSTRUMENT
    {   // size profile of the batch: block rows per cluster, by group
        int hist[3][12] = {};
        for (int j = 0; j < nj; ++j) {
            const int nbj = (tab[4 * j + 2] + dim_ * tab[4 * j + 3] + 31) / 32;
            hist[j < ncoop ? 0 : (j < n0 ? 1 : 2)][std::min(11, nbj / 8)]++;
        }
        for (int g = 0; g < 3; ++g) {
            fprintf(stderr, "[train] group %d (%s):", g, g == 0 ? "cooperative" : (g == 1 ? "8 waves" : "1 wave"));
            for (int b = 0; b < 12; ++b) if (hist[g][b]) fprintf(stderr, "  nb %d-%d: %d", 8 * b, 8 * b + 7, hist[g][b]);
            fprintf(stderr, "   coop workgroups %d\n", (int)cwork.size() / 3);
        }
    }
#endif
    // K3b work lists per group: one entry per (job, block column); columns longer than kLongCol rows first (a workgroup
    // of 8 pipelined wavefronts each), the rest one wavefront per column
    // clusters of at most ONGPIS_FUSED_MAX_K rows: one fused on-chip launch per size tier (ongpis_fused.hip) instead of
    // group 2's gather / build / factorise / invert chain
    const bool fused2 = use_fused;
    const int nsep = fused2 ? n0 : nj;         // jobs that take the separate kernels
    int n1 = n0;                               // fused tier boundary: [n0, n1) up to 8 block rows, [n1, nj) up to 5
    while (fused2 && n1 < nj && (tab[4 * n1 + 2] + dim_ * tab[4 * n1 + 3] + 31) / 32 > 5) ++n1;
    const int gbeg[4] = {0, ncoop, n0, nsep};
    std::vector<int> work;
    int wl_off[3] = {0, 0, 0}, wl_long[3] = {0, 0, 0}, wl_mid[3] = {0, 0, 0}, wl_short[3] = {0, 0, 0};
    const bool lazy = lazy_inverse;     // K3b deferred to the first use of the models (ensure_inverses)
    if (!lazy)
        for (int grp = 0; grp < 3; ++grp) build_inverse_work(tab, gbeg[grp], gbeg[grp + 1], kLongCol, work, wl_off[grp], wl_long[grp], wl_mid[grp], wl_short[grp]);
    if ((int)work.size() > cap_work_) {
        (void)hipFree(d_work_); d_work_ = nullptr;
        int cap = (int)work.size() * 3 / 2 + 1024;
        GPIS_HIP(hipMalloc(&d_work_, sizeof(int) * (size_t)cap));
        cap_work_ = cap;
    }
    if (ncoop > 0) {
        // behind the work list: three words per cooperative cluster (abort / workgroups done / offset of its flag words), then
        // 2 x rows progress words per cluster; everything but the offsets starts at zero
        coop_hdr_.assign(3 * (size_t)ncoop, 0);
        size_t flags = 0;
        for (int j = 0; j < ncoop; ++j) {
            const int Kj = tab[4 * j + 2] + dim_ * tab[4 * j + 3];
            coop_hdr_[3 * (size_t)j + 2] = (int)(3 * (size_t)ncoop + flags);
            flags += 2 * (size_t)((Kj + 32) / 32);
        }
        const size_t need = cwork.size() + 3 * (size_t)ncoop + flags;
        if ((int)need > cap_cwork_) {
            (void)hipFree(d_cwork_); d_cwork_ = nullptr;
            int cap = (int)need * 2 + 1024;
            GPIS_HIP(hipMalloc(&d_cwork_, sizeof(int) * (size_t)cap));
            cap_cwork_ = cap;
        }
        GPIS_HIP(hipMemcpyAsync(d_cwork_, cwork.data(), sizeof(int) * cwork.size(), hipMemcpyHostToDevice, s));
        GPIS_HIP(hipMemsetAsync(d_cwork_ + cwork.size(), 0, sizeof(int) * (3 * (size_t)ncoop + flags), s));
        GPIS_HIP(hipMemcpyAsync(d_cwork_ + cwork.size(), coop_hdr_.data(), sizeof(int) * coop_hdr_.size(), hipMemcpyHostToDevice, s));
    }
    if (ids) GPIS_HIP(hipMemcpyAsync(d_ids_, ids->data(), sizeof(int) * ids->size(), hipMemcpyHostToDevice, s));
    GPIS_HIP(hipMemcpyAsync(d_jobs_, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice, s));
    if (!work.empty()) GPIS_HIP(hipMemcpyAsync(d_work_, work.data(), sizeof(int) * work.size(), hipMemcpyHostToDevice, s));
    if (lazy) for (int j = 0; j < nsep; ++j) mark_stale(tab[4 * j]);
    if (!s2_) {   // side streams / events of the size groups (created once, outside the timed interval)
        // (lowest priority: in the pipelined map update the ObsGP queries of the next frame run beside these and must not wait)
        if (int src = ongpis_make_train_stream(&s2_, cu_reserve_)) return src;
        if (int src = ongpis_make_train_stream(&s3_, cu_reserve_)) return src;
    }
    if (!evf_) {
        GPIS_HIP(hipEventCreateWithFlags(&evf_, hipEventDisableTiming));
        GPIS_HIP(hipEventCreateWithFlags(&evj_, hipEventDisableTiming));
        GPIS_HIP(hipEventCreateWithFlags(&evj3_, hipEventDisableTiming));
    }
    if (profile) {
        if (!ev0_) { GPIS_HIP(hipEventCreate(&ev0_)); GPIS_HIP(hipEventCreate(&ev1_)); }
        GPIS_HIP(hipEventRecord(ev0_, s));
    }
    if (!d_err_) GPIS_HIP(hipMalloc(&d_err_, sizeof(int) * 4));
    {
        const int ctl[4] = {0, debug_inject, wait_limit_ticks, 0};
        GPIS_HIP(hipMemcpyAsync(d_err_, ctl, sizeof(ctl), hipMemcpyHostToDevice, s));
    }
    // fork: groups 1 and 2 on side streams, group 0 (or the first non-empty group) on the caller's stream.  Every group
    // gathers and builds its OWN kernel matrices at the head of its chain (one launch over all clusters kept the largest
    // clusters waiting for the kernel matrices of all the others: 1.3 ms in front of the cooperative factorisation).
    GPIS_HIP(hipEventRecord(evf_, s));
    hipStream_t gs[3] = {s, s3_, s2_};
    bool fused_launched = false;
    if (fused2 && nj > n0) {
        GPIS_HIP(hipStreamWaitEvent(s2_, evf_, 0));
        FusedTrainArgs fa;
        fa.models = d_models_; fa.ids = d_ids_; fa.pts = pts_.d; fa.cap = pts_.cap; fa.err = d_err_;
        if (n1 > n0) {
            fa.jobs = d_jobs_ + 4 * n0;
            const int nbmax = (tab[4 * n0 + 2] + dim_ * tab[4 * n0 + 3] + 31) / 32;
            int frc = ongpis_launch_train_fused(fa, n1 - n0, nbmax, s2_);
            if (frc) return frc;
        }
        if (nj > n1) {
            fa.jobs = d_jobs_ + 4 * n1;
            const int nbmax = (tab[4 * n1 + 2] + dim_ * tab[4 * n1 + 3] + 31) / 32;
            int frc = ongpis_launch_train_fused(fa, nj - n1, nbmax, s2_);
            if (frc) return frc;
        }
        fused_launched = true;   // (joined after the group loop: the caller's stream must not wait for the small clusters before it starts the largest ones)
    }
    for (int grp = 0; grp < 3; ++grp) {
        const int nbeg = gbeg[grp], ncnt = gbeg[grp + 1] - gbeg[grp];
        if (ncnt <= 0) continue;
        if (grp > 0) GPIS_HIP(hipStreamWaitEvent(gs[grp], evf_, 0));
        ongpis_launch_gather(d_models_, d_jobs_ + 4 * nbeg, ncnt, d_ids_, pts_.d, pts_.cap, gs[grp]);
        ongpis_launch_buildK(d_models_, d_jobs_ + 4 * nbeg, ncnt, gs[grp]);
        if (grp == 0) ongpis_launch_chol_flow(d_models_, d_jobs_, d_cwork_, (int)cwork.size() / 3, d_cwork_ + cwork.size(), d_err_, gs[0]);
#ifdef GPIS_EXPERIMENTS
        else if (grp == 1 && getenv("GPIS_ASYNC_CHOL") && atoi(getenv("GPIS_ASYNC_CHOL"))) ongpis_launch_chol_async(d_models_, d_jobs_ + 4 * nbeg, ncnt, d_err_, gs[grp]);
#endif
        else ongpis_launch_chol(d_models_, d_jobs_ + 4 * nbeg, ncnt, grp == 2 ? 1 : 0, gs[grp]);
        // K3b of the group: X = L^-1, re-tiled for K4.  The job index in the work list is the global one.
        if (!lazy) ongpis_launch_inverse(d_models_, d_jobs_, d_work_ + wl_off[grp], wl_long[grp], wl_mid[grp], wl_short[grp], d_err_, gs[grp]);
        if (grp == 1) { GPIS_HIP(hipEventRecord(evj3_, s3_)); GPIS_HIP(hipStreamWaitEvent(s, evj3_, 0)); }
        if (grp == 2) { GPIS_HIP(hipEventRecord(evj_, s2_)); GPIS_HIP(hipStreamWaitEvent(s, evj_, 0)); }
    }
    if (fused_launched) { GPIS_HIP(hipEventRecord(evj_, s2_)); GPIS_HIP(hipStreamWaitEvent(s, evj_, 0)); }
    GPIS_HIP(hipGetLastError());
    if (profile) GPIS_HIP(hipEventRecord(ev1_, s));
    if (!h_err_) GPIS_HIP(hipHostMalloc((void**)&h_err_, sizeof(int) * 4));
    for (int i = 0; i < 4; ++i) h_err_[i] = 0;
    GPIS_HIP(hipMemcpyAsync(h_err_, d_err_, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
    pend_active_ = true; pend_profile_ = profile; pend_stream_ = s; pend_lazy_ = lazy;
    enqueued_ = true;
    pend_models_.resize(nj);
    for (int j = 0; j < nj; ++j) pend_models_[j] = tab[4 * j];
    if (defer_finish) return deferred_rc;
    const int frc = train_finish();
    return frc ? frc : deferred_rc;
}

// Join the training batch in flight (no-op without one).
int OnGPISStore::train_finish() {
    if (!pend_active_) return GPIS_OK;
    pend_active_ = false;
    const hipError_t se = hipStreamSynchronize(pend_stream_);
    coop_give_back(coop_dev_, coop_held_); coop_held_ = 0;
    GPIS_HIP(se);
    if (pend_profile_) GPIS_HIP(hipEventElapsedTime(&last_train_ms, ev0_, ev1_));
    if (h_err_[0]) {
        // bit 0: a job the fused kernel cannot hold; bit 1: a wait of the cooperative factorisation expired; bit 2: a row
        // wait of the inverse expired.  The factors of this batch may be incomplete: every model of the batch is marked
        // untrained (test() treats the cells as having no GP) and the caller gets GPIS_ERR_STATE.
        fprintf(stderr, "[gpismap_amd] training kernels reported error word 0x%x: the %d models of this batch are dropped\n", h_err_[0], (int)pend_models_.size());
        for (int slot : pend_models_) {
            if (slot < 0 || slot >= (int)models_.size() || !live_[slot]) continue;
            ClusterModel& m = models_[slot];
            free_model_mem(m);
            std::memset(&m, 0, sizeof(ClusterModel));
            if (slot < (int)xstale_.size()) xstale_[slot] = 0;     // (no factor left that an inverse could be computed from)
            dropped_.push_back(slot);
        }
        dirty_ = true;
        (void)sync_models(pend_stream_);
        return GPIS_ERR_STATE;
    }
    // (eager inverse: X exists when the batch is through.  The mode the batch was ENQUEUED with decides, not the current
    // flag: a lazy batch has no X yet whatever the caller switched to since.)
    if (trim_scratch && !pend_lazy_) trim_models(pend_models_);
    return GPIS_OK;
}

int OnGPISStore::kernel_matrix(const float* x, const int* gidx, const float* sigx, const float* sigg, int N, float* K_out, hipStream_t s) {
    if (!x || !gidx || !sigx || !sigg || !K_out || N < 1) return GPIS_ERR_ARG;
    (void)train_finish();
    int ng = 0;
    for (int k = 0; k < N; ++k) {
        if (gidx[k] >= 0) { if (gidx[k] != ng) return GPIS_ERR_ARG; ++ng; }
    }
    const int slot = new_slot();
    int rc = alloc_model(slot, N, ng, kAllocFull);
    if (rc) { release_slot(slot); return rc; }
    const ClusterModel m = models_[slot];
    std::vector<float> x4((size_t)4 * N, 0.f), sig((size_t)2 * N), y((size_t)m.ld, 0.f);
    for (int k = 0; k < N; ++k) {
        for (int d = 0; d < dim_; ++d) x4[(size_t)4 * k + d] = x[(size_t)dim_ * k + d];
        sig[k] = sigx[k]; sig[N + k] = sigg[k];
    }
    rc = sync_models(s);
    if (rc) { release_slot(slot); return rc; }
    const int job[4] = {slot, 0, N, ng};
    int* d_job = nullptr;
    GPIS_HIP(hipMalloc(&d_job, sizeof(job)));
    GPIS_HIP(hipMemcpyAsync(d_job, job, sizeof(job), hipMemcpyHostToDevice, s));
    GPIS_HIP(hipMemcpyAsync(m.x4, x4.data(), sizeof(float) * x4.size(), hipMemcpyHostToDevice, s));
    GPIS_HIP(hipMemcpyAsync(m.sig, sig.data(), sizeof(float) * sig.size(), hipMemcpyHostToDevice, s));
    GPIS_HIP(hipMemcpyAsync(m.gidx, gidx, sizeof(int) * (size_t)N, hipMemcpyHostToDevice, s));
    GPIS_HIP(hipMemcpyAsync(m.y, y.data(), sizeof(float) * y.size(), hipMemcpyHostToDevice, s));
    // the row table the gather kernel writes (ongpis_train.hip): point | (gradient component + 1) << 28
    std::vector<int> rowinfo((size_t)m.ld, (int)(0xFu << 28));
    for (int k = 0; k < N; ++k) {
        rowinfo[k] = k;
        if (gidx[k] >= 0) for (int c = 0; c < dim_; ++c) rowinfo[(size_t)N + (size_t)c * ng + gidx[k]] = k | ((c + 1) << 28);
    }
    GPIS_HIP(hipMemcpyAsync(m.rowinfo, rowinfo.data(), sizeof(int) * rowinfo.size(), hipMemcpyHostToDevice, s));
    ongpis_launch_buildK(d_models_, d_job, 1, s);
    GPIS_HIP(hipGetLastError());
    std::vector<float> L((size_t)m.ld * m.ld);
    GPIS_HIP(hipMemcpyAsync(L.data(), m.L, sizeof(float) * L.size(), hipMemcpyDeviceToHost, s));
    GPIS_HIP(hipStreamSynchronize(s));
    (void)hipFree(d_job);
    const int K = m.K;
    for (int c = 0; c < K; ++c)
        for (int r = 0; r < K; ++r) K_out[r + (size_t)c * K] = (r >= c) ? L[r + (size_t)c * m.ld] : 0.f;
    release_slot(slot);
    return GPIS_OK;
}

size_t OnGPISStore::packed_bytes(const int* slots, int n) const {
    size_t mx = 0;
    for (int i = 0; i < n; ++i) {
        const ClusterModel* m = model(slots[i]);
        if (m && m->base) mx = std::max(mx, packed_model_bytes(m->ld, m->N));
    }
    return mx;
}

// Records of models that are NOT trained (allocation failure, refused size) travel as "absent" records -- a header with
// K = 0 -- so that every rank still reaches the collective and the receivers mark those slots untrained.
// Device scratch of a pack / unpack call: the slot list (n ints), the byte offset of every record (n x 8 bytes) and, for
// unpack, the gathered headers (16 n ints), in d_slots_.
static size_t pk_scratch_ints(int n) { return (size_t)n + 2 * (size_t)n + 16 * (size_t)n + 8; }

int OnGPISStore::pack_models(const int* slots, int n, void* d_buf, size_t stride, hipStream_t s, const size_t* offs, bool factors) {
    if (n <= 0) return GPIS_OK;
    if (factors) { const int frc = train_finish(); if (frc) return frc; }
    else { const int erc = ensure_inverses(s); if (erc) return erc; }
    // (factors: a model whose X is pending and whose factor is still here travels as kind 1; everything else has its X)
    auto as_factor = [&](int slot) {
        if (!factors || slot < 0 || slot >= (int)xstale_.size() || !xstale_[slot]) return false;
        const ClusterModel* m = model(slot);
        return m && m->scratch && m->Lt && m->alpha;
    };
    if (!offs && stride % 256 != 0) return GPIS_ERR_ARG;
    auto rec_off = [&](int i) { return offs ? offs[i] : (size_t)i * stride; };
    auto rec_room = [&](int i) { return offs ? offs[i + 1] - offs[i] : stride; };
    std::vector<int> present;
    std::vector<unsigned long long> poff;
    for (int i = 0; i < n; ++i) {
        if (offs && (offs[i] % 256 != 0 || offs[i + 1] < offs[i])) return GPIS_ERR_ARG;
        const ClusterModel* m = model(slots[i]);
        if (m && m->base && m->Xt) {
            if (packed_model_bytes(m->ld, m->N) > rec_room(i)) return GPIS_ERR_ARG;
            present.push_back(as_factor(slots[i]) ? (slots[i] | (1 << 30)) : slots[i]); poff.push_back((unsigned long long)rec_off(i));
        } else if (rec_room(i) < 64) return GPIS_ERR_ARG;
    }
    int rc = sync_models(s);
    if (rc) return rc;
    if ((int)present.size() < n) {      // absent records: zero headers (K = 0)
        for (int i = 0; i < n; ++i) {
            const ClusterModel* m = model(slots[i]);
            if (!(m && m->base && m->Xt)) GPIS_HIP(hipMemsetAsync((char*)d_buf + rec_off(i), 0, 64, s));
        }
    }
    if (!present.empty()) {
        const int np = (int)present.size();
        if ((int)pk_scratch_ints(n) > cap_slots_) {
            (void)hipFree(d_slots_); d_slots_ = nullptr; cap_slots_ = 0;
            const size_t cap = pk_scratch_ints(n + n / 2 + 64);
            GPIS_HIP(hipMalloc(&d_slots_, sizeof(int) * cap));
            cap_slots_ = (int)cap;
        }
        unsigned long long* d_offs = reinterpret_cast<unsigned long long*>(d_slots_ + ((n + 1) & ~1));
        GPIS_HIP(hipMemcpyAsync(d_slots_, present.data(), sizeof(int) * (size_t)np, hipMemcpyHostToDevice, s));
        GPIS_HIP(hipMemcpyAsync(d_offs, poff.data(), sizeof(unsigned long long) * (size_t)np, hipMemcpyHostToDevice, s));
        model_pack_launch(true, d_models_, d_slots_, np, (char*)d_buf, d_offs, s);
        GPIS_HIP(hipGetLastError());
    }
    GPIS_HIP(hipStreamSynchronize(s));     // (the host vectors above were the sources of asynchronous copies)
    return GPIS_OK;
}

int OnGPISStore::unpack_models(const void* d_buf, int n, size_t stride, int* slots, hipStream_t s, const size_t* offs) {
    if (n <= 0) return GPIS_OK;
    (void)train_finish();
    if (!offs && stride % 256 != 0) return GPIS_ERR_ARG;
    auto rec_off = [&](int i) { return offs ? offs[i] : (size_t)i * stride; };
    auto rec_room = [&](int i) { return offs ? offs[i + 1] - offs[i] : stride; };
    for (int i = 0; i < n; ++i) if (offs && (offs[i] % 256 != 0 || offs[i + 1] < offs[i] + 64)) return GPIS_ERR_ARG;
    if ((int)pk_scratch_ints(n) > cap_slots_) {
        (void)hipFree(d_slots_); d_slots_ = nullptr; cap_slots_ = 0;
        const size_t cap = pk_scratch_ints(n + n / 2 + 64);
        GPIS_HIP(hipMalloc(&d_slots_, sizeof(int) * cap));
        cap_slots_ = (int)cap;
    }
    unsigned long long* d_offs = reinterpret_cast<unsigned long long*>(d_slots_ + ((n + 1) & ~1));
    int* d_hdr = d_slots_ + ((n + 1) & ~1) + 2 * n;
    std::vector<unsigned long long> aoff(n);
    for (int i = 0; i < n; ++i) aoff[i] = (unsigned long long)rec_off(i);
    std::vector<int> hdr((size_t)16 * n);
    GPIS_HIP(hipMemcpyAsync(d_offs, aoff.data(), sizeof(unsigned long long) * (size_t)n, hipMemcpyHostToDevice, s));
    model_headers_launch((const char*)d_buf, d_offs, n, d_hdr, s);
    GPIS_HIP(hipGetLastError());
    GPIS_HIP(hipMemcpyAsync(hdr.data(), d_hdr, sizeof(int) * 16 * (size_t)n, hipMemcpyDeviceToHost, s));
    GPIS_HIP(hipStreamSynchronize(s));
    std::vector<int> created;            // slots made by this call: released again if a later record is refused
    auto bail = [&](int rc) { for (int sl : created) release_slot(sl); return rc; };
    std::vector<int> present;
    std::vector<unsigned long long> poff;
    last_unpack_factors = 0;
    for (int i = 0; i < n; ++i) {
        const int* h = &hdr[(size_t)16 * i];
        const int dim = h[0], N = h[1], ng = h[2], K = h[3], ld = h[4], kind = h[7];
        const bool absent = (K == 0 && N == 0);
        if (!absent && kind != 0 && kind != 1) return bail(GPIS_ERR_ARG);
        if (!absent && (dim != dim_ || N <= 0 || ng < 0 || ng > N || K != N + dim * ng || ld != (int)align_up((size_t)K + 1, 32) ||
                        packed_model_bytes(ld, N) > rec_room(i)))
            return bail(GPIS_ERR_ARG);
        if (slots[i] < 0) { slots[i] = new_slot(); created.push_back(slots[i]); }
        else if (slots[i] >= (int)models_.size() || !live_[slots[i]]) return bail(GPIS_ERR_ARG);
        if (absent) {                    // the owner could not train it: untrained here too (test() sees no GP in that cell)
            ClusterModel& m = models_[slots[i]];
            free_model_mem(m);
            std::memset(&m, 0, sizeof(ClusterModel));
            dirty_ = true;
            continue;
        }
        if (slots[i] < (int)xstale_.size()) xstale_[slots[i]] = 0;     // (a prediction record carries its X)
        int rc = alloc_model(slots[i], N, ng, kind == 1 ? kAllocFactorImport : kAllocPredictOnly);
        if (rc) return bail(rc);
        if (kind == 1) { mark_stale(slots[i]); ++last_unpack_factors; }   // its X is computed by the first prediction (ensure_inverses)
        present.push_back(kind == 1 ? (slots[i] | (1 << 30)) : slots[i]); poff.push_back(aoff[i]);
    }
    int rc = sync_models(s);
    if (rc) return bail(rc);
    if (!present.empty()) {
        const int np = (int)present.size();
        GPIS_HIP(hipMemcpyAsync(d_slots_, present.data(), sizeof(int) * (size_t)np, hipMemcpyHostToDevice, s));
        GPIS_HIP(hipMemcpyAsync(d_offs, poff.data(), sizeof(unsigned long long) * (size_t)np, hipMemcpyHostToDevice, s));
        model_pack_launch(false, d_models_, d_slots_, np, (char*)const_cast<void*>(d_buf), d_offs, s);
        GPIS_HIP(hipGetLastError());
    }
    GPIS_HIP(hipStreamSynchronize(s));
    return GPIS_OK;
}

// Job-level predict with host job arrays: sort by model, cut into tiles of ONGPIS_TILE_Q (16), launch per
// size class.  (The map-level test path bins on the device instead, see map_query.hip.)
int OnGPISStore::eval_jobs(const float* d_xq4, const int* h_job_q, const int* h_job_model, int njobs, float* d_out,
                           hipStream_t s) {
    if (njobs <= 0) return GPIS_OK;
    int rc = ensure_inverses(s);
    if (rc) return rc;
    rc = sync_models(s);
    if (rc) return rc;
    std::vector<int> order(njobs);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return h_job_model[a] < h_job_model[b]; });
    std::vector<int> jq(njobs), jo(njobs);
    std::vector<int> tmodel[ONGPIS_NCLASS], toff[ONGPIS_NCLASS], tcnt[ONGPIS_NCLASS];
    int maxN[ONGPIS_NCLASS] = {0}, maxLd[ONGPIS_NCLASS] = {0};
    for (int i = 0; i < njobs;) {
        int mslot = h_job_model[order[i]];
        const ClusterModel* m = model(mslot);
        if (!m || !m->base) return GPIS_ERR_ARG;
        int cls = ongpis_eval_class(m->ld / 32);
        int e = i;
        while (e < njobs && h_job_model[order[e]] == mslot) ++e;
        for (int t = i; t < e; t += ONGPIS_TILE_Q) {
            tmodel[cls].push_back(mslot); toff[cls].push_back(t); tcnt[cls].push_back(std::min(ONGPIS_TILE_Q, e - t));
        }
        maxN[cls] = std::max(maxN[cls], m->N); maxLd[cls] = std::max(maxLd[cls], m->ld);
        for (int t = i; t < e; ++t) { jq[t] = h_job_q[order[t]]; jo[t] = order[t]; }
        i = e;
    }
    size_t ntl = 0;
    for (int c = 0; c < ONGPIS_NCLASS; ++c) ntl += tmodel[c].size();
    size_t need = 2 * (size_t)njobs + 3 * ntl;
    if ((int)need > cap_ej_) {
        (void)hipFree(d_ej_); d_ej_ = nullptr;
        int cap = (int)need * 3 / 2 + 1024;
        GPIS_HIP(hipMalloc(&d_ej_, sizeof(int) * (size_t)cap));
        cap_ej_ = cap;
    }
    int* d_jq = d_ej_;
    int* d_jo = d_ej_ + njobs;
    int* d_t = d_ej_ + 2 * njobs;
    GPIS_HIP(hipMemcpyAsync(d_jq, jq.data(), sizeof(int) * njobs, hipMemcpyHostToDevice, s));
    GPIS_HIP(hipMemcpyAsync(d_jo, jo.data(), sizeof(int) * njobs, hipMemcpyHostToDevice, s));
    std::vector<int> tl;
    size_t base[ONGPIS_NCLASS];
    for (int c = 0; c < ONGPIS_NCLASS; ++c) {
        base[c] = tl.size();
        tl.insert(tl.end(), tmodel[c].begin(), tmodel[c].end());
        tl.insert(tl.end(), toff[c].begin(), toff[c].end());
        tl.insert(tl.end(), tcnt[c].begin(), tcnt[c].end());
    }
    if (!tl.empty()) GPIS_HIP(hipMemcpyAsync(d_t, tl.data(), sizeof(int) * tl.size(), hipMemcpyHostToDevice, s));
    if (profile) {
        if (!ev0_) { GPIS_HIP(hipEventCreate(&ev0_)); GPIS_HIP(hipEventCreate(&ev1_)); }
        GPIS_HIP(hipEventRecord(ev0_, s));
    }
    for (int c = 0; c < ONGPIS_NCLASS; ++c) {
        int nt = (int)tmodel[c].size();
        if (!nt) continue;
        EvalArgs a;
        a.models = d_models_; a.xq = reinterpret_cast<const float4*>(d_xq4);
        a.tile_model = d_t + base[c]; a.tile_off = d_t + base[c] + nt; a.tile_cnt = d_t + base[c] + 2 * nt;
        a.job_q = d_jq; a.job_out = d_jo; a.out = d_out; a.cb = 0; a.debug = debug_inject; a.err = eval_err(); a.trace = nullptr;
        rc = ongpis_eval_launch(c, nt, maxN[c], maxLd[c], a, s);
        if (rc) return rc;
    }
    if (profile) GPIS_HIP(hipEventRecord(ev1_, s));
    GPIS_HIP(hipStreamSynchronize(s));
    if (profile) GPIS_HIP(hipEventElapsedTime(&last_eval_ms, ev0_, ev1_));
    if (const int ew = take_eval_err()) {
        fprintf(stderr, "[gpismap_amd] prediction kernels reported error word 0x%x (a ring wait expired): the affected results are NaN\n", ew);
        return GPIS_ERR_STATE;
    }
    return GPIS_OK;
}

}  // namespace gpis
