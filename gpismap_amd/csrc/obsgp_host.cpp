// Host side of the device ObsGP: partition tables (reference ObsGP.cpp:204-265 for
// the 2-D tiling, :85-143 for the 1-D groups), buffer management, H2D/D2H.
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <map>
#include <mutex>
#include <set>
#include <vector>
#include <algorithm>
#include "obsgp.h"

namespace gpis {

// ---------------------------------------------------------------- DevPool ----
// Best-fit allocator with splitting and coalescing over large device chunks (one hipMalloc per 512 MiB, not one per
// cluster model).  Round 1-2 used exact size classes: a cluster model that grows by a few percent per frame (the usual
// case) left its class at every retraining, its freed block matched nobody, and the pool grew by 2-4 GB per frame on the
// synthetic sequence (25 GB after ten frames for ~5 GB of live models, a 256 MiB hipMalloc every few dozen models).  Freed
// neighbours now merge, so the blocks the shrinking-and-growing population leaves behind are found again.
struct DevPool {
    std::map<char*, size_t> free_addr_;            // free blocks by address (for merging)
    std::multimap<size_t, char*> free_size_;       // the same blocks by size (best fit)
    std::map<void*, size_t> live_;
    std::vector<void*> chunks_;
    std::vector<size_t> chunk_sizes_;
    int device = -1;                               // device the pool was created on (its chunks return to that device's cache)
    std::set<char*> chunk_base_;                   // blocks never merge across two hipMalloc regions
    size_t bytes = 0;
    void drop_free(char* a, size_t sz) {
        free_addr_.erase(a);
        auto r = free_size_.equal_range(sz);
        for (auto it = r.first; it != r.second; ++it) if (it->second == a) { free_size_.erase(it); break; }
    }
    void add_free(char* a, size_t sz) { free_addr_[a] = sz; free_size_.insert({sz, a}); }
};
// Standard-size chunks of destroyed pools are kept per device for the next pool of the process (a 512 MiB hipMalloc
// costs 15 ms whenever the driver has to wipe the pages first -- a map that is reset / re-created per sequence paid for its whole pool again, in the middle of its
// first frames); bounded by GPIS_POOL_CACHE_GB (default 16, 0 = give everything back at once; a 4 GB default was measured in round 5: the
// F = 5 bench map holds 7 GB of models, and every fusion after the first paid 3 x 5 ms of hipMalloc in its last frame again).  gpis_pool_cache_trim() (C-ABI,
// also run at process exit by the Python mirror) hands the cached chunks back to the driver; a hipMalloc of the library that
// fails retries once after trimming the cache, so cached chunks never make an allocation of this library fail.
namespace {
constexpr size_t kPoolChunk = (size_t)512 << 20;
std::mutex g_chunk_mu;
std::map<int, std::vector<void*>> g_chunk_cache;     // device -> free standard-size chunks
size_t chunk_cache_limit() {
    static const size_t lim = [] { const char* e = getenv("GPIS_POOL_CACHE_GB"); const double gb = e ? atof(e) : 16.0; return (size_t)(gb > 0 ? gb * 1024.0 * 1024.0 * 1024.0 / (double)kPoolChunk : 0); }();
    return lim;
}
// Round 6, measured (tools/ubench/malloc_busy.hip, profiles/r06_malloc_busy.txt): a 512 MiB hipMalloc costs 0.02 ms while the driver
// still has wiped pages to hand out and 15.1 ms (the wipe of 512 MiB) once it has not -- whichever thread asks, with or without
// kernels in flight.  A helper thread that kept max(8, as many as the pools hold) spare chunks ahead of the demand was built on
// the round-5 reading ("5-10 ms per chunk") and removed again: it doubled the process' demand, ran into the 15 ms allocations the
// pools alone never met, and those stalled update() by 30-220 ms a frame (the wipes share the copy engines with its small
// transfers).  The pools allocate for themselves, exactly what they need.
void* chunk_cache_take() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_chunk_mu);
    void* c = nullptr;
    auto it = g_chunk_cache.find(dev);
    if (it != g_chunk_cache.end() && !it->second.empty()) { c = it->second.back(); it->second.pop_back(); }
    return c;
}
bool chunk_cache_put(int dev, void* c) {
    std::lock_guard<std::mutex> lk(g_chunk_mu);
    auto& v = g_chunk_cache[dev];
    if (v.size() >= chunk_cache_limit()) return false;
    v.push_back(c);
    return true;
}
}  // namespace
size_t pool_cache_trim() {
    std::vector<std::pair<int, void*>> all;
    {
        std::lock_guard<std::mutex> lk(g_chunk_mu);
        for (auto& kv : g_chunk_cache) { for (void* c : kv.second) all.push_back({kv.first, c}); kv.second.clear(); }
    }
    for (auto& dc : all) { DeviceScope ds(dc.first); (void)hipFree(dc.second); }
    return all.size() * kPoolChunk;
}
DevPool* pool_create() {
    DevPool* p = new DevPool();
    if (hipGetDevice(&p->device) != hipSuccess) p->device = -1;
    return p;
}
void pool_destroy(DevPool* p) {
    if (!p) return;
    // (hipFree synchronises the device implicitly; a cached chunk skips it, so do it once, explicitly: nothing may still be
    // running on memory that the next pool of the process hands out)
    if (!p->chunks_.empty() && p->device >= 0) { DeviceScope ds(p->device); (void)hipDeviceSynchronize(); }
    for (size_t i = 0; i < p->chunks_.size(); ++i) {
        void* c = p->chunks_[i];
        if (!(p->chunk_sizes_[i] == kPoolChunk && p->device >= 0 && chunk_cache_put(p->device, c))) (void)hipFree(c);
    }
    delete p;
}
void* pool_alloc(DevPool* p, size_t bytes) {
    constexpr size_t kGran = 4u << 10, kMinSplit = 64u << 10, kChunk = kPoolChunk;
    size_t c = ((bytes ? bytes : 1) + kGran - 1) / kGran * kGran;
    auto it = p->free_size_.lower_bound(c);
    if (it == p->free_size_.end()) {
        const size_t chunk = std::max(c, kChunk);
        void* base = (chunk == kChunk) ? chunk_cache_take() : nullptr;
#ifdef GPIS_INSTRUMENT
        struct SlowAlloc {      // (instrumented builds: an allocation the driver had to wipe pages for shows up as ~15 ms per 512 MiB)
            std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now(); bool mine;
            ~SlowAlloc() { const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count(); if (mine && ms > 1.f) fprintf(stderr, "[pool] hipMalloc of a chunk took %.1f ms\n", ms); }
        } slow_alloc{std::chrono::steady_clock::now(), base == nullptr};
#endif
        if (!base && hipMalloc(&base, chunk) != hipSuccess) {
            (void)hipGetLastError();
            if (pool_cache_trim() == 0 || hipMalloc(&base, chunk) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        }
        p->chunks_.push_back(base);
        p->chunk_sizes_.push_back(chunk);
        p->chunk_base_.insert((char*)base);
        p->bytes += chunk;
        p->add_free((char*)base, chunk);
        it = p->free_size_.lower_bound(c);
    }
    char* a = it->second;
    const size_t sz = it->first;
    p->drop_free(a, sz);
    if (sz - c >= kMinSplit) p->add_free(a + c, sz - c);
    else c = sz;
    p->live_[a] = c;
    return a;
}
void pool_free(DevPool* p, void* ptr) {
    if (!ptr) return;
    auto it = p->live_.find(ptr);
    if (it == p->live_.end()) return;
    char* a = (char*)ptr;
    size_t sz = it->second;
    p->live_.erase(it);
    auto nx = p->free_addr_.find(a + sz);                       // merge with the free block behind it
    if (nx != p->free_addr_.end() && !p->chunk_base_.count(nx->first)) { const size_t nsz = nx->second; p->drop_free(a + sz, nsz); sz += nsz; }
    if (!p->chunk_base_.count(a)) {                             // merge with the free block in front of it
        auto pv = p->free_addr_.lower_bound(a);
        if (pv != p->free_addr_.begin()) {
            --pv;
            if (pv->first + pv->second == a) { char* pa = pv->first; const size_t psz = pv->second; p->drop_free(pa, psz); a = pa; sz += psz; }
        }
    }
    p->add_free(a, sz);
}
size_t pool_bytes(DevPool* p) { return p->bytes; }
size_t pool_block_size(DevPool* p, void* ptr) {      // bytes of a live block (0: not one)
    if (!ptr) return 0;
    auto it = p->live_.find(ptr);
    return it == p->live_.end() ? 0 : it->second;
}

// ------------------------------------------------------------ ObsGPDevice ----
static constexpr int OVERLAP2 = 3, GROUP2 = 5;   // params.h:108-109
static constexpr int OVERLAP1 = 6, GROUP1 = 20;  // params.h:103-104

ObsGPDevice::ObsGPDevice() { std::memset(&view_, 0, sizeof(view_)); }
ObsGPDevice::~ObsGPDevice() {
    (void)hipFree(d_x_); (void)hipFree(d_f_); (void)hipFree(d_idx_); (void)hipFree(d_tab_);
    (void)hipFree(d_q_); (void)hipFree(d_val_); (void)hipFree(d_var_);
    (void)hipHostFree(h_q_); (void)hipHostFree(h_val_); (void)hipHostFree(h_var_);
    if (b_pending_) (void)hipEventSynchronize(evb_);
    (void)hipHostFree(hb_q_); (void)hipHostFree(hb_val_); (void)hipHostFree(hb_var_);
    (void)hipFree(db_q_); (void)hipFree(db_val_); (void)hipFree(db_var_); (void)hipFree(d_bin_[0]); (void)hipFree(d_bin_[1]);
    if (evb_) (void)hipEventDestroy(evb_);
    (void)hipFree(view_.tn); (void)hipFree(view_.tx); (void)hipFree(view_.talpha); (void)hipFree(view_.tL);
}

int ObsGPDevice::ensure_groups(int ng) {
    if (ng <= cap_groups_) return GPIS_OK;
    (void)hipFree(view_.tn); (void)hipFree(view_.tx); (void)hipFree(view_.talpha); (void)hipFree(view_.tL);
    view_.tn = nullptr; view_.tx = nullptr; view_.talpha = nullptr; view_.tL = nullptr;
    cap_groups_ = 0;
    GPIS_HIP(hipMalloc(&view_.tn, sizeof(int) * ng));
    GPIS_HIP(hipMalloc(&view_.tx, sizeof(float) * 128 * (size_t)ng));
    GPIS_HIP(hipMalloc(&view_.talpha, sizeof(float) * 64 * (size_t)ng));
    GPIS_HIP(hipMalloc(&view_.tL, sizeof(float) * 4096 * (size_t)ng));
    cap_groups_ = ng;
    return GPIS_OK;
}
int ObsGPDevice::ensure_io(size_t nx, size_t nf) {
    if (nx > cap_x_) { (void)hipFree(d_x_); d_x_ = nullptr; cap_x_ = 0; GPIS_HIP(hipMalloc(&d_x_, sizeof(float) * nx)); cap_x_ = nx; }
    if (nf > cap_f_) { (void)hipFree(d_f_); d_f_ = nullptr; cap_f_ = 0; GPIS_HIP(hipMalloc(&d_f_, sizeof(float) * nf)); cap_f_ = nf; }
    return GPIS_OK;
}
int ObsGPDevice::ensure_q(int nq) {
    if (nq <= cap_q_) return GPIS_OK;
    (void)hipFree(d_q_); (void)hipFree(d_val_); (void)hipFree(d_var_);
    d_q_ = d_val_ = d_var_ = nullptr; cap_q_ = 0;
    int cap = 2 * nq + 4096;   // (geometric: hipMalloc / hipFree in the middle of a frame cost more than the memory)
    GPIS_HIP(hipMalloc(&d_q_, sizeof(float) * 2 * (size_t)cap));
    GPIS_HIP(hipMalloc(&d_val_, sizeof(float) * (size_t)cap));
    GPIS_HIP(hipMalloc(&d_var_, sizeof(float) * (size_t)cap));
    cap_q_ = cap;
    return GPIS_OK;
}

int ObsGPDevice::train2d(const float* xt, const float* f, int ni, int nj, hipStream_t s) {
    if (b_pending_) (void)wait_b();    // (a batch of the second staging set still reads the groups)
    trained_ = false;
    if (!(ni > 0 && nj > 0 && xt && f)) return GPIS_ERR_ARG;
    bool repart = (sz0_ != ni || sz1_ != nj || view_.mode != 2);
    if (repart) {  // ObsGP2D::computePartition
        sz0_ = ni; sz1_ = nj;
        int ng0 = (ni - OVERLAP2) / GROUP2 + 1, ng1 = (nj - OVERLAP2) / GROUP2 + 1;
        if (ng0 < 1 || ng1 < 1) return GPIS_ERR_ARG;
        h_i0_.clear(); h_i1_.clear(); h_j0_.clear(); h_j1_.clear(); h_vali_.clear(); h_valj_.clear();
        h_vali_.push_back(xt[0]);
        for (int n = 0; n < ng0; ++n) {
            int a = n * GROUP2, b = a + GROUP2 + OVERLAP2 - 1;
            if (n < ng0 - 1) h_vali_.push_back(xt[2 * (b - OVERLAP2 / 2)]);
            else { b = ni - 1; h_vali_.push_back(xt[2 * b]); }
            h_i0_.push_back(a); h_i1_.push_back(b);
        }
        h_valj_.push_back(xt[1]);
        for (int m = 0; m < ng1; ++m) {
            int a = m * GROUP2, b = a + GROUP2 + OVERLAP2 - 1;
            if (m < ng1 - 1) h_valj_.push_back(xt[2 * (size_t)(b - OVERLAP2 / 2) * ni + 1]);
            else { b = nj - 1; h_valj_.push_back(xt[2 * (size_t)b * ni + 1]); }
            h_j0_.push_back(a); h_j1_.push_back(b);
        }
        int nidx = 2 * ng0 + 2 * ng1, ntab = (ng0 + 1) + (ng1 + 1);
        if (nidx > cap_idx_) { (void)hipFree(d_idx_); d_idx_ = nullptr; GPIS_HIP(hipMalloc(&d_idx_, sizeof(int) * nidx)); cap_idx_ = nidx; }
        if (ntab > cap_tab_) { (void)hipFree(d_tab_); d_tab_ = nullptr; GPIS_HIP(hipMalloc(&d_tab_, sizeof(float) * ntab)); cap_tab_ = ntab; }
        std::vector<int> idx;
        idx.insert(idx.end(), h_i0_.begin(), h_i0_.end()); idx.insert(idx.end(), h_i1_.begin(), h_i1_.end());
        idx.insert(idx.end(), h_j0_.begin(), h_j0_.end()); idx.insert(idx.end(), h_j1_.begin(), h_j1_.end());
        std::vector<float> tab(h_vali_);
        tab.insert(tab.end(), h_valj_.begin(), h_valj_.end());
        GPIS_HIP(hipMemcpyAsync(d_idx_, idx.data(), sizeof(int) * nidx, hipMemcpyHostToDevice, s));
        GPIS_HIP(hipMemcpyAsync(d_tab_, tab.data(), sizeof(float) * ntab, hipMemcpyHostToDevice, s));
        GPIS_HIP(hipStreamSynchronize(s));  // staging vectors go out of scope
        view_.mode = 2; view_.ni = ni; view_.nj = nj; view_.ng0 = ng0; view_.ng1 = ng1; view_.ngroups = ng0 * ng1;
        view_.i0 = d_idx_; view_.i1 = d_idx_ + ng0; view_.j0 = d_idx_ + 2 * ng0; view_.j1 = d_idx_ + 2 * ng0 + ng1;
        view_.ga = nullptr; view_.glen = nullptr;
        view_.vali = d_tab_; view_.valj = d_tab_ + (ng0 + 1);
        int rc = ensure_groups(view_.ngroups);
        if (rc) return rc;
    }
    size_t npx = (size_t)ni * nj;
    int rc = ensure_io(2 * npx, npx);
    if (rc) return rc;
    GPIS_HIP(hipMemcpyAsync(d_x_, xt, sizeof(float) * 2 * npx, hipMemcpyHostToDevice, s));
    GPIS_HIP(hipMemcpyAsync(d_f_, f, sizeof(float) * npx, hipMemcpyHostToDevice, s));
    view_.x = d_x_; view_.f = d_f_;
    obsgp_launch_train(view_, s);
    GPIS_HIP(hipGetLastError());
    GPIS_HIP(hipStreamSynchronize(s));
    trained_ = true;
    return GPIS_OK;
}

int ObsGPDevice::train1d(const float* xt, const float* f, int N, hipStream_t s) {
    trained_ = false;
    if (!(N > 0 && xt && f)) return GPIS_ERR_ARG;
    sz0_ = sz1_ = 0;  // a later train2d must re-partition
    int nGroup = N / GROUP1 + 1;
    std::vector<int> ga, glen;
    std::vector<float> range;
    range.push_back(xt[0]);
    for (int n = 0; n < nGroup - 1; ++n) {
        if (n < nGroup - 2) {
            int a = n * GROUP1, b = a + GROUP1 + OVERLAP1;
            range.push_back(xt[b - OVERLAP1 / 2]);
            ga.push_back(a); glen.push_back(GROUP1 + OVERLAP1);
        } else {
            int a = n * GROUP1;
            int b = a + (N - a) / 2 + OVERLAP1;
            range.push_back(xt[b - OVERLAP1 / 2]);
            ga.push_back(a); glen.push_back(b - a + 1);
            ++n;
            a = a + (N - a) / 2;
            b = N - 1;
            range.push_back(xt[b]);
            ga.push_back(a); glen.push_back(b - a + 1);
        }
    }
    int ng = (int)ga.size();
    if (ng < 1) return GPIS_ERR_ARG;
    for (int g = 0; g < ng; ++g)
        if (glen[g] > 64 || ga[g] + glen[g] > N || glen[g] < 1) return GPIS_ERR_LIMIT;
    int nidx = 2 * ng, ntab = ng + 1;
    if (nidx > cap_idx_) { (void)hipFree(d_idx_); d_idx_ = nullptr; GPIS_HIP(hipMalloc(&d_idx_, sizeof(int) * nidx)); cap_idx_ = nidx; }
    if (ntab > cap_tab_) { (void)hipFree(d_tab_); d_tab_ = nullptr; GPIS_HIP(hipMalloc(&d_tab_, sizeof(float) * ntab)); cap_tab_ = ntab; }
    std::vector<int> idx(ga);
    idx.insert(idx.end(), glen.begin(), glen.end());
    GPIS_HIP(hipMemcpyAsync(d_idx_, idx.data(), sizeof(int) * nidx, hipMemcpyHostToDevice, s));
    GPIS_HIP(hipMemcpyAsync(d_tab_, range.data(), sizeof(float) * ntab, hipMemcpyHostToDevice, s));
    int rc = ensure_io((size_t)N, (size_t)N);
    if (rc) return rc;
    GPIS_HIP(hipMemcpyAsync(d_x_, xt, sizeof(float) * N, hipMemcpyHostToDevice, s));
    GPIS_HIP(hipMemcpyAsync(d_f_, f, sizeof(float) * N, hipMemcpyHostToDevice, s));
    view_.mode = 1; view_.ni = N; view_.nj = 1; view_.ng0 = ng; view_.ng1 = 1; view_.ngroups = ng;
    view_.i0 = view_.i1 = view_.j0 = view_.j1 = nullptr;
    view_.ga = d_idx_; view_.glen = d_idx_ + ng;
    view_.vali = d_tab_; view_.valj = nullptr;
    view_.x = d_x_; view_.f = d_f_;
    rc = ensure_groups(ng);
    if (rc) return rc;
    obsgp_launch_train(view_, s);
    GPIS_HIP(hipGetLastError());
    GPIS_HIP(hipStreamSynchronize(s));
    trained_ = true;
    return GPIS_OK;
}

int ObsGPDevice::query_device(const float* d_q, int nq, float* d_val, float* d_var, hipStream_t s) {
    if (!trained_) return GPIS_ERR_STATE;
    { const int lrc = launch_query(0, d_q, nq, d_val, d_var, s); if (lrc) return lrc; }
    GPIS_HIP(hipGetLastError());
    return GPIS_OK;
}

int ObsGPDevice::query(const float* q, int nq, float* val, float* var, hipStream_t s) {
    if (!trained_) return GPIS_ERR_STATE;
    if (nq <= 0) return GPIS_OK;
    int rc = ensure_q(nq);
    if (rc) return rc;
    int per = (view_.mode == 2) ? 2 : 1;
    GPIS_HIP(hipMemcpyAsync(d_q_, q, sizeof(float) * per * (size_t)nq, hipMemcpyHostToDevice, s));
    GPIS_HIP(hipMemcpyAsync(d_val_, val, sizeof(float) * (size_t)nq, hipMemcpyHostToDevice, s));
    { const int lrc = launch_query(0, d_q_, nq, d_val_, d_var_, s); if (lrc) return lrc; }
    GPIS_HIP(hipGetLastError());
    GPIS_HIP(hipMemcpyAsync(val, d_val_, sizeof(float) * (size_t)nq, hipMemcpyDeviceToHost, s));
    GPIS_HIP(hipMemcpyAsync(var, d_var_, sizeof(float) * (size_t)nq, hipMemcpyDeviceToHost, s));
    GPIS_HIP(hipStreamSynchronize(s));
    return GPIS_OK;
}

float* ObsGPDevice::stage_q(int nq) {
    if (nq > cap_hq_) {
        (void)hipHostFree(h_q_); (void)hipHostFree(h_val_); (void)hipHostFree(h_var_);
        h_q_ = h_val_ = h_var_ = nullptr; cap_hq_ = 0;
        const int cap = 2 * nq + 4096;     // (page-locked allocations cost about a millisecond: grow geometrically -- the re-evaluation batches grow with the map)
        if (hipHostMalloc(&h_q_, sizeof(float) * 2 * (size_t)cap, hipHostMallocDefault) != hipSuccess) { h_q_ = nullptr; return nullptr; }
        if (hipHostMalloc(&h_val_, sizeof(float) * (size_t)cap, hipHostMallocDefault) != hipSuccess) { h_val_ = nullptr; return nullptr; }
        if (hipHostMalloc(&h_var_, sizeof(float) * (size_t)cap, hipHostMallocDefault) != hipSuccess) { h_var_ = nullptr; return nullptr; }
        cap_hq_ = cap;
    }
    return h_q_;
}

int ObsGPDevice::query_staged(int nq, hipStream_t s) {
    if (!trained_) return GPIS_ERR_STATE;
    if (nq <= 0) return GPIS_OK;
    if (nq > cap_hq_ || !h_q_ || !h_val_ || !h_var_) return GPIS_ERR_STATE;
    // Small batches (the line-search rounds of the 2-D re-evaluation, late re-evaluations in 3-D: dependent round trips of a
    // few hundred queries): the kernel reads the queries from and writes the answers into the page-locked staging itself --
    // one launch and one synchronisation instead of a copy in, a memset, the launch and two copies out.
    constexpr int kZeroCopyMax = 4096;      // (below the binned path's threshold; a lane reads 4-8 bytes and writes 8 over the host link)
    if (nq < kZeroCopyMax) {
        float *zq = nullptr, *zval = nullptr, *zvar = nullptr;
        if (hipHostGetDevicePointer((void**)&zq, h_q_, 0) == hipSuccess && hipHostGetDevicePointer((void**)&zval, h_val_, 0) == hipSuccess &&
            hipHostGetDevicePointer((void**)&zvar, h_var_, 0) == hipSuccess) {
            std::memset(h_val_, 0, sizeof(float) * (size_t)nq);
            obsgp_launch_query(view_, zq, nq, zval, zvar, s);
            GPIS_HIP(hipGetLastError());
            GPIS_HIP(hipStreamSynchronize(s));
            return GPIS_OK;
        }
        (void)hipGetLastError();
    }
    int rc = ensure_q(nq);
    if (rc) return rc;
    const int per = (view_.mode == 2) ? 2 : 1;
    GPIS_HIP(hipMemcpyAsync(d_q_, h_q_, sizeof(float) * per * (size_t)nq, hipMemcpyHostToDevice, s));
    GPIS_HIP(hipMemsetAsync(d_val_, 0, sizeof(float) * (size_t)nq, s));
    { const int lrc = launch_query(0, d_q_, nq, d_val_, d_var_, s); if (lrc) return lrc; }
    GPIS_HIP(hipGetLastError());
    GPIS_HIP(hipMemcpyAsync(h_val_, d_val_, sizeof(float) * (size_t)nq, hipMemcpyDeviceToHost, s));
    GPIS_HIP(hipMemcpyAsync(h_var_, d_var_, sizeof(float) * (size_t)nq, hipMemcpyDeviceToHost, s));
    GPIS_HIP(hipStreamSynchronize(s));
    return GPIS_OK;
}

int ObsGPDevice::trained_groups(hipStream_t s) {
    if (!trained_) return 0;
    std::vector<int> tn(view_.ngroups);
    if (hipMemcpyAsync(tn.data(), view_.tn, sizeof(int) * view_.ngroups, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
    if (hipStreamSynchronize(s) != hipSuccess) return -1;
    int c = 0;
    for (int n : tn) c += n > 0;
    return c;
}

int ObsGPDevice::get_group(int g, int* n, float* x, float* alpha, float* L, hipStream_t s) {
    if (!trained_ || g < 0 || g >= view_.ngroups) return GPIS_ERR_ARG;
    GPIS_HIP(hipMemcpyAsync(n, view_.tn + g, sizeof(int), hipMemcpyDeviceToHost, s));
    if (x) GPIS_HIP(hipMemcpyAsync(x, view_.tx + (size_t)g * 128, sizeof(float) * 128, hipMemcpyDeviceToHost, s));
    if (alpha) GPIS_HIP(hipMemcpyAsync(alpha, view_.talpha + (size_t)g * 64, sizeof(float) * 64, hipMemcpyDeviceToHost, s));
    if (L) GPIS_HIP(hipMemcpyAsync(L, view_.tL + (size_t)g * 4096, sizeof(float) * 4096, hipMemcpyDeviceToHost, s));
    GPIS_HIP(hipStreamSynchronize(s));
    return GPIS_OK;
}

}  // namespace gpis

namespace gpis {

// K2 launch: batches of kBinMin queries or more are sorted by group on the device first (obsgp.hip)
int ObsGPDevice::launch_query(int set, const float* d_q, int nq, float* d_val, float* d_var, hipStream_t s) {
    constexpr int kBinMin = 4096;
    if (nq < kBinMin) { obsgp_launch_query(view_, d_q, nq, d_val, d_var, s); return GPIS_OK; }
    const size_t need = 2 * (size_t)nq + 2 * (size_t)(view_.ngroups + 1);
    if (need > cap_bin_[set]) {
        (void)hipFree(d_bin_[set]); d_bin_[set] = nullptr; cap_bin_[set] = 0;
        const size_t cap = 2 * need + 4096;
        GPIS_HIP(hipMalloc(&d_bin_[set], sizeof(int) * cap));
        cap_bin_[set] = cap;
    }
    obsgp_launch_query_binned(view_, d_q, nq, d_val, d_var, d_bin_[set], s);
    return GPIS_OK;
}

float* ObsGPDevice::stage_qb(int nq) {
    if (b_pending_) (void)wait_b();
    if (nq > cap_hqb_) {
        (void)hipHostFree(hb_q_); (void)hipHostFree(hb_val_); (void)hipHostFree(hb_var_);
        hb_q_ = hb_val_ = hb_var_ = nullptr; cap_hqb_ = 0;
        const int cap = nq + nq / 4 + 1024;
        if (hipHostMalloc(&hb_q_, sizeof(float) * 2 * (size_t)cap, hipHostMallocDefault) != hipSuccess) { hb_q_ = nullptr; return nullptr; }
        if (hipHostMalloc(&hb_val_, sizeof(float) * (size_t)cap, hipHostMallocDefault) != hipSuccess) { hb_val_ = nullptr; return nullptr; }
        if (hipHostMalloc(&hb_var_, sizeof(float) * (size_t)cap, hipHostMallocDefault) != hipSuccess) { hb_var_ = nullptr; return nullptr; }
        cap_hqb_ = cap;
    }
    return hb_q_;
}

int ObsGPDevice::query_staged_b_async(int nq, hipStream_t s) {
    if (!trained_) return GPIS_ERR_STATE;
    if (nq <= 0) return GPIS_OK;
    if (nq > cap_hqb_ || !hb_q_ || !hb_val_ || !hb_var_) return GPIS_ERR_STATE;
    if (nq > cap_qb_) {
        (void)hipFree(db_q_); (void)hipFree(db_val_); (void)hipFree(db_var_);
        db_q_ = db_val_ = db_var_ = nullptr; cap_qb_ = 0;
        const int cap = nq + nq / 4 + 1024;
        GPIS_HIP(hipMalloc(&db_q_, sizeof(float) * 2 * (size_t)cap));
        GPIS_HIP(hipMalloc(&db_val_, sizeof(float) * (size_t)cap));
        GPIS_HIP(hipMalloc(&db_var_, sizeof(float) * (size_t)cap));
        cap_qb_ = cap;
    }
    if (!evb_) GPIS_HIP(hipEventCreateWithFlags(&evb_, hipEventDisableTiming));
    const int per = (view_.mode == 2) ? 2 : 1;
    GPIS_HIP(hipMemcpyAsync(db_q_, hb_q_, sizeof(float) * per * (size_t)nq, hipMemcpyHostToDevice, s));
    GPIS_HIP(hipMemsetAsync(db_val_, 0, sizeof(float) * (size_t)nq, s));
    { const int lrc = launch_query(1, db_q_, nq, db_val_, db_var_, s); if (lrc) return lrc; }
    GPIS_HIP(hipGetLastError());
    GPIS_HIP(hipMemcpyAsync(hb_val_, db_val_, sizeof(float) * (size_t)nq, hipMemcpyDeviceToHost, s));
    GPIS_HIP(hipMemcpyAsync(hb_var_, db_var_, sizeof(float) * (size_t)nq, hipMemcpyDeviceToHost, s));
    GPIS_HIP(hipEventRecord(evb_, s));
    b_pending_ = true;
    return GPIS_OK;
}

int ObsGPDevice::wait_b() {
    if (!b_pending_) return GPIS_OK;
    b_pending_ = false;
    GPIS_HIP(hipEventSynchronize(evb_));
    return GPIS_OK;
}

}  // namespace gpis
