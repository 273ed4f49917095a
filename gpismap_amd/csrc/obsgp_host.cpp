// Host side of the device ObsGP: partition tables (reference ObsGP.cpp:204-265 for
// the 2-D tiling, :85-143 for the 1-D groups), buffer management, H2D/D2H.
#include <cstring>
#include <map>
#include <vector>
#include <algorithm>
#include "obsgp.h"

namespace gpis {

// ---------------------------------------------------------------- DevPool ----
// Size-class free lists over large device chunks: a block is carved from the current chunk (bump pointer) the
// first time its class is needed and recycled through its class list afterwards -- one hipMalloc per 256 MiB
// instead of one per cluster model (the first frame allocates several hundred models).
struct DevPool {
    std::multimap<size_t, void*> free_;
    std::map<void*, size_t> live_;
    std::vector<void*> chunks_;
    char* cur_ = nullptr;
    size_t left_ = 0;
    size_t bytes = 0;
};
DevPool* pool_create() { return new DevPool(); }
void pool_destroy(DevPool* p) {
    if (!p) return;
    for (void* c : p->chunks_) (void)hipFree(c);
    delete p;
}
static size_t pool_class(size_t b) {  // 256 KiB granules above 1 MiB, 4 KiB below
    size_t g = b > (1u << 20) ? (256u << 10) : (4u << 10);
    return (b + g - 1) / g * g;
}
void* pool_alloc(DevPool* p, size_t bytes) {
    size_t c = pool_class(bytes ? bytes : 1);
    auto it = p->free_.find(c);
    void* ptr = nullptr;
    if (it != p->free_.end()) { ptr = it->second; p->free_.erase(it); }
    else {
        if (c > p->left_) {
            // the rest of the old chunk stays usable for small classes through the free lists
            while (p->left_ >= (4u << 10)) {
                size_t piece = p->left_ >= (256u << 10) ? (256u << 10) : (4u << 10);
                p->free_.insert({piece, p->cur_});
                p->cur_ += piece; p->left_ -= piece;
            }
            const size_t chunk = std::max<size_t>(c, (size_t)256 << 20);
            void* base = nullptr;
            if (hipMalloc(&base, chunk) != hipSuccess) return nullptr;
            p->chunks_.push_back(base);
            p->cur_ = (char*)base; p->left_ = chunk;
            p->bytes += chunk;
        }
        ptr = p->cur_;
        p->cur_ += c; p->left_ -= c;
    }
    p->live_[ptr] = c;
    return ptr;
}
void pool_free(DevPool* p, void* ptr) {
    if (!ptr) return;
    auto it = p->live_.find(ptr);
    if (it == p->live_.end()) return;
    p->free_.insert({it->second, ptr});
    p->live_.erase(it);
}
size_t pool_bytes(DevPool* p) { return p->bytes; }

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
    (void)hipFree(db_q_); (void)hipFree(db_val_); (void)hipFree(db_var_);
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
    obsgp_launch_query(view_, d_q, nq, d_val, d_var, s);
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
    obsgp_launch_query(view_, d_q_, nq, d_val_, d_var_, s);
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
    int rc = ensure_q(nq);
    if (rc) return rc;
    const int per = (view_.mode == 2) ? 2 : 1;
    GPIS_HIP(hipMemcpyAsync(d_q_, h_q_, sizeof(float) * per * (size_t)nq, hipMemcpyHostToDevice, s));
    GPIS_HIP(hipMemsetAsync(d_val_, 0, sizeof(float) * (size_t)nq, s));
    obsgp_launch_query(view_, d_q_, nq, d_val_, d_var_, s);
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
    obsgp_launch_query(view_, db_q_, nq, db_val_, db_var_, s);
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
