// K5 + test() driver: per-query cluster lookup, device-side binning of queries by
// cluster, two evaluation passes through K4 and the variance-weighted blend.
// Reference: GPisMap3::test_kernel cpp/src/GPisMap3.cpp:794-902 and
// GPisMap::test_kernel cpp/src/GPisMap.cpp:665-763.
#pragma once
#include <vector>
#include "ongpis.h"

namespace gpis {

struct ClusterEntry {  // host description of one non-empty cluster cell (tree traversal order)
    float c[3];
    float lo[3], hi[3];
    int model;         // store slot or -1
    int parent;        // index into the ancestor table or -1
};

// Internal tree node above the cluster level.  The reference reaches a cluster cell only through
// a top-down walk that prunes on EVERY ancestor's box (octree.cpp:864-866); ancestor and child
// bounds are rounded independently, so for queries aligned with cell boundaries the ancestor test
// can fail where the cell's own test passes.  The lookup kernel therefore re-checks the chain.
struct AncestorEntry {
    float lo[3], hi[3];
    int parent;        // next ancestor or -1 (root)
};

struct ClusterTableView {
    int n;
    const float4* c;   // centre
    const float4* lo;
    const float4* hi;
    const int* model;
    const int* parent; // [n] first ancestor (index into anc_*) or -1
    const float4* anc_lo;
    const float4* anc_hi;
    const int* anc_parent;
    const int* grid;   // dense lattice: cell -> table index or -1
    int gx, gy, gz;
    double ox, oy, oz; // lattice origin (lower corner of cell 0)
    double pitch;      // cluster cell size
};

class MapQuery {
public:
    MapQuery(int dim, float search_half, float var_thre, float prior_var);
    ~MapQuery();
    // Rebuild the cluster table + lattice from the host tree (after every update()).
    int set_clusters(const std::vector<ClusterEntry>& cl, const std::vector<AncestorEntry>& anc, double pitch, hipStream_t s);
    // test(): x device [n][dim] interleaved, res device [n][2(1+dim)]; only the entries the
    // reference writes are touched.
    int run(OnGPISStore& store, const float* d_x, int n, float* d_res, hipStream_t s);
    int num_clusters() const { return ncl_; }
    // statistics of the last run
    long long last_evals = 0, last_flops = 0, last_touched = 0;
    int last_launches = 0;      // K4 launches of the last run
    float last_eval_ms = 0.f;   // time inside the K4 launches (hipEvents on the stream)
    int chunk = 1 << 22;
    bool profile = false;

private:
    int ensure_scratch(int n, int nmodels);
    int run_chunk(OnGPISStore& store, const float* d_x, int n, float* d_res, hipStream_t s);
    int eval_pass(OnGPISStore& store, int njobs, int shift, int rec_base, int nmodels, hipStream_t s);
    int dim_;
    float search_half_, var_thre_, prior_var_;
    int ncl_ = 0;
    ClusterTableView tv_{};
    void* d_tab_ = nullptr; size_t cap_tab_ = 0;
    int* d_grid_ = nullptr; size_t cap_grid_ = 0;
    // per-chunk scratch
    int cap_n_ = 0, cap_models_ = 0;
    float4* d_xq_ = nullptr;
    int* d_cand_ = nullptr;     // [3][cap]
    int* d_ncand_ = nullptr;    // [cap]
    int* d_jm_ = nullptr;       // [2*cap] job -> model (-1 inactive)
    int* d_jq_ = nullptr;       // [2*cap] sorted query index
    int* d_jo_ = nullptr;       // [2*cap] sorted output record
    float* d_out_ = nullptr;    // [3*cap][8]
    int* d_cnt_ = nullptr;      // [models] job count
    int* d_base_ = nullptr;     // [models]
    int* d_cursor_ = nullptr;   // [models]
    int* d_tbase_ = nullptr;    // [models]
    int* d_tile_ = nullptr;     // [3][tile_cap]
    int tile_cap_ = 0;
    int* d_tot_ = nullptr;      // [16] totals
    int* d_tie_ = nullptr;      // [1] queries with an exact distance tie among their nearest candidates (list: d_jq_, free at that time)
    std::vector<int> h_maxN_, h_maxLd_;   // per class: largest N and ld (LDS sizing of K4)
    hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
    // K4 launches of one pass (one per size class, disjoint tiles and outputs) run beside each other: the first on the caller's
    // stream, the rest on side streams forked from and joined to it
    static constexpr int kSide = 3;
    hipStream_t side_[kSide] = {nullptr, nullptr, nullptr};
    hipEvent_t evfork_ = nullptr, evjoin_[kSide] = {nullptr, nullptr, nullptr};
    bool side_off_ = false;
};

}  // namespace gpis
