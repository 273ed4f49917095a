// GPisMap3: MI355X-native drop-in for reference cpp/src/GPisMap3.cpp.
//
// The reference interleaves single-point ObsGP queries with octree mutations
// (evalPoints :580-696, reEvalPoints :321-569).  The observation GP is frozen
// for the whole frame, so every query is a pure function of the point being
// processed: here they are evaluated SPECULATIVELY in batches on the GPU (K2)
// and the tree mutations are replayed on the host in the reference's order.
// Points that move into a not-yet-visited cluster during the pass (the reference
// re-evaluates them because it fetches each cluster's node list lazily, :308-311)
// are handled by an on-demand batch when that cluster is reached.
#include <algorithm>
#include <thread>
#include <string>
#include <unistd.h>
#include <cstdlib>
#include <array>
#include <cmath>
#include <chrono>
#include <cstring>
#include <functional>
#include <memory>
#include <unordered_map>
#include "../../include/GPisMap3.h"
#include "flat_tree.h"
#include "host_pool.h"
#include "map_query.h"
#include "obsgp.h"
#include "ongpis.h"

// (defined at the end of this file; used by the several-devices paths above them)
GPisMap3* gpis3_impl_create_on(const GPisMap3Param& par, const camParam& c, const int* devices, int n);
int gpis3_impl_shard_info(GPisMap3* g, int* out, int n);
long long gpis3_impl_shard_bytes(GPisMap3* g, int owner);
int gpis3_impl_shard_pack(GPisMap3* g, void* d_buf, void* stream);
int gpis3_impl_shard_unpack(GPisMap3* g, int owner, const void* d_buf, void* stream);
int gpis3_impl_shard_finish(GPisMap3* g);

using namespace gpis;

// Fine-grained host timing of update() for tools/update_profile.py: only in builds with -DGPIS_INSTRUMENT.
#ifdef GPIS_INSTRUMENT
struct UpdLap {
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void operator()(const char* what) {
        auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[upd] %-28s %7.2f ms\n", what, std::chrono::duration<float, std::milli>(n - t).count());
        t = n;
    }
};
#else
struct UpdLap { void operator()(const char*) {} };
#endif

namespace {

inline float occ_test(float rinv, float rinv0, float a) {  // GPisMap3.cpp:38-41
    return (float)(2.0 * (1.0 / (1.0 + std::exp((double)(-a * (rinv - rinv0)))) - 0.5));
}
inline float saturate(float v, float lo, float hi) { return std::min(std::max(v, lo), hi); }

std::array<float, 9> quat2dcm(const float q[4]) {  // GPisMap3.cpp:48-63
    std::array<float, 9> d;
    d[0] = q[0] * q[0] + q[1] * q[1] - q[2] * q[2] - q[3] * q[3];
    d[1] = (float)(2.0 * (double)(q[1] * q[2] + q[0] * q[3]));
    d[2] = (float)(2.0 * (double)(q[1] * q[3] - q[0] * q[2]));
    d[3] = (float)(2.0 * (double)(q[1] * q[2] - q[0] * q[3]));
    d[4] = q[0] * q[0] - q[1] * q[1] + q[2] * q[2] - q[3] * q[3];
    d[5] = (float)(2.0 * (double)(q[0] * q[1] + q[2] * q[3]));
    d[6] = (float)(2.0 * (double)(q[1] * q[3] + q[0] * q[2]));
    d[7] = (float)(2.0 * (double)(q[2] * q[3] - q[0] * q[1]));
    d[8] = q[0] * q[0] - q[1] * q[1] - q[2] * q[2] + q[3] * q[3];
    return d;
}

constexpr float kRtimes = 2.0f;    // params.h:39 GPISMAP3_RTIMES
constexpr float kCleng = 0.025f;   // params.h:40 GPISMAP3_TREE_CLUSTER_HALF_LENGTH

FlatTreeParam tree_param3() {
    FlatTreeParam p;
    p.min_half = (float)(0.0125 / 2.0);  // params.h:41
    p.max_half = (float)1.6;
    p.init_half = (float)0.4;
    p.cluster_half = kCleng;
    p.min_half_sq = p.min_half * p.min_half;
    p.cluster_eps = 1e-6;
    p.qleaf_eps_plain = 0.0001;
    p.qleaf_eps_dist = 0.001;
    p.qdesc_eps = 0.001;
    return p;
}

}  // namespace

struct GPisMap3::Impl {
    int device = -1;   // HIP device this map lives on (current device at construction)
    int upd_rc = 0;    // first device-side failure inside the last update() (0: none); update() itself is void like the reference's
    int fail_rc = 0;   // last device-side failure of test()/testDevice() (0: none) -- the C-ABI reports it instead of "false"
    using T3 = FlatTree<3>;
    GPisMap3Param setting;
    camParam cam;
    T3 tree;
    T3::Set activeSet;
    ObsGPDevice gpo;
    OnGPISStore store;
    MapQuery mq;
    hipStream_t stream = nullptr;
    // Pipelined update (the default; GPIS_PIPELINE_UPDATE=0 or gpis3_set_pipeline(map, 0) select the reference's
    // synchronous update(), SURVEY 8(b) "Threading"): update() returns once the frame's OnGPIS training is ENQUEUED on
    // `train_stream`; the join -- wait, error word, dropped batch -- happens where the result is first needed: the next
    // update()'s updateGPs, test()/testDevice(), statistics, the sharded exchange, gpis3_sync().  The host work of the next
    // frame (preprocessing, the ObsGP batches on `stream`, tree replay) then runs beside the factorisations of this one.
    // Map state and results do not depend on the mode.
    hipStream_t train_stream = nullptr;
    hipStream_t batch_stream = nullptr;   // the new-pixel ObsGP batch: issued right after the ObsGP training, collected by evalPoints()
    bool batch_inflight = false, batch_launched = false;
    void launch_pixel_batch();
    bool pipeline = false;
    bool device_gather = true;   // K6 range part on the device (gpis3_set_host_gather: the host walk, kept for the cross-check)
    int finish_training();
    bool ok = false;        // device objects usable
    bool has_tree = false;  // reference: t != 0
    bool gpo_created = false;

    float u_obs_limit[2] = {0, 0}, v_obs_limit[2] = {0, 0};
    std::vector<float> vu_grid, obs_zinv, obs_valid_u, obs_valid_v, obs_valid_xyzlocal, obs_valid_xyzglobal;
    struct NewPoint { bool keep = false; float grad[3] = {0, 0, 0}; float noise = 0.f, gnoise = 0.f; };
    NewPoint pixel_point(const float* pv, const float* pr, const float* xl) const;
    std::vector<int> pre_wnode, pre_wpt;   // evalPoints: witnesses of the frozen is_not_new pre-pass
    std::vector<NewPoint> pre_new;         // ... and the data of the pixels that were new as of the pre-pass
    std::vector<float> mirror_soa;         // updateGPs: host image of the point mirror (9 SoA rows)
    std::vector<int> pix_off;              // preprocData: first valid-pixel index of each column range (pix_parts + 1 entries)
    int pix_parts = 0;
    float pose_tr[3] = {0, 0, 0}, pose_R[9] = {0};
    int obs_numdata = 0;
    float range_obs_max = 0.f;

    // host<->device staging for test()
    float* d_x = nullptr; float* d_res = nullptr; size_t cap_x = 0, cap_res = 0;

    // statistics
    long stat_obs_queries = 0, stat_clusters_trained = 0, stat_late = 0;
    double stat_model_bytes = 0;  // sum over live models of 4 [dN + K + K(K+1)/2] (SURVEY 8d model_bytes)
    float last_update_ms[6] = {0, 0, 0, 0, 0, 0};

    // sharded training (gpis3_set_shard): owner rank of every cluster job of the last update(), in job order
    int shard_rank = 0, shard_world = 1;
    struct ShardJob { int slot, n, ng, owner; };
    std::vector<ShardJob> shard_jobs;
    bool table_pending = false;   // update() trained the local share only: the cluster table waits for the exchange
    void build_cluster_table();
    // One process per GPU with the host logic run ONCE (round 6; gpis3_set_frame_export / gpis3_frame_record / gpis3_apply_frame).
    // The LEAD process runs update() as ever, writes down what the frame decided -- the slot operations in order, the point mirror,
    // the cell lists and cluster descriptors of the K6 pass, the training jobs with their owners, the cluster table's entries --
    // and puts the training of its own share off until gpis3_train_deferred(), so that the record can travel first.  A WORKER
    // process never replays the frame: gpis3_apply_frame() mirrors the slot operations on its store (the ids must come out as
    // recorded), uploads the mirror, runs K6 on its own device, trains its share and keeps the table entries for the exchange
    // that follows (gpis3_shard_pack / _unpack / _finish, unchanged).  A worker holds no tree: getAllPoints is the lead's business.
    bool export_frames = false;
    std::vector<char> frame_rec;
    std::vector<int> slot_ops;                   // this frame's slot operations in order: ~slot released, slot >= 0 taken
    std::vector<TrainJob> deferred_jobs;         // the lead's own share, waiting for gpis3_train_deferred()
    std::vector<int> deferred_ids;
    bool deferred_train = false, deferred_dev = true;
    bool remote_index = false;                   // worker process: the cluster table is built from the lead's entries
    std::vector<ClusterEntry> frame_ent;
    std::vector<AncestorEntry> frame_anc;
    void collect_cluster_entries(std::vector<ClusterEntry>& ent, std::vector<AncestorEntry>& anc) const;
    int train_deferred();
    int apply_frame(const char* buf, size_t bytes);

    // ---- several devices behind ONE map object (GPIS_DEVICES=0,1,... or gpis3_create_multi): this instance is rank 0,
    // `peers` are ranks 1..n-1, each a complete map on its own device.  update(): every rank runs the same deterministic
    // host logic on its own host thread and trains its K^3-balanced share of the frame's clusters; the packed models
    // travel device to device (hipMemcpyPeer), are unpacked as predict-only models and every rank builds its table.
    // test(): the queries are dealt to the ranks in blocks of kQueryBlock rows round-robin and answered concurrently;
    // per-query arithmetic does not depend on the cut, so the result is bit-identical to a single-device map.
    // (Reference: the fan-out inside the call over host threads, GPisMap3.cpp:759-784 and :904-949.)
    std::vector<GPisMap3*> peers;
    // Round 5: the host logic of update() -- preprocessing, ObsGP regression and queries, the tree replay -- runs ONCE, on rank 0
    // (the lead); the peers are device workers: the lead mirrors every slot operation on their stores, hands each its share of
    // the frame's training jobs (point mirror + K6 on the worker's own device) and, after the exchange, its cluster table
    // (built from the lead's tree).  A worker's own tree / ObsGP objects stay empty.
    // What travels in the model exchange of a sharded update: -1 (default) follows the inverse mode -- with the lazy inverse the
    // owner ships FACTOR records (model_pack.hip kind 1: Lt + alpha, the same bytes) and every receiver inverts what it predicts
    // with, when it predicts; with the eager inverse, prediction records (X).  0 / 1 force one or the other (gpis3_set_shard_factors).
    int shard_factors_mode = -1;
    bool ship_factors() const { return shard_factors_mode < 0 ? store.lazy_inverse : shard_factors_mode != 0; }
    int stat_deferred_inverses = 0;              // factor records this rank received in the last exchange (their X is still to come)
    Impl* lead = nullptr;                        // workers: the rank-0 instance
    int stat_host_replays = 0;                   // host replays (update_one executions) of the last update() call, all ranks
    template <class F> void each_store(F f) {    // f(store) on this rank's store and on every worker's, each on its device
        f(store);
        for (GPisMap3* q : peers) { DeviceScope ds(q->impl()->device); f(q->impl()->store); }
    }
    int new_slot_all() {                         // the same slot id on every rank (the stores see the same sequence of operations)
        const int s = store.new_slot();
        if (export_frames) slot_ops.push_back(s);
        for (GPisMap3* q : peers) {
            DeviceScope ds(q->impl()->device);
            if (q->impl()->store.new_slot() != s && !upd_rc) { upd_rc = GPIS_ERR_STATE; fprintf(stderr, "[gpismap_amd] model slots of the devices diverged\n"); }
        }
        return s;
    }
    void* d_send = nullptr; size_t cap_send = 0;
    void* d_recv = nullptr; size_t cap_recv = 0;
    float* h_xstage = nullptr; size_t cap_xstage = 0;     // multi-device test(): page-locked staging of this rank's query blocks ...
    float* h_rstage = nullptr; size_t cap_rstage = 0;     // ... and of their results
    std::vector<hipStream_t> peer_streams;       // in-library multi-device exchange: one copy stream per source rank
    double stat_exchange_bytes = 0.0;            // bytes this rank received in the last in-library exchange
    static constexpr int kQueryBlock = 65536;

    Impl(const GPisMap3Param& par, const camParam& c)
        : cam(c), tree(tree_param3()), store(3, par.map_scale_param),
          mq(3, (float)((double)kCleng * 3.0), 0.5f, (float)(1.0 + (double)par.map_noise_param)) {
        setting = par;      // assignment: the public struct's only copy constructor takes a non-const reference (as the reference's)
        int pr_least = 0, pr_greatest = 0;
        ok = (hipGetDevice(&device) == hipSuccess) && (hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest) == hipSuccess) &&
             (hipStreamCreateWithPriority(&stream, hipStreamDefault, pr_greatest) == hipSuccess) &&
             (hipStreamCreateWithPriority(&train_stream, hipStreamNonBlocking, pr_least) == hipSuccess) &&
             (hipStreamCreateWithFlags(&batch_stream, hipStreamNonBlocking) == hipSuccess);
        if (const char* e = getenv("GPIS_PIPELINE_RESERVE_CUS")) { pipeline_reserve = std::max(0, atoi(e)); pipeline_reserve_set = true; }
        // update() is pipelined by default: it returns once the frame's training is enqueued; whatever needs the models (the
        // next training, test(), the getters, gpis3_sync) joins it.  GPIS_PIPELINE_UPDATE=0 / gpis3_set_pipeline(map, 0): every
        // update() joins its own training before it returns, like the reference's.
        if (const char* e = getenv("GPIS_PIPELINE_UPDATE")) want_pipeline = atoi(e) != 0;
        if (ok) apply_pipeline();
        if (const char* e = getenv("GPIS_EAGER_INVERSE")) if (atoi(e) != 0) store.lazy_inverse = false;
        store.trim_scratch = true;     // a cluster keeps only what prediction reads once its inverse exists (gpis3_set_keep_factors: keep all)
        if (!ok) device = -1;
        if (!ok) fprintf(stderr, "[gpismap_amd] GPisMap3: no usable HIP device; update()/test() will fail\n");
    }
    ~Impl() {
        (void)store.train_finish();
        (void)hipFree(d_x); (void)hipFree(d_res); (void)hipFree(d_send); (void)hipFree(d_recv);
        for (hipStream_t ps : peer_streams) if (ps) (void)hipStreamDestroy(ps);
        if (h_xstage) (void)hipHostFree(h_xstage);
        if (h_rstage) (void)hipHostFree(h_rstage);
        if (stream) (void)hipStreamDestroy(stream);
        if (train_stream) (void)hipStreamDestroy(train_stream);
        if (batch_stream) (void)hipStreamDestroy(batch_stream);
    }

    // Pipelined update: the training of frame f runs beside the host work and the ObsGP batches of frame f + 1.  A factorisation
    // workgroup holds its CU for milliseconds and nothing pre-empts it, so the training streams are kept off
    // `pipeline_reserve` CUs (spread evenly over the XCDs): the ObsGP kernels -- highest priority, unmasked -- start at once
    // there (measured, tools/ubench/cumask_probe.hip: 6 us instead of 2 ms beside a busy unmasked stream).  Round 6: a quarter of
    // the device (64 CUs) instead of an eighth -- the next frame's large K2 batches (stored points x 7 queries, the pixel batch)
    // run on the reserved CUs only while a training is in flight, and on 32 of them they took 2.4 ms where the idle device takes
    // 1.0; pipelined frames 2..6 of the synthetic sequence, one call, reserve 32 / 64 / 96: 10.0-13.7 / 8.7-11.5 / 8.3-11.4 ms,
    // per frame with the drain charged 14.0-15.0 / 12.1-12.4 / 13.1-13.3 ms (the drain grows with the reserve: 11.6 / 12.4 / 13.7 ms).
    int pipeline_reserve = 64;
    bool pipeline_reserve_set = false;
    // What the caller asked for (gpis3_set_pipeline / GPIS_PIPELINE_UPDATE) and what is in force: a map that shards its training over
    // ranks or devices trains synchronously (every frame ends with the exchange) and sets no CUs aside, whatever was asked; the
    // wish is remembered, so gpis3_set_shard(0, 1) restores it.
    bool want_pipeline = true;
    void apply_pipeline() { set_pipeline(want_pipeline && peers.empty() && shard_world == 1); }
    void set_pipeline(bool on) {
        (void)store.train_finish();
        int want = 0;
        if (on) {
            int ncu = 0;
            (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device);
            want = pipeline_reserve_set ? pipeline_reserve : std::min(pipeline_reserve, ncu / 4);    // default: a quarter of the device, at most 64 CUs
        }
        if (want != store.cu_reserve()) {
            (void)store.set_cu_reserve(want);
            hipStream_t ns = nullptr;
            if (ongpis_make_train_stream(&ns, want) == GPIS_OK && ns) {
                if (train_stream) { (void)hipStreamSynchronize(train_stream); (void)hipStreamDestroy(train_stream); }
                train_stream = ns;
            }
        }
        pipeline = on;
    }

    void reset() {  // GPisMap3.cpp:99-115
        tree.clear(); has_tree = false;
        store.clear();
        gpo.reset_trained(); gpo_created = false;
        obs_numdata = 0;
        activeSet.clear();
        std::vector<ClusterEntry> none;
        std::vector<AncestorEntry> nanc;
        mq.set_clusters(none, nanc, 2.0 * kCleng, stream);
    }

    bool preprocData(const float* dataz, int N, const std::vector<float>& pose);
    bool regressObs();
    void updateMapPoints();
    void evalPoints();
    void updateGPs();
    int try_insert(int pid, T3::InsSet& ins, bool known_new = false);

    struct Stage2 {   // per point, after the centre query
        bool go = false;        // survives the var / occupancy gates
        float x_new[3], abs_oc = 0.f;
        float grad_loc[3];
    };
    void reeval_batch(const std::vector<int>& ids, std::vector<Stage2>& st, std::vector<float>& pval,
                      std::vector<float>& pvar);
    // Stage 3 of a stored point's re-evaluation in two halves: reeval_math is a pure function of the point's state, its ObsGP
    // answers, the pose and the settings (data-parallel over the points of the frame: host_pool.h); reeval_commit applies the
    // outcome to the tree (sequential, in the reference's order).  reeval_apply = both, for the late arrivals.
    struct ReevalOut { int kind = 0; float pos[3], grad[3], noise = 0.f, gnoise = 0.f; };   // 0 nothing, 1 inflate the noises, 2 move the point
    // Per-frame work arrays of update() kept across frames (round 6): ~9 MB of them were allocated and freed every frame, and glibc
    // gives freed memory of that size back to the system -- every frame paid the page faults and the zeroing again (a run with
    // MALLOC_MMAP_THRESHOLD_ / MALLOC_TRIM_THRESHOLD_ raised: about 1 ms per frame).  Same contents, capacity retained.
    std::vector<int> ws_ids, ws_slot, ws_nodes, ws_late;
    std::vector<Stage2> ws_st;
    std::vector<float> ws_pval, ws_pvar, ws_val, ws_var;
    std::vector<ReevalOut> ws_outs;
    std::vector<char> ws_front;
    std::vector<std::array<float, 3>> ws_loc;
    T3::CellLists ws_cells;
    std::vector<std::vector<int>> ws_touched, ws_plist;
    ReevalOut reeval_math(const FlatPoint<3>& nd, const Stage2& st, const float* pval, const float* pvar) const;
    void reeval_commit(int pid, const ReevalOut& o);
    void reeval_apply(int pid, const Stage2& st, const float* pval, const float* pvar);
    std::unique_ptr<HostPool> hpool;
    HostPool& pool() { if (!hpool) hpool.reset(new HostPool()); return *hpool; }
};

// ------------------------------------------------------------------ preprocess ----
bool GPisMap3::Impl::preprocData(const float* dataz, int N, const std::vector<float>& pose) {  // :125-216
    if (!dataz || N < 1) return false;
    obs_numdata = 0; pix_parts = 0;     // (the pixel arrays keep their storage: sizes follow obs_numdata)
    range_obs_max = 0.0f;
    if (pose.size() != 12) return false;
    for (int i = 0; i < 3; ++i) pose_tr[i] = pose[i];
    for (int i = 0; i < 9; ++i) pose_R[i] = pose[3 + i];
    int n = cam.width / setting.obs_skip;
    int m = cam.height / setting.obs_skip;
    if (vu_grid.empty()) {
        if (cam.width * cam.height != N) return false;
        vu_grid.resize((size_t)2 * n * m);
        int col = 0, row = 0;
        for (int n_ = 0; n_ < n; ++n_) {
            col = n_ * setting.obs_skip;
            for (int m_ = 0; m_ < m; ++m_) {
                row = m_ * setting.obs_skip;
                int j = 2 * (m * n_ + m_);
                vu_grid[j] = ((float)row - cam.cy) / cam.fy;
                vu_grid[j + 1] = ((float)col - cam.cx) / cam.fx;
            }
        }
        u_obs_limit[0] = -cam.cx / cam.fx;
        u_obs_limit[1] = ((float)col - cam.cx) / cam.fx;
        v_obs_limit[0] = -cam.cy / cam.fy;
        v_obs_limit[1] = ((float)row - cam.cy) / cam.fy;
    }
    // Two passes over column ranges on the host threads: count the valid pixels per range, then every range writes its
    // pixels at its offset -- the order (columns, then rows) and every value are those of the sequential loop.  The ranges
    // are kept (pix_off): the query build and the pre-pass of evalPoints() hand each range to the same thread again.
    obs_numdata = 0;
    obs_zinv.resize((size_t)n * m);
    const int parts = std::max(1, std::min(pool().size(), n / 16));
    pix_parts = parts;
    pix_off.assign((size_t)parts + 1, 0);
    std::vector<float> rmax((size_t)parts, 0.f);
    auto valid = [&](int k) { return k < N && (double)dataz[k] < 4e0 && (double)dataz[k] > 4e-1; };   // isRangeValid :33-36
    pool().run_parts(parts, [&](int p) {
        const int c0 = (int)((long long)n * p / parts), c1 = (int)((long long)n * (p + 1) / parts);
        int cnt = 0;
        for (int n_ = c0; n_ < c1; ++n_) {
            const int col = n_ * setting.obs_skip;
            for (int m_ = 0; m_ < m; ++m_) cnt += valid(col * cam.height + m_ * setting.obs_skip) ? 1 : 0;
        }
        pix_off[(size_t)p + 1] = cnt;
    });
    for (int p = 0; p < parts; ++p) pix_off[(size_t)p + 1] += pix_off[(size_t)p];
    const int total = pix_off[(size_t)parts];
    obs_valid_u.resize((size_t)total); obs_valid_v.resize((size_t)total);
    obs_valid_xyzlocal.resize((size_t)3 * total); obs_valid_xyzglobal.resize((size_t)3 * total);
    pool().run_parts(parts, [&](int p) {
        const int c0 = (int)((long long)n * p / parts), c1 = (int)((long long)n * (p + 1) / parts);
        size_t o = (size_t)pix_off[(size_t)p];
        float rm = 0.f;
        for (int n_ = c0; n_ < c1; ++n_) {
            const int col = n_ * setting.obs_skip;
            for (int m_ = 0; m_ < m; ++m_) {
                const int row = m_ * setting.obs_skip;
                const int k = col * cam.height + row;
                if (valid(k)) {
                    const int j = 2 * (m * n_ + m_);
                    const float z = dataz[k];
                    if (rm < z) rm = z;
                    obs_zinv[(size_t)m * n_ + m_] = (float)(1.0 / (double)z);
                    const float u = vu_grid[j + 1], v = vu_grid[j];
                    obs_valid_u[o] = u; obs_valid_v[o] = v;
                    const float xloc = u * z, yloc = v * z;
                    obs_valid_xyzlocal[3 * o] = xloc; obs_valid_xyzlocal[3 * o + 1] = yloc; obs_valid_xyzlocal[3 * o + 2] = z;
                    obs_valid_xyzglobal[3 * o] = pose_R[0] * xloc + pose_R[3] * yloc + pose_R[6] * z + pose_tr[0];
                    obs_valid_xyzglobal[3 * o + 1] = pose_R[1] * xloc + pose_R[4] * yloc + pose_R[7] * z + pose_tr[1];
                    obs_valid_xyzglobal[3 * o + 2] = pose_R[2] * xloc + pose_R[5] * yloc + pose_R[8] * z + pose_tr[2];
                    ++o;
                } else obs_zinv[(size_t)m * n_ + m_] = -1.0f;
            }
        }
        rmax[(size_t)p] = rm;
    });
    for (int p = 0; p < parts; ++p) if (range_obs_max < rmax[(size_t)p]) range_obs_max = rmax[(size_t)p];
    obs_numdata = total;
    return obs_numdata > 1;
}

bool GPisMap3::Impl::regressObs() {  // :239-256 -> K1
    gpo_created = true;
    if (2 * obs_zinv.size() != vu_grid.size()) return false;
    int ni = cam.height / setting.obs_skip, nj = cam.width / setting.obs_skip;
    int rc = gpo.train2d(vu_grid.data(), obs_zinv.data(), ni, nj, stream);
    if (rc != GPIS_OK) { fprintf(stderr, "[gpismap_amd] ObsGP training failed (%d)\n", rc); if (!upd_rc) upd_rc = rc; return false; }
    return gpo.trained();
}

// GPisMap3.cpp:544-556 / :611-623.  Returns 2 when the point was stored and a cluster cell was
// reported (the caller then fills in its data), 1 when it was stored through the set-less
// root-growth path (octree.cpp:151-212: it stays in the tree with default data, as in the
// reference), 0 when it was not stored (the point object is released).
int GPisMap3::Impl::try_insert(int pid, T3::InsSet& ins, bool known_new) {
    bool ok_ = false;
    // (known_new: the caller has just asked is_not_new() for this position and nothing touched the tree since -- the
    // reference's second IsNotNew inside its insert helper would give the same answer)
    if (known_new || !tree.is_not_new_cached(tree.pts[pid].pos)) {
        ok_ = tree.insert_cached(pid, &ins);
        if (ok_ && !tree.is_root(tree.root)) tree.root = tree.get_root(tree.root);
    }
    if (!ok_) { tree.drop_point(pid); return 0; }
    return ins.empty() ? 1 : 2;
}

// ----------------------------------------------------------- re-evaluation ----
// Stage 1+2 for a batch of existing points: centre query (K2), gates, the (query-free)
// line search, then the six perturbation queries (K2).  GPisMap3.cpp:334-446.
void GPisMap3::Impl::reeval_batch(const std::vector<int>& ids, std::vector<Stage2>& st, std::vector<float>& pval,
                                  std::vector<float>& pvar) {
    const int n = (int)ids.size();
    st.assign(n, Stage2());
    pval.assign((size_t)6 * n, 0.f); pvar.assign((size_t)6 * n, 1e6f);
    if (n == 0) return;
    // both K2 batches go through the ObsGP object's page-locked staging, sized once for the larger (6 n) one
    float* q = gpo.stage_q(6 * n);
    if (!q) { fprintf(stderr, "[gpismap_amd] ObsGP staging allocation failed\n"); if (!upd_rc) upd_rc = GPIS_ERR_HIP; return; }
    std::vector<char>& front = ws_front;
    std::vector<std::array<float, 3>>& loc = ws_loc;
    front.assign(n, 0); loc.resize(n);
    // (both host loops below are pure per point: on the host threads, host_pool.h)
    pool().parallel_for(n, [&](int lo_, int hi_) {
    for (int i = lo_; i < hi_; ++i) {
        const float* pos = tree.pts[ids[i]].pos;
        float x_loc = pose_R[0] * (pos[0] - pose_tr[0]) + pose_R[1] * (pos[1] - pose_tr[1]) + pose_R[2] * (pos[2] - pose_tr[2]);
        float y_loc = pose_R[3] * (pos[0] - pose_tr[0]) + pose_R[4] * (pos[1] - pose_tr[1]) + pose_R[5] * (pos[2] - pose_tr[2]);
        float z_loc = pose_R[6] * (pos[0] - pose_tr[0]) + pose_R[7] * (pos[1] - pose_tr[1]) + pose_R[8] * (pos[2] - pose_tr[2]);
        loc[i] = {x_loc, y_loc, z_loc};
        front[i] = !(z_loc < 0.0);
        q[2 * i] = front[i] ? y_loc / z_loc : 1e30f;  // behind the camera: never queried by the reference
        q[2 * i + 1] = front[i] ? x_loc / z_loc : 1e30f;
    }
    });
    int rc = gpo.query_staged(n, stream);
    if (rc != GPIS_OK) { fprintf(stderr, "[gpismap_amd] ObsGP query failed (%d)\n", rc); if (!upd_rc) upd_rc = rc; return; }
    stat_obs_queries += n;
    const float delx = setting.delx;
    // centre answers: copied out, the staging is reused for the perturbation batch
    std::vector<float>& val = ws_val; std::vector<float>& var = ws_var;
    val.assign(gpo.staged_val(), gpo.staged_val() + n); var.assign(gpo.staged_var(), gpo.staged_var() + n);
    float* q2 = q;
    std::fill(q2, q2 + (size_t)12 * n, 1e30f);
    pool().parallel_for(n, [&](int lo_, int hi_) {
    for (int i = lo_; i < hi_; ++i) {
        if (!front[i]) continue;
        if (var[i] > setting.obs_var_thre) continue;
        float x_loc = loc[i][0], y_loc = loc[i][1], z_loc = loc[i][2];
        float rinv = (float)(1.0 / (double)z_loc);
        float rinv0 = val[i];
        float oc = occ_test(rinv, rinv0, (float)((double)z_loc * 30.0));
        if ((double)oc < -0.02) continue;
        Stage2& s = st[i];
        const float* grad = tree.pts[ids[i]].grad;
        s.grad_loc[0] = pose_R[0] * grad[0] + pose_R[1] * grad[1] + pose_R[2] * grad[2];
        s.grad_loc[1] = pose_R[3] * grad[0] + pose_R[4] * grad[1] + pose_R[5] * grad[2];
        s.grad_loc[2] = pose_R[6] * grad[0] + pose_R[7] * grad[1] + pose_R[8] * grad[2];
        float abs_oc = (float)std::fabs((double)oc);
        float dx = delx;
        float x_new[3] = {x_loc, y_loc, z_loc};
        for (int it = 0; it < 10 && (double)abs_oc > 0.02; ++it) {
            if (oc < 0) for (int d = 0; d < 3; ++d) x_new[d] += s.grad_loc[d] * dx;
            else for (int d = 0; d < 3; ++d) x_new[d] -= s.grad_loc[d] * dx;
            // The reference re-queries the ORIGINAL location (:390-393): same input, same
            // (rinv0, var) -- no new query is needed and var <= thre still holds.
            float r_new = z_loc;
            float oc_new = occ_test((float)(1.0 / (double)r_new), rinv0, (float)((double)r_new * 30.0));
            float abs_oc_new = (float)std::fabs((double)oc_new);
            if ((double)abs_oc_new < 0.02 || (double)oc < -0.02) break;
            else if ((double)(oc * oc_new) < 0.0) dx = (float)(0.5 * (double)dx);
            else dx = (float)(1.1 * (double)dx);
            abs_oc = abs_oc_new;
            oc = oc_new;
        }
        s.go = true; s.abs_oc = abs_oc;
        for (int d = 0; d < 3; ++d) s.x_new[d] = x_new[d];
        static const float pert[3][6] = {{1, -1, 0, 0, 0, 0}, {0, 0, 1, -1, 0, 0}, {0, 0, 0, 0, 1, -1}};
        for (int k = 0; k < 6; ++k) {
            float X = x_new[0] + delx * pert[0][k];
            float Y = x_new[1] + delx * pert[1][k];
            float Z = x_new[2] + delx * pert[2][k];
            q2[(size_t)12 * i + 2 * k] = Y / Z;
            q2[(size_t)12 * i + 2 * k + 1] = X / Z;
        }
    }
    });
    rc = gpo.query_staged(6 * n, stream);
    if (rc != GPIS_OK) { fprintf(stderr, "[gpismap_amd] ObsGP query failed (%d)\n", rc); if (!upd_rc) upd_rc = rc; return; }
    pval.assign(gpo.staged_val(), gpo.staged_val() + (size_t)6 * n);
    pvar.assign(gpo.staged_var(), gpo.staged_var() + (size_t)6 * n);
    stat_obs_queries += 6 * (long)n;
}

// Stage 3 for one point: the fusion arithmetic (pure) ...  :410-566
GPisMap3::Impl::ReevalOut GPisMap3::Impl::reeval_math(const FlatPoint<3>& nd, const Stage2& s, const float* pval, const float* pvar) const {
    ReevalOut out;
    if (!s.go) return out;
    const float w = (float)(1.0 / 6.0);
    const float delx = setting.delx;
    static const float pert[3][6] = {{1, -1, 0, 0, 0, 0}, {0, 0, 1, -1, 0, 0}, {0, 0, 0, 0, 1, -1}};
    float occ[6] = {-1, -1, -1, -1, -1, -1};
    float occ_mean = 0.f, r0_mean = 0.f, r0_sqr_sum = 0.f;
    float r_new = s.x_new[2];
    float var = 0.f;
    for (int i = 0; i < 6; ++i) {
        float Z = s.x_new[2] + delx * pert[2][i];
        r_new = Z;
        var = pvar[i];
        if (var > setting.obs_var_thre) break;
        float rinv0 = pval[i];
        occ[i] = occ_test((float)(1.0 / (double)r_new), rinv0, (float)((double)r_new * 30.0));
        occ_mean += w * occ[i];
        float r0 = (float)(1.0 / (double)rinv0);
        r0_sqr_sum += r0 * r0;
        r0_mean += w * r0;
    }
    if (var > setting.obs_var_thre) return out;

    const float* pos = nd.pos;
    const float* grad = nd.grad;
    float gl[3] = {(occ[0] - occ[1]) / delx, (occ[2] - occ[3]) / delx, (occ[4] - occ[5]) / delx};
    float norm_g = std::sqrt(gl[0] * gl[0] + gl[1] * gl[1] + gl[2] * gl[2]);
    if ((double)norm_g < 1e-3) {  // uncertainty increased
        out.kind = 1;
        out.noise = (float)(2.0 * (double)nd.sigx);
        out.gnoise = (float)(2.0 * (double)nd.sigg);
        return out;
    }
    float r_var = (float)((double)r0_sqr_sum / 5.0 - (double)(r0_mean * r0_mean) * 6.0 / 5.0);
    r_var /= delx;
    float noise = 100.0f, grad_noise = 1.0f;
    if ((double)norm_g > 1e-6) {
        for (int d = 0; d < 3; ++d) gl[d] = gl[d] / norm_g;
        noise = setting.min_position_noise * saturate(r_new * r_new, 1.0f, noise);
        grad_noise = saturate(std::fabs(occ_mean) + r_var, setting.min_grad_noise, grad_noise);
    } else noise = setting.min_position_noise * noise;
    const float* x_new = s.x_new;
    float dist = std::sqrt(x_new[0] * x_new[0] + x_new[1] * x_new[1] + x_new[2] * x_new[2]);
    float view_ang = std::max(-(x_new[0] * gl[0] + x_new[1] * gl[1] + x_new[2] * gl[2]) / dist, (float)1e-1);
    float view_ang2 = view_ang * view_ang;
    float view_noise = (float)((double)setting.min_position_noise * ((1.0 - (double)view_ang2) / (double)view_ang2));
    noise += view_noise + s.abs_oc;
    grad_noise = (float)((double)grad_noise + 0.1 * (double)view_noise);

    float pos_new[3], grad_new[3];
    for (int d = 0; d < 3; ++d) {
        pos_new[d] = pose_R[d] * x_new[0] + pose_R[3 + d] * x_new[1] + pose_R[6 + d] * x_new[2] + pose_tr[d];
        grad_new[d] = pose_R[d] * gl[0] + pose_R[3 + d] * gl[1] + pose_R[6 + d] * gl[2];
    }
    float noise_old = nd.sigx, grad_noise_old = nd.sigg;
    float pos_noise_sum = noise_old + noise;
    float grad_noise_sum = grad_noise_old + grad_noise;
    if ((double)grad_noise_old > 0.5 || (double)grad_noise_old > 0.6) {
        ;
    } else {
        for (int d = 0; d < 3; ++d) pos_new[d] = (noise * pos[d] + noise_old * pos_new[d]) / pos_noise_sum;
        float dist2 = (float)(0.5 * (double)std::sqrt((pos[0] - pos_new[0]) * (pos[0] - pos_new[0]) +
                                                       (pos[1] - pos_new[1]) * (pos[1] - pos_new[1]) +
                                                       (pos[2] - pos_new[2]) * (pos[2] - pos_new[2])));
        float axis[3];
        axis[0] = grad_new[1] * grad[2] - grad_new[2] * grad[1];
        axis[1] = -grad_new[0] * grad[2] + grad_new[2] * grad[0];
        axis[2] = grad_new[0] * grad[1] - grad_new[1] * grad[0];
        float ang = (float)std::acos((double)(grad_new[0] * grad[0] + grad_new[1] * grad[1] + grad_new[2] * grad[2]));
        ang = ang * noise / pos_noise_sum;
        float q[4] = {1.0f, 0.0f, 0.0f, 0.0f};
        if (ang > 1 - 6) {  // sic, GPisMap3.cpp:517
            q[0] = (float)std::cos((double)ang / 2.0);
            float sina = (float)std::sin((double)ang / 2.0);
            q[1] = axis[0] * sina; q[2] = axis[1] * sina; q[3] = axis[2] * sina;
        }
        auto Rot = quat2dcm(q);
        grad_new[0] = Rot[0] * grad[0] + Rot[1] * grad[1] + Rot[2] * grad[2];
        grad_new[1] = Rot[3] * grad[0] + Rot[4] * grad[1] + Rot[5] * grad[2];
        grad_new[2] = Rot[6] * grad[0] + Rot[7] * grad[1] + Rot[8] * grad[2];
        grad_noise = std::min((float)1.0, std::max(grad_noise * grad_noise_old / grad_noise_sum + dist2, setting.map_noise_param));
        noise = std::max((noise * noise_old / pos_noise_sum + dist2), setting.map_noise_param);
    }
    out.kind = 2;
    for (int d = 0; d < 3; ++d) { out.pos[d] = pos_new[d]; out.grad[d] = grad_new[d]; }
    out.noise = noise; out.gnoise = grad_noise;
    return out;
}

// ... and the tree mutation, in the reference's order
void GPisMap3::Impl::reeval_commit(int pid, const ReevalOut& o) {
    if (o.kind == 0) return;
    if (o.kind == 1) { tree.pts[pid].sigx = o.noise; tree.pts[pid].sigg = o.gnoise; return; }
    float old_pos[3] = {tree.pts[pid].pos[0], tree.pts[pid].pos[1], tree.pts[pid].pos[2]};   // copy: the point object is replaced below
    tree.remove_cached(old_pos, &activeSet);
    if ((double)o.noise > 1.0 && (double)o.gnoise > 0.61) return;
    int np = tree.new_point(o.pos);
    T3::InsSet ins;
    if (try_insert(np, ins) != 2) return;
    FlatPoint<3>& p = tree.pts[np];
    p.val = -setting.fbias; p.sigx = o.noise; p.sigg = o.gnoise; p.type = 1;
    for (int d = 0; d < 3; ++d) p.grad[d] = o.grad[d];
    ins.for_each([&](int c) { activeSet.insert(c); });
}

void GPisMap3::Impl::reeval_apply(int pid, const Stage2& s, const float* pval, const float* pvar) {
    const FlatPoint<3> nd = tree.pts[pid];
    reeval_commit(pid, reeval_math(nd, s, pval, pvar));
}

// ------------------------------------------------------------ updateMapPoints ----
void GPisMap3::Impl::updateMapPoints() {  // GPisMap3.cpp:258-319
    if (!has_tree || !gpo_created) return;
    UpdLap ulap;
    std::vector<int> oc;
    tree.query_clusters(tree.root, pose_tr, range_obs_max, oc, nullptr);
    if (oc.empty()) return;
    float r2 = range_obs_max * range_obs_max;
    std::vector<int> sel;  // clusters passing the range / frustum gates, in visiting order
    for (int c : oc) {
        const T3::TNode& cn = tree.nodes[c];
        const float* ct = cn.c;
        float l = cn.h;
        float sqr_range = (ct[0] - pose_tr[0]) * (ct[0] - pose_tr[0]) + (ct[1] - pose_tr[1]) * (ct[1] - pose_tr[1]) +
                          (ct[2] - pose_tr[2]) * (ct[2] - pose_tr[2]);
        if (sqr_range > (r2 + 2 * l * l)) continue;
        int within_angle = 0;  // overwritten per corner, not accumulated (:298)
        for (int i = 0; i < 8; ++i) {
            float e[3] = {(i & 1) ? cn.hi[0] : cn.lo[0], (i & 2) ? cn.lo[1] : cn.hi[1], (i & 4) ? cn.lo[2] : cn.hi[2]};
            float x_loc = pose_R[0] * (e[0] - pose_tr[0]) + pose_R[1] * (e[1] - pose_tr[1]) + pose_R[2] * (e[2] - pose_tr[2]);
            float y_loc = pose_R[3] * (e[0] - pose_tr[0]) + pose_R[4] * (e[1] - pose_tr[1]) + pose_R[5] * (e[2] - pose_tr[2]);
            float z_loc = pose_R[6] * (e[0] - pose_tr[0]) + pose_R[7] * (e[1] - pose_tr[1]) + pose_R[8] * (e[2] - pose_tr[2]);
            if (z_loc > 0) {
                float xv = x_loc / z_loc, yv = y_loc / z_loc;
                within_angle = int((xv > u_obs_limit[0]) && (xv < u_obs_limit[1]) && (yv > v_obs_limit[0]) && (yv < v_obs_limit[1]));
            }
        }
        if (within_angle == 0) continue;
        sel.push_back(c);
    }
    if (sel.empty()) return;

    // speculative batch over every point currently stored in the selected clusters
    std::vector<int>& ids = ws_ids;
    ids.clear();
    for (int c : sel) tree.all_points(c, ids);
    std::vector<Stage2>& st = ws_st;
    std::vector<float>& pval = ws_pval; std::vector<float>& pvar = ws_pvar;
    ulap("reEvalPoints: select");
    reeval_batch(ids, st, pval, pvar);
    ulap("reEvalPoints: K2 batches");
    launch_pixel_batch();      // the new-pixel batch runs on the device while the host replays the re-evaluation below
    std::vector<int>& slot = ws_slot;
    slot.assign(tree.pts.size(), -1);
    for (size_t i = 0; i < ids.size(); ++i) slot[ids[i]] = (int)i;
    // the fusion arithmetic of every point of the batch on the host threads (a point's outcome depends on its own state only,
    // and no commit below touches another stored point's data); the tree replay then applies the outcomes in order
    std::vector<ReevalOut>& outs = ws_outs;
    outs.resize(ids.size());
    pool().parallel_for((int)ids.size(), [&](int lo, int hi) {
        for (int i = lo; i < hi; ++i) outs[i] = reeval_math(tree.pts[ids[i]], st[i], &pval[(size_t)6 * i], &pvar[(size_t)6 * i]);
    });
    ulap("reEvalPoints: fusion arithmetic");

    // replay in the reference's order; node lists are fetched lazily per cluster (:308-311)
    std::vector<int>& nodes = ws_nodes; std::vector<int>& late = ws_late;
#ifdef GPIS_INSTRUMENT
    float late_ms = 0.f; int late_batches = 0;
#endif
    for (int c : sel) {
        nodes.clear();
        tree.all_points(c, nodes);
        late.clear();
        for (int pid : nodes) if (pid >= (int)slot.size() || slot[pid] < 0) late.push_back(pid);
        std::vector<Stage2> lst;
        std::vector<float> lval, lvar;
        if (!late.empty()) {
#ifdef GPIS_INSTRUMENT
            auto t0 = std::chrono::steady_clock::now();
#endif
            reeval_batch(late, lst, lval, lvar); stat_late += (long)late.size();
#ifdef GPIS_INSTRUMENT
            late_ms += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count(); ++late_batches;
#endif
        }
        size_t li = 0;
        for (int pid : nodes) {
            if (pid < (int)slot.size() && slot[pid] >= 0) {
                reeval_commit(pid, outs[slot[pid]]);
            } else {
                reeval_apply(pid, lst[li], &lval[6 * li], &lvar[6 * li]);
                ++li;
            }
        }
    }
    ulap("reEvalPoints: apply");
#ifdef GPIS_INSTRUMENT
    fprintf(stderr, "[upd] late re-evaluations: %d batches, %.2f ms of the apply lap\n", late_batches, late_ms);
#endif
}

// ------------------------------------------------------------------ evalPoints ----
static const float kPert3[3][6] = {{1, -1, 0, 0, 0, 0}, {0, 0, 1, -1, 0, 0}, {0, 0, 0, 0, 1, -1}};

// The speculative K2 batch of evalPoints() -- centre + 6 perturbations per valid pixel -- depends on the frame's ObsGP and
// pixels only, not on the tree: it is built and ENQUEUED (own stream, second staging set of the ObsGP object) right after
// the re-evaluation batches of the stored points, runs on the device while the host replays that re-evaluation, and
// evalPoints() collects it.  (Issued BEFORE the re-evaluation batches it delayed them by more than it saved: measured.)
void GPisMap3::Impl::launch_pixel_batch() {
    if (batch_launched) return;
    batch_launched = true;
    batch_inflight = false;
    if (obs_numdata < 1) return;
    const float delx = setting.delx;
    const int n = obs_numdata;
    UpdLap ulap;
    float* q = gpo.stage_qb(7 * n);   // page-locked staging of the ObsGP object: filled in place, answers read in place
    if (!q) { fprintf(stderr, "[gpismap_amd] ObsGP staging allocation failed\n"); if (!upd_rc) upd_rc = GPIS_ERR_HIP; return; }
    pool().run_parts(std::max(1, pix_parts), [&](int p) {
        const int k0 = pix_parts ? pix_off[(size_t)p] : 0, k1 = pix_parts ? pix_off[(size_t)p + 1] : n;
        for (int k = k0; k < k1; ++k) {
            const float* xl = &obs_valid_xyzlocal[3 * (size_t)k];
            q[(size_t)14 * k] = obs_valid_v[k];
            q[(size_t)14 * k + 1] = obs_valid_u[k];
            for (int i = 0; i < 6; ++i) {
                float X = xl[0] + delx * kPert3[0][i];
                float Y = xl[1] + delx * kPert3[1][i];
                float Z = xl[2] + delx * kPert3[2][i];
                q[(size_t)14 * k + 2 + 2 * i] = Y / Z;
                q[(size_t)14 * k + 3 + 2 * i] = X / Z;
            }
        }
    });
    ulap("evalPoints: build queries");
    int rc = gpo.query_staged_b_async(7 * n, batch_stream);
    if (rc != GPIS_OK) { fprintf(stderr, "[gpismap_amd] ObsGP query failed (%d)\n", rc); if (!upd_rc) upd_rc = rc; return; }
    batch_inflight = true;
}

// The data of a new surface point from its pixel's seven ObsGP answers (GPisMap3.cpp:624-690): pure per pixel.
GPisMap3::Impl::NewPoint GPisMap3::Impl::pixel_point(const float* pv, const float* pr, const float* xl) const {
    NewPoint o;
    const float w = (float)(1.0 / 6.0);
    const float delx = setting.delx;
    const float (&pert)[3][6] = kPert3;
    float occ[6] = {-1, -1, -1, -1, -1, -1};
    float occ_mean = 0.f;
    float v = pr[0];
    for (int i = 0; i < 6; ++i) {
        float Z = xl[2] + delx * pert[2][i];
        v = pr[1 + i];
        if (v > setting.obs_var_thre) break;
        occ[i] = occ_test((float)(1.0 / (double)Z), pv[1 + i], (float)((double)Z * 30.0));
        occ_mean += w * occ[i];
    }
    if (v > setting.obs_var_thre) return o;      // keep == false: the point is taken out of the tree again
    o.keep = true;
    float noise = 100.0f, grad_noise = 1.00f;
    float g[3] = {(occ[0] - occ[1]) / delx, (occ[2] - occ[3]) / delx, (occ[4] - occ[5]) / delx};
    float norm_grad = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
    if ((double)norm_grad > 1e-6) {
        norm_grad = std::sqrt(norm_grad);
        float gx = g[0] / norm_grad, gy = g[1] / norm_grad, gz = g[2] / norm_grad;
        g[0] = pose_R[0] * gx + pose_R[3] * gy + pose_R[6] * gz;
        g[1] = pose_R[1] * gx + pose_R[4] * gy + pose_R[7] * gz;
        g[2] = pose_R[2] * gx + pose_R[5] * gy + pose_R[8] * gz;
        float dist = std::sqrt(xl[0] * xl[0] + xl[1] * xl[1] + xl[2] * xl[2]);
        noise = setting.min_position_noise * saturate(dist, 1.0f, noise);
        grad_noise = saturate(std::fabs(occ_mean), setting.min_grad_noise, grad_noise);
        float view_ang = std::max(-(xl[0] * gx + xl[1] * gy + xl[2] * gz) / dist, (float)1e-1);
        float view_ang2 = view_ang * view_ang;
        float view_noise = (float)((double)setting.min_position_noise * ((1.0 - (double)view_ang2) / (double)view_ang2));
        noise += view_noise;
    }
    o.noise = noise; o.gnoise = grad_noise;
    for (int d = 0; d < 3; ++d) o.grad[d] = g[d];
    return o;
}

void GPisMap3::Impl::evalPoints() {  // GPisMap3.cpp:580-696
    if (!has_tree || obs_numdata < 1) return;
    const int n = obs_numdata;
    launch_pixel_batch();              // (not yet issued when there was nothing to re-evaluate)
    UpdLap ulap;
    if (!batch_inflight) return;       // (launch_pixel_batch reported why)
    batch_inflight = false;
    int rc = gpo.wait_b();
    if (rc != GPIS_OK) { fprintf(stderr, "[gpismap_amd] ObsGP query failed (%d)\n", rc); if (!upd_rc) upd_rc = rc; return; }
    stat_obs_queries += 7 * (long)n;
    const float* val = gpo.staged_val_b();
    const float* var = gpo.staged_var_b();
    ulap("evalPoints: K2 batch");
    // Pre-pass over all pixels on the host threads (column ranges of preprocData; the tree is not touched):
    //  * the variance gate of the centre query (witness node kGated);
    //  * "not new" against the tree as it stands now, each answer with its witness (leaf, point id): nine pixels out of ten
    //    end there because of a point that was in the map before this pass.  The ordered pass below accepts such an answer
    //    only while the witness still holds and asks the tree again otherwise (flat_tree.h: is_not_new_frozen) -- round 2's
    //    plain pre-filter was not exact, because an insert can subdivide the witness leaf;
    //  * for the pixels that are new as of now, the data of the point they would become (pure arithmetic of the pixel's
    //    seven answers).  A pre-computed "new" itself is never trusted: the ordered pass asks the tree.
    constexpr int kGated = -2;
    pre_wnode.resize((size_t)n); pre_wpt.resize((size_t)n); pre_new.resize((size_t)n);
    pool().run_parts(std::max(1, pix_parts), [&](int p) {
        const int k0 = pix_parts ? pix_off[(size_t)p] : 0, k1 = pix_parts ? pix_off[(size_t)p + 1] : n;
        int cell = -1;
        for (int k = k0; k < k1; ++k) {
            if (var[(size_t)7 * k] > setting.obs_var_thre) { pre_wnode[k] = kGated; continue; }
            if (!tree.is_not_new_frozen(&obs_valid_xyzglobal[3 * (size_t)k], &cell, &pre_wnode[k], &pre_wpt[k]))
                pre_new[k] = pixel_point(&val[(size_t)7 * k], &var[(size_t)7 * k], &obs_valid_xyzlocal[3 * (size_t)k]);
        }
    });
    ulap("evalPoints: frozen pre-pass");

#ifdef GPIS_INSTRUMENT
    long dbg_full = 0, dbg_lost = 0, dbg_ins = 0;
#endif
    for (int k = 0; k < n; ++k) {
        const int w = pre_wnode[k];
        if (w == kGated) continue;
        // (the reference allocates the node before IsNotNew and discards it when the test says "not new", GPisMap3.cpp:611-623:
        // no side effect, so the nine pixels out of ten that end here never take a point object)
        if (tree.witness_holds(w, pre_wpt[k])) continue;
#ifdef GPIS_INSTRUMENT
        ++dbg_full; if (w >= 0) ++dbg_lost;
#endif
        if (tree.is_not_new_cached(&obs_valid_xyzglobal[3 * (size_t)k])) continue;
        int pid = tree.new_point(&obs_valid_xyzglobal[3 * (size_t)k]);
        T3::InsSet ins;
        if (try_insert(pid, ins, true) != 2) continue;
#ifdef GPIS_INSTRUMENT
        ++dbg_ins;
#endif
        // (a pixel whose witness was lost on the way has no pre-computed data)
        const NewPoint np = (w < 0) ? pre_new[k] : pixel_point(&val[(size_t)7 * k], &var[(size_t)7 * k], &obs_valid_xyzlocal[3 * (size_t)k]);
        if (!np.keep) {
            tree.remove(tree.root, tree.pts[pid].pos, nullptr);
            continue;
        }
        FlatPoint<3>& p = tree.pts[pid];
        p.val = -setting.fbias; p.sigx = np.noise; p.sigg = np.gnoise; p.type = 1;
        for (int d = 0; d < 3; ++d) p.grad[d] = np.grad[d];
        ins.for_each([&](int c) { activeSet.insert(c); });
    }
    ulap("evalPoints: insert pass");
#ifdef GPIS_INSTRUMENT
    fprintf(stderr, "[upd]   pixels %d: asked the tree %ld (witness lost %ld), stored %ld\n", n, dbg_full, dbg_lost, dbg_ins);
#endif
}

// -------------------------------------------------------------------- updateGPs ----
// ---- frame records (one process per GPU, host logic once: see Impl::export_frames) -------------------------------------------
namespace {
constexpr unsigned kFrameMagic = 0x46335047u;      // "GP3F"
struct FrameHeader {
    unsigned magic, version;
    int rc, device_gather, total, nothing;       // nothing = 1: the lead's update() returned before it touched the map (no valid pixel ...)
    unsigned long long np, n_slot_ops, n_cell_pts, n_cr, n_desc, n_counts, n_jobs, n_ids, n_ent, n_anc;
};
struct FrameWriter {
    std::vector<char>& b;
    template <class T> void put(const T* p, size_t n) { const size_t o = b.size(); b.resize(o + ((sizeof(T) * n + 7) & ~(size_t)7)); if (n) std::memcpy(b.data() + o, p, sizeof(T) * n); }
};
struct FrameReader {
    const char* p; size_t left; bool ok = true;
    template <class T> const T* take(size_t n) {
        const size_t need = (sizeof(T) * n + 7) & ~(size_t)7;
        if (!ok || need > left) { ok = false; return nullptr; }
        const T* r = reinterpret_cast<const T*>(p); p += need; left -= need; return r;
    }
};
}  // namespace

void GPisMap3::Impl::updateGPs() {  // GPisMap3.cpp:698-792 -> K6 + K3
    UpdLap ulap;
    shard_jobs.clear();
    table_pending = false;
    deferred_train = false; deferred_jobs.clear(); deferred_ids.clear();
    // the frame's record for worker processes (export_frames): written once the frame's decisions are complete
    auto write_record = [&](int rc_, size_t np_, const std::vector<int>* cell_pts, const std::vector<int>* cr_, const std::vector<int>* desc_,
                            int total_, const std::vector<int>* counts_, const std::vector<TrainJob>* jobs_, const std::vector<int>* ids_) {
        static const std::vector<int> none;
        static const std::vector<TrainJob> nojobs;
        std::vector<ClusterEntry> ent; std::vector<AncestorEntry> anc;
        collect_cluster_entries(ent, anc);
        FrameHeader h;
        std::memset(&h, 0, sizeof(h));
        h.magic = kFrameMagic; h.version = 1; h.rc = rc_; h.device_gather = device_gather ? 1 : 0; h.total = total_;
        const std::vector<int>& cp = cell_pts ? *cell_pts : none; const std::vector<int>& crr = cr_ ? *cr_ : none;
        const std::vector<int>& ds_ = desc_ ? *desc_ : none; const std::vector<int>& cn = counts_ ? *counts_ : none;
        const std::vector<TrainJob>& jb = jobs_ ? *jobs_ : nojobs; const std::vector<int>& idv = ids_ ? *ids_ : none;
        h.np = np_; h.n_slot_ops = slot_ops.size(); h.n_cell_pts = cp.size(); h.n_cr = crr.size(); h.n_desc = ds_.size();
        h.n_counts = cn.size(); h.n_jobs = jb.size(); h.n_ids = idv.size(); h.n_ent = ent.size(); h.n_anc = anc.size();
        frame_rec.clear();
        FrameWriter w{frame_rec};
        w.put(&h, 1);
        w.put(slot_ops.data(), slot_ops.size());
        w.put(mirror_soa.data(), np_ ? 9 * np_ : 0);
        w.put(cp.data(), cp.size()); w.put(crr.data(), crr.size()); w.put(ds_.data(), ds_.size()); w.put(cn.data(), cn.size());
        w.put(jb.data(), jb.size());
        std::vector<int> owners(jb.size(), 0);
        for (size_t j = 0; j < jb.size() && j < shard_jobs.size(); ++j) owners[j] = shard_jobs[j].owner;
        w.put(owners.data(), owners.size());
        w.put(idv.data(), idv.size());
        w.put(ent.data(), ent.size()); w.put(anc.data(), anc.size());
    };
    bool record_written = false;
    T3::Set updateSet(activeSet);
    std::vector<int> qs;
    for (int a : activeSet) {
        qs.clear();
        tree.query_clusters(tree.root, tree.nodes[a].c, kRtimes * tree.nodes[a].h, qs, nullptr);
        for (int c : qs) updateSet.insert(c);
    }
    ulap("updateGPs: neighbour sets");
    // the previous frame's batch: joined before any model slot is released or allocated -- but as late as possible: what only
    // reads the tree (the mirror image, the cell lists) is done beside it when update() is pipelined
    auto join_previous = [&]() {
        finish_training();
        ulap("updateGPs: join");
        for (int m : tree.released_models) { each_store([&](OnGPISStore& st) { st.release_slot(m); }); if (export_frames) slot_ops.push_back(~m); }
        tree.released_models.clear();
    };
    if (updateSet.empty()) join_previous();
    else {
        std::vector<int> todo(updateSet.begin(), updateSet.end());
        std::sort(todo.begin(), todo.end());
        std::vector<TrainJob> jobs;
        std::vector<int> ids, res;
        T3::CellLists& cell_lists = ws_cells;   // the points of every touched cell listed once for the whole batch (flat_tree.h)
        cell_lists.reset(tree.nodes.size());
        store.defer_finish = pipeline && shard_world == 1 && peers.empty();
        int rc = GPIS_OK;
        // mirror of the map points in HBM: 9 SoA rows indexed by point id (the range gather below reads the positions)
        {
            size_t np = tree.pts.size();
            mirror_soa.resize(9 * np);
            float* soa = mirror_soa.data();
            pool().parallel_for((int)np, [&](int lo, int hi) {
                for (size_t i = (size_t)lo; i < (size_t)hi; ++i) {
                    const FlatPoint<3>& p = tree.pts[i];
                    for (int d = 0; d < 3; ++d) { soa[d * np + i] = p.pos[d]; soa[(3 + d) * np + i] = p.grad[d]; }
                    soa[6 * np + i] = p.val; soa[7 * np + i] = p.sigx; soa[8 * np + i] = p.sigg;
                }
            }, 4096);
            ulap("updateGPs: point mirror");
        }
        const size_t np_mirror = tree.pts.size();
        if (!device_gather) { join_previous(); rc = store.upload_points(mirror_soa.data(), (int)np_mirror, train_stream); }
        std::vector<int> cr, desc, counts, cl_of;   // (device gather; kept for the workers' own K6 pass below)
        int total = 0;
        if (rc == GPIS_OK && device_gather) {
            // K6 range part on the device: the host only names the cells (traversal order) and lists each touched cell once
            // the cell walks of the frame's clusters on the host threads (read-only, independent), the listing in cluster order
            std::vector<std::vector<int>>& touched = ws_touched;
            if (touched.size() < todo.size()) touched.resize(todo.size());
            pool().parallel_for((int)todo.size(), [&](int lo_, int hi_) {
                for (int i = lo_; i < hi_; ++i) {
                    const int c = todo[i];
                    touched[i].clear();
                    tree.query_clusters(tree.root, tree.nodes[c].c, tree.nodes[c].h * kRtimes, touched[i], nullptr);
                }
            }, 16);
            {
                // ... and so does listing every touched cell's points (a subtree walk per cell): the distinct cells in the order the
                // serial pass would first meet them, their point lists in parallel, laid out back to back by a prefix sum --
                // cell_lists then holds exactly what range_cells_from() would have built on the fly
                std::vector<int> order;
                for (size_t ti = 0; ti < todo.size(); ++ti)
                    for (int cell : touched[ti]) if (cell_lists.begin[cell] == -1) { cell_lists.begin[cell] = -2; order.push_back(cell); }
                std::vector<std::vector<int>>& plist = ws_plist;
                if (plist.size() < order.size()) plist.resize(order.size());
                pool().parallel_for((int)order.size(), [&](int lo_, int hi_) {
                    for (int i = lo_; i < hi_; ++i) { plist[i].clear(); tree.all_points(order[i], plist[i]); }
                }, 16);
                size_t tot = cell_lists.pts.size();
                for (size_t i = 0; i < order.size(); ++i) tot += plist[i].size();
                cell_lists.pts.reserve(tot);
                for (size_t i = 0; i < order.size(); ++i) {
                    cell_lists.begin[order[i]] = (int)cell_lists.pts.size();
                    cell_lists.pts.insert(cell_lists.pts.end(), plist[i].begin(), plist[i].end());
                    cell_lists.end[order[i]] = (int)cell_lists.pts.size();
                }
            }
            for (size_t ti = 0; ti < todo.size(); ++ti) {
                const int c = todo[ti];
                const float h = tree.nodes[c].h * kRtimes;
                const int e0 = (int)cr.size() / 2;
                const int capc = tree.range_cells_from(touched[ti], cell_lists, cr);
                if (capc == 0) { cr.resize((size_t)2 * e0); continue; }      // no point in any touched cell: the reference's empty result
                union { float f; int i; } u;
                desc.push_back(e0); desc.push_back((int)cr.size() / 2 - e0); desc.push_back(total);
                for (int d = 0; d < 3; ++d) { u.f = tree.nodes[c].c[d]; desc.push_back(u.i); }
                u.f = h * h; desc.push_back(u.i); desc.push_back(0);
                cl_of.push_back(c);
                total += capc;
            }
            counts.assign(2 * cl_of.size(), 0);
            ulap("updateGPs: cell lists");
            join_previous();
            rc = store.upload_points(mirror_soa.data(), (int)np_mirror, train_stream);
            if (rc == GPIS_OK && !cl_of.empty())
                rc = store.gather_ranges(cell_lists.pts.data(), (int)cell_lists.pts.size(), cr.data(), (int)cr.size() / 2, desc.data(),
                                         (int)cl_of.size(), total, counts.data(), train_stream);
            for (size_t i = 0; i < cl_of.size() && rc == GPIS_OK; ++i) {
                if (counts[2 * i] == 0) continue;
                const int c = cl_of[i];
                if (tree.nodes[c].model < 0) tree.nodes[c].model = new_slot_all();
                TrainJob j;
                j.model = tree.nodes[c].model; j.off = desc[8 * i + 2]; j.n = counts[2 * i]; j.ng = counts[2 * i + 1];
                jobs.push_back(j);
            }
        } else if (rc == GPIS_OK) {
            for (int c : todo) {
                res.clear();
                tree.query_range_cells(tree.nodes[c].c, tree.nodes[c].h * kRtimes, cell_lists, res);
                if (res.empty()) continue;
                int ng = 0;
                for (int pid : res) {  // OnGPIS.cpp:122-125
                    const FlatPoint<3>& p = tree.pts[pid];
                    bool tiny = ((double)std::fabs(p.grad[0]) < 1e-6) && ((double)std::fabs(p.grad[1]) < 1e-6) && ((double)std::fabs(p.grad[2]) < 1e-6);
                    if (!(((double)p.sigg > 0.1001) || tiny)) ++ng;
                }
                if (tree.nodes[c].model < 0) tree.nodes[c].model = new_slot_all();
                TrainJob j;
                j.model = tree.nodes[c].model; j.off = (int)ids.size(); j.n = (int)res.size(); j.ng = ng;
                jobs.push_back(j);
                ids.insert(ids.end(), res.begin(), res.end());
            }
        }
        ulap("updateGPs: range queries");
        auto train = [&](const std::vector<TrainJob>& js) { return device_gather ? store.train_batch_dev(js, train_stream) : store.train_batch(js, ids, train_stream); };
        if (!jobs.empty() || rc != GPIS_OK) {
            if (shard_world > 1) {
                // Greedy longest-processing-time partition of the frame's clusters by their K^3 factorisation cost
                // (ties by job order): every rank computes the same owners and trains only its own share.
                std::vector<int> ord(jobs.size());
                for (size_t i = 0; i < ord.size(); ++i) ord[i] = (int)i;
                auto cost = [&](int j) { const double K = jobs[j].n + 3.0 * jobs[j].ng; return K * K * K; };
                std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return cost(a) > cost(b); });
                std::vector<double> load(shard_world, 0.0);
                std::vector<int> owner(jobs.size(), 0);
                for (int j : ord) {
                    int best = 0;
                    for (int r = 1; r < shard_world; ++r) if (load[r] < load[best]) best = r;
                    owner[j] = best; load[best] += cost(j);
                }
                std::vector<TrainJob> mine;
                for (size_t j = 0; j < jobs.size(); ++j) {
                    shard_jobs.push_back({jobs[j].model, jobs[j].n, jobs[j].ng, owner[j]});
                    if (owner[j] == shard_rank) mine.push_back(jobs[j]);
                }
                if (!peers.empty()) {
                    // Lead mode: every worker gets the point mirror, runs the K6 range pass on its own device (same cell lists, same
                    // offsets: its id buffer then holds what the lead's holds) and trains its share -- each on its own host thread,
                    // beside the lead's share.
                    std::vector<int> wrc(1 + peers.size(), GPIS_OK);
                    std::vector<std::thread> th;
                    const int rc0 = rc;       // (the lead's rc is written below while the workers run: they look at a copy)
                    for (size_t r = 1; r <= peers.size(); ++r)
                        th.emplace_back([&, r, rc0] {
                            Impl& w = *peers[r - 1]->impl();
                            DeviceScope ds(w.device);
                            // (job list and pending table as on the lead, failed frame or not: the exchange that follows builds every
                            // rank's table from the lead's index either way)
                            w.shard_jobs = shard_jobs; w.table_pending = true; w.has_tree = true; w.upd_rc = 0;
                            if (rc0 != GPIS_OK) return;
                            std::vector<TrainJob> wj;
                            for (size_t j = 0; j < jobs.size(); ++j) if (owner[j] == (int)r) wj.push_back(jobs[j]);
                            int q = w.store.upload_points(mirror_soa.data(), (int)np_mirror, w.train_stream);
                            if (q == GPIS_OK && device_gather && !cl_of.empty()) {
                                std::vector<int> wcounts(counts.size(), 0);
                                q = w.store.gather_ranges(cell_lists.pts.data(), (int)cell_lists.pts.size(), cr.data(), (int)cr.size() / 2, desc.data(),
                                                          (int)cl_of.size(), total, wcounts.data(), w.train_stream);
                                if (q == GPIS_OK && wcounts != counts) q = GPIS_ERR_STATE;
                            }
                            if (q == GPIS_OK && !wj.empty()) q = device_gather ? w.store.train_batch_dev(wj, w.train_stream) : w.store.train_batch(wj, ids, w.train_stream);
                            wrc[r] = q;
                            if (q != GPIS_OK) w.upd_rc = q;
                        });
                    if (rc == GPIS_OK && !mine.empty()) rc = train(mine);
                    for (auto& t : th) t.join();
                    for (size_t r = 1; r < wrc.size(); ++r) if (wrc[r] != GPIS_OK && rc == GPIS_OK) rc = wrc[r];
                } else if (export_frames) {
                    // one process per GPU, this is the lead: the record first, the own share when the caller says so (gpis3_train_deferred)
                    write_record(rc, np_mirror, &cell_lists.pts, &cr, &desc, total, &counts, &jobs, &ids);
                    record_written = true;
                    if (rc == GPIS_OK && !mine.empty()) { deferred_jobs = mine; deferred_ids = ids; deferred_dev = device_gather; deferred_train = true; }
                } else if (rc == GPIS_OK && !mine.empty()) rc = train(mine);
                table_pending = true;
            } else if (rc == GPIS_OK) rc = train(jobs);
            if (rc != GPIS_OK) { fprintf(stderr, "[gpismap_amd] OnGPIS training failed (%d)\n", rc); if (!upd_rc) upd_rc = rc; }
            stat_clusters_trained += (long)jobs.size();
            ulap("updateGPs: train_batch");
        }
    }
    activeSet.clear();
    if (export_frames && !record_written) write_record(upd_rc, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr);   // (a frame without training: slot operations and the table only)
    slot_ops.clear();
    if (!table_pending) {
        build_cluster_table();
        // (nothing was trained, but cells may have come or gone: the workers' tables follow the lead's index too)
        for (GPisMap3* q : peers) { Impl& w = *q->impl(); DeviceScope ds(w.device); w.has_tree = true; w.build_cluster_table(); if (w.upd_rc && !upd_rc) upd_rc = w.upd_rc; }
    }
    ulap("updateGPs: cluster table");
}

// The lead process' own share of a frame whose record was exported (gpis3_train_deferred): what update() would have done in place.
int GPisMap3::Impl::train_deferred() {
    if (!deferred_train) return GPIS_OK;
    deferred_train = false;
    int rc = deferred_dev ? store.train_batch_dev(deferred_jobs, train_stream) : store.train_batch(deferred_jobs, deferred_ids, train_stream);
    deferred_jobs.clear(); deferred_ids.clear();
    if (rc != GPIS_OK) { fprintf(stderr, "[gpismap_amd] OnGPIS training failed (%d)\n", rc); if (!upd_rc) upd_rc = rc; }
    return rc;
}

// A worker process' frame: the lead's record instead of a replay (Impl::export_frames).  Everything is checked before it is
// used -- a record that does not fit this store (another slot sequence, counts that the device pass does not reproduce) is
// refused with GPIS_ERR_STATE and leaves the map's models as they were.
int GPisMap3::Impl::apply_frame(const char* buf, size_t bytes) {
    FrameReader rd{buf, bytes};
    const FrameHeader* hp = rd.take<FrameHeader>(1);
    if (!hp || hp->magic != kFrameMagic || hp->version != 1) return GPIS_ERR_ARG;
    const FrameHeader h = *hp;
    const int* ops = rd.take<int>((size_t)h.n_slot_ops);
    const float* soa = rd.take<float>(h.np ? (size_t)(9 * h.np) : 0);
    const int* cell_pts = rd.take<int>((size_t)h.n_cell_pts);
    const int* cr = rd.take<int>((size_t)h.n_cr);
    const int* desc = rd.take<int>((size_t)h.n_desc);
    const int* counts = rd.take<int>((size_t)h.n_counts);
    const TrainJob* jobs = rd.take<TrainJob>((size_t)h.n_jobs);
    const int* owners = rd.take<int>((size_t)h.n_jobs);
    const int* ids = rd.take<int>((size_t)h.n_ids);
    const ClusterEntry* ent = rd.take<ClusterEntry>((size_t)h.n_ent);
    const AncestorEntry* anc = rd.take<AncestorEntry>((size_t)h.n_anc);
    if (!rd.ok || h.n_desc % 8 != 0 || h.n_counts != h.n_desc / 4 || h.n_cr % 2 != 0) return GPIS_ERR_ARG;
    {   // the index structures the device pass will follow: every range inside its array (a damaged record is refused, not walked)
        const long long ncr = (long long)(h.n_cr / 2), ncp = (long long)h.n_cell_pts, npts = (long long)h.np;
        for (size_t i = 0; i < h.n_cell_pts; ++i) if (cell_pts[i] < 0 || cell_pts[i] >= npts) return GPIS_ERR_ARG;
        for (long long i = 0; i < ncr; ++i) if (cr[2 * i] < 0 || cr[2 * i] > cr[2 * i + 1] || cr[2 * i + 1] > ncp) return GPIS_ERR_ARG;
        long long cap_sum = 0;
        for (size_t i = 0; i < h.n_desc / 8; ++i) {
            const long long e0 = desc[8 * i], ne = desc[8 * i + 1], off = desc[8 * i + 2];
            if (e0 < 0 || ne < 0 || e0 + ne > ncr || off != cap_sum) return GPIS_ERR_ARG;
            for (long long e = e0; e < e0 + ne; ++e) cap_sum += cr[2 * e + 1] - cr[2 * e];
        }
        if (h.total < 0 || cap_sum != (long long)h.total) return GPIS_ERR_ARG;
        for (size_t j = 0; j < h.n_jobs; ++j)
            if (jobs[j].n <= 0 || jobs[j].ng < 0 || jobs[j].ng > jobs[j].n || jobs[j].off < 0 ||
                (long long)jobs[j].off + jobs[j].n > (h.device_gather ? (long long)h.total : (long long)h.n_ids)) return GPIS_ERR_ARG;
        for (size_t i = 0; i < h.n_ids; ++i) if (ids[i] < 0 || ids[i] >= npts) return GPIS_ERR_ARG;
        for (size_t i = 0; i < h.n_ent; ++i) if (ent[i].parent < -1 || ent[i].parent >= (int)h.n_anc) return GPIS_ERR_ARG;
        for (size_t i = 0; i < h.n_anc; ++i) if (anc[i].parent < -1 || anc[i].parent >= (int)h.n_anc) return GPIS_ERR_ARG;
    }
    stat_host_replays = 0;
    if (h.nothing) { shard_jobs.clear(); table_pending = false; upd_rc = 0; return GPIS_OK; }
    stat_deferred_inverses = 0;
    upd_rc = 0;
    shard_jobs.clear();
    table_pending = false;
    (void)finish_training();
    // the slot operations, in the lead's order: this store must hand out the ids the lead's did
    for (size_t i = 0; i < h.n_slot_ops; ++i) {
        if (ops[i] < 0) store.release_slot(~ops[i]);
        else if (store.new_slot() != ops[i]) { fprintf(stderr, "[gpismap_amd] apply_frame: model slots diverged from the lead's\n"); upd_rc = GPIS_ERR_STATE; return GPIS_ERR_STATE; }
    }
    remote_index = true; has_tree = true;
    frame_ent.assign(ent, ent + h.n_ent); frame_anc.assign(anc, anc + h.n_anc);
    int rc = h.rc;
    std::vector<TrainJob> mine;
    for (size_t j = 0; j < h.n_jobs; ++j) {
        if (owners[j] < 0 || owners[j] >= shard_world) return GPIS_ERR_ARG;
        shard_jobs.push_back({jobs[j].model, jobs[j].n, jobs[j].ng, owners[j]});
        if (owners[j] == shard_rank) mine.push_back(jobs[j]);
    }
    if (h.n_jobs > 0) {
        if (rc == GPIS_OK) rc = store.upload_points(soa, (int)h.np, train_stream);
        const int ncl = (int)(h.n_desc / 8);
        if (rc == GPIS_OK && h.device_gather && ncl > 0) {
            std::vector<int> wcounts((size_t)h.n_counts, 0);
            rc = store.gather_ranges(cell_pts, (int)h.n_cell_pts, cr, (int)(h.n_cr / 2), desc, ncl, h.total, wcounts.data(), train_stream);
            if (rc == GPIS_OK && std::memcmp(wcounts.data(), counts, sizeof(int) * wcounts.size()) != 0) rc = GPIS_ERR_STATE;   // (this device's K6 pass must reproduce the lead's)
        }
        if (rc == GPIS_OK && !mine.empty()) {
            store.defer_finish = false;
            if (h.device_gather) rc = store.train_batch_dev(mine, train_stream);
            else { std::vector<int> idv(ids, ids + h.n_ids); rc = store.train_batch(mine, idv, train_stream); }
        }
        stat_clusters_trained += (long)h.n_jobs;
        table_pending = true;
    }
    if (rc != GPIS_OK) { fprintf(stderr, "[gpismap_amd] apply_frame: training failed (%d)\n", rc); if (!upd_rc) upd_rc = rc; }
    if (!table_pending) build_cluster_table();
    return upd_rc;
}

int GPisMap3::Impl::finish_training() {
    if (!store.train_pending()) return GPIS_OK;
    const int rc = store.train_finish();
    if (rc != GPIS_OK) {
        fprintf(stderr, "[gpismap_amd] OnGPIS training failed (%d)\n", rc);
        if (!upd_rc) upd_rc = rc;
        if (!table_pending) build_cluster_table();   // the batch was dropped: its cells have no GP any more
    }
    return rc;
}

void GPisMap3::Impl::collect_cluster_entries(std::vector<ClusterEntry>& ent, std::vector<AncestorEntry>& anc) const {
    const T3& tree = lead ? lead->tree : this->tree;    // (a device worker answers test() from the lead's index; its model slots are the lead's)
    ent.clear(); anc.clear();
    if (!(lead ? lead->has_tree : has_tree)) return;
    // cluster table for test(): every non-empty cluster cell in traversal order
    std::vector<int> cl;
    tree.all_clusters(cl);
    ent.resize(cl.size());
    // ancestor chains (up to the root the map holds), shared between sibling cells
    std::unordered_map<int, int> anc_of;
    std::function<int(int)> anc_index = [&](int node) -> int {
        if (node < 0) return -1;
        auto it = anc_of.find(node);
        if (it != anc_of.end()) return it->second;
        const T3::TNode& a = tree.nodes[node];
        int up = (node == tree.root) ? -1 : anc_index(a.par);
        AncestorEntry e;
        for (int d = 0; d < 3; ++d) { e.lo[d] = a.lo[d]; e.hi[d] = a.hi[d]; }
        e.parent = up;
        anc.push_back(e);
        anc_of[node] = (int)anc.size() - 1;
        return (int)anc.size() - 1;
    };
    for (size_t i = 0; i < cl.size(); ++i) {
        const T3::TNode& t = tree.nodes[cl[i]];
        for (int d = 0; d < 3; ++d) { ent[i].c[d] = t.c[d]; ent[i].lo[d] = t.lo[d]; ent[i].hi[d] = t.hi[d]; }
        ent[i].model = t.model;          // (the slot; build_cluster_table keeps it only where this rank's store holds a trained model)
        ent[i].parent = anc_index(t.par);
    }
}

void GPisMap3::Impl::build_cluster_table() {
    table_pending = false;
    std::vector<ClusterEntry> ent;
    std::vector<AncestorEntry> anc;
    if (remote_index) { ent = frame_ent; anc = frame_anc; }      // (worker process: the lead's entries of the last frame record)
    else collect_cluster_entries(ent, anc);
    stat_model_bytes = 0;
    for (ClusterEntry& e : ent) {
        // a cell whose training failed (allocation) has a live slot without a factor: no GP for test() (prior only)
        const ClusterModel* mm = store.model(e.model);
        if (mm && mm->base) stat_model_bytes += 4.0 * (3.0 * mm->N + mm->K + 0.5 * (double)mm->K * (mm->K + 1));
        else e.model = -1;
    }
    int rc = mq.set_clusters(ent, anc, 2.0 * (double)kCleng, stream);
    if (rc != GPIS_OK) { fprintf(stderr, "[gpismap_amd] cluster table upload failed (%d)\n", rc); if (!upd_rc) upd_rc = rc; }
}

// --------------------------------------------------------------- public surface ----
// The reference's gateways call the class methods directly (mexGPisMap3.cpp:70,102,150; mexGPisMap.cpp:70,108):
// nothing may propagate out of them into MATLAB.  Every public method is a function-try-block.
static void nothrow_report(const char* where, const char* what) {
    fprintf(stderr, "[gpismap_amd] %s: exception contained (%s)\n", where, what);
}

// Device list of a map object: GPIS_DEVICES=0,1,2,... (a device may repeat: logical shards on one GPU); empty = the
// device current in the calling thread, one device.
static std::vector<int> env_devices() {
    std::vector<int> d;
    const char* e = getenv("GPIS_DEVICES");
    if (!e || !*e) return d;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) return d;
    const char* p = e;
    while (*p) {
        char* end = nullptr;
        long v = strtol(p, &end, 10);
        if (end == p) break;
        if (v < 0 || v >= ndev) { fprintf(stderr, "[gpismap_amd] GPIS_DEVICES: device %ld out of range (%d visible): ignored\n", v, ndev); d.clear(); return d; }
        d.push_back((int)v);
        p = end;
        while (*p == ',' || *p == ' ') ++p;
    }
    return d;
}
static GPisMap3::Impl* make_impl(const GPisMap3Param& par, const camParam& c, const std::vector<int>& devs) {
    GPisMap3::Impl* m;
    {
        DeviceScope ds(devs.empty() ? -1 : devs[0]);
        m = new GPisMap3::Impl(par, c);
    }
    if (devs.size() > 1) {
        std::vector<int> one(1);
        for (size_t i = 1; i < devs.size(); ++i) {
            one[0] = devs[i];
            GPisMap3* peer = gpis3_impl_create_on(par, c, one.data(), 1);
            if (!peer) break;
            m->peers.push_back(peer);
        }
        const int world = 1 + (int)m->peers.size();
        m->shard_rank = 0; m->shard_world = world;
        for (int r = 1; r < world; ++r) { m->peers[r - 1]->impl()->shard_rank = r; m->peers[r - 1]->impl()->shard_world = world; m->peers[r - 1]->impl()->lead = m; }
        // several devices behind one map train synchronously (every frame ends with the model exchange): no CUs set aside
        if (world > 1) {
            { DeviceScope ds(m->device); m->apply_pipeline(); }
            for (GPisMap3* q : m->peers) { DeviceScope ds(q->impl()->device); q->impl()->apply_pipeline(); }
        }
    }
    return m;
}
GPisMap3::GPisMap3() : p_(make_impl(GPisMap3Param(), camParam(), env_devices())) {}
GPisMap3::GPisMap3(GPisMap3Param par) : p_(make_impl(par, camParam(), env_devices())) {}
GPisMap3::GPisMap3(GPisMap3Param par, camParam c) : p_(make_impl(par, c, env_devices())) {}
GPisMap3::~GPisMap3() {
    for (GPisMap3* q : p_->peers) delete q;
    p_->peers.clear();
    DeviceScope dev_scope_(p_->device);
    delete p_;
}
// explicit device list (C-ABI gpis3_create_multi); n = 1: one map on that device, GPIS_DEVICES not consulted
GPisMap3* gpis3_impl_create_on(const GPisMap3Param& par, const camParam& c, const int* devices, int n) {
    int ndev = 0;
    if (!devices || n < 1 || hipGetDeviceCount(&ndev) != hipSuccess) return nullptr;
    std::vector<int> devs(devices, devices + n);
    for (int d : devs) if (d < 0 || d >= ndev) return nullptr;
    try { return new GPisMap3(par, c, devs.data(), n); } catch (...) { return nullptr; }
}
GPisMap3::GPisMap3(const GPisMap3Param& par, camParam c, const int* devices, int n)
    : p_(make_impl(par, c, std::vector<int>(devices, devices + n))) {}

template <class F>
static void for_each_rank(GPisMap3::Impl& m, F f) {     // rank r on its own host thread (rank 0 on the caller's)
    const int world = 1 + (int)m.peers.size();
    std::vector<std::thread> th;
    for (int r = 1; r < world; ++r) th.emplace_back([&, r] { f(r, m.peers[r - 1]); });
    f(0, (GPisMap3*)nullptr);
    for (auto& t : th) t.join();
}

void GPisMap3::reset() try {
    for (GPisMap3* q : p_->peers) q->reset();
    DeviceScope dev_scope_(p_->device);
    p_->reset();
} catch (const std::exception& e) { nothrow_report("GPisMap3::reset", e.what()); } catch (...) { nothrow_report("GPisMap3::reset", "unknown exception"); }

void GPisMap3::resetCam(camParam c) try {  // GPisMap3.cpp:117-123
    for (GPisMap3* q : p_->peers) q->resetCam(c);
    DeviceScope dev_scope_(p_->device);
    p_->cam = c;
    p_->vu_grid.clear();
} catch (const std::exception& e) { nothrow_report("GPisMap3::resetCam", e.what()); } catch (...) { nothrow_report("GPisMap3::resetCam", "unknown exception"); }

// The models every rank trained travel to every other rank: pack on the owner's device (records back to back at their own
// sizes), hipMemcpyPeerAsync into the receiver's buffer -- one stream per source, so the copies of the later sources run
// under the unpacking of the earlier ones --, unpack as predict-only models, then every rank builds its cluster table.
// Bytes: a rank sends its records (2 K^2 + 20 K bytes each) once to each of the other n-1 ranks and nothing else
// (synthetic scene, F = 5: 1.3 GB of records per frame in total; round 3 padded every record to the largest: 5.8 GB).
static int exchange_models_multi(GPisMap3* self) {
    GPisMap3::Impl& m0 = *self->impl();
    const int world = 1 + (int)m0.peers.size();
    auto inst = [&](int r) { return r == 0 ? self : m0.peers[r - 1]; };
    std::vector<size_t> bytes(world);                          // record bytes of rank r (the same on every rank)
    for (int r = 0; r < world; ++r) { const long long b = gpis3_impl_shard_bytes(self, r); if (b < 0) return GPIS_ERR_STATE; bytes[r] = (size_t)b; }
    std::vector<int> rc(world, GPIS_OK);
    auto ensure = [](void*& p, size_t& cap, size_t need) -> int {
        if (need <= cap) return GPIS_OK;
        (void)hipFree(p); p = nullptr; cap = 0;
        if (hipMalloc(&p, need + need / 4) != hipSuccess) return GPIS_ERR_HIP;
        cap = need + need / 4;
        return GPIS_OK;
    };
    for_each_rank(m0, [&](int r, GPisMap3*) {                 // pack, every rank on its device
        GPisMap3::Impl& m = *inst(r)->impl();
        DeviceScope ds(m.device);
        if (bytes[r] == 0) return;
        rc[r] = ensure(m.d_send, m.cap_send, bytes[r]);
        if (!rc[r]) rc[r] = gpis3_impl_shard_pack(inst(r), m.d_send, nullptr);
    });
    for (int r = 0; r < world; ++r) if (rc[r]) return rc[r];
    for_each_rank(m0, [&](int q, GPisMap3*) {                 // receive + unpack, every rank on its device
        GPisMap3::Impl& m = *inst(q)->impl();
        DeviceScope ds(m.device);
        size_t total = 0;
        for (int r = 0; r < world; ++r) if (r != q) total += bytes[r];
        m.stat_exchange_bytes = (double)total;
        if (total) rc[q] = ensure(m.d_recv, m.cap_recv, total);
        if (rc[q]) return;
        if ((int)m.peer_streams.size() < world) {
            m.peer_streams.resize(world, nullptr);
            for (int r = 0; r < world; ++r) {
                if (r == q) continue;
                if (hipStreamCreateWithFlags(&m.peer_streams[r], hipStreamNonBlocking) != hipSuccess) { rc[q] = GPIS_ERR_HIP; return; }
                const int sd = inst(r)->impl()->device;
                if (sd != m.device) { (void)hipDeviceEnablePeerAccess(sd, 0); (void)hipGetLastError(); }   // (once; "already enabled" is fine)
            }
        }
        std::vector<size_t> off(world, 0);
        size_t o = 0;
        for (int r = 0; r < world; ++r) {                      // all copies in flight first
            if (r == q || bytes[r] == 0) continue;
            off[r] = o; o += bytes[r];
            GPisMap3::Impl& src = *inst(r)->impl();
            if (hipMemcpyPeerAsync((char*)m.d_recv + off[r], m.device, src.d_send, src.device, bytes[r], m.peer_streams[r]) != hipSuccess) { rc[q] = GPIS_ERR_HIP; return; }
        }
        for (int r = 0; r < world && !rc[q]; ++r) {
            if (r == q || bytes[r] == 0) continue;
            if (hipStreamSynchronize(m.peer_streams[r]) != hipSuccess) { rc[q] = GPIS_ERR_HIP; break; }
            rc[q] = gpis3_impl_shard_unpack(inst(q), r, (char*)m.d_recv + off[r], nullptr);
        }
        if (!rc[q]) rc[q] = gpis3_impl_shard_finish(inst(q));
    });
    for (int r = 0; r < world; ++r) if (rc[r]) return rc[r];
    return GPIS_OK;
}

void GPisMap3::update(float* dataz, int N, std::vector<float>& pose) try {  // GPisMap3.cpp:218-237
    if (p_->peers.empty() || p_->shard_rank != 0) { update_one(dataz, N, pose); return; }
    // several devices: the host logic runs once, here (the lead); updateGPs hands the workers their shares of the training, then
    // the trained models are exchanged and every rank builds its cluster table from the lead's index
    for (GPisMap3* q : p_->peers) { Impl& w = *q->impl(); w.upd_rc = 0; w.shard_jobs.clear(); w.table_pending = false; w.stat_host_replays = 0; w.stat_deferred_inverses = 0; }
    update_one(dataz, N, pose);
    int rc = p_->upd_rc;
    for (GPisMap3* q : p_->peers) if (!rc) rc = q->impl()->upd_rc;
    if (!rc && p_->table_pending) rc = exchange_models_multi(this);
    if (rc) { p_->upd_rc = rc; fprintf(stderr, "[gpismap_amd] GPisMap3::update (%d devices): failed (%d)\n", 1 + (int)p_->peers.size(), rc); }
} catch (const std::exception& e) { nothrow_report("GPisMap3::update", e.what()); p_->upd_rc = GPIS_ERR_STATE; } catch (...) { nothrow_report("GPisMap3::update", "unknown exception"); p_->upd_rc = GPIS_ERR_STATE; }

void GPisMap3::update_one(float* dataz, int N, std::vector<float>& pose) try {
    DeviceScope dev_scope_(p_->device);
    Impl& m = *p_;
    m.upd_rc = 0;
    m.stat_host_replays = 1;
    m.stat_deferred_inverses = 0;
    m.shard_jobs.clear();       // an update that returns early (no valid pixel, failed regression) must not leave the
    m.table_pending = false;    // previous frame's job list to a later exchange
    if (m.remote_index) { m.upd_rc = GPIS_ERR_STATE; fprintf(stderr, "[gpismap_amd] GPisMap3::update: this map applies the lead's frame records (gpis3_apply_frame)\n"); return; }
    if (m.deferred_train) (void)m.train_deferred();      // (the caller never asked for the previous frame's own share: train it now)
    if (m.export_frames) {      // (an update that returns early still leaves a record: "nothing happened")
        FrameHeader h;
        std::memset(&h, 0, sizeof(h));
        h.magic = kFrameMagic; h.version = 1; h.nothing = 1;
        m.frame_rec.assign((const char*)&h, (const char*)&h + sizeof(h));
        m.slot_ops.clear();
    }
    if (!m.ok) { m.upd_rc = GPIS_ERR_HIP; fprintf(stderr, "[gpismap_amd] GPisMap3::update: HIP device unavailable\n"); return; }
    m.tree.recycle();
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](int i) {
        auto t1 = std::chrono::steady_clock::now();
        m.last_update_ms[i] = std::chrono::duration<float, std::milli>(t1 - t0).count();
        t0 = t1;
    };
    for (float& v : m.last_update_ms) v = 0.f;
    if (!m.preprocData(dataz, N, pose)) return;
    lap(0);
    if (m.regressObs()) {
        lap(1);
        m.batch_launched = false;
        m.updateMapPoints();
        lap(2);
        if (!m.has_tree) {  // addNewMeas :571-578
            float c[3] = {0.f, 0.f, 0.f};
            m.tree.make_root(c);
            m.has_tree = true;
        }
        m.evalPoints();
        lap(3);
        m.updateGPs();
        lap(4);
    }
} catch (const std::exception& e) { nothrow_report("GPisMap3::update", e.what()); p_->upd_rc = GPIS_ERR_STATE; } catch (...) { nothrow_report("GPisMap3::update", "unknown exception"); p_->upd_rc = GPIS_ERR_STATE; }

// Several devices behind one map: a device that had to drop models (error word of a training or inverse pass -- with factor
// records every receiver inverts for itself, so ONE rank can lose models its owner still holds) must not leave test() answers
// depending on which rank evaluates a block.  The union of what any rank dropped is dropped everywhere and every table is
// rebuilt; the status surfaces on the lead (upd_rc), as a failed training does.
static void reconcile_dropped(GPisMap3::Impl& m) {
    if (m.peers.empty()) { (void)m.store.take_dropped(); return; }
    std::vector<int> all = m.store.take_dropped();
    for (GPisMap3* q : m.peers) { std::vector<int> d = q->impl()->store.take_dropped(); all.insert(all.end(), d.begin(), d.end()); }
    if (all.empty()) return;
    std::sort(all.begin(), all.end());
    all.erase(std::unique(all.begin(), all.end()), all.end());
    fprintf(stderr, "[gpismap_amd] %d models dropped on one device are dropped on every device of the map\n", (int)all.size());
    if (!m.upd_rc) m.upd_rc = GPIS_ERR_STATE;
    for_each_rank(m, [&](int r, GPisMap3* q) {
        GPisMap3::Impl& w = r == 0 ? m : *q->impl();
        DeviceScope ds(w.device);
        (void)w.store.drop_models(all, w.stream);
        if (!w.table_pending) w.build_cluster_table();
    });
}

bool GPisMap3::testDevice(const float* d_x, int leng, float* d_res, void* hip_stream) try {
    DeviceScope dev_scope_(p_->device);
    Impl& m = *p_;
    m.fail_rc = 0;
    if (!m.ok || !d_x || !d_res || leng < 1) return false;
    if (!m.has_tree) return false;  // the reference dereferences a null tree here
    if (m.table_pending) { m.fail_rc = GPIS_ERR_STATE; fprintf(stderr, "[gpismap_amd] GPisMap3::testDevice: sharded update not finished (gpis3_shard_finish)\n"); return false; }
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : m.stream;
    m.fail_rc = 0;
    m.finish_training();
    const int rc = m.mq.run(m.store, d_x, leng, d_res, s);
    if (rc != GPIS_OK) { m.fail_rc = rc; fprintf(stderr, "[gpismap_amd] GPisMap3::testDevice: device path failed (%d)\n", rc); }
    if (rc == GPIS_ERR_STATE) m.build_cluster_table();   // (models dropped by the inverse pass: their cells have no GP any more)
    return rc == GPIS_OK;
} catch (const std::exception& e) { nothrow_report("GPisMap3::testDevice", e.what()); p_->fail_rc = GPIS_ERR_STATE; return false; } catch (...) { nothrow_report("GPisMap3::testDevice", "unknown exception"); p_->fail_rc = GPIS_ERR_STATE; return false; }

bool GPisMap3::test(float* x, int dim, int leng, float* res) try {  // GPisMap3.cpp:904-949
    if (p_->peers.empty() || p_->shard_rank != 0 || x == 0 || dim != 3 || leng < 1) return test_one(x, dim, leng, res);
    // several devices: blocks of kQueryBlock queries dealt round-robin (GPisMap3.cpp:904-949 partitions over host threads)
    Impl& m0 = *p_;
    const int world = 1 + (int)m0.peers.size(), B = Impl::kQueryBlock;
    const int nblk = (leng + B - 1) / B;
    std::vector<char> okv(world, 1);
    std::vector<int> frc(world, 0);
    for_each_rank(m0, [&](int r, GPisMap3* q) {
        GPisMap3* g = r == 0 ? this : q;
        Impl& mr = *g->impl();
        // this rank's blocks gathered into PAGE-LOCKED staging (the copies to and from the device are then true DMA transfers
        // straight out of / into these buffers; pageable vectors cost one more pass through the runtime's own staging each way)
        size_t nq = 0;
        for (int b = r; b < nblk; b += world) nq += (size_t)(std::min(leng, (b + 1) * B) - b * B);
        if (nq == 0) return;
        DeviceScope ds(mr.device);
        auto ensure_pinned = [](float*& p, size_t& cap, size_t need) -> bool {
            if (need <= cap) return true;
            if (p) (void)hipHostFree(p);
            p = nullptr; cap = 0;
            const size_t want = need + need / 2;
            if (hipHostMalloc((void**)&p, sizeof(float) * want) != hipSuccess) { p = nullptr; return false; }
            cap = want;
            return true;
        };
        if (!ensure_pinned(mr.h_xstage, mr.cap_xstage, 3 * nq) || !ensure_pinned(mr.h_rstage, mr.cap_rstage, 8 * nq)) { okv[r] = 0; frc[r] = GPIS_ERR_HIP; return; }
        size_t o = 0;
        for (int b = r; b < nblk; b += world) {
            const int lo = b * B, hi = std::min(leng, lo + B);
            std::memcpy(mr.h_xstage + 3 * o, x + (size_t)3 * lo, sizeof(float) * 3 * (size_t)(hi - lo));
            std::memcpy(mr.h_rstage + 8 * o, res + (size_t)8 * lo, sizeof(float) * 8 * (size_t)(hi - lo));   // (callers pre-fill res)
            o += (size_t)(hi - lo);
        }
        okv[r] = g->test_one(mr.h_xstage, 3, (int)nq, mr.h_rstage) ? 1 : 0;
        frc[r] = mr.fail_rc;
        if (!okv[r]) return;
        o = 0;
        for (int b = r; b < nblk; b += world) {
            const int lo = b * B, hi = std::min(leng, lo + B);
            std::memcpy(res + (size_t)8 * lo, mr.h_rstage + 8 * o, sizeof(float) * 8 * (size_t)(hi - lo));
            o += (size_t)(hi - lo);
        }
    });
    reconcile_dropped(m0);
    for (int r = 0; r < world; ++r) if (!okv[r]) { p_->fail_rc = frc[r]; return false; }
    return true;
} catch (const std::exception& e) { nothrow_report("GPisMap3::test", e.what()); p_->fail_rc = GPIS_ERR_STATE; return false; } catch (...) { nothrow_report("GPisMap3::test", "unknown exception"); p_->fail_rc = GPIS_ERR_STATE; return false; }

bool GPisMap3::test_one(float* x, int dim, int leng, float* res) try {
    DeviceScope dev_scope_(p_->device);
    Impl& m = *p_;
    m.fail_rc = 0;
    if (x == 0 || dim != 3 || leng < 1) return false;
    if (!m.ok) { fprintf(stderr, "[gpismap_amd] GPisMap3::test: HIP device unavailable\n"); return false; }
    if (!m.has_tree) return false;
    if (m.table_pending) { m.fail_rc = GPIS_ERR_STATE; fprintf(stderr, "[gpismap_amd] GPisMap3::test: sharded update not finished (gpis3_shard_finish)\n"); return false; }
    m.fail_rc = 0;
    m.finish_training();
    auto fail = [&](int rc) { m.fail_rc = rc; fprintf(stderr, "[gpismap_amd] GPisMap3::test: device path failed (%d)\n", rc); return false; };
    size_t nx = (size_t)3 * leng, nr = (size_t)8 * leng;
    if (nx > m.cap_x) { (void)hipFree(m.d_x); m.d_x = nullptr; m.cap_x = 0; if (hipMalloc(&m.d_x, sizeof(float) * nx) != hipSuccess) return fail(GPIS_ERR_HIP); m.cap_x = nx; }
    if (nr > m.cap_res) { (void)hipFree(m.d_res); m.d_res = nullptr; m.cap_res = 0; if (hipMalloc(&m.d_res, sizeof(float) * nr) != hipSuccess) return fail(GPIS_ERR_HIP); m.cap_res = nr; }
    if (hipMemcpyAsync(m.d_x, x, sizeof(float) * nx, hipMemcpyHostToDevice, m.stream) != hipSuccess) return fail(GPIS_ERR_HIP);
    // only some entries are written (callers pre-fill res, mexGPisMap3.cpp:99): start from the caller's buffer
    if (hipMemcpyAsync(m.d_res, res, sizeof(float) * nr, hipMemcpyHostToDevice, m.stream) != hipSuccess) return fail(GPIS_ERR_HIP);
    {
        const int rc = m.mq.run(m.store, m.d_x, leng, m.d_res, m.stream);
        if (rc == GPIS_ERR_STATE) m.build_cluster_table();   // (models dropped by the inverse pass: their cells have no GP any more)
        if (rc != GPIS_OK) return fail(rc);
    }
    if (hipMemcpyAsync(res, m.d_res, sizeof(float) * nr, hipMemcpyDeviceToHost, m.stream) != hipSuccess) return fail(GPIS_ERR_HIP);
    if (hipStreamSynchronize(m.stream) != hipSuccess) return fail(GPIS_ERR_HIP);
    return true;
} catch (const std::exception& e) { nothrow_report("GPisMap3::test", e.what()); p_->fail_rc = GPIS_ERR_STATE; return false; } catch (...) { nothrow_report("GPisMap3::test", "unknown exception"); p_->fail_rc = GPIS_ERR_STATE; return false; }

void GPisMap3::getAllPoints(std::vector<float>& pos) try {  // GPisMap3.cpp:951-972
    pos.clear();
    Impl& m = *p_;
    if (!m.has_tree || m.tree.root < 0) return;      // (a worker -- device or process -- answers test() from the lead's index and holds no tree)
    std::vector<int> ids;
    m.tree.all_points(m.tree.root, ids);
    pos.reserve(ids.size() * 3);
    for (int id : ids) for (int d = 0; d < 3; ++d) pos.push_back(m.tree.pts[id].pos[d]);
} catch (const std::exception& e) { nothrow_report("GPisMap3::getAllPoints", e.what()); } catch (...) { nothrow_report("GPisMap3::getAllPoints", "unknown exception"); }

void GPisMap3::getAllNodes(std::vector<float>& out) try {
    out.clear();
    Impl& m = *p_;
    if (!m.has_tree || m.tree.root < 0) return;
    std::vector<int> ids;
    m.tree.all_points(m.tree.root, ids);
    out.reserve(ids.size() * 9);
    for (int id : ids) {
        const FlatPoint<3>& p = m.tree.pts[id];
        for (int d = 0; d < 3; ++d) out.push_back(p.pos[d]);
        for (int d = 0; d < 3; ++d) out.push_back(p.grad[d]);
        out.push_back(p.val); out.push_back(p.sigx); out.push_back(p.sigg);
    }
} catch (const std::exception& e) { nothrow_report("GPisMap3::getAllNodes", e.what()); } catch (...) { nothrow_report("GPisMap3::getAllNodes", "unknown exception"); }

// ------------------------------------------------------------------ checkpoint ----
// File: header, tree parameters, the flat tree's vectors as they are (nodes, points, free lists), the cluster cells that carry a
// trained model with the byte offsets of their records, then the packed records themselves (model_pack.hip: what prediction
// reads -- row table, points, X = L^-1 with alpha).  The models travel as they are, not as something to retrain: the
// reference's update leaves a model stale when a point is removed without its cell being re-activated, and a reloaded map
// must answer exactly as the saved one did.
namespace {
struct CkptHeader {
    char magic[8];
    unsigned version, dim, sz_node, sz_point, sz_param;
    int root, has_tree;
    unsigned long long n_nodes, n_pts, n_free_nodes, n_free_pts, n_pending, n_models, model_bytes;
    unsigned long long checksum;     // FNV-1a (64 bit) over the header (this field zero) and every byte that follows it
};
const char kCkptMagic[8] = {'G', 'P', 'I', 'S', '3', 'C', 'K', '3'};
constexpr unsigned kCkptVersion = 3;
struct Fnv64 {
    unsigned long long h = 1469598103934665603ull;
    void add(const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } }
};
template <class T> bool wr_vec(FILE* f, const std::vector<T>& v, Fnv64& ck) {
    ck.add(v.data(), sizeof(T) * v.size());
    return v.empty() || fwrite(v.data(), sizeof(T), v.size(), f) == v.size();
}
template <class T> bool rd_vec(FILE* f, std::vector<T>& v, size_t n, Fnv64& ck) {
    v.resize(n);
    if (n != 0 && fread(v.data(), sizeof(T), n, f) != n) return false;
    ck.add(v.data(), sizeof(T) * n);
    return true;
}
// The raw images of the tree are only swapped into the live map after every index in them has been checked: a file of the right
// size with a damaged payload must be refused, not walked.  Checks: every child / parent / point / free-list index in
// range; parent and child agree; live nodes are not on the free list and free entries are dead; a live leaf's point is alive;
// the root is a live node without a parent; cluster cells with a model are live.
bool ckpt_tree_consistent(const std::vector<FlatTree<3>::TNode>& nodes, const std::vector<FlatPoint<3>>& pts, const std::vector<int>& free_nodes,
                          const std::vector<int>& free_pts, const std::vector<int>& pending, const std::vector<int>& cells, int root, bool has_tree) {
    const int nn = (int)nodes.size(), np = (int)pts.size();
    std::vector<char> node_free((size_t)nn, 0), pt_free((size_t)np, 0);
    for (int i : free_nodes) { if (i < 0 || i >= nn || node_free[i] || nodes[i].alive) return false; node_free[i] = 1; }
    for (int i : free_pts) { if (i < 0 || i >= np || pt_free[i] || pts[i].alive) return false; pt_free[i] = 1; }
    for (int i : pending) { if (i < 0 || i >= np || pt_free[i] || pts[i].alive) return false; pt_free[i] = 1; }
    for (int i = 0; i < nn; ++i) {
        const FlatTree<3>::TNode& t = nodes[i];
        if (!t.alive) continue;
        if (t.par < -1 || t.par >= nn || t.pt < -1 || t.pt >= np) return false;
        if (t.par >= 0) {
            if (!nodes[t.par].alive || nodes[t.par].leaf) return false;
            bool found = false;
            for (int k = 0; k < FlatTree<3>::NC; ++k) found = found || nodes[t.par].ch[k] == i;
            if (!found) return false;
        }
        if (t.pt >= 0 && !pts[t.pt].alive) return false;
        for (int k = 0; k < FlatTree<3>::NC; ++k) {
            const int c = t.ch[k];
            if (c < -1 || c >= nn) return false;
            if (!t.leaf && c >= 0 && (!nodes[c].alive || nodes[c].par != i)) return false;
        }
        if (!(t.h > 0.f)) return false;
    }
    if (has_tree && (root < 0 || root >= nn || !nodes[root].alive || nodes[root].par != -1)) return false;
    // every live node hangs off the root within a bounded depth (parent and child links agree, see above, so a walk down the
    // child links from the root finds exactly the nodes whose parent chain ends at the root): a cycle of live nodes beside the
    // tree, or a crafted chain deep enough to overflow the recursive walks over the index, is refused here
    {
        constexpr int kMaxDepth = 64;       // (max_half / min_half = 2^8 in the product's geometry)
        std::vector<int> depth((size_t)nn, -1), stack;
        size_t reached = 0, live = 0;
        for (int i = 0; i < nn; ++i) live += nodes[i].alive ? 1 : 0;
        if (has_tree) { depth[root] = 0; stack.push_back(root); }
        while (!stack.empty()) {
            const int i = stack.back(); stack.pop_back();
            ++reached;
            if (nodes[i].leaf) continue;
            for (int k = 0; k < FlatTree<3>::NC; ++k) {
                const int c = nodes[i].ch[k];
                if (c < 0) continue;
                if (depth[c] >= 0 || depth[i] + 1 > kMaxDepth) return false;
                depth[c] = depth[i] + 1;
                stack.push_back(c);
            }
        }
        if (reached != live) return false;
    }
    for (int c : cells) if (c < 0 || c >= nn || !nodes[c].alive) return false;
    return true;
}
// A checkpoint read and validated on the host, before anything of it touches a map.
struct CkptImage {
    CkptHeader h;
    std::vector<FlatTree<3>::TNode> nodes; std::vector<FlatPoint<3>> pts;
    std::vector<int> free_nodes, free_pts, pending, cells;
    std::vector<unsigned long long> offs;
    std::vector<char> bytes;
};
bool ckpt_read(const char* path, const FlatTreeParam& q, CkptImage& im) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    struct Closer { FILE* f; ~Closer() { if (f) fclose(f); } } closer{f};
    CkptHeader& h = im.h;
    FlatTreeParam prm;
    Fnv64 ck;
    if (fread(&h, sizeof(h), 1, f) != 1 || std::memcmp(h.magic, kCkptMagic, 8) != 0 || h.version != kCkptVersion || h.dim != 3 ||
        h.sz_node != sizeof(FlatTree<3>::TNode) || h.sz_point != sizeof(FlatPoint<3>) || h.sz_param != sizeof(FlatTreeParam)) return false;
    { CkptHeader h0 = h; h0.checksum = 0; ck.add(&h0, sizeof(h0)); }
    // another tree geometry (field by field: the struct has padding)
    if (fread(&prm, sizeof(prm), 1, f) != 1) return false;
    ck.add(&prm, sizeof(prm));
    if (prm.init_half != q.init_half || prm.min_half != q.min_half || prm.min_half_sq != q.min_half_sq ||
        prm.max_half != q.max_half || prm.cluster_half != q.cluster_half || prm.cluster_eps != q.cluster_eps ||
        prm.qleaf_eps_plain != q.qleaf_eps_plain || prm.qleaf_eps_dist != q.qleaf_eps_dist || prm.qdesc_eps != q.qdesc_eps) return false;
    const unsigned long long lim = 1ull << 31;
    if (h.n_nodes >= lim || h.n_pts >= lim || h.n_free_nodes > h.n_nodes || h.n_free_pts > h.n_pts || h.n_pending > h.n_pts || h.n_models > h.n_nodes) return false;
    if (!rd_vec(f, im.nodes, (size_t)h.n_nodes, ck) || !rd_vec(f, im.pts, (size_t)h.n_pts, ck) || !rd_vec(f, im.free_nodes, (size_t)h.n_free_nodes, ck) ||
        !rd_vec(f, im.free_pts, (size_t)h.n_free_pts, ck) || !rd_vec(f, im.pending, (size_t)h.n_pending, ck) || !rd_vec(f, im.cells, (size_t)h.n_models, ck) ||
        !rd_vec(f, im.offs, (size_t)h.n_models + 1, ck) || im.offs.back() != h.model_bytes || !rd_vec(f, im.bytes, (size_t)h.model_bytes, ck)) return false;
    if (ck.h != h.checksum) return false;       // damaged header or payload
    for (size_t i = 0; i + 1 < im.offs.size(); ++i) if (im.offs[i] > im.offs[i + 1]) return false;
    return ckpt_tree_consistent(im.nodes, im.pts, im.free_nodes, im.free_pts, im.pending, im.cells, h.root, h.has_tree != 0);
}
}  // namespace

bool GPisMap3::saveMap(const char* path) try {
    Impl& m = *p_;
    if (!path || !m.ok || m.remote_index) return false;      // (a worker process holds no tree: the checkpoint is the lead's to write)
    DeviceScope ds(m.device);
    if (m.finish_training() != GPIS_OK || m.table_pending) return false;
    std::vector<int> cl, cells;
    std::vector<int> slots;
    std::vector<unsigned long long> offs(1, 0);
    if (m.has_tree) m.tree.all_clusters(cl);
    for (int c : cl) {
        const int slot = m.tree.nodes[c].model;
        const ClusterModel* md = slot >= 0 ? m.store.model(slot) : nullptr;
        if (!md || md->N <= 0 || md->K <= 0) continue;        // (a cell whose training failed has a slot without a model: no GP there, saved as none)
        cells.push_back(c); slots.push_back(slot);
        offs.push_back(offs.back() + packed_model_bytes(md->ld, md->N));
    }
    const size_t total = (size_t)offs.back();
    std::vector<char> bytes(total);
    if (total) {
        void* d_buf = nullptr;
        if (hipMalloc(&d_buf, total) != hipSuccess) return false;
        std::vector<size_t> o(offs.begin(), offs.end());
        int rc = m.store.pack_models(slots.data(), (int)slots.size(), d_buf, 0, m.stream, o.data());
        if (rc == GPIS_OK && hipStreamSynchronize(m.stream) != hipSuccess) rc = GPIS_ERR_HIP;
        if (rc == GPIS_OK && hipMemcpy(bytes.data(), d_buf, total, hipMemcpyDeviceToHost) != hipSuccess) rc = GPIS_ERR_HIP;
        (void)hipFree(d_buf);
        if (rc != GPIS_OK) return false;
    }
    // written beside the target and renamed over it once complete: a failed save (disk full, pack error) leaves the previous
    // checkpoint of that name intact
    const std::string tmp_path = std::string(path) + ".tmp";
    FILE* f = fopen(tmp_path.c_str(), "wb");
    if (!f) return false;
    CkptHeader h;
    std::memset(&h, 0, sizeof(h));
    std::memcpy(h.magic, kCkptMagic, 8);
    h.version = kCkptVersion; h.dim = 3; h.sz_node = (unsigned)sizeof(FlatTree<3>::TNode); h.sz_point = (unsigned)sizeof(FlatPoint<3>); h.sz_param = (unsigned)sizeof(FlatTreeParam);
    h.root = m.tree.root; h.has_tree = m.has_tree ? 1 : 0;
    h.n_nodes = m.tree.nodes.size(); h.n_pts = m.tree.pts.size(); h.n_free_nodes = m.tree.free_nodes.size();
    h.n_free_pts = m.tree.free_pts.size(); h.n_pending = m.tree.pending_free_pts.size();
    h.n_models = cells.size(); h.model_bytes = total;
    // images with the padding bytes of the structures zeroed (the same map always gives the same file) and the model slots,
    // which mean nothing to another process, cleared
    std::vector<FlatTree<3>::TNode> nimg(m.tree.nodes.size());
    std::vector<FlatPoint<3>> pimg(m.tree.pts.size());
    if (!nimg.empty()) std::memset((void*)nimg.data(), 0, sizeof(nimg[0]) * nimg.size());
    if (!pimg.empty()) std::memset((void*)pimg.data(), 0, sizeof(pimg[0]) * pimg.size());
    for (size_t i = 0; i < nimg.size(); ++i) {
        const FlatTree<3>::TNode& a = m.tree.nodes[i];
        FlatTree<3>::TNode& b = nimg[i];
        for (int d = 0; d < 3; ++d) { b.c[d] = a.c[d]; b.lo[d] = a.lo[d]; b.hi[d] = a.hi[d]; }
        b.h = a.h; b.hsq = a.hsq;
        for (int k = 0; k < FlatTree<3>::NC; ++k) b.ch[k] = a.ch[k];
        b.par = a.par; b.pt = a.pt; b.model = -1;
        b.leaf = a.leaf; b.maxDepth = a.maxDepth; b.rootLimit = a.rootLimit; b.alive = a.alive;
    }
    for (size_t i = 0; i < pimg.size(); ++i) {
        const FlatPoint<3>& a = m.tree.pts[i];
        FlatPoint<3>& b = pimg[i];
        for (int d = 0; d < 3; ++d) { b.pos[d] = a.pos[d]; b.grad[d] = a.grad[d]; }
        b.val = a.val; b.sigx = a.sigx; b.sigg = a.sigg; b.type = a.type; b.alive = a.alive;
    }
    FlatTreeParam pimg_prm;
    std::memset((void*)&pimg_prm, 0, sizeof(pimg_prm));
    {
        const FlatTreeParam& q = m.tree.prm;
        pimg_prm.init_half = q.init_half; pimg_prm.min_half = q.min_half; pimg_prm.min_half_sq = q.min_half_sq; pimg_prm.max_half = q.max_half;
        pimg_prm.cluster_half = q.cluster_half; pimg_prm.cluster_eps = q.cluster_eps; pimg_prm.qleaf_eps_plain = q.qleaf_eps_plain;
        pimg_prm.qleaf_eps_dist = q.qleaf_eps_dist; pimg_prm.qdesc_eps = q.qdesc_eps;
    }
    Fnv64 ck;
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1;      // (rewritten below with the checksum)
    ck.add(&h, sizeof(h));                           // (checksum field still zero)
    ck.add(&pimg_prm, sizeof(pimg_prm));
    ok = ok && fwrite(&pimg_prm, sizeof(FlatTreeParam), 1, f) == 1 &&
         wr_vec(f, nimg, ck) && wr_vec(f, pimg, ck) && wr_vec(f, m.tree.free_nodes, ck) && wr_vec(f, m.tree.free_pts, ck) &&
         wr_vec(f, m.tree.pending_free_pts, ck) && wr_vec(f, cells, ck) && wr_vec(f, offs, ck) && wr_vec(f, bytes, ck);
    h.checksum = ck.h;
    ok = ok && fseek(f, 0, SEEK_SET) == 0 && fwrite(&h, sizeof(h), 1, f) == 1;
    ok = ok && fflush(f) == 0 && fsync(fileno(f)) == 0;
    ok = (fclose(f) == 0) && ok;
    ok = ok && rename(tmp_path.c_str(), path) == 0;
    if (!ok) (void)remove(tmp_path.c_str());      // never leave a half-written file behind; the target is untouched
    return ok;
} catch (const std::exception& e) { nothrow_report("GPisMap3::saveMap", e.what()); return false; } catch (...) { nothrow_report("GPisMap3::saveMap", "unknown exception"); return false; }

// loadMap in three steps.  (1) The file is read and validated on the host, once (ckpt_read).  (2) STAGE: every device unpacks
// the records into NEW slots of its live store; until that has succeeded everywhere the map is untouched, and a failure on
// any device releases what the call created on every device.  Several devices behind one map: the workers' cluster tables are
// built from the LEAD's index, which names the lead's slots -- the slot ids every device obtained must be the lead's, and
// this is checked, not assumed (a divergent free list is a refused load, not a test() that reads the wrong models).
// (3) COMMIT (cannot fail half-way across devices: nothing in it allocates): the models the stores held before go back to
// them, the lead swaps the index in -- a worker's own tree stays empty -- and every device builds its cluster table.
namespace {
int ckpt_stage(GPisMap3::Impl& m, const CkptImage& im, std::vector<int>& slots) {
    DeviceScope ds(m.device);
    (void)m.finish_training();
    slots.assign(im.cells.size(), -1);
    if (im.cells.empty()) return GPIS_OK;
    int rc = GPIS_OK;
    void* d_buf = nullptr;
    if (hipMalloc(&d_buf, im.bytes.size()) != hipSuccess) rc = GPIS_ERR_HIP;
    if (rc == GPIS_OK && hipMemcpy(d_buf, im.bytes.data(), im.bytes.size(), hipMemcpyHostToDevice) != hipSuccess) rc = GPIS_ERR_HIP;
    std::vector<size_t> o(im.offs.begin(), im.offs.end());
    if (rc == GPIS_OK) rc = m.store.unpack_models(d_buf, (int)im.cells.size(), 0, slots.data(), m.stream, o.data());
    if (rc == GPIS_OK && hipStreamSynchronize(m.stream) != hipSuccess) rc = GPIS_ERR_HIP;
    if (d_buf) (void)hipFree(d_buf);
    if (rc != GPIS_OK) {
        for (int& sl : slots) { if (sl >= 0) m.store.release_slot(sl); sl = -1; }
        (void)hipGetLastError();
    }
    return rc;
}
void ckpt_unstage(GPisMap3::Impl& m, std::vector<int>& slots) {
    DeviceScope ds(m.device);
    for (int& sl : slots) { if (sl >= 0) m.store.release_slot(sl); sl = -1; }
    // devices that got further than others before the refusal hold different free lists now; in ascending order every store
    // hands out the same ids again (ids a store never created lie above all of its free ones)
    m.store.canonical_free_slots();
}
void ckpt_commit(GPisMap3::Impl& m, CkptImage* im /* the lead's: swapped in; nullptr on a worker */, const std::vector<int>& old_slots,
                 const std::vector<int>& cells, const std::vector<int>& slots) {
    DeviceScope ds(m.device);
    // (by slot, not by walking the tree: a device worker's models hang off the lead's tree)
    for (int sl : old_slots) m.store.release_slot(sl);
    m.upd_rc = 0;
    m.tree.clear(); m.has_tree = false;
    m.gpo.reset_trained(); m.gpo_created = false;
    m.obs_numdata = 0;
    m.activeSet.clear();
    m.shard_jobs.clear(); m.table_pending = false;
    if (!im) return;
    for (FlatTree<3>::TNode& t : im->nodes) t.model = -1;
    m.tree.nodes.swap(im->nodes); m.tree.pts.swap(im->pts); m.tree.free_nodes.swap(im->free_nodes); m.tree.free_pts.swap(im->free_pts);
    m.tree.pending_free_pts.swap(im->pending); m.tree.released_models.clear();
    m.tree.root = im->h.has_tree ? im->h.root : -1; m.tree.last_cell = -1;
    m.has_tree = im->h.has_tree != 0;
    for (size_t i = 0; i < cells.size(); ++i) m.tree.nodes[cells[i]].model = slots[i];
}
}  // namespace

bool GPisMap3::loadMap(const char* path) try {
    if (p_->peers.empty() || p_->shard_rank != 0) return loadMap_one(path);
    Impl& m = *p_;
    if (!path || !m.ok) return false;
    CkptImage im;
    if (!ckpt_read(path, m.tree.prm, im)) return false;
    const int world = 1 + (int)m.peers.size();
    auto rank_impl = [&](int r) -> Impl& { return r == 0 ? m : *m.peers[r - 1]->impl(); };
    std::vector<std::vector<int>> old_slots(world), slots(world);
    std::vector<int> rcv(world, GPIS_OK);
    for_each_rank(m, [&](int r, GPisMap3*) {
        Impl& w = rank_impl(r);
        DeviceScope ds(w.device);
        (void)w.finish_training();
        old_slots[r] = w.store.live_slots();
        rcv[r] = ckpt_stage(w, im, slots[r]);
    });
    bool ok = true;
    for (int r = 0; r < world; ++r) ok = ok && rcv[r] == GPIS_OK && slots[r] == slots[0];
    if (!ok) {
        for (int r = 0; r < world; ++r)
            if (rcv[r] == GPIS_OK && slots[r] != slots[0]) { fprintf(stderr, "[gpismap_amd] loadMap: model slots of the devices diverged; the map is kept as it was\n"); break; }
        for_each_rank(m, [&](int r, GPisMap3*) { ckpt_unstage(rank_impl(r), slots[r]); });
        return false;
    }
    const std::vector<int> cells = im.cells;
    ckpt_commit(m, &im, old_slots[0], cells, slots[0]);          // the lead's index first: the workers' tables are built from it
    for_each_rank(m, [&](int r, GPisMap3*) {
        Impl& w = rank_impl(r);
        if (r > 0) ckpt_commit(w, nullptr, old_slots[r], cells, slots[r]);
        DeviceScope ds(w.device);
        w.build_cluster_table();
    });
    for (int r = 0; r < world; ++r) if (rank_impl(r).upd_rc != 0) return false;
    return true;
} catch (const std::exception& e) { nothrow_report("GPisMap3::loadMap", e.what()); return false; } catch (...) { nothrow_report("GPisMap3::loadMap", "unknown exception"); return false; }

bool GPisMap3::loadMap_one(const char* path) try {
    Impl& m = *p_;
    if (!path || !m.ok) return false;
    DeviceScope ds(m.device);
    CkptImage im;
    if (!ckpt_read(path, m.tree.prm, im)) return false;
    (void)m.finish_training();
    const std::vector<int> old_slots = m.store.live_slots();
    std::vector<int> slots;
    if (ckpt_stage(m, im, slots) != GPIS_OK) return false;
    const std::vector<int> cells = im.cells;
    ckpt_commit(m, &im, old_slots, cells, slots);
    m.build_cluster_table();
    return m.upd_rc == 0;
} catch (const std::exception& e) { nothrow_report("GPisMap3::loadMap", e.what()); return false; } catch (...) { nothrow_report("GPisMap3::loadMap", "unknown exception"); return false; }

// accessors used by the C-ABI (capi.cpp)
int gpis3_impl_fail(GPisMap3* g) { return g->impl()->fail_rc; }
int gpis3_impl_device(GPisMap3* g) { return g->impl()->device; }
int gpis3_impl_num_devices(GPisMap3* g) { return 1 + (int)g->impl()->peers.size(); }
int gpis3_impl_set_shard(GPisMap3* g, int rank, int world) {
    GPisMap3::Impl& m = *g->impl();
    if (m.table_pending) return GPIS_ERR_STATE;
    m.shard_rank = rank; m.shard_world = world;
    { DeviceScope ds(m.device); m.apply_pipeline(); }   // sharded training is synchronous (no CUs set aside); back at world 1 the caller's wish applies again
    return GPIS_OK;
}
int gpis3_impl_shard_info(GPisMap3* g, int* out, int n) {
    GPisMap3::Impl& m = *g->impl();
    if (n < 2 + m.shard_world) return GPIS_ERR_ARG;
    out[0] = (int)m.shard_jobs.size(); out[1] = 0;
    for (int r = 0; r < m.shard_world; ++r) out[2 + r] = 0;
    for (auto& j : m.shard_jobs) { ++out[2 + j.owner]; if (j.owner == m.shard_rank) ++out[1]; }
    return GPIS_OK;
}
// The records of one owner, in the frame's job order, back to back at their own sizes: slots and byte offsets (n + 1
// entries).  Every rank can lay out every rank's buffer: N and ng of all jobs of the frame are known everywhere.
static void shard_layout(GPisMap3::Impl& m, int owner, std::vector<int>& slots, std::vector<size_t>& offs) {
    slots.clear(); offs.assign(1, 0);
    for (auto& j : m.shard_jobs)
        if (j.owner == owner) {
            const int K = j.n + 3 * j.ng, ld = (K + 1 + 31) / 32 * 32;
            slots.push_back(j.slot);
            offs.push_back(offs.back() + packed_model_bytes(ld, j.n));
        }
}
long long gpis3_impl_shard_bytes(GPisMap3* g, int owner) {
    GPisMap3::Impl& m = *g->impl();
    if (owner < 0 || owner >= m.shard_world) return GPIS_ERR_ARG;
    std::vector<int> slots; std::vector<size_t> offs;
    shard_layout(m, owner, slots, offs);
    return (long long)offs.back();
}
int gpis3_impl_shard_pack(GPisMap3* g, void* d_buf, void* stream) {
    GPisMap3::Impl& m = *g->impl();
    DeviceScope ds(m.device);
    std::vector<int> slots; std::vector<size_t> offs;
    shard_layout(m, m.shard_rank, slots, offs);
    if (slots.empty()) return GPIS_OK;
    if (!d_buf) return GPIS_ERR_ARG;
    return m.store.pack_models(slots.data(), (int)slots.size(), d_buf, 0, stream ? (hipStream_t)stream : m.stream, offs.data(), m.ship_factors());
}
int gpis3_impl_shard_unpack(GPisMap3* g, int owner, const void* d_buf, void* stream) {
    GPisMap3::Impl& m = *g->impl();
    DeviceScope ds(m.device);
    if (owner < 0 || owner >= m.shard_world || owner == m.shard_rank) return GPIS_ERR_ARG;
    std::vector<int> slots; std::vector<size_t> offs;
    shard_layout(m, owner, slots, offs);
    if (slots.empty()) return GPIS_OK;
    if (!d_buf) return GPIS_ERR_ARG;
    const int rc = m.store.unpack_models(d_buf, (int)slots.size(), 0, slots.data(), stream ? (hipStream_t)stream : m.stream, offs.data());
    if (rc == GPIS_OK) m.stat_deferred_inverses += m.store.last_unpack_factors;
    return rc;
}
int gpis3_impl_shard_finish(GPisMap3* g) {
    GPisMap3::Impl& m = *g->impl();
    DeviceScope ds(m.device);
    if (!m.table_pending) return GPIS_OK;
    m.upd_rc = 0;
    m.build_cluster_table();
    return m.upd_rc;
}
int gpis3_impl_update_fail(GPisMap3* g) { return g->impl()->upd_rc; }
// host logic once across processes (Impl::export_frames)
int gpis3_impl_set_frame_export(GPisMap3* g, int on) {
    GPisMap3::Impl& m = *g->impl();
    if (!m.peers.empty() || m.lead) return GPIS_ERR_STATE;      // (several devices behind one map share the lead's memory: nothing to export)
    m.export_frames = on != 0;
    if (!m.export_frames) { m.frame_rec.clear(); m.slot_ops.clear(); }
    return GPIS_OK;
}
long long gpis3_impl_frame_record(GPisMap3* g, void* buf, long long cap) {     // bytes of the last update()'s record; copied when buf holds them
    GPisMap3::Impl& m = *g->impl();
    if (!m.export_frames) return GPIS_ERR_STATE;
    const long long n = (long long)m.frame_rec.size();
    if (buf && cap >= n && n > 0) std::memcpy(buf, m.frame_rec.data(), (size_t)n);
    return n;
}
int gpis3_impl_train_deferred(GPisMap3* g) {
    GPisMap3::Impl& m = *g->impl();
    DeviceScope ds(m.device);
    return m.train_deferred();
}
int gpis3_impl_apply_frame(GPisMap3* g, const void* buf, long long bytes) {
    GPisMap3::Impl& m = *g->impl();
    if (!buf || bytes < (long long)sizeof(FrameHeader)) return GPIS_ERR_ARG;
    if (!m.ok || !m.peers.empty() || m.lead || m.export_frames || (m.has_tree && !m.remote_index)) return GPIS_ERR_STATE;   // (a map that replays frames itself is not a worker)
    DeviceScope ds(m.device);
    try { return m.apply_frame((const char*)buf, (size_t)bytes); } catch (...) { m.upd_rc = GPIS_ERR_STATE; return GPIS_ERR_STATE; }
}
void gpis3_impl_stats(GPisMap3* g, double* out, int n) {
    GPisMap3::Impl& m = *g->impl();
    DeviceScope ds(m.device);
    m.finish_training();      // (the training time of the last batch is read off its events)
    double v[28] = {(double)m.gpo.trained_groups(m.stream), (double)m.stat_obs_queries, (double)m.stat_clusters_trained,
                    (double)m.stat_late, (double)m.mq.num_clusters(), (double)m.mq.last_evals, (double)m.mq.last_eval_ms,
                    (double)m.store.device_bytes(), (double)m.mq.last_flops, (double)m.mq.last_launches,
                    (double)m.store.last_train_ms, (double)m.stat_model_bytes,
                    m.last_update_ms[0], m.last_update_ms[1], m.last_update_ms[2], m.last_update_ms[3], m.last_update_ms[4],
                    m.store.last_train_flops, m.store.last_train_bytes, (double)m.store.last_train_jobs, (double)m.store.last_train_maxK,
                    (double)m.store.last_inverse_ms, (double)m.store.last_inverse_jobs, m.stat_exchange_bytes,
                    m.pipeline ? 1.0 : 0.0, (double)m.store.cu_reserve(), 0.0, (double)m.stat_deferred_inverses};
    v[26] = (double)m.stat_host_replays;
    for (GPisMap3* q : m.peers) v[26] += (double)q->impl()->stat_host_replays;
    for (int i = 0; i < n && i < 28; ++i) out[i] = v[i];
}
// join the training the last update() left in flight; returns the update status (0: fine)
int gpis3_impl_sync(GPisMap3* g) {
    GPisMap3::Impl& m = *g->impl();
    for (GPisMap3* q : m.peers) { int rc = gpis3_impl_sync(q); if (rc && !m.upd_rc) m.upd_rc = rc; }
    DeviceScope ds(m.device);
    m.finish_training();
    return m.upd_rc;
}
void gpis3_impl_set_pipeline(GPisMap3* g, int on) {
    GPisMap3::Impl& m = *g->impl();
    for (GPisMap3* q : m.peers) gpis3_impl_set_pipeline(q, on);
    DeviceScope ds(m.device);
    m.finish_training();
    m.want_pipeline = on != 0;
    m.apply_pipeline();
}
// join the training in flight and compute the inverses it left to the first prediction; returns the update status
int gpis3_impl_prepare_test(GPisMap3* g) {
    GPisMap3::Impl& m = *g->impl();
    for (GPisMap3* q : m.peers) { int rc = gpis3_impl_prepare_test(q); if (rc && !m.upd_rc) m.upd_rc = rc; }
    DeviceScope ds(m.device);
    m.finish_training();
    const int rc = m.store.ensure_inverses(m.stream);
    if (rc && !m.upd_rc) m.upd_rc = rc;
    if (rc == GPIS_ERR_STATE && !m.table_pending) m.build_cluster_table();
    if (!m.lead) reconcile_dropped(m);
    return m.upd_rc;
}
void gpis3_impl_set_lazy_inverse(GPisMap3* g, int on) {
    GPisMap3::Impl& m = *g->impl();
    for (GPisMap3* q : m.peers) gpis3_impl_set_lazy_inverse(q, on);
    DeviceScope ds(m.device);
    m.finish_training();       // a batch in flight is joined in the mode it was enqueued with
    m.store.lazy_inverse = on != 0;
}
void gpis3_impl_set_host_gather(GPisMap3* g, int on) {
    GPisMap3::Impl& m = *g->impl();
    for (GPisMap3* q : m.peers) gpis3_impl_set_host_gather(q, on);
    m.device_gather = on == 0;
}
void gpis3_impl_set_shard_factors(GPisMap3* g, int mode) {
    GPisMap3::Impl& m = *g->impl();
    for (GPisMap3* q : m.peers) gpis3_impl_set_shard_factors(q, mode);
    m.shard_factors_mode = mode < 0 ? -1 : (mode != 0);
}
void gpis3_impl_set_keep_factors(GPisMap3* g, int on) {
    GPisMap3::Impl& m = *g->impl();
    for (GPisMap3* q : m.peers) gpis3_impl_set_keep_factors(q, on);
    DeviceScope ds(m.device);
    m.finish_training();
    m.store.trim_scratch = on == 0;
}
void gpis3_impl_profile(GPisMap3* g, int on) {
    GPisMap3::Impl& m = *g->impl();
    for (GPisMap3* q : m.peers) gpis3_impl_profile(q, on);
    m.mq.profile = on != 0;
    m.store.profile = on != 0;
}
