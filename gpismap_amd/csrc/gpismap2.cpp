// GPisMap: MI355X-native drop-in for reference cpp/src/GPisMap.cpp (2-D laser scans).
// Same restructuring as gpismap3.cpp: ObsGP (1-D, frozen per frame) queries are evaluated in
// batches on the GPU (K2) and tree mutations are replayed on the host in the reference's order.
// The 2-D line search (GPisMap.cpp:277-317) really is sequential per point (it re-queries the
// moved location), so it runs as up to ten batch rounds over the points still iterating.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <unordered_map>
#include "../../include/GPisMap.h"
#include "flat_tree.h"
#include "map_query.h"
#include "obsgp.h"
#include "ongpis.h"

using namespace gpis;

namespace {

inline float occ_test(float rinv, float rinv0, float a) {  // GPisMap.cpp:39-42
    return (float)(2.0 * (1.0 / (1.0 + std::exp((double)(-a * (rinv - rinv0)))) - 0.5));
}
inline float saturate(float v, float lo, float hi) { return std::min(std::max(v, lo), hi); }
inline void polar2Cart(float a, float r, float& x, float& y) {  // GPisMap.cpp:44-49 (double cos/sin)
    x = (float)((double)r * std::cos((double)a));
    y = (float)((double)r * std::sin((double)a));
}
inline void cart2polar(float x, float y, float& a, float& r) {  // GPisMap.cpp:50-55
    a = (float)std::atan2((double)y, (double)x);
    r = (float)std::sqrt((double)(x * x + y * y));
}

FlatTreeParam tree_param2() {
    FlatTreeParam p;
    p.min_half = (float)0.2;  // params.h:34-37
    p.max_half = (float)102.4;
    p.init_half = (float)12.8;
    p.cluster_half = (float)0.8;
    p.min_half_sq = p.min_half * p.min_half;
    p.cluster_eps = 1e-3;
    p.qleaf_eps_plain = 0.0001;
    p.qleaf_eps_dist = 0.001;
    p.qdesc_eps = 0.001;
    return p;
}

}  // namespace

struct GPisMap::Impl {
    int device = -1;   // HIP device this map lives on (current device at construction)
    int upd_rc = 0;    // first device-side failure inside the last update() (0: none); update() itself is void like the reference's
    int fail_rc = 0;   // last device-side failure of test()/testDevice() (0: none) -- the C-ABI reports it instead of "false"
    using T2 = FlatTree<2>;
    GPisMapParam setting;
    T2 tree;
    T2::Set activeSet;
    ObsGPDevice gpo;
    OnGPISStore store;
    MapQuery mq;
    hipStream_t stream = nullptr;
    bool ok = false, has_tree = false, gpo_created = false;
    std::vector<float> obs_theta, obs_range, obs_f, obs_xylocal, obs_xyglobal;
    float pose_tr[2] = {0, 0}, pose_R[4] = {0, 0, 0, 0};
    int obs_numdata = 0;
    float range_obs_max = 0.f;
    float* d_x = nullptr; float* d_res = nullptr; size_t cap_x = 0, cap_res = 0;
    long stat_obs_queries = 0, stat_clusters_trained = 0, stat_late = 0;
    // update() is pipelined like the 3-D map's (round 6): it returns once the frame's training is enqueued; the next update's
    // first touch of the store, test(), gpis2_stats and gpis2_sync join it (the store joins by itself wherever it needs the models).
    // GPIS_PIPELINE_UPDATE=0 / gpis2_set_pipeline(map, 0): every update() joins its own training (the reference's behaviour).  The
    // 2-D clusters are small (K <= ~220 on data/2D: the fused kernel, 0.3 ms per frame), so no CUs are set aside for it.
    bool pipeline = true;
    int join_training() {
        if (!store.train_pending()) return GPIS_OK;
        const int rc = store.train_finish();
        if (rc != GPIS_OK) { fprintf(stderr, "[gpismap_amd] OnGPIS training failed (%d)\n", rc); if (!upd_rc) upd_rc = rc; rebuild_table(); }
        return rc;
    }
    void rebuild_table();

    explicit Impl(const GPisMapParam& par)
        : tree(tree_param2()), store(2, par.map_scale_param),
          mq(2, (float)((double)par.map_scale_param * 4.0), 0.4f, (float)(1.0 + (double)par.map_noise_param)) {
        setting = par;      // assignment: the public struct's only copy constructor takes a non-const reference (as the reference's)
        ok = (hipGetDevice(&device) == hipSuccess) && (hipStreamCreate(&stream) == hipSuccess);
        if (const char* e = getenv("GPIS_PIPELINE_UPDATE")) pipeline = atoi(e) != 0;
        if (!ok) device = -1;
        if (!ok) fprintf(stderr, "[gpismap_amd] GPisMap: no usable HIP device; update()/test() will fail\n");
    }
    ~Impl() {
        (void)hipFree(d_x); (void)hipFree(d_res);
        if (stream) (void)hipStreamDestroy(stream);
    }
    void reset() {
        tree.clear(); has_tree = false;
        store.clear();
        gpo.reset_trained(); gpo_created = false;
        obs_numdata = 0;
        activeSet.clear();
        std::vector<ClusterEntry> none;
        std::vector<AncestorEntry> nanc;
        mq.set_clusters(none, nanc, 2.0 * (double)tree.prm.cluster_half, stream);
    }

    bool preproData(const float* datax, const float* dataf, int N, const std::vector<float>& pose);
    void updateMapPoints();
    void evalPoints();
    void updateGPs();
    int try_insert(int pid, T2::InsSet& ins);

    struct Work {        // state of one point during re-evaluation
        bool go = false;
        float x_new[2], abs_oc = 0.f, r_new = 0.f;
        float pval[4], pvar[4];
    };
    void reeval_batch(const std::vector<int>& ids, std::vector<Work>& w);
    void reeval_apply(int pid, const Work& w);
    bool query(std::vector<float>& q, std::vector<float>& val, std::vector<float>& var) {
        val.assign(q.size(), 0.f); var.assign(q.size(), 1e6f);
        if (q.empty()) return true;
        // through the ObsGP object's page-locked staging (a re-evaluation is up to a dozen dependent round trips: copies out of
        // pageable vectors are staged by the runtime, 20-40 us each)
        const int nq = (int)q.size();
        float* hq = gpo.stage_q(nq);
        int rc = hq ? GPIS_OK : GPIS_ERR_HIP;
        if (rc == GPIS_OK) { std::memcpy(hq, q.data(), sizeof(float) * (size_t)nq); rc = gpo.query_staged(nq, stream); }
        if (rc == GPIS_OK) { val.assign(gpo.staged_val(), gpo.staged_val() + nq); var.assign(gpo.staged_var(), gpo.staged_var() + nq); }
        if (rc != GPIS_OK) { fprintf(stderr, "[gpismap_amd] ObsGP query failed (%d)\n", rc); if (!upd_rc) upd_rc = rc; return false; }
        stat_obs_queries += (long)q.size();
        return true;
    }
};

bool GPisMap::Impl::preproData(const float* datax, const float* dataf, int N, const std::vector<float>& pose) {  // :105-149
    if (!datax || !dataf || N < 1) return false;
    obs_theta.clear(); obs_range.clear(); obs_f.clear(); obs_xylocal.clear(); obs_xyglobal.clear();
    range_obs_max = 0.0f;
    if (pose.size() != 6) return false;
    pose_tr[0] = pose[0]; pose_tr[1] = pose[1];
    for (int i = 0; i < 4; ++i) pose_R[i] = pose[2 + i];
    obs_numdata = 0;
    for (int k = 0; k < N; ++k) {
        float xloc = 0.f, yloc = 0.f;
        if ((double)dataf[k] < 3e1 && (double)dataf[k] > 2e-1) {
            if (range_obs_max < dataf[k]) range_obs_max = dataf[k];
            obs_theta.push_back(datax[k]);
            obs_range.push_back(dataf[k]);
            obs_f.push_back((float)(1.0 / (double)std::sqrt(dataf[k])));
            polar2Cart(datax[k], dataf[k], xloc, yloc);
            obs_xylocal.push_back(xloc); obs_xylocal.push_back(yloc);
            xloc += setting.sensor_offset[0];
            yloc += setting.sensor_offset[1];
            obs_xyglobal.push_back(pose_R[0] * xloc + pose_R[2] * yloc + pose_tr[0]);
            obs_xyglobal.push_back(pose_R[1] * xloc + pose_R[3] * yloc + pose_tr[1]);
            ++obs_numdata;
        }
    }
    return obs_numdata > 1;
}

int GPisMap::Impl::try_insert(int pid, T2::InsSet& ins) {  // GPisMap.cpp:433-443 / :497-507
    bool ok_ = false;
    if (!tree.is_not_new_cached(tree.pts[pid].pos)) {
        ok_ = tree.insert_cached(pid, &ins);
        if (ok_ && !tree.is_root(tree.root)) tree.root = tree.get_root(tree.root);
    }
    if (!ok_) { tree.drop_point(pid); return 0; }
    return ins.empty() ? 1 : 2;
}

// GPisMap.cpp:243-345 for a batch of points: centre query, gates, line search in rounds, then
// the four perturbation queries.
void GPisMap::Impl::reeval_batch(const std::vector<int>& ids, std::vector<Work>& w) {
    const int n = (int)ids.size();
    w.assign(n, Work());
    if (n == 0) return;
    const float delx = setting.delx;
    std::vector<float> q(n), val, var;
    std::vector<float> xl(n), yl(n), rr(n);
    for (int i = 0; i < n; ++i) {
        const float* pos = tree.pts[ids[i]].pos;
        float x_loc = pose_R[0] * (pos[0] - pose_tr[0]) + pose_R[1] * (pos[1] - pose_tr[1]);
        float y_loc = pose_R[2] * (pos[0] - pose_tr[0]) + pose_R[3] * (pos[1] - pose_tr[1]);
        x_loc -= setting.sensor_offset[0];
        y_loc -= setting.sensor_offset[1];
        float ang, r;
        cart2polar(x_loc, y_loc, ang, r);
        xl[i] = x_loc; yl[i] = y_loc; rr[i] = r; q[i] = ang;
    }
    if (!query(q, val, var)) return;
    struct Iter { int i; float oc, abs_oc, dx, gl0, gl1; };
    std::vector<Iter> act;
    for (int i = 0; i < n; ++i) {
        if (var[i] > setting.obs_var_thre) continue;
        float r = rr[i];
        float oc = occ_test((float)(1.0 / (double)std::sqrt(r)), val[i], (float)((double)r * 30.0));
        if ((double)oc < -0.1) continue;
        const float* grad = tree.pts[ids[i]].grad;
        Work& wk = w[i];
        wk.go = true;
        wk.x_new[0] = xl[i]; wk.x_new[1] = yl[i];
        wk.r_new = r;
        wk.abs_oc = (float)std::fabs((double)oc);
        Iter it;
        it.i = i; it.oc = oc; it.abs_oc = wk.abs_oc; it.dx = delx;
        it.gl0 = pose_R[0] * grad[0] + pose_R[1] * grad[1];
        it.gl1 = pose_R[2] * grad[0] + pose_R[3] * grad[1];
        if ((double)it.abs_oc > 0.02) act.push_back(it);
    }
    // line search: round k performs iteration k of every point still in its loop
    for (int round = 0; round < 10 && !act.empty(); ++round) {
        std::vector<float> qa(act.size()), va, vr;
        std::vector<float> rnew(act.size());
        for (size_t a = 0; a < act.size(); ++a) {
            Iter& it = act[a];
            Work& wk = w[it.i];
            if (it.oc < 0) { wk.x_new[0] += it.gl0 * it.dx; wk.x_new[1] += it.gl1 * it.dx; }
            else { wk.x_new[0] -= it.gl0 * it.dx; wk.x_new[1] -= it.gl1 * it.dx; }
            float ang;
            cart2polar(wk.x_new[0], wk.x_new[1], ang, rnew[a]);
            wk.r_new = rnew[a];
            qa[a] = ang;
        }
        if (!query(qa, va, vr)) return;
        std::vector<Iter> next;
        for (size_t a = 0; a < act.size(); ++a) {
            Iter it = act[a];
            Work& wk = w[it.i];
            if (vr[a] > setting.obs_var_thre) continue;  // break
            float r_new = rnew[a];
            float oc_new = occ_test((float)(1.0 / (double)std::sqrt(r_new)), va[a], (float)((double)r_new * 30.0));
            float abs_oc_new = (float)std::fabs((double)oc_new);
            if ((double)abs_oc_new < 0.02 || (double)it.oc < -0.1) continue;  // break (abs_oc / oc keep their old values)
            else if ((double)(it.oc * oc_new) < 0.0) it.dx = (float)(0.5 * (double)it.dx);
            else it.dx = (float)(1.1 * (double)it.dx);
            it.abs_oc = abs_oc_new; it.oc = oc_new;
            wk.abs_oc = abs_oc_new;
            if ((double)it.abs_oc > 0.02) next.push_back(it);
        }
        act.swap(next);
    }
    // perturbations around x_new
    static const float pert[2][4] = {{1, -1, 0, 0}, {0, 0, 1, -1}};
    std::vector<float> qp((size_t)4 * n, 1e30f), vp, rp;
    for (int i = 0; i < n; ++i) {
        if (!w[i].go) continue;
        for (int k = 0; k < 4; ++k) {
            float X = w[i].x_new[0] + delx * pert[0][k];
            float Y = w[i].x_new[1] + delx * pert[1][k];
            float a, r_;
            cart2polar(X, Y, a, r_);
            qp[(size_t)4 * i + k] = a;
        }
    }
    if (!query(qp, vp, rp)) return;
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 4; ++k) { w[i].pval[k] = vp[(size_t)4 * i + k]; w[i].pvar[k] = rp[(size_t)4 * i + k]; }
}

void GPisMap::Impl::reeval_apply(int pid, const Work& s) {  // GPisMap.cpp:319-453
    if (!s.go) return;
    const float delx = setting.delx;
    static const float pert[2][4] = {{1, -1, 0, 0}, {0, 0, 1, -1}};
    float occ[4] = {-1, -1, -1, -1};
    float occ_mean = 0.f, r0_mean = 0.f, r0_sqr_sum = 0.f;
    float var = 0.f;
    for (int i = 0; i < 4; ++i) {
        float X = s.x_new[0] + delx * pert[0][i];
        float Y = s.x_new[1] + delx * pert[1][i];
        float a, r_;
        cart2polar(X, Y, a, r_);
        var = s.pvar[i];
        if (var > setting.obs_var_thre) break;
        float rinv0 = s.pval[i];
        occ[i] = occ_test((float)(1.0 / (double)std::sqrt(r_)), rinv0, (float)((double)r_ * 30.0));
        occ_mean = (float)((double)occ_mean + 0.25 * (double)occ[i]);
        float r0 = (float)(1.0 / (double)(rinv0 * rinv0));
        r0_sqr_sum += r0 * r0;
        r0_mean = (float)((double)r0_mean + 0.25 * (double)r0);
    }
    if (var > setting.obs_var_thre) return;
    FlatPoint<2> nd = tree.pts[pid];
    const float* pos = nd.pos;
    const float* grad = nd.grad;
    float gl[2] = {(occ[0] - occ[1]) / delx, (occ[2] - occ[3]) / delx};
    float norm_g = std::sqrt(gl[0] * gl[0] + gl[1] * gl[1]);
    if ((double)norm_g < 1e-3) {
        tree.pts[pid].sigx = (float)(2.0 * (double)nd.sigx);
        tree.pts[pid].sigg = (float)(2.0 * (double)nd.sigg);
        return;
    }
    float r_var = (float)((double)r0_sqr_sum / 3.0 - (double)(r0_mean * r0_mean) * 4.0 / 3.0);
    r_var /= delx;
    float noise = 100.0f, grad_noise = 1.0f;
    float r_new = s.r_new;
    if ((double)norm_g > 1e-6) {
        gl[0] = gl[0] / norm_g; gl[1] = gl[1] / norm_g;
        noise = setting.min_position_noise * saturate(r_new * r_new, 1.0f, noise);
        grad_noise = saturate(std::fabs(occ_mean) + r_var, setting.min_grad_noise, grad_noise);
    } else noise = setting.min_position_noise * noise;
    float x_new[2] = {s.x_new[0], s.x_new[1]};
    float dist = std::sqrt(x_new[0] * x_new[0] + x_new[1] * x_new[1]);
    float view_ang = std::max(-(x_new[0] * gl[0] + x_new[1] * gl[1]) / dist, (float)1e-1);
    float view_ang2 = view_ang * view_ang;
    float view_noise = (float)((double)setting.min_position_noise * ((1.0 - (double)view_ang2) / (double)view_ang2));
    noise += view_noise + s.abs_oc;
    grad_noise = (float)((double)grad_noise + 0.1 * (double)view_noise);
    float pos_new[2], grad_new[2];
    x_new[0] += setting.sensor_offset[0];
    x_new[1] += setting.sensor_offset[1];
    pos_new[0] = pose_R[0] * x_new[0] + pose_R[2] * x_new[1] + pose_tr[0];
    pos_new[1] = pose_R[1] * x_new[0] + pose_R[3] * x_new[1] + pose_tr[1];
    grad_new[0] = pose_R[0] * gl[0] + pose_R[2] * gl[1];
    grad_new[1] = pose_R[1] * gl[0] + pose_R[3] * gl[1];
    float noise_old = nd.sigx, grad_noise_old = nd.sigg;
    float pos_noise_sum = noise_old + noise;
    float grad_noise_sum = grad_noise_old + grad_noise;
    if ((double)grad_noise_old > 0.5 || (double)grad_noise_old > 0.6) {
        ;
    } else {
        pos_new[0] = (noise * pos[0] + noise_old * pos_new[0]) / pos_noise_sum;
        pos_new[1] = (noise * pos[1] + noise_old * pos_new[1]) / pos_noise_sum;
        float dist2 = (float)(0.5 * (double)std::sqrt((pos[0] - pos_new[0]) * (pos[0] - pos_new[0]) +
                                                       (pos[1] - pos_new[1]) * (pos[1] - pos_new[1])));
        float tv[2];
        tv[0] = grad[0] * grad_new[0] + grad[1] * grad_new[1];
        tv[1] = -grad[1] * grad_new[0] + grad[0] * grad_new[1];
        float ang_dist = (float)std::atan2((double)tv[1], (double)tv[0]) * noise / pos_noise_sum;
        float sina = (float)std::sin((double)ang_dist);
        float cosa = (float)std::cos((double)ang_dist);
        grad_new[0] = cosa * grad[0] - sina * grad[1];
        grad_new[1] = sina * grad[0] + cosa * grad[1];
        grad_noise = std::min((float)1.0, std::max(grad_noise * grad_noise_old / grad_noise_sum + dist2, setting.map_noise_param));
        noise = std::max((noise * noise_old / pos_noise_sum + dist2), setting.map_noise_param);
    }
    tree.remove(tree.root, nd.pos, &activeSet);
    if ((double)noise > 1.0 && (double)grad_noise > 0.61) return;
    int np = tree.new_point(pos_new);
    T2::InsSet ins;
    if (try_insert(np, ins) != 2) return;
    FlatPoint<2>& p = tree.pts[np];
    p.val = -setting.fbias; p.sigx = noise; p.sigg = grad_noise; p.type = 1;
    p.grad[0] = grad_new[0]; p.grad[1] = grad_new[1];
    ins.for_each([&](int c) { activeSet.insert(c); });
}

void GPisMap::Impl::updateMapPoints() {  // GPisMap.cpp:181-233
    if (!has_tree || !gpo_created) return;
    std::vector<int> quads;
    tree.query_clusters(tree.root, pose_tr, range_obs_max, quads, nullptr);
    float r2 = range_obs_max * range_obs_max;
    std::vector<int> sel;
    for (int c : quads) {
        const T2::TNode& cn = tree.nodes[c];
        float l = cn.h;
        float sqr_range = (cn.c[0] - pose_tr[0]) * (cn.c[0] - pose_tr[0]) + (cn.c[1] - pose_tr[1]) * (cn.c[1] - pose_tr[1]);
        if (sqr_range > (r2 + 2 * l * l)) continue;
        int within_angle = 0;
        for (int i = 0; i < 4; ++i) {
            float e[2] = {(i & 1) ? cn.hi[0] : cn.lo[0], (i & 2) ? cn.lo[1] : cn.hi[1]};
            float x_loc = pose_R[0] * (e[0] - pose_tr[0]) + pose_R[1] * (e[1] - pose_tr[1]);
            float y_loc = pose_R[2] * (e[0] - pose_tr[0]) + pose_R[3] * (e[1] - pose_tr[1]);
            x_loc -= setting.sensor_offset[0];
            y_loc -= setting.sensor_offset[1];
            float ang = 0.f, r = 0.f;
            cart2polar(x_loc, y_loc, ang, r);
            within_angle += int((ang > setting.angle_obs_limit[0]) && (ang < setting.angle_obs_limit[1]));
        }
        if (within_angle == 0) continue;
        sel.push_back(c);
    }
    if (sel.empty()) return;
    std::vector<int> ids;
    for (int c : sel) tree.all_points(c, ids);
    std::vector<Work> w;
    reeval_batch(ids, w);
    std::vector<int> slot(tree.pts.size(), -1);
    for (size_t i = 0; i < ids.size(); ++i) slot[ids[i]] = (int)i;
    std::vector<int> nodes, late;
    for (int c : sel) {
        nodes.clear();
        tree.all_points(c, nodes);
        late.clear();
        for (int pid : nodes) if (pid >= (int)slot.size() || slot[pid] < 0) late.push_back(pid);
        std::vector<Work> lw;
        if (!late.empty()) { reeval_batch(late, lw); stat_late += (long)late.size(); }
        size_t li = 0;
        for (int pid : nodes) {
            if (pid < (int)slot.size() && slot[pid] >= 0) reeval_apply(pid, w[slot[pid]]);
            else reeval_apply(pid, lw[li++]);
        }
    }
}

void GPisMap::Impl::evalPoints() {  // GPisMap.cpp:466-572
    if (!has_tree || obs_numdata < 1) return;
    const float delx = setting.delx;
    const int n = obs_numdata;
    static const float pert[2][4] = {{1, -1, 0, 0}, {0, 0, 1, -1}};
    std::vector<float> q((size_t)5 * n), val, var, rs((size_t)4 * n);
    for (int k = 0; k < n; ++k) {
        q[(size_t)5 * k] = obs_theta[k];
        for (int i = 0; i < 4; ++i) {
            float X = obs_xylocal[2 * k] + delx * pert[0][i];
            float Y = obs_xylocal[2 * k + 1] + delx * pert[1][i];
            float a, r;
            cart2polar(X, Y, a, r);
            q[(size_t)5 * k + 1 + i] = a;
            rs[(size_t)4 * k + i] = r;
        }
    }
    if (!query(q, val, var)) return;
    for (int k = 0; k < n; ++k) {
        const float* pv = &val[(size_t)5 * k];
        const float* pr = &var[(size_t)5 * k];
        if (pr[0] > setting.obs_var_thre) continue;
        int pid = tree.new_point(&obs_xyglobal[2 * (size_t)k]);
        T2::InsSet ins;
        if (try_insert(pid, ins) != 2) continue;
        float occ[4] = {-1, -1, -1, -1};
        float occ_mean = 0.f;
        float v = pr[0];
        for (int i = 0; i < 4; ++i) {
            float r = rs[(size_t)4 * k + i];
            v = pr[1 + i];
            if (v > setting.obs_var_thre) break;
            occ[i] = occ_test((float)(1.0 / (double)std::sqrt(r)), pv[1 + i], (float)((double)r * 30.0));
            occ_mean = (float)((double)occ_mean + 0.25 * (double)occ[i]);
        }
        if (v > setting.obs_var_thre) { tree.remove(tree.root, tree.pts[pid].pos, nullptr); continue; }
        float noise = 100.0f, grad_noise = 1.00f;
        float g[2] = {(occ[0] - occ[1]) / delx, (occ[2] - occ[3]) / delx};
        float norm_grad = g[0] * g[0] + g[1] * g[1];
        if ((double)norm_grad > 1e-6) {
            norm_grad = std::sqrt(norm_grad);
            float gx = g[0] / norm_grad, gy = g[1] / norm_grad;
            g[0] = pose_R[0] * gx + pose_R[2] * gy;
            g[1] = pose_R[1] * gx + pose_R[3] * gy;
            noise = setting.min_position_noise * saturate(obs_range[k] * obs_range[k], 1.0f, noise);
            grad_noise = saturate(std::fabs(occ_mean), setting.min_grad_noise, grad_noise);
            float xl = obs_xylocal[2 * k], yl = obs_xylocal[2 * k + 1];
            float dist = std::sqrt(xl * xl + yl * yl);
            float view_ang = std::max(-(xl * gx + yl * gy) / dist, (float)1e-1);
            float view_ang2 = view_ang * view_ang;
            float view_noise = (float)((double)setting.min_position_noise * ((1.0 - (double)view_ang2) / (double)view_ang2));
            noise += view_noise;
        }
        FlatPoint<2>& p = tree.pts[pid];
        p.val = -setting.fbias; p.sigx = noise; p.sigg = grad_noise; p.type = 1;
        p.grad[0] = g[0]; p.grad[1] = g[1];
        ins.for_each([&](int c) { activeSet.insert(c); });
    }
}

void GPisMap::Impl::updateGPs() {  // GPisMap.cpp:574-663 -> K6 + K3
    T2::Set updateSet(activeSet);
    std::vector<int> qs;
    for (int a : activeSet) {
        qs.clear();
        tree.query_clusters(tree.root, tree.nodes[a].c, (float)(4.0 * (double)tree.nodes[a].h), qs, nullptr);
        for (int c : qs) updateSet.insert(c);
    }
    (void)join_training();      // the previous frame's batch, before any model slot is released or allocated
    for (int m : tree.released_models) store.release_slot(m);
    tree.released_models.clear();
    if (!updateSet.empty()) {  // (the reference divides by zero on an empty set, SURVEY B-5)
        std::vector<int> todo(updateSet.begin(), updateSet.end());
        std::sort(todo.begin(), todo.end());
        std::vector<TrainJob> jobs;
        std::vector<int> ids, res;
        T2::CellLists cell_lists;   // the points of every touched cell listed once for the whole batch (flat_tree.h)
        cell_lists.reset(tree.nodes.size());
        for (int c : todo) {
            res.clear();
            tree.query_range_cells(tree.nodes[c].c, (float)((double)tree.nodes[c].h * 4.0), cell_lists, res);
            if (res.empty()) continue;
            int ng = 0;
            for (int pid : res) {
                const FlatPoint<2>& p = tree.pts[pid];
                bool tiny = ((double)std::fabs(p.grad[0]) < 1e-6) && ((double)std::fabs(p.grad[1]) < 1e-6);
                if (!(((double)p.sigg > 0.1001) || tiny)) ++ng;
            }
            if (tree.nodes[c].model < 0) tree.nodes[c].model = store.new_slot();
            TrainJob j;
            j.model = tree.nodes[c].model; j.off = (int)ids.size(); j.n = (int)res.size(); j.ng = ng;
            jobs.push_back(j);
            ids.insert(ids.end(), res.begin(), res.end());
        }
        if (!jobs.empty()) {
            size_t np = tree.pts.size();
            std::vector<float> soa(9 * np, 0.f);
            for (size_t i = 0; i < np; ++i) {
                const FlatPoint<2>& p = tree.pts[i];
                for (int d = 0; d < 2; ++d) { soa[d * np + i] = p.pos[d]; soa[(3 + d) * np + i] = p.grad[d]; }
                soa[6 * np + i] = p.val; soa[7 * np + i] = p.sigx; soa[8 * np + i] = p.sigg;
            }
            int rc = store.upload_points(soa.data(), (int)np, stream);
            store.defer_finish = pipeline;
            if (rc == GPIS_OK) rc = store.train_batch(jobs, ids, stream);
            if (rc != GPIS_OK) { fprintf(stderr, "[gpismap_amd] OnGPIS training failed (%d)\n", rc); if (!upd_rc) upd_rc = rc; }
            stat_clusters_trained += (long)jobs.size();
        }
    }
    activeSet.clear();
    rebuild_table();
}

void GPisMap::Impl::rebuild_table() {
    std::vector<int> cl;
    tree.all_clusters(cl);
    std::vector<ClusterEntry> ent(cl.size());
    std::vector<AncestorEntry> anc;
    std::unordered_map<int, int> anc_of;
    std::function<int(int)> anc_index = [&](int node) -> int {
        if (node < 0) return -1;
        auto it = anc_of.find(node);
        if (it != anc_of.end()) return it->second;
        const T2::TNode& a = tree.nodes[node];
        int up = (node == tree.root) ? -1 : anc_index(a.par);
        AncestorEntry e;
        for (int d = 0; d < 2; ++d) { e.lo[d] = a.lo[d]; e.hi[d] = a.hi[d]; }
        e.lo[2] = 0.f; e.hi[2] = 0.f;
        e.parent = up;
        anc.push_back(e);
        anc_of[node] = (int)anc.size() - 1;
        return (int)anc.size() - 1;
    };
    for (size_t i = 0; i < cl.size(); ++i) {
        const T2::TNode& t = tree.nodes[cl[i]];
        for (int d = 0; d < 2; ++d) { ent[i].c[d] = t.c[d]; ent[i].lo[d] = t.lo[d]; ent[i].hi[d] = t.hi[d]; }
        ent[i].c[2] = 0.f; ent[i].lo[2] = 0.f; ent[i].hi[2] = 0.f;
        { const ClusterModel* mm = store.model(t.model); ent[i].model = (mm && mm->base) ? t.model : -1; }
        ent[i].parent = anc_index(t.par);
    }
    int rc = mq.set_clusters(ent, anc, 2.0 * (double)tree.prm.cluster_half, stream);
    if (rc != GPIS_OK) { fprintf(stderr, "[gpismap_amd] cluster table upload failed (%d)\n", rc); if (!upd_rc) upd_rc = rc; }
}

// The reference's gateways call the class methods directly (mexGPisMap3.cpp:70,102,150; mexGPisMap.cpp:70,108):
// nothing may propagate out of them into MATLAB.  Every public method is a function-try-block.
static void nothrow_report(const char* where, const char* what) {
    fprintf(stderr, "[gpismap_amd] %s: exception contained (%s)\n", where, what);
}

GPisMap::GPisMap() : p_(new Impl(GPisMapParam())) {}
GPisMap::GPisMap(GPisMapParam par) : p_(new Impl(par)) {}
GPisMap::~GPisMap() {
    DeviceScope dev_scope_(p_->device);
    delete p_;
}
void GPisMap::reset() try {
    DeviceScope dev_scope_(p_->device);
    p_->reset();
} catch (const std::exception& e) { nothrow_report("GPisMap::reset", e.what()); } catch (...) { nothrow_report("GPisMap::reset", "unknown exception"); }

void GPisMap::update(float* datax, float* dataf, int N, std::vector<float>& pose) try {  // GPisMap.cpp:151-167
    DeviceScope dev_scope_(p_->device);
    Impl& m = *p_;
    m.upd_rc = 0;
    if (!m.ok) { m.upd_rc = GPIS_ERR_HIP; fprintf(stderr, "[gpismap_amd] GPisMap::update: HIP device unavailable\n"); return; }
    m.tree.recycle();
    if (!m.preproData(datax, dataf, N, pose)) return;
    m.gpo_created = true;
    int rc = m.gpo.train1d(m.obs_theta.data(), m.obs_f.data(), m.obs_numdata, m.stream);  // regressObs :169-179 -> K1
    if (rc != GPIS_OK || !m.gpo.trained()) { if (rc) { fprintf(stderr, "[gpismap_amd] ObsGP training failed (%d)\n", rc); if (!m.upd_rc) m.upd_rc = rc; } return; }
    m.updateMapPoints();
    if (!m.has_tree) {
        float c[2] = {0.f, 0.f};
        m.tree.make_root(c);
        m.has_tree = true;
    }
    m.evalPoints();
    m.updateGPs();
} catch (const std::exception& e) { nothrow_report("GPisMap::update", e.what()); p_->upd_rc = GPIS_ERR_STATE; } catch (...) { nothrow_report("GPisMap::update", "unknown exception"); p_->upd_rc = GPIS_ERR_STATE; }

bool GPisMap::testDevice(const float* d_x, int leng, float* d_res, void* hip_stream) try {
    DeviceScope dev_scope_(p_->device);
    Impl& m = *p_;
    m.fail_rc = 0;
    if (!m.ok || !d_x || !d_res || leng < 1 || !m.has_tree) return false;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : m.stream;
    m.fail_rc = 0;
    (void)m.join_training();
    const int rc = m.mq.run(m.store, d_x, leng, d_res, s);
    if (rc != GPIS_OK) { m.fail_rc = rc; fprintf(stderr, "[gpismap_amd] GPisMap::testDevice: device path failed (%d)\n", rc); }
    return rc == GPIS_OK;
} catch (const std::exception& e) { nothrow_report("GPisMap::testDevice", e.what()); p_->fail_rc = GPIS_ERR_STATE; return false; } catch (...) { nothrow_report("GPisMap::testDevice", "unknown exception"); p_->fail_rc = GPIS_ERR_STATE; return false; }

bool GPisMap::test(float* x, int dim, int leng, float* res) try {  // GPisMap.cpp:765-810
    DeviceScope dev_scope_(p_->device);
    Impl& m = *p_;
    m.fail_rc = 0;
    if (x == 0 || dim != 2 || leng < 1) return false;
    if (!m.ok) { fprintf(stderr, "[gpismap_amd] GPisMap::test: HIP device unavailable\n"); return false; }
    if (!m.has_tree) return false;
    m.fail_rc = 0;
    auto fail = [&](int rc) { m.fail_rc = rc; fprintf(stderr, "[gpismap_amd] GPisMap::test: device path failed (%d)\n", rc); return false; };
    (void)m.join_training();
    size_t nx = (size_t)2 * leng, nr = (size_t)6 * leng;
    if (nx > m.cap_x) { (void)hipFree(m.d_x); m.d_x = nullptr; m.cap_x = 0; if (hipMalloc(&m.d_x, sizeof(float) * nx) != hipSuccess) return fail(GPIS_ERR_HIP); m.cap_x = nx; }
    if (nr > m.cap_res) { (void)hipFree(m.d_res); m.d_res = nullptr; m.cap_res = 0; if (hipMalloc(&m.d_res, sizeof(float) * nr) != hipSuccess) return fail(GPIS_ERR_HIP); m.cap_res = nr; }
    if (hipMemcpyAsync(m.d_x, x, sizeof(float) * nx, hipMemcpyHostToDevice, m.stream) != hipSuccess) return fail(GPIS_ERR_HIP);
    if (hipMemcpyAsync(m.d_res, res, sizeof(float) * nr, hipMemcpyHostToDevice, m.stream) != hipSuccess) return fail(GPIS_ERR_HIP);
    { const int rc = m.mq.run(m.store, m.d_x, leng, m.d_res, m.stream); if (rc != GPIS_OK) return fail(rc); }
    if (hipMemcpyAsync(res, m.d_res, sizeof(float) * nr, hipMemcpyDeviceToHost, m.stream) != hipSuccess) return fail(GPIS_ERR_HIP);
    if (hipStreamSynchronize(m.stream) != hipSuccess) return fail(GPIS_ERR_HIP);
    return true;
} catch (const std::exception& e) { nothrow_report("GPisMap::test", e.what()); p_->fail_rc = GPIS_ERR_STATE; return false; } catch (...) { nothrow_report("GPisMap::test", "unknown exception"); p_->fail_rc = GPIS_ERR_STATE; return false; }

void GPisMap::getAllNodes(std::vector<float>& out) try {
    out.clear();
    Impl& m = *p_;
    if (!m.has_tree) return;
    std::vector<int> ids;
    m.tree.all_points(m.tree.root, ids);
    for (int id : ids) {
        const FlatPoint<2>& p = m.tree.pts[id];
        out.push_back(p.pos[0]); out.push_back(p.pos[1]); out.push_back(p.grad[0]); out.push_back(p.grad[1]);
        out.push_back(p.val); out.push_back(p.sigx); out.push_back(p.sigg);
    }
} catch (const std::exception& e) { nothrow_report("GPisMap::getAllNodes", e.what()); } catch (...) { nothrow_report("GPisMap::getAllNodes", "unknown exception"); }

int gpis2_impl_sync(GPisMap* g) {      // join the training the last update() left in flight; the update status
    GPisMap::Impl& m = *g->impl();
    DeviceScope ds(m.device);
    (void)m.join_training();
    return m.upd_rc;
}
void gpis2_impl_set_pipeline(GPisMap* g, int on) {
    GPisMap::Impl& m = *g->impl();
    DeviceScope ds(m.device);
    (void)m.join_training();
    m.pipeline = on != 0;
}
void gpis2_impl_stats(GPisMap* g, double* out, int n) {
    GPisMap::Impl& m = *g->impl();
    { DeviceScope ds(m.device); (void)m.join_training(); }      // (the training time of the last batch is read off its events)
    double v[12] = {(double)m.gpo.trained_groups(m.stream), (double)m.stat_obs_queries, (double)m.stat_clusters_trained,
                    (double)m.stat_late, (double)m.mq.num_clusters(), (double)m.mq.last_evals, (double)m.mq.last_eval_ms,
                    (double)m.store.device_bytes(), (double)m.mq.last_flops, (double)m.mq.last_launches,
                    (double)m.store.last_train_ms, 0.0};
    for (int i = 0; i < n && i < 12; ++i) out[i] = v[i];
}

int gpis2_impl_fail(GPisMap* g) { return g->impl()->fail_rc; }
int gpis2_impl_device(GPisMap* g) { return g->impl()->device; }
int gpis2_impl_update_fail(GPisMap* g) { return g->impl()->upd_rc; }
