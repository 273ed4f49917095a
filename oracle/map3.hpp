// ORACLE (test infrastructure only; parity unpinned -- see linalg.hpp).
// CPU restatement of the 3-D map orchestrator, reference cpp/src/GPisMap3.cpp
// (+ cpp/include/GPisMap3.h, params.h).  Sequential host logic as in the
// reference; ObsGP single-point queries are evaluated inline instead of
// spawning a std::thread per query (ObsGP.cpp:418-457) -- identical results.
#pragma once
#include <array>
#include <atomic>
#include <cstring>
#include <thread>
#include "tree.hpp"

namespace orc {

struct CamParam {  // GPisMap3.h:29-46
    float fx = 568.0f, fy = 568.0f, cx = 310.f, cy = 224.f;
    int width = 640, height = 480;
};

struct Map3Param {  // GPisMap3.h:48-81, params.h:77-93
    float delx = (float)1e-3;
    float fbias = (float)0.2;
    float obs_var_thre = (float)0.04;
    int obs_skip = 2;
    float min_position_noise = (float)1e-3;
    float min_grad_noise = (float)1e-2;
    float map_scale_param = (float)0.04;
    float map_noise_param = (float)5e-3;
};

template <class F>
static inline void parallel_for(int n, int nthreads, F&& fn) {
    if (n <= 0) return;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > n) nthreads = n;
    if (nthreads == 1) { fn(0, n); return; }
    std::vector<std::thread> th;
    int base = n / nthreads, rem = n % nthreads, cur = 0;
    for (int i = 0; i < nthreads; ++i) {
        int len = base + (i < rem ? 1 : 0);
        th.emplace_back([=, &fn] { fn(cur, cur + len); });
        cur += len;
    }
    for (auto& t : th) t.join();
}

static inline float occ_test(float rinv, float rinv0, float a) {  // GPisMap3.cpp:38-41
    return (float)(2.0 * (1.0 / (1.0 + std::exp((double)(-a * (rinv - rinv0)))) - 0.5));
}
static inline float saturate(float v, float lo, float hi) { return std::min(std::max(v, lo), hi); }

struct Map3Stats {
    long obsgp_tiles = 0, obsgp_queries = 0, clusters_trained = 0, sumK = 0, maxK = 0, gp_evals = 0;
};

class GPisMap3 {
public:
    using T3 = Tree<3>;
    using NodeP = T3::NodeP;

    Map3Param setting;
    CamParam cam;
    int nthreads = (int)std::thread::hardware_concurrency();
    Map3Stats stats;

    GPisMap3() { init(); }
    explicit GPisMap3(const CamParam& c) : cam(c) { init(); }
    ~GPisMap3() { reset(); }

    void reset() {  // GPisMap3.cpp:99-115
        delete t; t = nullptr;
        gpo.reset();
        obs_numdata = 0;
        activeSet.clear();
    }
    void resetCam(const CamParam& c) { cam = c; vu_grid.clear(); }  // :117-123

    void update(const float* dataz, int N, const float* pose, int npose) {  // :218-237
        if (!preprocData(dataz, N, pose, npose)) return;
        if (regressObs()) {
            updateMapPoints();
            addNewMeas();
            updateGPs();
        }
    }

    bool test(const float* x, int dim, int leng, float* res) {  // :904-949
        if (!x || dim != 3 || leng < 1) return false;
        if (!t) return false;  // reference dereferences a null tree here (SURVEY B-9)
        std::atomic<long> ev{0};
        parallel_for(leng, nthreads, [&](int a, int b) {
            long e = 0;
            for (int i = a; i < b; ++i) e += test_one(x + 3 * (size_t)i, res + 8 * (size_t)i);
            ev += e;
        });
        stats.gp_evals += ev.load();
        return true;
    }

    // per-query ambiguity flags (bit0 distance tie, bit1 variance near the 0.5 gate, bit2 near-equal
    // candidate variances); the outputs themselves are discarded
    void testFlags(const float* x, int leng, int* flags) {
        if (!t) return;
        parallel_for(leng, nthreads, [&](int a, int b) {
            float r[8];
            for (int i = a; i < b; ++i) { for (float& v : r) v = 0.f; flags[i] = 0; test_one(x + 3 * (size_t)i, r, &flags[i]); }
        });
    }

    void getAllPoints(std::vector<float>& pos) {  // :951-972
        pos.clear();
        if (!t) return;
        std::vector<NodeP> nodes;
        t->allNodes(nodes);
        for (auto& n : nodes) { pos.push_back(n->pos[0]); pos.push_back(n->pos[1]); pos.push_back(n->pos[2]); }
    }
    // Full point state (pos3 grad3 val sigx sigg) in tree order -- for parity of the map itself.
    void getAllNodes(std::vector<float>& out) {
        out.clear();
        if (!t) return;
        std::vector<NodeP> nodes;
        t->allNodes(nodes);
        for (auto& n : nodes) {
            for (int d = 0; d < 3; ++d) out.push_back(n->pos[d]);
            for (int d = 0; d < 3; ++d) out.push_back(n->grad[d]);
            out.push_back(n->val); out.push_back(n->sigx); out.push_back(n->sigg);
        }
    }
    // every trained cluster re-factorised in the current arithmetic mode on its stored training set (gp.hpp retrain)
    int retrainAll() {
        if (!t) return 0;
        float c0[3] = {0, 0, 0};
        std::vector<T3*> q;
        t->queryClusters(Box<3>(c0, 1e9f), q, nullptr);
        parallel_for((int)q.size(), nthreads, [&](int a, int b) {
            for (int i = a; i < b; ++i) if (q[i]->gp) q[i]->gp->retrain();
        });
        int n = 0;
        for (T3* c : q) if (c->gp && c->gp->trained) ++n;
        return n;
    }
    // training set of the i-th trained cluster (traversal order): n x 9 floats pos3 grad3 val sigx sigg; returns n
    int clusterSamples(int i, float* out9, int cap) {
        if (!t) return 0;
        float c0[3] = {0, 0, 0};
        std::vector<T3*> q;
        t->queryClusters(Box<3>(c0, 1e9f), q, nullptr);
        int k = 0;
        for (T3* c : q) {
            if (!c->gp || !c->gp->trained) continue;
            if (k++ != i) continue;
            const OnGPIS& g = *c->gp;
            const int n = (int)g.tr_val.size();
            if (out9 && n <= cap)
                for (int j = 0; j < n; ++j) {
                    for (int d = 0; d < 3; ++d) { out9[9 * j + d] = g.tr_pos[3 * j + d]; out9[9 * j + 3 + d] = g.tr_grad[3 * j + d]; }
                    out9[9 * j + 6] = g.tr_val[j]; out9[9 * j + 7] = g.tr_sx[j]; out9[9 * j + 8] = g.tr_sg[j];
                }
            return n;
        }
        return 0;
    }
    int numClusters() {
        if (!t) return 0;
        float c[3] = {0, 0, 0};
        std::vector<T3*> q;
        t->queryClusters(Box<3>(c, 1e9f), q, nullptr);
        return (int)q.size();
    }

    // exposed for component-level parity tests
    std::unique_ptr<ObsGP2D> gpo;
    std::vector<float> vu_grid, obs_zinv;
    T3* t = nullptr;

private:
    static constexpr float Rtimes = 2.0f;    // params.h:39
    static constexpr float C_leng = 0.025f;  // params.h:40
    TreeParam tprm;
    float u_obs_limit[2] = {0, 0}, v_obs_limit[2] = {0, 0};
    T3::Set activeSet;
    std::vector<float> obs_valid_u, obs_valid_v, obs_valid_xyzlocal, obs_valid_xyzglobal;
    float pose_tr[3] = {0, 0, 0}, pose_R[9] = {0};
    int obs_numdata = 0;
    float range_obs_max = 0.f;

    void init() {
        tprm.min_half = (float)(0.0125 / 2.0);  // params.h:41, GPisMap3.cpp:28-31
        tprm.max_half = (float)1.6;
        tprm.init_half = (float)0.4;
        tprm.cluster_half = C_leng;
        tprm.min_half_sq = tprm.min_half * tprm.min_half;
        tprm.cluster_eps = 1e-6;       // octree.cpp:325
        tprm.qleaf_eps_plain = 0.0001; // octree.cpp:837
        tprm.qleaf_eps_dist = 0.001;   // octree.cpp:870
        tprm.qdesc_eps = 0.001;        // octree.cpp:842,875
    }

    bool preprocData(const float* dataz, int N, const float* pose, int npose) {  // :125-216
        if (!dataz || N < 1) return false;
        obs_valid_xyzlocal.clear(); obs_valid_xyzglobal.clear();
        obs_valid_u.clear(); obs_valid_v.clear(); obs_zinv.clear();
        range_obs_max = 0.0f;
        if (npose != 12) return false;
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
        obs_numdata = 0;
        for (int n_ = 0; n_ < n; ++n_) {
            int col = n_ * setting.obs_skip;
            for (int m_ = 0; m_ < m; ++m_) {
                int row = m_ * setting.obs_skip;
                int k = col * cam.height + row;
                // isRangeValid compares against double literals 4e0 / 4e-1 (GPisMap3.cpp:33-36)
                if (k < N && (double)dataz[k] < 4e0 && (double)dataz[k] > 4e-1) {
                    int j = 2 * (m * n_ + m_);
                    float z = dataz[k];
                    if (range_obs_max < z) range_obs_max = z;
                    obs_zinv.push_back((float)(1.0 / (double)z));
                    float u = vu_grid[j + 1], v = vu_grid[j];
                    obs_valid_u.push_back(u); obs_valid_v.push_back(v);
                    float xloc = u * z, yloc = v * z;
                    obs_valid_xyzlocal.push_back(xloc); obs_valid_xyzlocal.push_back(yloc); obs_valid_xyzlocal.push_back(z);
                    obs_valid_xyzglobal.push_back(pose_R[0] * xloc + pose_R[3] * yloc + pose_R[6] * z + pose_tr[0]);
                    obs_valid_xyzglobal.push_back(pose_R[1] * xloc + pose_R[4] * yloc + pose_R[7] * z + pose_tr[1]);
                    obs_valid_xyzglobal.push_back(pose_R[2] * xloc + pose_R[5] * yloc + pose_R[8] * z + pose_tr[2]);
                    ++obs_numdata;
                } else obs_zinv.push_back(-1.0f);
            }
        }
        return obs_numdata > 1;
    }

    bool regressObs() {  // :239-256
        if (!gpo) gpo = std::make_unique<ObsGP2D>();
        if (2 * obs_zinv.size() != vu_grid.size()) return false;
        int ni = cam.height / setting.obs_skip, nj = cam.width / setting.obs_skip;
        gpo->reset();
        gpo->train(vu_grid.data(), obs_zinv.data(), ni, nj);
        if (gpo->trained) for (auto& g : gpo->gps) if (g) ++stats.obsgp_tiles;
        return gpo->trained;
    }

    void obs_query(float v, float u, float& rinv0, float& var) { gpo->test1(v, u, rinv0, var); }

    void updateMapPoints() {  // :258-319
        if (!t || !gpo) return;
        std::vector<T3*> oc;
        t->queryClusters(Box<3>(pose_tr, range_obs_max), oc, nullptr);
        float r2 = range_obs_max * range_obs_max;
        for (T3* c : oc) {
            const float* ct = c->box.c;
            float l = c->box.h;
            float sqr_range = (ct[0] - pose_tr[0]) * (ct[0] - pose_tr[0]) + (ct[1] - pose_tr[1]) * (ct[1] - pose_tr[1]) +
                              (ct[2] - pose_tr[2]) * (ct[2] - pose_tr[2]);
            if (sqr_range > (r2 + 2 * l * l)) continue;
            // corners in the order NWF,NEF,SWF,SEF,NWB,NEB,SWB,SEB; the flag is
            // overwritten, not accumulated (:298, SURVEY B-13)
            int within_angle = 0;
            for (int i = 0; i < 8; ++i) {
                float e[3] = {(i & 1) ? c->box.hi[0] : c->box.lo[0], (i & 2) ? c->box.lo[1] : c->box.hi[1],
                              (i & 4) ? c->box.lo[2] : c->box.hi[2]};
                float x_loc = pose_R[0] * (e[0] - pose_tr[0]) + pose_R[1] * (e[1] - pose_tr[1]) + pose_R[2] * (e[2] - pose_tr[2]);
                float y_loc = pose_R[3] * (e[0] - pose_tr[0]) + pose_R[4] * (e[1] - pose_tr[1]) + pose_R[5] * (e[2] - pose_tr[2]);
                float z_loc = pose_R[6] * (e[0] - pose_tr[0]) + pose_R[7] * (e[1] - pose_tr[1]) + pose_R[8] * (e[2] - pose_tr[2]);
                if (z_loc > 0) {
                    float xv = x_loc / z_loc, yv = y_loc / z_loc;
                    within_angle = int((xv > u_obs_limit[0]) && (xv < u_obs_limit[1]) && (yv > v_obs_limit[0]) && (yv < v_obs_limit[1]));
                }
            }
            if (within_angle == 0) continue;
            std::vector<NodeP> nodes;
            c->allNodes(nodes);
            reEvalPoints(nodes);
        }
    }

    static std::array<float, 9> quat2dcm(const float q[4]) {  // :48-63
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

    bool try_insert(const NodeP& p, T3::Set& ins) {  // :544-556, :611-623
        bool ok = false;
        if (!t->isNotNew(p)) {
            ok = t->insert(p, &ins);
            if (ok && !t->isRoot()) t = t->root();
        }
        return ok && !ins.empty();
    }

    void reEvalPoints(std::vector<NodeP>& nodes) {  // :321-569
        float rinv0 = 0.f, var = 0.f;
        const float w = (float)(1.0 / 6.0);
        const float delx = setting.delx;
        for (auto& nd : nodes) {
            const float* pos = nd->pos;
            float x_loc = pose_R[0] * (pos[0] - pose_tr[0]) + pose_R[1] * (pos[1] - pose_tr[1]) + pose_R[2] * (pos[2] - pose_tr[2]);
            float y_loc = pose_R[3] * (pos[0] - pose_tr[0]) + pose_R[4] * (pos[1] - pose_tr[1]) + pose_R[5] * (pos[2] - pose_tr[2]);
            float z_loc = pose_R[6] * (pos[0] - pose_tr[0]) + pose_R[7] * (pos[1] - pose_tr[1]) + pose_R[8] * (pos[2] - pose_tr[2]);
            if (z_loc < 0.0) continue;
            float v = y_loc / z_loc, u = x_loc / z_loc;
            float rinv = (float)(1.0 / (double)z_loc);
            obs_query(v, u, rinv0, var);
            if (var > setting.obs_var_thre) continue;
            float oc = occ_test(rinv, rinv0, (float)((double)z_loc * 30.0));
            if ((double)oc < -0.02) continue;

            const float* grad = nd->grad;
            float grad_loc[3];
            grad_loc[0] = pose_R[0] * grad[0] + pose_R[1] * grad[1] + pose_R[2] * grad[2];
            grad_loc[1] = pose_R[3] * grad[0] + pose_R[4] * grad[1] + pose_R[5] * grad[2];
            grad_loc[2] = pose_R[6] * grad[0] + pose_R[7] * grad[1] + pose_R[8] * grad[2];

            float abs_oc = (float)std::fabs((double)oc);
            float dx = delx;
            float x_new[3] = {x_loc, y_loc, z_loc};
            float r_new = z_loc;
            for (int i = 0; i < 10 && (double)abs_oc > 0.02; ++i) {
                if (oc < 0) for (int d = 0; d < 3; ++d) x_new[d] += grad_loc[d] * dx;
                else for (int d = 0; d < 3; ++d) x_new[d] -= grad_loc[d] * dx;
                // the reference re-queries the ORIGINAL location here (:390-393, SURVEY B-3)
                r_new = z_loc;
                obs_query(y_loc / z_loc, x_loc / z_loc, rinv0, var);
                if (var > setting.obs_var_thre) break;
                float oc_new = occ_test((float)(1.0 / (double)r_new), rinv0, (float)((double)r_new * 30.0));
                float abs_oc_new = (float)std::fabs((double)oc_new);
                if ((double)abs_oc_new < 0.02 || (double)oc < -0.02) break;
                else if ((double)(oc * oc_new) < 0.0) dx = (float)(0.5 * (double)dx);
                else dx = (float)(1.1 * (double)dx);
                abs_oc = abs_oc_new;
                oc = oc_new;
            }

            float pert[3][6] = {{1, -1, 0, 0, 0, 0}, {0, 0, 1, -1, 0, 0}, {0, 0, 0, 0, 1, -1}};
            float occ[6] = {-1, -1, -1, -1, -1, -1};
            float occ_mean = 0.f, r0_mean = 0.f, r0_sqr_sum = 0.f;
            for (int i = 0; i < 6; ++i) {
                float X = x_new[0] + delx * pert[0][i];
                float Y = x_new[1] + delx * pert[1][i];
                float Z = x_new[2] + delx * pert[2][i];
                r_new = Z;
                obs_query(Y / Z, X / Z, rinv0, var);
                if (var > setting.obs_var_thre) break;
                occ[i] = occ_test((float)(1.0 / (double)r_new), rinv0, (float)((double)r_new * 30.0));
                occ_mean += w * occ[i];
                float r0 = (float)(1.0 / (double)rinv0);
                r0_sqr_sum += r0 * r0;
                r0_mean += w * r0;
            }
            if (var > setting.obs_var_thre) continue;

            float gl[3] = {(occ[0] - occ[1]) / delx, (occ[2] - occ[3]) / delx, (occ[4] - occ[5]) / delx};
            float norm_g = std::sqrt(gl[0] * gl[0] + gl[1] * gl[1] + gl[2] * gl[2]);
            if ((double)norm_g < 1e-3) {
                nd->sigx = (float)(2.0 * (double)nd->sigx);
                nd->sigg = (float)(2.0 * (double)nd->sigg);
                continue;
            }
            float r_var = (float)((double)r0_sqr_sum / 5.0 - (double)(r0_mean * r0_mean) * 6.0 / 5.0);
            r_var /= delx;
            float noise = 100.0f, grad_noise = 1.0f;
            if ((double)norm_g > 1e-6) {
                for (int d = 0; d < 3; ++d) gl[d] = gl[d] / norm_g;
                noise = setting.min_position_noise * saturate(r_new * r_new, 1.0f, noise);
                grad_noise = saturate(std::fabs(occ_mean) + r_var, setting.min_grad_noise, grad_noise);
            } else noise = setting.min_position_noise * noise;

            float dist = std::sqrt(x_new[0] * x_new[0] + x_new[1] * x_new[1] + x_new[2] * x_new[2]);
            float view_ang = std::max(-(x_new[0] * gl[0] + x_new[1] * gl[1] + x_new[2] * gl[2]) / dist, (float)1e-1);
            float view_ang2 = view_ang * view_ang;
            float view_noise = (float)((double)setting.min_position_noise * ((1.0 - (double)view_ang2) / (double)view_ang2));
            noise += view_noise + abs_oc;
            grad_noise = (float)((double)grad_noise + 0.1 * (double)view_noise);

            float pos_new[3], grad_new[3];
            for (int d = 0; d < 3; ++d) {
                pos_new[d] = pose_R[d] * x_new[0] + pose_R[3 + d] * x_new[1] + pose_R[6 + d] * x_new[2] + pose_tr[d];
                grad_new[d] = pose_R[d] * gl[0] + pose_R[3 + d] * gl[1] + pose_R[6 + d] * gl[2];
            }
            float noise_old = nd->sigx, grad_noise_old = nd->sigg;
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
                // unqualified acos/cos/sin on float args resolve to the double versions
                float ang = (float)std::acos((double)(grad_new[0] * grad[0] + grad_new[1] * grad[1] + grad_new[2] * grad[2]));
                ang = ang * noise / pos_noise_sum;
                float q[4] = {1.0f, 0.0f, 0.0f, 0.0f};
                if (ang > 1 - 6) {  // sic (:517) -- always true unless ang is NaN
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
            t->remove(nd, &activeSet);
            if ((double)noise > 1.0 && (double)grad_noise > 0.61) continue;
            NodeP p = std::make_shared<MapNode<3>>(pos_new);
            T3::Set ins;
            if (!try_insert(p, ins)) continue;
            p->val = -setting.fbias; p->sigx = noise; p->sigg = grad_noise; p->type = 1;
            for (int d = 0; d < 3; ++d) p->grad[d] = grad_new[d];
            for (T3* c : ins) activeSet.insert(c);
        }
    }

    void addNewMeas() {  // :571-578
        if (!t) { float c[3] = {0, 0, 0}; t = T3::make_root(&tprm, c); }
        evalPoints();
    }

    void evalPoints() {  // :580-696
        if (!t || obs_numdata < 1) return;
        const float w = (float)(1.0 / 6.0);
        const float delx = setting.delx;
        for (int k = 0; k < obs_numdata; ++k) {
            int k3 = 3 * k;
            float rinv0 = 0.f, var = 0.f;
            obs_query(obs_valid_v[k], obs_valid_u[k], rinv0, var);
            if (var > setting.obs_var_thre) continue;
            NodeP p = std::make_shared<MapNode<3>>(&obs_valid_xyzglobal[k3]);
            T3::Set ins;
            if (!try_insert(p, ins)) continue;
            const float* xl = &obs_valid_xyzlocal[k3];
            float pert[3][6] = {{1, -1, 0, 0, 0, 0}, {0, 0, 1, -1, 0, 0}, {0, 0, 0, 0, 1, -1}};
            float occ[6] = {-1, -1, -1, -1, -1, -1};
            float occ_mean = 0.f;
            for (int i = 0; i < 6; ++i) {
                float X = xl[0] + delx * pert[0][i];
                float Y = xl[1] + delx * pert[1][i];
                float Z = xl[2] + delx * pert[2][i];
                obs_query(Y / Z, X / Z, rinv0, var);
                if (var > setting.obs_var_thre) break;
                occ[i] = occ_test((float)(1.0 / (double)Z), rinv0, (float)((double)Z * 30.0));
                occ_mean += w * occ[i];
            }
            if (var > setting.obs_var_thre) { t->remove(p, nullptr); continue; }
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
            p->val = -setting.fbias; p->sigx = noise; p->sigg = grad_noise; p->type = 1;
            for (int d = 0; d < 3; ++d) p->grad[d] = g[d];
            for (T3* c : ins) activeSet.insert(c);
        }
    }

    void updateGPs() {  // :698-792
        T3::Set updateSet(activeSet);
        for (T3* a : activeSet) {
            std::vector<T3*> qs;
            t->queryClusters(Box<3>(a->box.c, Rtimes * a->box.h), qs, nullptr);
            for (T3* q : qs) updateSet.insert(q);
        }
        if (updateSet.empty()) return;
        std::vector<T3*> todo(updateSet.begin(), updateSet.end());
        std::vector<long> ks(todo.size(), 0);
        parallel_for((int)todo.size(), nthreads, [&](int a, int b) {
            std::vector<NodeP> res;
            std::vector<float> pos, grad, val, sx, sg;
            for (int i = a; i < b; ++i) {
                T3* c = todo[i];
                res.clear();
                t->queryRange(Box<3>(c->box.c, c->box.h * Rtimes), res);
                if (res.empty()) continue;
                size_t n = res.size();
                pos.resize(3 * n); grad.resize(3 * n); val.resize(n); sx.resize(n); sg.resize(n);
                for (size_t k = 0; k < n; ++k) {
                    for (int d = 0; d < 3; ++d) { pos[3 * k + d] = res[k]->pos[d]; grad[3 * k + d] = res[k]->grad[d]; }
                    val[k] = res[k]->val; sx[k] = res[k]->sigx; sg[k] = res[k]->sigg;
                }
                auto gp = std::make_shared<OnGPIS>(3, setting.map_scale_param);
                gp->train(pos.data(), grad.data(), val.data(), sx.data(), sg.data(), (int)n);
                c->gp = gp;
                ks[i] = gp->K;
            }
        });
        for (long k : ks) if (k) { ++stats.clusters_trained; stats.sumK += k; stats.maxK = std::max(stats.maxK, k); }
        activeSet.clear();
    }

    // One query, GPisMap3.cpp:794-902.  Returns the number of GP evaluations.
    int test_one(const float* xt, float* res, int* flag = nullptr) {
        const float var_thre = 0.5f;
        int ev = 0;
        std::vector<T3*> quads;
        std::vector<float> sqdst;
        t->queryClusters(Box<3>(xt, (float)((double)C_leng * 3.0)), quads, &sqdst);
        res[4] = (float)(1.0 + (double)setting.map_noise_param);
        if (quads.size() == 1) {
            auto& gp = quads[0]->gp;
            if (gp) { gp->test1(xt, res, res + 4); ++ev; }
        } else if (sqdst.size() > 1) {
            std::vector<int> idx(sqdst.size());
            for (size_t i = 0; i < idx.size(); ++i) idx[i] = (int)i;
            std::sort(idx.begin(), idx.end(), [&](int a, int b) { return sqdst[a] < sqdst[b]; });
            if (flag) {  // order-ambiguous: an exact distance tie among the (up to) four nearest cells
                size_t lim = std::min<size_t>(idx.size(), 4);
                for (size_t i = 1; i < lim; ++i) if (sqdst[idx[i]] == sqdst[idx[i - 1]]) *flag |= 1;
                // bit3: std::sort (unstable beyond 16 elements) and a stable sort disagree on the three
                // nearest cells -- the only queries whose result depends on the sort implementation
                std::vector<int> ids(sqdst.size());
                for (size_t i = 0; i < ids.size(); ++i) ids[i] = (int)i;
                std::stable_sort(ids.begin(), ids.end(), [&](int a, int b) { return sqdst[a] < sqdst[b]; });
                size_t l3 = std::min<size_t>(idx.size(), 3);
                for (size_t i = 0; i < l3; ++i) if (ids[i] != idx[i]) *flag |= 8;
                if (idx.size() > 16) *flag |= 16;
            }
            auto gp = quads[idx[0]]->gp;
            if (gp) { gp->test1(xt, res, res + 4); ++ev; }
            if (flag && std::fabs(res[4] - var_thre) < 1e-3f) *flag |= 2;  // branch-ambiguous
            if (res[4] > var_thre) {
                float f2[8], grad2[8 * 3], var2[8 * 4];
                var2[0] = res[4];
                int numc = (int)sqdst.size();
                if (numc > 3) numc = 3;
                for (int m = 0; m < numc - 1; ++m) {
                    int m1 = m + 1;
                    float mv[4] = {0, 0, 0, 0};
                    auto g2 = quads[idx[m1]]->gp;
                    g2->test1(xt, mv, &var2[m1 * 4]); ++ev;
                    f2[m1] = mv[0]; grad2[m1 * 3] = mv[1]; grad2[m1 * 3 + 1] = mv[2]; grad2[m1 * 3 + 2] = mv[3];
                }
                f2[0] = res[0]; grad2[0] = res[1]; grad2[1] = res[2]; grad2[2] = res[3];
                var2[1] = res[5]; var2[2] = res[6]; var2[3] = res[7];
                std::vector<int> id2(numc);
                for (int i = 0; i < numc; ++i) id2[i] = i;
                std::sort(id2.begin(), id2.end(), [&](int a, int b) { return var2[a * 4] < var2[b * 4]; });
                int b0 = id2[0];
                if (flag) {
                    if (std::fabs(var2[b0 * 4] - var_thre) < 1e-3f) *flag |= 2;
                    for (int i = 1; i < numc; ++i) if (std::fabs(var2[id2[i] * 4] - var2[id2[i - 1] * 4]) < 1e-4f) *flag |= 4;
                }
                if (var2[b0 * 4] < var_thre) {
                    res[0] = f2[b0];
                    for (int d = 0; d < 3; ++d) res[1 + d] = grad2[b0 * 3 + d];
                    for (int d = 0; d < 4; ++d) res[4 + d] = var2[b0 * 4 + d];
                } else {
                    int b1 = id2[1];
                    float w1 = var2[b0 * 4] - var_thre, w2 = var2[b1 * 4] - var_thre, w12 = w1 + w2;
                    res[0] = (w2 * f2[b0] + w1 * f2[b1]) / w12;
                    for (int d = 0; d < 3; ++d) res[1 + d] = (w2 * grad2[b0 * 3 + d] + w1 * grad2[b1 * 3 + d]) / w12;
                    for (int d = 0; d < 4; ++d) res[4 + d] = (w2 * var2[b0 * 4 + d] + w1 * var2[b1 * 4 + d]) / w12;
                }
            }
        }
        return ev;
    }
};

}  // namespace orc
