// ORACLE (test infrastructure only; parity unpinned -- see linalg.hpp).
// CPU restatement of the 2-D map orchestrator, reference cpp/src/GPisMap.cpp
// (+ cpp/include/GPisMap.h, params.h:57-74).  Sequential host logic as in the
// reference; ObsGP queries inline (no thread per query).
#pragma once
#include "map3.hpp"

namespace orc {

struct Map2Param {  // GPisMap.h:29-67, params.h:57-74
    float delx = (float)1e-2;
    float fbias = (float)0.2;
    float sensor_offset[2] = {(float)0.08, (float)0.0};
    float angle_obs_limit[2] = {(float)(-135.0 * M_PI / 180.0), (float)(135.0 * M_PI / 180.0)};
    float obs_var_thre = (float)0.1;
    float min_position_noise = (float)1e-2;
    float min_grad_noise = (float)1e-2;
    float map_scale_param = (float)1.2;
    float map_noise_param = (float)1e-2;
};

// unqualified cos/sin/atan2/sqrt on float arguments resolve to the double versions (GPisMap.cpp:44-55)
static inline void polar2Cart(float a, float r, float& x, float& y) {
    x = (float)((double)r * std::cos((double)a));
    y = (float)((double)r * std::sin((double)a));
}
static inline void cart2polar(float x, float y, float& a, float& r) {
    a = (float)std::atan2((double)y, (double)x);
    r = (float)std::sqrt((double)(x * x + y * y));
}

class GPisMap2 {
public:
    using T2 = Tree<2>;
    using NodeP = T2::NodeP;
    Map2Param setting;
    int nthreads = (int)std::thread::hardware_concurrency();
    Map3Stats stats;

    GPisMap2() { init(); }
    ~GPisMap2() { reset(); }

    void reset() {  // GPisMap.cpp:90-103
        delete t; t = nullptr;
        gpo.reset();
        obs_numdata = 0;
        activeSet.clear();
    }

    void update(const float* datax, const float* dataf, int N, const float* pose, int npose) {  // :151-167
        if (!preproData(datax, dataf, N, pose, npose)) return;
        if (regressObs()) {
            updateMapPoints();
            addNewMeas();
            updateGPs();
        }
    }

    bool test(const float* x, int dim, int leng, float* res) {  // :765-810
        if (!x || dim != 2 || leng < 1) return false;
        if (!t) return false;
        std::atomic<long> ev{0};
        parallel_for(leng, nthreads, [&](int a, int b) {
            long e = 0;
            for (int i = a; i < b; ++i) e += test_one(x + 2 * (size_t)i, res + 6 * (size_t)i, nullptr);
            ev += e;
        });
        stats.gp_evals += ev.load();
        return true;
    }
    void testFlags(const float* x, int leng, int* flags) {
        if (!t) return;
        parallel_for(leng, nthreads, [&](int a, int b) {
            float r[6];
            for (int i = a; i < b; ++i) { for (float& v : r) v = 0.f; flags[i] = 0; test_one(x + 2 * (size_t)i, r, &flags[i]); }
        });
    }

    void getAllNodes(std::vector<float>& out) {  // pos2 grad2 val sigx sigg, tree order
        out.clear();
        if (!t) return;
        std::vector<NodeP> nodes;
        t->allNodes(nodes);
        for (auto& n : nodes) {
            out.push_back(n->pos[0]); out.push_back(n->pos[1]);
            out.push_back(n->grad[0]); out.push_back(n->grad[1]);
            out.push_back(n->val); out.push_back(n->sigx); out.push_back(n->sigg);
        }
    }

    // every trained cluster re-factorised in the current arithmetic mode on its stored training set (gp.hpp retrain)
    int retrainAll() {
        if (!t) return 0;
        float c0[2] = {0, 0};
        std::vector<T2*> q;
        t->queryClusters(Box<2>(c0, 1e9f), q, nullptr);
        parallel_for((int)q.size(), nthreads, [&](int a, int b) {
            for (int i = a; i < b; ++i) if (q[i]->gp) q[i]->gp->retrain();
        });
        int n = 0;
        for (T2* c : q) if (c->gp && c->gp->trained) ++n;
        return n;
    }

    std::unique_ptr<ObsGP1D> gpo;
    T2* t = nullptr;

private:
    TreeParam tprm;
    T2::Set activeSet;
    std::vector<float> obs_theta, obs_range, obs_f, obs_xylocal, obs_xyglobal;
    float pose_tr[2] = {0, 0}, pose_R[4] = {0, 0, 0, 0};
    int obs_numdata = 0;
    float range_obs_max = 0.f;

    void init() {
        tprm.min_half = (float)0.2;   // params.h:34-37, GPisMap.cpp:26-29
        tprm.max_half = (float)102.4;
        tprm.init_half = (float)12.8;
        tprm.cluster_half = (float)0.8;
        tprm.min_half_sq = tprm.min_half * tprm.min_half;
        tprm.cluster_eps = 1e-3;        // quadtree.cpp:238
        tprm.qleaf_eps_plain = 0.0001;  // quadtree.cpp:623
        tprm.qleaf_eps_dist = 0.001;    // quadtree.cpp:652
        tprm.qdesc_eps = 0.001;         // quadtree.cpp:628,657
    }

    bool preproData(const float* datax, const float* dataf, int N, const float* pose, int npose) {  // :105-149
        if (!datax || !dataf || N < 1) return false;
        obs_theta.clear(); obs_range.clear(); obs_f.clear(); obs_xylocal.clear(); obs_xyglobal.clear();
        range_obs_max = 0.0f;
        if (npose != 6) return false;
        pose_tr[0] = pose[0]; pose_tr[1] = pose[1];
        for (int i = 0; i < 4; ++i) pose_R[i] = pose[2 + i];
        obs_numdata = 0;
        for (int k = 0; k < N; ++k) {
            float xloc = 0.f, yloc = 0.f;
            if ((double)dataf[k] < 3e1 && (double)dataf[k] > 2e-1) {  // isRangeValid :34-37
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

    bool regressObs() {  // :169-179
        if (!gpo) gpo = std::make_unique<ObsGP1D>();
        gpo->train(obs_theta.data(), obs_f.data(), obs_numdata);
        if (gpo->trained) stats.obsgp_tiles += (long)gpo->gps.size();
        return gpo->trained;
    }

    void obs_query(float ang, float& rinv0, float& var) { gpo->test1(ang, rinv0, var); }

    void updateMapPoints() {  // :181-233
        if (!t || !gpo) return;
        std::vector<T2*> quads;
        t->queryClusters(Box<2>(pose_tr, range_obs_max), quads, nullptr);
        float r2 = range_obs_max * range_obs_max;
        for (T2* c : quads) {
            const float* ct = c->box.c;
            float l = c->box.h;
            float sqr_range = (ct[0] - pose_tr[0]) * (ct[0] - pose_tr[0]) + (ct[1] - pose_tr[1]) * (ct[1] - pose_tr[1]);
            if (sqr_range > (r2 + 2 * l * l)) continue;
            int within_angle = 0;  // accumulated in 2-D (:217)
            for (int i = 0; i < 4; ++i) {  // NW, NE, SW, SE
                float e[2] = {(i & 1) ? c->box.hi[0] : c->box.lo[0], (i & 2) ? c->box.lo[1] : c->box.hi[1]};
                float x_loc = pose_R[0] * (e[0] - pose_tr[0]) + pose_R[1] * (e[1] - pose_tr[1]);
                float y_loc = pose_R[2] * (e[0] - pose_tr[0]) + pose_R[3] * (e[1] - pose_tr[1]);
                x_loc -= setting.sensor_offset[0];
                y_loc -= setting.sensor_offset[1];
                float ang = 0.f, r = 0.f;
                cart2polar(x_loc, y_loc, ang, r);
                within_angle += int((ang > setting.angle_obs_limit[0]) && (ang < setting.angle_obs_limit[1]));
            }
            if (within_angle == 0) continue;
            std::vector<NodeP> nodes;
            c->allNodes(nodes);
            reEvalPoints(nodes);
        }
    }

    bool try_insert(const NodeP& p, T2::Set& ins) {
        bool ok = false;
        if (!t->isNotNew(p)) {
            ok = t->insert(p, &ins);
            if (ok && !t->isRoot()) t = t->root();
        }
        return ok && !ins.empty();
    }

    void reEvalPoints(std::vector<NodeP>& nodes) {  // :235-455
        float rinv0 = 0.f, var = 0.f, ang = 0.f, r = 0.f;
        const float delx = setting.delx;
        for (auto& nd : nodes) {
            const float* pos = nd->pos;
            float x_loc = pose_R[0] * (pos[0] - pose_tr[0]) + pose_R[1] * (pos[1] - pose_tr[1]);
            float y_loc = pose_R[2] * (pos[0] - pose_tr[0]) + pose_R[3] * (pos[1] - pose_tr[1]);
            x_loc -= setting.sensor_offset[0];
            y_loc -= setting.sensor_offset[1];
            cart2polar(x_loc, y_loc, ang, r);
            obs_query(ang, rinv0, var);
            if (var > setting.obs_var_thre) continue;
            float oc = occ_test((float)(1.0 / (double)std::sqrt(r)), rinv0, (float)((double)r * 30.0));
            if ((double)oc < -0.1) continue;

            const float* grad = nd->grad;
            float grad_loc[2];
            grad_loc[0] = pose_R[0] * grad[0] + pose_R[1] * grad[1];
            grad_loc[1] = pose_R[2] * grad[0] + pose_R[3] * grad[1];

            float abs_oc = (float)std::fabs((double)oc);
            float dx = delx;
            float x_new[2] = {x_loc, y_loc};
            float r_new = r;
            for (int i = 0; i < 10 && (double)abs_oc > 0.02; ++i) {
                if (oc < 0) { x_new[0] += grad_loc[0] * dx; x_new[1] += grad_loc[1] * dx; }
                else { x_new[0] -= grad_loc[0] * dx; x_new[1] -= grad_loc[1] * dx; }
                cart2polar(x_new[0], x_new[1], ang, r_new);
                obs_query(ang, rinv0, var);
                if (var > setting.obs_var_thre) break;
                float oc_new = occ_test((float)(1.0 / (double)std::sqrt(r_new)), rinv0, (float)((double)r_new * 30.0));
                float abs_oc_new = (float)std::fabs((double)oc_new);
                if ((double)abs_oc_new < 0.02 || (double)oc < -0.1) break;
                else if ((double)(oc * oc_new) < 0.0) dx = (float)(0.5 * (double)dx);
                else dx = (float)(1.1 * (double)dx);
                abs_oc = abs_oc_new;
                oc = oc_new;
            }

            float pert[2][4] = {{1, -1, 0, 0}, {0, 0, 1, -1}};
            float occ[4] = {-1, -1, -1, -1};
            float occ_mean = 0.f, r0_mean = 0.f, r0_sqr_sum = 0.f;
            for (int i = 0; i < 4; ++i) {
                float X = x_new[0] + delx * pert[0][i];
                float Y = x_new[1] + delx * pert[1][i];
                float r_;
                cart2polar(X, Y, ang, r_);
                obs_query(ang, rinv0, var);
                if (var > setting.obs_var_thre) break;
                occ[i] = occ_test((float)(1.0 / (double)std::sqrt(r_)), rinv0, (float)((double)r_ * 30.0));
                occ_mean = (float)((double)occ_mean + 0.25 * (double)occ[i]);
                float r0 = (float)(1.0 / (double)(rinv0 * rinv0));
                r0_sqr_sum += r0 * r0;
                r0_mean = (float)((double)r0_mean + 0.25 * (double)r0);
            }
            if (var > setting.obs_var_thre) continue;

            float gl[2] = {(occ[0] - occ[1]) / delx, (occ[2] - occ[3]) / delx};
            float norm_g = std::sqrt(gl[0] * gl[0] + gl[1] * gl[1]);
            if ((double)norm_g < 1e-3) {
                nd->sigx = (float)(2.0 * (double)nd->sigx);
                nd->sigg = (float)(2.0 * (double)nd->sigg);
                continue;
            }
            float r_var = (float)((double)r0_sqr_sum / 3.0 - (double)(r0_mean * r0_mean) * 4.0 / 3.0);
            r_var /= delx;
            float noise = 100.0f, grad_noise = 1.0f;
            if ((double)norm_g > 1e-6) {
                gl[0] = gl[0] / norm_g; gl[1] = gl[1] / norm_g;
                noise = setting.min_position_noise * saturate(r_new * r_new, 1.0f, noise);
                grad_noise = saturate(std::fabs(occ_mean) + r_var, setting.min_grad_noise, grad_noise);
            } else noise = setting.min_position_noise * noise;

            float dist = std::sqrt(x_new[0] * x_new[0] + x_new[1] * x_new[1]);
            float view_ang = std::max(-(x_new[0] * gl[0] + x_new[1] * gl[1]) / dist, (float)1e-1);
            float view_ang2 = view_ang * view_ang;
            float view_noise = (float)((double)setting.min_position_noise * ((1.0 - (double)view_ang2) / (double)view_ang2));
            noise += view_noise + abs_oc;
            grad_noise = (float)((double)grad_noise + 0.1 * (double)view_noise);

            float pos_new[2], grad_new[2];
            x_new[0] += setting.sensor_offset[0];
            x_new[1] += setting.sensor_offset[1];
            pos_new[0] = pose_R[0] * x_new[0] + pose_R[2] * x_new[1] + pose_tr[0];
            pos_new[1] = pose_R[1] * x_new[0] + pose_R[3] * x_new[1] + pose_tr[1];
            grad_new[0] = pose_R[0] * gl[0] + pose_R[2] * gl[1];
            grad_new[1] = pose_R[1] * gl[0] + pose_R[3] * gl[1];

            float noise_old = nd->sigx, grad_noise_old = nd->sigg;
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
            t->remove(nd, &activeSet);
            if ((double)noise > 1.0 && (double)grad_noise > 0.61) continue;
            NodeP p = std::make_shared<MapNode<2>>(pos_new);
            T2::Set ins;
            if (!try_insert(p, ins)) continue;
            p->val = -setting.fbias; p->sigx = noise; p->sigg = grad_noise; p->type = 1;
            p->grad[0] = grad_new[0]; p->grad[1] = grad_new[1];
            for (T2* c : ins) activeSet.insert(c);
        }
    }

    void addNewMeas() {  // :457-464
        if (!t) { float c[2] = {0, 0}; t = T2::make_root(&tprm, c); }
        evalPoints();
    }

    void evalPoints() {  // :466-572
        if (!t || obs_numdata < 1) return;
        const float delx = setting.delx;
        for (int k = 0; k < obs_numdata; ++k) {
            int k2 = 2 * k;
            float rinv0 = 0.f, var = 0.f;
            obs_query(obs_theta[k], rinv0, var);
            if (var > setting.obs_var_thre) continue;
            NodeP p = std::make_shared<MapNode<2>>(&obs_xyglobal[k2]);
            T2::Set ins;
            if (!try_insert(p, ins)) continue;
            float pert[2][4] = {{1, -1, 0, 0}, {0, 0, 1, -1}};
            float occ[4] = {-1, -1, -1, -1};
            float occ_mean = 0.f;
            for (int i = 0; i < 4; ++i) {
                float X = obs_xylocal[k2] + delx * pert[0][i];
                float Y = obs_xylocal[k2 + 1] + delx * pert[1][i];
                float a, r;
                cart2polar(X, Y, a, r);
                obs_query(a, rinv0, var);
                if (var > setting.obs_var_thre) break;
                occ[i] = occ_test((float)(1.0 / (double)std::sqrt(r)), rinv0, (float)((double)r * 30.0));
                occ_mean = (float)((double)occ_mean + 0.25 * (double)occ[i]);
            }
            if (var > setting.obs_var_thre) { t->remove(p, nullptr); continue; }
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
                float dist = std::sqrt(obs_xylocal[k2] * obs_xylocal[k2] + obs_xylocal[k2 + 1] * obs_xylocal[k2 + 1]);
                float view_ang = std::max(-(obs_xylocal[k2] * gx + obs_xylocal[k2 + 1] * gy) / dist, (float)1e-1);
                float view_ang2 = view_ang * view_ang;
                float view_noise = (float)((double)setting.min_position_noise * ((1.0 - (double)view_ang2) / (double)view_ang2));
                noise += view_noise;
            }
            p->val = -setting.fbias; p->sigx = noise; p->sigg = grad_noise; p->type = 1;
            p->grad[0] = g[0]; p->grad[1] = g[1];
            for (T2* c : ins) activeSet.insert(c);
        }
    }

    void updateGPs() {  // :574-663
        T2::Set updateSet(activeSet);
        for (T2* a : activeSet) {
            std::vector<T2*> qs;
            t->queryClusters(Box<2>(a->box.c, (float)(4.0 * (double)a->box.h)), qs, nullptr);
            for (T2* q : qs) updateSet.insert(q);
        }
        if (updateSet.empty()) { return; }  // the reference divides by zero here (SURVEY B-5)
        std::vector<T2*> todo(updateSet.begin(), updateSet.end());
        std::vector<long> ks(todo.size(), 0);
        parallel_for((int)todo.size(), nthreads, [&](int a, int b) {
            std::vector<NodeP> res;
            std::vector<float> pos, grad, val, sx, sg;
            for (int i = a; i < b; ++i) {
                T2* c = todo[i];
                res.clear();
                t->queryRange(Box<2>(c->box.c, (float)((double)c->box.h * 4.0)), res);
                if (res.empty()) continue;
                size_t n = res.size();
                pos.resize(2 * n); grad.resize(2 * n); val.resize(n); sx.resize(n); sg.resize(n);
                for (size_t k = 0; k < n; ++k) {
                    for (int d = 0; d < 2; ++d) { pos[2 * k + d] = res[k]->pos[d]; grad[2 * k + d] = res[k]->grad[d]; }
                    val[k] = res[k]->val; sx[k] = res[k]->sigx; sg[k] = res[k]->sigg;
                }
                auto gp = std::make_shared<OnGPIS>(2, setting.map_scale_param);
                gp->train(pos.data(), grad.data(), val.data(), sx.data(), sg.data(), (int)n);
                c->gp = gp;
                ks[i] = gp->K;
            }
        });
        for (long k : ks) if (k) { ++stats.clusters_trained; stats.sumK += k; stats.maxK = std::max(stats.maxK, k); }
        activeSet.clear();
    }

    int test_one(const float* xt, float* res, int* flag) {  // GPisMap.cpp:665-763
        const float var_thre = 0.4f;
        int ev = 0;
        std::vector<T2*> quads;
        std::vector<float> sqdst;
        t->queryClusters(Box<2>(xt, (float)((double)setting.map_scale_param * 4.0)), quads, &sqdst);
        res[3] = (float)(1.0 + (double)setting.map_noise_param);
        if (quads.size() == 1) {
            auto& gp = quads[0]->gp;
            if (gp) { gp->test1(xt, res, res + 3); ++ev; }
        } else if (sqdst.size() > 1) {
            std::vector<int> idx(sqdst.size());
            for (size_t i = 0; i < idx.size(); ++i) idx[i] = (int)i;
            std::sort(idx.begin(), idx.end(), [&](int a, int b) { return sqdst[a] < sqdst[b]; });
            auto gp = quads[idx[0]]->gp;
            if (gp) { gp->test1(xt, res, res + 3); ++ev; }
            if (flag && std::fabs(res[3] - var_thre) < 1e-3f) *flag |= 2;
            if (res[3] > var_thre) {
                float f2[4], grad2[4 * 2], var2[4 * 3];
                var2[0] = res[3];
                int numc = (int)sqdst.size();
                if (numc > 3) numc = 3;
                for (int m = 0; m < numc - 1; ++m) {
                    int m1 = m + 1;
                    float mv[3] = {0, 0, 0};
                    quads[idx[m1]]->gp->test1(xt, mv, &var2[m1 * 3]); ++ev;
                    f2[m1] = mv[0]; grad2[m1 * 2] = mv[1]; grad2[m1 * 2 + 1] = mv[2];
                }
                f2[0] = res[0]; grad2[0] = res[1]; grad2[1] = res[2];
                var2[1] = res[4]; var2[2] = res[5];
                std::vector<int> id2(numc);
                for (int i = 0; i < numc; ++i) id2[i] = i;
                std::sort(id2.begin(), id2.end(), [&](int a, int b) { return var2[a * 3] < var2[b * 3]; });
                int b0 = id2[0];
                if (flag) {
                    if (std::fabs(var2[b0 * 3] - var_thre) < 1e-3f) *flag |= 2;
                    for (int i = 1; i < numc; ++i) if (std::fabs(var2[id2[i] * 3] - var2[id2[i - 1] * 3]) < 1e-4f) *flag |= 4;
                }
                if (var2[b0 * 3] < var_thre) {
                    res[0] = f2[b0]; res[1] = grad2[b0 * 2]; res[2] = grad2[b0 * 2 + 1];
                    for (int d = 0; d < 3; ++d) res[3 + d] = var2[b0 * 3 + d];
                } else {
                    int b1 = id2[1];
                    float w1 = var2[b0 * 3] - var_thre, w2 = var2[b1 * 3] - var_thre, w12 = w1 + w2;
                    res[0] = (w2 * f2[b0] + w1 * f2[b1]) / w12;
                    res[1] = (w2 * grad2[b0 * 2] + w1 * grad2[b1 * 2]) / w12;
                    res[2] = (w2 * grad2[b0 * 2 + 1] + w1 * grad2[b1 * 2 + 1]) / w12;
                    for (int d = 0; d < 3; ++d) res[3 + d] = (w2 * var2[b0 * 3 + d] + w1 * var2[b1 * 3 + d]) / w12;
                }
            }
        }
        return ev;
    }
};

}  // namespace orc
