// ORACLE (test infrastructure only; parity unpinned -- see linalg.hpp).
// Flat C-ABI over the CPU restatement so tests/ and bench.py's cpu_baseline leg
// can drive it through ctypes.  Nothing in gpismap_amd/ may link or load this.
#include <cstring>
#include "map2.hpp"

using namespace orc;

extern "C" {

// ---- map level (3-D) -------------------------------------------------------
void* orc3_create(const double* cam6) {
    if (cam6) {
        CamParam c;
        c.fx = (float)cam6[0]; c.fy = (float)cam6[1]; c.cx = (float)cam6[2]; c.cy = (float)cam6[3];
        c.width = (int)cam6[4]; c.height = (int)cam6[5];
        return new GPisMap3(c);
    }
    return new GPisMap3();
}
void orc3_destroy(void* h) { delete (GPisMap3*)h; }
void orc3_reset(void* h) { ((GPisMap3*)h)->reset(); }
void orc3_set_threads(void* h, int n) { ((GPisMap3*)h)->nthreads = n; }
void orc3_set_camera(void* h, const double* cam6) {
    CamParam c;
    c.fx = (float)cam6[0]; c.fy = (float)cam6[1]; c.cx = (float)cam6[2]; c.cy = (float)cam6[3];
    c.width = (int)cam6[4]; c.height = (int)cam6[5];
    ((GPisMap3*)h)->resetCam(c);
}
void orc3_update(void* h, const float* depth, int n, const float* pose12) {
    ((GPisMap3*)h)->update(depth, n, pose12, 12);
}
int orc3_test(void* h, const float* x, int dim, int n, float* res) {
    return ((GPisMap3*)h)->test(x, dim, n, res) ? 1 : 0;
}
void orc3_test_flags(void* h, const float* x, int n, int* flags) { ((GPisMap3*)h)->testFlags(x, n, flags); }
// observation grid of the last frame (for component-level K1/K2 parity)
void orc3_obs_dims(void* h, int* ni, int* nj) {
    auto* m = (GPisMap3*)h;
    *ni = m->cam.height / m->setting.obs_skip; *nj = m->cam.width / m->setting.obs_skip;
}
void orc3_get_obs(void* h, float* vu, float* zinv) {
    auto* m = (GPisMap3*)h;
    std::memcpy(vu, m->vu_grid.data(), m->vu_grid.size() * sizeof(float));
    std::memcpy(zinv, m->obs_zinv.data(), m->obs_zinv.size() * sizeof(float));
}
int orc3_obsgp_num_tiles(void* h) { auto* m = (GPisMap3*)h; return m->gpo ? (int)m->gpo->gps.size() : 0; }
// returns n (0 = untrained); L is n x n column-major
int orc3_obsgp_tile(void* h, int tile, float* x, float* alpha, float* L) {
    auto* m = (GPisMap3*)h;
    if (!m->gpo || tile < 0 || tile >= (int)m->gpo->gps.size() || !m->gpo->gps[tile]) return 0;
    const GPou& g = *m->gpo->gps[tile];
    if (x) std::memcpy(x, g.x.data(), g.x.size() * sizeof(float));
    if (alpha) std::memcpy(alpha, g.alpha.data(), g.alpha.size() * sizeof(float));
    if (L) std::memcpy(L, g.L.data(), g.L.size() * sizeof(float));
    return g.n;
}
int orc3_num_points(void* h) {
    std::vector<float> p; ((GPisMap3*)h)->getAllPoints(p); return (int)(p.size() / 3);
}
int orc3_get_points(void* h, float* out, int cap) {
    std::vector<float> p; ((GPisMap3*)h)->getAllPoints(p);
    int n = (int)(p.size() / 3);
    if (out && n <= cap) std::memcpy(out, p.data(), p.size() * sizeof(float));
    return n;
}
int orc3_get_nodes(void* h, float* out9, int cap) {
    std::vector<float> p; ((GPisMap3*)h)->getAllNodes(p);
    int n = (int)(p.size() / 9);
    if (out9 && n <= cap) std::memcpy(out9, p.data(), p.size() * sizeof(float));
    return n;
}
// same-map comparison of arithmetic modes: re-factorise every trained cluster on its stored training set in the
// CURRENT mode (orc_set_arith_mode); the map (points, tree, cluster sets) stays what the original mode built
int orc3_retrain_all(void* h) { return ((GPisMap3*)h)->retrainAll(); }
int orc2_retrain_all(void* h) { return ((GPisMap2*)h)->retrainAll(); }
int orc3_cluster_samples(void* h, int i, float* out9, int cap) { return ((GPisMap3*)h)->clusterSamples(i, out9, cap); }
int orc3_num_clusters(void* h) { return ((GPisMap3*)h)->numClusters(); }
void orc3_stats(void* h, long* out6) {
    auto& s = ((GPisMap3*)h)->stats;
    out6[0] = s.obsgp_tiles; out6[1] = ((GPisMap3*)h)->gpo ? ((GPisMap3*)h)->gpo->n_queries : 0;
    out6[2] = s.clusters_trained; out6[3] = s.sumK; out6[4] = s.maxK; out6[5] = s.gp_evals;
}
// ObsGP of the last frame: batched single-point queries (v,u) -> (val,var).
void orc3_obsgp_query(void* h, const float* vu, int n, float* val, float* var) {
    auto* m = (GPisMap3*)h;
    for (int i = 0; i < n; ++i) m->gpo->test1(vu[2 * i], vu[2 * i + 1], val[i], var[i]);
}

// ---- map level (2-D) -------------------------------------------------------
void* orc2_create() { return new GPisMap2(); }
void orc2_destroy(void* h) { delete (GPisMap2*)h; }
void orc2_reset(void* h) { ((GPisMap2*)h)->reset(); }
void orc2_set_threads(void* h, int n) { ((GPisMap2*)h)->nthreads = n; }
void orc2_update(void* h, const float* theta, const float* range, int n, const float* pose6) {
    ((GPisMap2*)h)->update(theta, range, n, pose6, 6);
}
int orc2_test(void* h, const float* x, int dim, int n, float* res) { return ((GPisMap2*)h)->test(x, dim, n, res) ? 1 : 0; }
void orc2_test_flags(void* h, const float* x, int n, int* flags) { ((GPisMap2*)h)->testFlags(x, n, flags); }
int orc2_get_nodes(void* h, float* out7, int cap) {
    std::vector<float> p; ((GPisMap2*)h)->getAllNodes(p);
    int n = (int)(p.size() / 7);
    if (out7 && n <= cap) std::memcpy(out7, p.data(), p.size() * sizeof(float));
    return n;
}
void orc2_stats(void* h, long* out6) {
    auto* m = (GPisMap2*)h;
    auto& s = m->stats;
    out6[0] = s.obsgp_tiles; out6[1] = m->gpo ? m->gpo->n_queries : 0;
    out6[2] = s.clusters_trained; out6[3] = s.sumK; out6[4] = s.maxK; out6[5] = s.gp_evals;
}
int orc2_obsgp_sizes(void* h, int* out, int cap) {
    auto* m = (GPisMap2*)h;
    if (!m->gpo) return 0;
    int n = (int)m->gpo->gps.size();
    for (int i = 0; i < n && i < cap; ++i) out[i] = m->gpo->gps[i]->n;
    return n;
}

// ---- component level -------------------------------------------------------
// arithmetic variant of every subsequent training / prediction (linalg.hpp): 0 tiled (default), 1 natural, 2 fp64acc, 3 eigen33
void orc_set_arith_mode(int m) { arith_mode() = (m >= 1 && m <= 3) ? m : 0; }
int orc_get_arith_mode() { return arith_mode(); }
void orc_chol_lower(float* A, int n, int ld) { chol_lower(A, n, ld); }
void orc_fwd_subst(const float* L, int n, int ld, float* B, int nrhs, int ldb) { fwd_subst(L, n, ld, B, nrhs, ldb); }
// the matrix-rhs variant used by the prediction (blocked, inverted 32x32 diagonal blocks; linalg.hpp)
void orc_fwd_subst_blocked(const float* L, int n, int ld, float* B, int nrhs, int ldb) {
    std::vector<float> inv;
    blocked_diag_inverses(L, n, ld, inv);
    fwd_subst_blocked(L, inv.data(), n, ld, B, nrhs, ldb);
}
void orc_bwd_subst(const float* L, int n, int ld, float* b) { bwd_subst(L, n, ld, b); }

// OU GP on one group (dim x n, n <= 64): outputs L (n x n col-major, lower) and alpha.
void orc_gpou_train(const float* x, const float* f, int dim, int n, float* L, float* alpha) {
    GPou g; g.train(x, f, dim, n);
    std::memcpy(L, g.L.data(), sizeof(float) * n * n);
    std::memcpy(alpha, g.alpha.data(), sizeof(float) * n);
}
void orc_gpou_test(const float* x, const float* f, int dim, int n, const float* xq, int nq, float* val, float* var) {
    GPou g; g.train(x, f, dim, n);
    for (int i = 0; i < nq; ++i) g.test1(xq + (size_t)dim * i, val[i], var[i]);
}

// OnGPIS: train on n samples (pos dim*n, grad dim*n, val, sigx, sigg); returns K.
// If Kout/Lout/alpha/gidx are non-null they receive K x K (col-major lower K matrix),
// the factor, alpha (K) and the gradient index (n).
int orc_ongpis_train(int dim, float scale, const float* pos, const float* grad, const float* val,
                     const float* sx, const float* sg, int n, float* Lout, float* alpha, int* gidx) {
    OnGPIS gp(dim, scale);
    gp.train(pos, grad, val, sx, sg, n);
    if (Lout) std::memcpy(Lout, gp.L.data(), sizeof(float) * (size_t)gp.K * gp.K);
    if (alpha) std::memcpy(alpha, gp.alpha.data(), sizeof(float) * gp.K);
    if (gidx) std::memcpy(gidx, gp.gidx.data(), sizeof(int) * n);
    return gp.K;
}
// Kernel matrix only (lower triangle, col-major ld=K); gidx in, as produced above.
void orc_matern32_train(int dim, int n, const float* x, const int* gidx, int ng, float scale,
                        const float* sigx, const float* sigg, float* K) {
    matern32_train_lower(dim, n, x, gidx, ng, scale, sigx, sigg, K, n + dim * ng);
}
void orc_matern32_cross(int dim, int n, const float* x, const int* gidx, int ng, float scale,
                        const float* xq, float* out) {
    matern32_cross1(dim, n, x, gidx, ng, scale, xq, out, n + dim * ng);
}
// train + predict nq queries: out is nq x 2(1+dim) (mean(1+dim), var(1+dim)).
int orc_ongpis_predict(int dim, float scale, const float* pos, const float* grad, const float* val,
                       const float* sx, const float* sg, int n, const float* xq, int nq, float* out) {
    OnGPIS gp(dim, scale);
    gp.train(pos, grad, val, sx, sg, n);
    int nc = 1 + dim;
    for (int i = 0; i < nq; ++i) gp.test1(xq + (size_t)dim * i, out + (size_t)2 * nc * i, out + (size_t)2 * nc * i + nc);
    return gp.K;
}

}  // extern "C"
