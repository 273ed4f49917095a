// ORACLE (test infrastructure only; parity unpinned -- see linalg.hpp).
// CPU restatement of the GP regressors on GPisMap's hot path:
//   covariance functions   reference cpp/src/covFnc.cpp
//   GPou / ObsGP1D/2D      reference cpp/src/ObsGP.cpp
//   OnGPIS                 reference cpp/src/OnGPIS.cpp
// Written from the semantics of those files (C++ usual-arithmetic-conversion
// rules included: the reference mixes double literals into float expressions),
// not copied from them.  Compile with -ffp-contract=off so that only the
// explicit fmaf chains below are fused.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <memory>
#include <vector>
#include "linalg.hpp"

namespace orc {

// ---------------------------------------------------------------------------
// Scalar kernels.  covFnc.cpp:29-33.  `exp` there resolves to ::exp(double)
// (only <cmath> is included, no using-declaration), so the exponential is taken
// in double and the product rounded once to float.
// ---------------------------------------------------------------------------
static inline float m32_kf(float r, float a) {
    return (float)((1.0 + (double)(a * r)) * std::exp((double)(-a * r)));
}
static inline float m32_kf1(float r, float dx, float a) {
    return (float)((double)(a * a * dx) * std::exp((double)(-a * r)));
}
static inline float m32_kf2(float r, float dx1, float dx2, float delta, float a) {
    return (float)((double)(a * a * (delta - a * dx1 * dx2 / r)) * std::exp((double)(-a * r)));
}
static inline float ou_k(float r, float a) { return (float)std::exp((double)(-a * r)); }

// Eigen's (a-b).norm() for 1..3 element float vectors: sequential sum of
// squares, then sqrt, all in float.
static inline float dist_n(const float* a, const float* b, int dim) {
    float s = 0.f;
    for (int d = 0; d < dim; ++d) {
        float t = a[d] - b[d];
        s = (d == 0) ? t * t : s + t * t;
    }
    return std::sqrt(s);
}

// ---------------------------------------------------------------------------
// Ornstein-Uhlenbeck GP on <= 64 points.  ObsGP.cpp:32-62, covFnc.cpp:47-68,93-109.
// x is dim x n column-major (point j at x[dim*j..]).
// ---------------------------------------------------------------------------
struct GPou {
    int dim = 0, n = 0;
    std::vector<float> x, L, alpha;
    bool trained = false;
    static constexpr float scale = 0.5f;   // params.h:99
    static constexpr float noise = 0.01f;  // params.h:100

    void train(const float* xt, const float* f, int dim_, int n_) {
        if (n_ <= 0) return;
        dim = dim_; n = n_;
        x.assign(xt, xt + (size_t)dim * n);
        float a = 1 / scale;
        L.assign((size_t)n * n, 0.f);
        for (int k = 0; k < n; ++k)
            for (int j = k; j < n; ++j) {
                float v;
                if (k == j) v = (float)(1.0 + (double)noise);  // covFnc.cpp:57
                else v = ou_k(dist_n(&x[(size_t)dim * k], &x[(size_t)dim * j], dim), a);
                L[j + (size_t)k * n] = v;  // lower triangle only
            }
        chol_lower_m(L.data(), n, n);
        alpha.assign(f, f + n);
        if (arith_mode() == ARITH_EIGEN33) eig::fwd_vec(L.data(), n, n, alpha.data());   // vector right-hand side (ObsGP.cpp:43)
        else fwd_subst_m(L.data(), n, n, alpha.data(), 1, n);
        bwd_subst_m(L.data(), n, n, alpha.data());
        trained = true;
    }

    // Single query.  Mean = k*^T alpha summed with a 64-slot xor butterfly
    // (order O4: zero-padded products, offsets 32,16,..,1); variance chain (O5)
    // acc = fmaf(v_j, v_j, acc) over ascending j.  ObsGP.cpp:50-62.
    void test1(const float* xt, float& f, float& var) const {
        float ks[64];
        float p[64];
        float a = 1 / scale;
        for (int i = 0; i < 64; ++i) { ks[i] = 0.f; p[i] = 0.f; }
        for (int i = 0; i < n; ++i) {
            ks[i] = ou_k(dist_n(&x[(size_t)dim * i], xt, dim), a);
            p[i] = ks[i] * alpha[i];
        }
        if (arith_mode() == ARITH_EIGEN33) {   // ObsGP.cpp:54-59 in Eigen 3.3's orders (linalg.hpp)
            f = eig::dot(ks, alpha.data(), n);
            eig::fwd_mat(L.data(), n, n, ks, 1, 64);
            var = (1 + noise) - eig::sum_sq_seq(ks, n);
            return;
        }
        if (arith_mode() != ARITH_TILED) {   // order variants (linalg.hpp): sequential sums, plain substitution
            const bool d64 = arith_mode() == ARITH_FP64ACC;
            f = d64 ? dot_nat<double>(ks, alpha.data(), n) : dot_nat<float>(ks, alpha.data(), n);
            fwd_subst_m(L.data(), n, n, ks, 1, 64);
            const float ss = d64 ? dot_nat<double>(ks, ks, n) : dot_nat<float>(ks, ks, n);
            var = d64 ? (float)((double)(1 + noise) - (double)ss) : (1 + noise) - ss;
            return;
        }
        for (int off = 32; off >= 1; off >>= 1) {
            float q[64];
            for (int i = 0; i < 64; ++i) q[i] = p[i] + p[i ^ off];
            for (int i = 0; i < 64; ++i) p[i] = q[i];
        }
        f = p[0];
        float acc = 0.f;
        for (int k = 0; k < n; ++k) {
            const float* col = &L[(size_t)k * n];
            float vk = ks[k] / col[k];
            float nvk = -vk;
            for (int j = k + 1; j < n; ++j) ks[j] = fmaf(col[j], nvk, ks[j]);
            acc = fmaf(vk, vk, acc);
        }
        var = (1 + noise) - acc;  // ObsGP.cpp:61 (int + float -> float)
    }
};

// ---------------------------------------------------------------------------
// ObsGP2D: overlapping 8x8-pixel tiles.  ObsGP.cpp:197-463, params.h:107-110.
// Members start zeroed (the reference relies on `new ObsGP2D()` value-
// initialisation) and the base-class reset never sets `repartition`, so the
// boundary tables are computed once per grid size (SURVEY B-7/B-15).
// ---------------------------------------------------------------------------
struct ObsGP2D {
    static constexpr float margin = 0.005f;
    static constexpr int overlap = 3, group = 5;
    int nGroup[2] = {0, 0};
    int szSamples[2] = {0, 0};
    bool repartition = false;
    bool trained = false;
    std::vector<int> i0, i1, j0, j1;
    std::vector<float> Val_i, Val_j;
    std::vector<std::unique_ptr<GPou>> gps;
    long n_queries = 0;

    void reset() { trained = false; gps.clear(); }  // ObsGP::reset, ObsGP.cpp:68-72

    void computePartition(const float* val, int ni, int nj) {  // ObsGP.cpp:204-265
        szSamples[0] = ni; szSamples[1] = nj;
        nGroup[0] = (ni - overlap) / group + 1;
        nGroup[1] = (nj - overlap) / group + 1;
        i0.clear(); i1.clear(); j0.clear(); j1.clear(); Val_i.clear(); Val_j.clear();
        Val_i.push_back(val[0]);
        for (int n = 0; n < nGroup[0]; ++n) {
            int a = n * group, b = a + group + overlap - 1;
            if (n < nGroup[0] - 1) Val_i.push_back(val[2 * (b - overlap / 2)]);
            else { b = ni - 1; Val_i.push_back(val[2 * b]); }
            i0.push_back(a); i1.push_back(b);
        }
        Val_j.push_back(val[1]);
        for (int m = 0; m < nGroup[1]; ++m) {
            int a = m * group, b = a + group + overlap - 1;
            if (m < nGroup[1] - 1) Val_j.push_back(val[2 * (b - overlap / 2) * ni + 1]);
            else { b = nj - 1; Val_j.push_back(val[2 * b * ni + 1]); }
            j0.push_back(a); j1.push_back(b);
        }
        if (!i0.empty() && !j0.empty()) repartition = false;
    }

    void train(const float* xt, const float* f, int ni, int nj) {  // ObsGP.cpp:280-342
        if (!(ni > 0 && nj > 0 && xt)) return;
        if (szSamples[0] != ni || szSamples[1] != nj || repartition) computePartition(xt, ni, nj);
        if (repartition) return;
        reset();
        gps.resize((size_t)nGroup[0] * nGroup[1]);
        std::vector<float> xv, fv;
        for (int m = 0; m < nGroup[1]; ++m)
            for (int n = 0; n < nGroup[0]; ++n) {
                xv.clear(); fv.clear();
                for (int j = j0[m]; j <= j1[m]; ++j)
                    for (int i = i0[n]; i <= i1[n]; ++i) {
                        int ind = j * szSamples[0] + i;
                        if (f[ind] > 0) {
                            xv.push_back(xt[ind * 2]); xv.push_back(xt[ind * 2 + 1]);
                            fv.push_back(f[ind]);
                        }
                    }
                if (xv.size() > 1) {
                    auto g = std::make_unique<GPou>();
                    g->train(xv.data(), fv.data(), 2, (int)fv.size());
                    gps[(size_t)m * nGroup[0] + n] = std::move(g);
                }
            }
        trained = true;
    }

    // Tile lookup, ObsGP.cpp:359-406.  Returns tile index or -1 (var = 1e6).
    int lookup(float v, float u) const {
        if (v < Val_i.front() + margin) return -1;
        if (v > Val_i.back() - margin) return -1;
        if (u < Val_j.front() + margin) return -1;
        if (u > Val_j.back() - margin) return -1;
        int n = 0;
        for (size_t k = 1; k < Val_i.size(); ++k, ++n) if (v < Val_i[k]) break;
        int m = 0;
        for (size_t k = 1; k < Val_j.size(); ++k, ++m) if (u < Val_j[k]) break;
        int ind = m * nGroup[0] + n;
        if (ind < (int)gps.size() && gps[ind] && gps[ind]->trained) return ind;
        return -1;
    }

    // Single-point query (all reference call sites pass one point).  val is
    // left untouched when no tile answers; var is then 1e6.
    void test1(float v, float u, float& val, float& var) {
        ++n_queries;
        if (!trained) return;
        var = 1e6f;
        int ind = lookup(v, u);
        if (ind < 0) return;
        float q[2] = {v, u};
        gps[ind]->test1(q, val, var);
    }
};

// ---------------------------------------------------------------------------
// ObsGP1D: overlapping 26-beam groups.  ObsGP.cpp:75-187, params.h:103-105.
// ---------------------------------------------------------------------------
struct ObsGP1D {
    static constexpr float margin = 0.0175f;
    static constexpr int overlap = 6, group = 20;
    int nGroup = 0, nSamples = 0;
    bool trained = false;
    std::vector<float> range;
    std::vector<std::unique_ptr<GPou>> gps;
    long n_queries = 0;

    void reset() { trained = false; gps.clear(); range.clear(); nSamples = 0; }

    void train(const float* xt, const float* f, int N) {  // ObsGP.cpp:85-143
        // NB: the reference calls ObsGP::reset() through the base pointer first
        // (GPisMap.cpp:176) and then ObsGP1D::reset() inside train(): both clear.
        reset();
        if (!(N > 0 && xt)) return;
        nSamples = N;
        nGroup = nSamples / group + 1;
        range.push_back(xt[0]);
        for (int n = 0; n < nGroup - 1; ++n) {
            if (n < nGroup - 2) {
                int a = n * group, b = a + group + overlap;
                range.push_back(xt[b - overlap / 2]);
                auto g = std::make_unique<GPou>();
                g->train(xt + a, f + a, 1, group + overlap);
                gps.push_back(std::move(g));
            } else {
                int a = n * group;
                int b = a + (nSamples - a) / 2 + overlap;
                range.push_back(xt[b - overlap / 2]);
                auto g = std::make_unique<GPou>();
                g->train(xt + a, f + a, 1, b - a + 1);
                gps.push_back(std::move(g));
                ++n;
                a = a + (nSamples - a) / 2;
                b = nSamples - 1;
                range.push_back(xt[b]);
                auto gl = std::make_unique<GPou>();
                gl->train(xt + a, f + a, 1, b - a + 1);
                gps.push_back(std::move(gl));
            }
        }
        trained = true;
    }

    int lookup(float t) const {  // ObsGP.cpp:154-183
        float liml = range.front() + margin, limr = range.back() - margin;
        if (t < liml || t > limr) return -1;
        for (size_t k = 1; k < range.size(); ++k)
            if (t > range[k - 1] && t < range[k]) {
                size_t j = k - 1;
                if (j < gps.size() && gps[j]->trained) return (int)j;
                return -1;
            }
        return -1;
    }

    void test1(float t, float& val, float& var) {
        ++n_queries;
        if (!trained) return;
        var = 1e6f;
        int j = lookup(t);
        if (j < 0) return;
        gps[j]->test1(&t, val, var);
    }
};

// ---------------------------------------------------------------------------
// Matern-3/2 covariance with first-derivative observations.
// Train matrix: covFnc.cpp:142-256 (3-D), :317-402 (2-D).  Row/col order
// [f(N); d/dx (ng); d/dy (ng); d/dz (ng)], gidx[k] = running index of point k
// among the gradient-bearing points or -1.  Only the lower triangle is filled
// (column-major, ld = K); the reference fills both halves symmetrically.
// ---------------------------------------------------------------------------
static inline float sqr3L_of(float scale) { return (float)(std::sqrt(3.0) / (double)scale); }

static inline void matern32_train_lower(int dim, int N, const float* x, const int* gidx, int ng,
                                        float scale, const float* sigx, const float* sigg,
                                        float* K, int ld) {
    const float a = sqr3L_of(scale);
    const float a2 = a * a;
    const int Kn = N + dim * ng;
    for (int c = 0; c < Kn; ++c)
        for (int r = c; r < Kn; ++r) K[r + (size_t)c * ld] = 0.f;
    auto put = [&](int r, int c, float v) {  // symmetric store into the lower half
        if (r >= c) K[r + (size_t)c * ld] = v; else K[c + (size_t)r * ld] = v;
    };
    for (int k = 0; k < N; ++k) {
        const float* xk = x + (size_t)dim * k;
        int kg = gidx[k];
        int kind[3] = {N + kg, N + kg + ng, N + kg + 2 * ng};
        // diagonal blocks, covFnc.cpp:171-190 / :345-357
        put(k, k, (float)(1.0 + (double)sigx[k]));
        if (kg >= 0) {
            for (int c = 0; c < dim; ++c) {
                put(kind[c], k, 0.f);
                for (int c2 = 0; c2 < dim; ++c2) if (c2 != c) put(kind[c], kind[c2], 0.f);
            }
            if (dim == 3) {
                for (int c = 0; c < 3; ++c) put(kind[c], kind[c], a2 + sigg[k]);
            } else {
                // covFnc.cpp:352 quirk; unqualified sqrt(float) there is ::sqrt(double)
                put(kind[0], kind[0], (float)((double)a2 + std::sqrt((double)(sigx[k] * sigg[k]))));
                put(kind[1], kind[1], a2 + sigg[k]);
            }
        }
        for (int j = k + 1; j < N; ++j) {
            const float* xj = x + (size_t)dim * j;
            int jg = gidx[j];
            int jind[3] = {N + jg, N + jg + ng, N + jg + 2 * ng};
            float r = dist_n(xk, xj, dim);
            float d[3] = {0, 0, 0};
            for (int c = 0; c < dim; ++c) d[c] = xk[c] - xj[c];
            put(j, k, m32_kf(r, a));
            if (kg >= 0) {
                float g1[3];
                for (int c = 0; c < dim; ++c) { g1[c] = -m32_kf1(r, d[c], a); put(kind[c], j, g1[c]); }
                if (jg >= 0) {
                    for (int c = 0; c < dim; ++c) put(k, jind[c], -g1[c]);
                    // kf2 block: upper-wedge entries computed, the rest mirrored
                    // (covFnc.cpp:217-236) so that (c2,c1) reuses (c1,c2), c1 < c2.
                    for (int c1 = 0; c1 < dim; ++c1)
                        for (int c2 = c1; c2 < dim; ++c2) {
                            float v = m32_kf2(r, d[c1], d[c2], c1 == c2 ? 1.0f : 0.0f, a);
                            put(kind[c1], jind[c2], v);
                            if (c2 != c1) put(kind[c2], jind[c1], v);
                        }
                }
            } else if (jg >= 0) {
                for (int c = 0; c < dim; ++c) put(k, jind[c], m32_kf1(r, d[c], a));
            }
        }
    }
}

// Cross covariance for ONE query: out is K x (1+dim) column-major (ld = K).
// covFnc.cpp:258-314 (3-D), :404-450 (2-D).  Delta = x_train - x_query.
static inline void matern32_cross1(int dim, int N, const float* x, const int* gidx, int ng,
                                   float scale, const float* xq, float* out, int ld) {
    const float a = sqr3L_of(scale);
    const int Kn = N + dim * ng;
    for (int i = 0; i < Kn * (1 + dim); ++i) out[(i % Kn) + (size_t)(i / Kn) * ld] = 0.f;
    for (int k = 0; k < N; ++k) {
        const float* xk = x + (size_t)dim * k;
        float r = dist_n(xk, xq, dim);
        float d[3] = {0, 0, 0};
        for (int c = 0; c < dim; ++c) d[c] = xk[c] - xq[c];
        out[k] = m32_kf(r, a);
        float g1[3];
        for (int c = 0; c < dim; ++c) { g1[c] = m32_kf1(r, d[c], a); out[k + (size_t)(1 + c) * ld] = g1[c]; }
        int kg = gidx[k];
        if (kg >= 0) {
            int kind[3] = {N + kg, N + kg + ng, N + kg + 2 * ng};
            for (int c = 0; c < dim; ++c) out[kind[c]] = -g1[c];
            for (int c1 = 0; c1 < dim; ++c1)
                for (int c2 = c1; c2 < dim; ++c2) {
                    float v = m32_kf2(r, d[c1], d[c2], c1 == c2 ? 1.0f : 0.0f, a);
                    out[kind[c1] + (size_t)(1 + c2) * ld] = v;
                    if (c2 != c1) out[kind[c2] + (size_t)(1 + c1) * ld] = v;
                }
        }
    }
}

// ---------------------------------------------------------------------------
// OnGPIS local regressor.  OnGPIS.cpp:34-149 (train), :177-263 (predict).
// Sample i is 9 floats in 3-D: pos(3) grad(3) val sigx sigg ; 7 in 2-D.
// ---------------------------------------------------------------------------
struct OnGPIS {
    int dim = 3, N = 0, ng = 0, K = 0;
    float scale = 1.f;
    float three_over_scale = 3.f;
    bool trained = false;
    std::vector<float> x;
    std::vector<int> gidx;
    std::vector<float> L, alpha;
    std::vector<float> Linv;   // inverted 32x32 diagonal blocks of L (linalg.hpp, fwd_subst_blocked)
    std::vector<float> X;      // tiled mode: explicit inverse X = L^-1, column-major K x K (lower), see train()
    int mode = ARITH_TILED;    // arithmetic variant the model was trained in: predictions use the same one
    std::vector<float> tr_pos, tr_grad, tr_val, tr_sx, tr_sg;   // the training set as given (retrain(): same-map comparisons)

    OnGPIS(int dim_, float s) : dim(dim_), scale(s), three_over_scale((float)(3.0 / (double)(s * s))) {}

    void train(const float* pos, const float* grad, const float* val, const float* sx,
               const float* sg, int n) {
        trained = false; N = 0;
        if (n <= 0) return;
        N = n;
        if (pos != tr_pos.data()) {
            tr_pos.assign(pos, pos + (size_t)dim * n); tr_grad.assign(grad, grad + (size_t)dim * n);
            tr_val.assign(val, val + n); tr_sx.assign(sx, sx + n); tr_sg.assign(sg, sg + n);
        }
        x.assign(pos, pos + (size_t)dim * N);
        gidx.assign(N, -1);
        std::vector<float> sigx(sx, sx + N), sigg(sg, sg + N);
        ng = 0;
        for (int k = 0; k < N; ++k) {
            bool tiny = true;
            for (int c = 0; c < dim; ++c) tiny = tiny && (std::fabs(grad[(size_t)dim * k + c]) < 1e-6);
            if (sigg[k] > 0.1001 || tiny) { sigx[k] = 2.0f; }  // OnGPIS.cpp:122-125
            else gidx[k] = ng++;
        }
        K = N + dim * ng;
        std::vector<float> y(K);
        for (int k = 0; k < N; ++k) {
            y[k] = val[k];
            if (gidx[k] >= 0)
                for (int c = 0; c < dim; ++c) y[N + c * ng + gidx[k]] = grad[(size_t)dim * k + c];
        }
        L.assign((size_t)K * K, 0.f);
        matern32_train_lower(dim, N, x.data(), gidx.data(), ng, scale, sigx.data(), sigg.data(), L.data(), K);
        mode = arith_mode();
        chol_lower_m(L.data(), K, K);
        alpha = y;
        if (mode == ARITH_EIGEN33) eig::fwd_vec(L.data(), K, K, alpha.data());   // vector right-hand side (OnGPIS.cpp:142)
        else fwd_subst_m(L.data(), K, K, alpha.data(), 1, K);
        bwd_subst_m(L.data(), K, K, alpha.data());
        if (arith_mode() == ARITH_TILED) {
            // Explicit inverse of the factor (tiled mode).  A prediction needs V = L^-1 k*; with X = L^-1 formed once
            // per training this is a triangular matrix product without any dependency between its rows -- the form
            // the batched GPU predictor runs on the matrix cores.  X is computed by the blocked forward substitution
            // of linalg.hpp on the columns of the identity (32 x 32 diagonal blocks through their inverses, orders
            // (O1)/(O6)); a column starts at its own block, the rows above are exact zeros.  Accuracy against an fp64
            // solve with the same factor: DESIGN.md section 2 (var_f error <= 5e-6 on trained cluster factors).
            blocked_diag_inverses(L.data(), K, K, Linv);
            X.assign((size_t)K * K, 0.f);
            std::vector<float> e(K);
            for (int j = 0; j < K; ++j) {
                const int r0 = (j / 32) * 32;
                std::fill(e.begin(), e.end(), 0.f);
                e[j] = 1.f;
                fwd_subst_blocked(L.data() + r0 + (size_t)r0 * K, Linv.data() + (size_t)(r0 / 32) * 1024, K - r0, K, e.data() + r0, 1, K);
                for (int i = j; i < K; ++i) X[(size_t)j * K + i] = e[i];
            }
        }
        trained = true;
    }

    // Train again on the stored training set in the CURRENT arithmetic mode (linalg.hpp): a map built in one mode can be
    // re-factorised in another, so that two modes are compared on identical maps and identical training sets.
    void retrain() {
        if (tr_val.empty()) return;
        train(tr_pos.data(), tr_grad.data(), tr_val.data(), tr_sx.data(), tr_sg.data(), (int)tr_val.size());
    }

    // Reduction order (O3) of the sum of squares ||V||^2 (tiled mode).  Eigen evaluates it with packet-wise
    // interleaved partial sums (order unspecified); this restatement fixes the order in which a 32x32-tiled
    // triangular product over W cooperating wavefronts meets the rows.  W = 1, 2, 4, 8 for nbx <= 4, 8, 16, more,
    // nbx = ceil((K+1)/32) (the leading dimension of the stored inverse: row K of it carries alpha, the mean).  The
    // nbv = ceil(K/32) block rows of V are dealt to the wavefronts from the LARGEST down, in groups of 4W, snake-wise
    // (block row b costs b+1 tile products, the snake balances the wavefronts): the i-th largest row b = nbv-1-i has
    // group g = i / 4W, slot t = (i % 4W) / W, position q = i % W and belongs to wavefront w = (t odd) ? W-1-q : q.
    // (K not a multiple of 32: nbv = nbx, the mean row rides in the last block row.  K a multiple of 32: the mean row
    // would be a block row of its own -- 31 zero rows multiplied for nothing; it is a chain on the vector ALU instead and
    // only the nbv rows of V are dealt.  Round 6; until round 5 the deal ran over nbx rows in that case too.)
    // Inside a 32-row block the two lane halves h = 0, 1 own the rows (r & 3) + 8 (r >> 2) + 4 h, r = 0..15.  Chain (w, h)
    // takes fmaf(v, v, .) over its rows in the order (g, t, r) ascending; then (w,0)+(w,1) are added and the W sums
    // accumulated in ascending w.
    static int chains_W(int nbx) { return nbx <= 4 ? 1 : (nbx <= 8 ? 2 : (nbx <= 16 ? 4 : 8)); }
    static float reduce_ss(int K, const float* v) {
        const int nbx = (K + 1 + 31) / 32, nbv = (K + 31) / 32, W = chains_W(nbx), RG = 4 * W;
        float P[8][2];
        for (int w = 0; w < 8; ++w) P[w][0] = P[w][1] = 0.f;
        for (int i = 0; i < nbv; ++i) {            // i ascending = (g, t) ascending for every wavefront
            const int b = nbv - 1 - i, ii = i % RG, t = ii / W, q = ii % W;
            const int w = (t & 1) ? W - 1 - q : q;
            for (int h = 0; h < 2; ++h)
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row < K) P[w][h] = fmaf(v[row], v[row], P[w][h]);
                }
        }
        float s = 0.f;
        for (int w = 0; w < W; ++w) s += (P[w][0] + P[w][1]);
        return s;
    }

    // out[0..dim] = f, grad ; out[1+dim .. 2(1+dim)-1] = variances.
    void test1(const float* xq, float* mean, float* var) const {
        if (!trained) return;
        const int nc = 1 + dim;
        std::vector<float> ks((size_t)K * nc);
        matern32_cross1(dim, N, x.data(), gidx.data(), ng, scale, xq, ks.data(), K);
        if (mode == ARITH_EIGEN33) {   // OnGPIS.cpp:187-213 in Eigen 3.3's orders (linalg.hpp)
            std::vector<float> zero(nc, 0.f);
            // K^T alpha: row-major GEMV kernel over the nc columns of k* (contiguous, leading dimension K)
            eig::gemv_row(nc, K, ks.data(), K, alpha.data(), zero.data(), 1.f);
            for (int c = 0; c < nc; ++c) mean[c] = zero[c];
            eig::fwd_mat(L.data(), K, K, ks.data(), nc, K);
            for (int c = 0; c < nc; ++c) {
                const float s = eig::sum_sq_seq(&ks[(size_t)c * K], K);
                if (dim == 3) var[c] = (c == 0) ? (float)(1.001 - (double)s) : (float)((double)three_over_scale + 0.001 - (double)s);
                else var[c] = (c == 0) ? (float)(1.01 - (double)s) : (float)((double)three_over_scale + 0.1 - (double)s);
            }
            return;
        }
        if (mode != ARITH_TILED) {   // order variants (linalg.hpp): sequential sums, plain substitution
            const bool d64 = mode == ARITH_FP64ACC;
            for (int c = 0; c < nc; ++c) {
                const float* col = &ks[(size_t)c * K];
                mean[c] = d64 ? dot_nat<double>(col, alpha.data(), K) : dot_nat<float>(col, alpha.data(), K);
            }
            if (d64) fwd_subst_nat<double>(L.data(), K, K, ks.data(), nc, K); else fwd_subst_nat<float>(L.data(), K, K, ks.data(), nc, K);
            for (int c = 0; c < nc; ++c) {
                const float* col = &ks[(size_t)c * K];
                const float s = d64 ? dot_nat<double>(col, col, K) : dot_nat<float>(col, col, K);
                if (dim == 3) var[c] = (c == 0) ? (float)(1.001 - (double)s) : (float)((double)three_over_scale + 0.001 - (double)s);
                else var[c] = (c == 0) ? (float)(1.01 - (double)s) : (float)((double)three_over_scale + 0.1 - (double)s);
            }
            return;
        }
        // Tiled mode.  Mean: ONE fmaf chain over ascending rows (alpha rides along as row K of the inverse, so the
        // matrix product delivers k*^T alpha in the natural order).  V = X k*: every element one fmaf chain from zero
        // over ascending k (terms beyond the diagonal are exact zeros).  Sum of squares: reduce_ss (O3).
        std::vector<float> v(K);
        for (int c = 0; c < nc; ++c) {
            const float* col = &ks[(size_t)c * K];
            float m = 0.f;
            for (int r = 0; r < K; ++r) m = fmaf(alpha[r], col[r], m);
            mean[c] = m;
            // per element r: a = 0; a = fmaf(X[r][k], col[k], a) for k = 0..r -- evaluated k-outer so that the inner loop
            // runs over contiguous rows (same chains, vectorisable)
            std::fill(v.begin(), v.end(), 0.f);
            for (int k = 0; k < K; ++k) {
                const float* xk = &X[(size_t)k * K];
                const float ck = col[k];
                for (int r = k; r < K; ++r) v[r] = fmaf(xk[r], ck, v[r]);
            }
            const float s = reduce_ss(K, v.data());
            if (dim == 3)  // OnGPIS.cpp:208-213
                var[c] = (c == 0) ? (float)(1.001 - (double)s)
                                  : (float)((double)three_over_scale + 0.001 - (double)s);
            else           // OnGPIS.cpp:235-237
                var[c] = (c == 0) ? (float)(1.01 - (double)s)
                                  : (float)((double)three_over_scale + 0.1 - (double)s);
        }
    }
};

}  // namespace orc
