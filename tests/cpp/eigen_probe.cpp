// TEST INFRASTRUCTURE: the reference's linear algebra executed by the REAL Eigen
// library, wherever a box has one.  Eigen is the reference's only dependency (header-only, not vendored, version not
// pinned: README.md:25, mex/make_GPisMap3.m:1 -> /usr/include/eigen3) and is absent from the image this repository is
// developed in, so the oracle's `eigen33` arithmetic mode (oracle/linalg.hpp) restates Eigen 3.3's orders from memory.
// This program is the one thing that can pin it: own code that makes exactly the Eigen calls of
//     GPou::train / test          cpp/src/ObsGP.cpp:32-62      (llt(), two triangular solveInPlace, K^T alpha,
//     OnGPIS::train / testSinglePoint  cpp/src/OnGPIS.cpp:139-143, :177-216   solveInPlace on a matrix, array().pow(2), colwise().sum())
// on the committed F2 / F3 inputs (tests/golden/fixtures_gp.npz) and writes L, alpha and the predictions, which
// tests/test_eigen_probe.py compares with the committed eigen33 fixtures.  Kernel MATRICES come from the oracle's
// restatement of covFnc.cpp (bit-checked against fixture F1); only the linear algebra is Eigen's.
// Built by the test with the reference's flags (`-O -std=c++11`; mex/make_GPisMap3.m:15 passes CXXFLAGS -std=c++11 and
// mex's default -O) when <Eigen/Dense> is found; never built or loaded by the product.
//
//   eigen_probe <in.bin> <out.bin>
//   in : int32 ncase, then per case  int32 kind (0 GPou dim 2 | 2, 3 OnGPIS of that dimension), n, nq, float32 scale,
//        kind 0: x[n][2] f[n] q[nq][2]   else: samples[n][3 dim + 3] (pos, grad, val, sigx, sigg) xq[nq][dim]
//   out: int32 EIGEN_WORLD, MAJOR, MINOR, vectorised (0/1); per case  kind 0: L[n*n] col-major, alpha[n], val[nq], var[nq]
//        else: int32 K, alpha[K], pred[nq][2 (1 + dim)]
#include <Eigen/Dense>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gp.hpp"      // oracle/: kernel functions only (ou_k, dist_n, matern32_train_lower, matern32_cross1)

typedef Eigen::MatrixXf EMatrixX;
typedef Eigen::VectorXf EVectorX;
using Eigen::Lower;
using Eigen::Upper;

static void rd(FILE* f, void* p, size_t n) { if (fread(p, 1, n, f) != n) { fprintf(stderr, "eigen_probe: short read\n"); exit(2); } }
static void wr(FILE* f, const void* p, size_t n) { if (fwrite(p, 1, n, f) != n) { fprintf(stderr, "eigen_probe: short write\n"); exit(2); } }

// ObsGP.cpp:32-62 with the kernel matrix of covFnc.cpp:47-68 / :93-109
static void case_gpou(FILE* in, FILE* out, int n, int nq) {
    const int dim = 2;
    std::vector<float> x((size_t)dim * n), f(n), q((size_t)dim * nq);
    rd(in, x.data(), 4 * x.size()); rd(in, f.data(), 4 * f.size()); rd(in, q.data(), 4 * q.size());
    const float scale = orc::GPou::scale, noise = orc::GPou::noise, a = 1 / scale;
    EMatrixX K(n, n);
    for (int k = 0; k < n; ++k)
        for (int j = k; j < n; ++j) {
            const float v = (k == j) ? (float)(1.0 + (double)noise) : orc::ou_k(orc::dist_n(&x[(size_t)dim * k], &x[(size_t)dim * j], dim), a);
            K(j, k) = v; K(k, j) = v;
        }
    EMatrixX L = K.llt().matrixL();                                              // ObsGP.cpp:41
    EVectorX alpha = Eigen::Map<EVectorX>(f.data(), n);                          // :42
    L.template triangularView<Lower>().solveInPlace(alpha);                      // :43
    L.transpose().template triangularView<Upper>().solveInPlace(alpha);          // :44
    std::vector<float> Lout((size_t)n * n, 0.f);
    for (int c = 0; c < n; ++c) for (int r = c; r < n; ++r) Lout[r + (size_t)c * n] = L(r, c);
    wr(out, Lout.data(), 4 * Lout.size());
    wr(out, alpha.data(), 4 * (size_t)n);
    std::vector<float> val(nq), var(nq);
    for (int i = 0; i < nq; ++i) {            // one query per call, as ObsGP2D::test drives GPou::test (ObsGP.cpp:352-408)
        EMatrixX Ks(n, 1);
        for (int k = 0; k < n; ++k) Ks(k, 0) = orc::ou_k(orc::dist_n(&x[(size_t)dim * k], &q[(size_t)dim * i], dim), a);
        EVectorX fm = Ks.transpose() * alpha;                                    // :54
        L.template triangularView<Lower>().solveInPlace(Ks);                     // :56
        Ks = Ks.array().pow(2);                                                  // :58
        EVectorX v = Ks.colwise().sum();                                         // :59
        EVectorX vr;
        vr = 1 + noise - v.head(1).array();                                      // :61
        val[i] = fm(0); var[i] = vr(0);
    }
    wr(out, val.data(), 4 * (size_t)nq); wr(out, var.data(), 4 * (size_t)nq);
}

// OnGPIS.cpp:91-149 (train), :177-216 / :218-263 (testSinglePoint / test2Dpoint)
static void case_ongpis(FILE* in, FILE* out, int dim, int n, int nq, float scale) {
    std::vector<float> s((size_t)(2 * dim + 3) * n), xq((size_t)dim * nq);   // pos(dim) grad(dim) val sigx sigg per sample
    rd(in, s.data(), 4 * s.size()); rd(in, xq.data(), 4 * xq.size());
    const int st = 2 * dim + 3;
    std::vector<float> x((size_t)dim * n), sigx(n), sigg(n);
    std::vector<int> gidx(n, -1);
    int ng = 0;
    for (int k = 0; k < n; ++k) {
        for (int c = 0; c < dim; ++c) x[(size_t)dim * k + c] = s[(size_t)st * k + c];
        sigx[k] = s[(size_t)st * k + 2 * dim + 1]; sigg[k] = s[(size_t)st * k + 2 * dim + 2];
        bool tiny = true;
        for (int c = 0; c < dim; ++c) tiny = tiny && (std::fabs(s[(size_t)st * k + dim + c]) < 1e-6);
        if (sigg[k] > 0.1001 || tiny) sigx[k] = 2.0f;                            // OnGPIS.cpp:122-125
        else gidx[k] = ng++;
    }
    const int K = n + dim * ng;
    EVectorX y(K);
    for (int k = 0; k < n; ++k) {
        y(k) = s[(size_t)st * k + 2 * dim];
        if (gidx[k] >= 0) for (int c = 0; c < dim; ++c) y(n + c * ng + gidx[k]) = s[(size_t)st * k + dim + c];
    }
    std::vector<float> Kl((size_t)K * K, 0.f);
    orc::matern32_train_lower(dim, n, x.data(), gidx.data(), ng, scale, sigx.data(), sigg.data(), Kl.data(), K);
    EMatrixX Km(K, K);
    for (int c = 0; c < K; ++c) for (int r = c; r < K; ++r) { Km(r, c) = Kl[r + (size_t)c * K]; Km(c, r) = Kl[r + (size_t)c * K]; }
    EMatrixX L = Km.llt().matrixL();                                             // OnGPIS.cpp:139
    EVectorX alpha = y;                                                          // :141
    L.template triangularView<Eigen::Lower>().solveInPlace(alpha);               // :142
    L.transpose().template triangularView<Eigen::Upper>().solveInPlace(alpha);   // :143
    wr(out, &K, 4);
    wr(out, alpha.data(), 4 * (size_t)K);
    const int nc = 1 + dim;
    const float tos = (float)(3.0 / (double)(scale * scale));                    // OnGPIS.h:58
    std::vector<float> pred((size_t)2 * nc * nq), ks((size_t)K * nc);
    for (int i = 0; i < nq; ++i) {
        orc::matern32_cross1(dim, n, x.data(), gidx.data(), ng, scale, &xq[(size_t)dim * i], ks.data(), K);
        EMatrixX Ks = Eigen::Map<EMatrixX>(ks.data(), K, nc);
        EVectorX res = Ks.transpose() * alpha;                                   // :187
        L.template triangularView<Eigen::Lower>().solveInPlace(Ks);              // :199
        Ks = Ks.array().pow(2);                                                  // :200
        EVectorX v = Ks.colwise().sum();                                         // :201
        float* o = &pred[(size_t)2 * nc * i];
        for (int c = 0; c < nc; ++c) o[c] = res(c);
        if (dim == 2) { o[nc] = 1.01 - v(0); o[nc + 1] = tos + 0.1 - v(1); o[nc + 2] = tos + 0.1 - v(2); }                                   // :235-237
        else { o[nc] = 1.001 - v(0); o[nc + 1] = tos + 0.001 - v(1); o[nc + 2] = tos + 0.001 - v(2); o[nc + 3] = tos + 0.001 - v(3); }       // :208-213
    }
    wr(out, pred.data(), 4 * pred.size());
}

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: eigen_probe in.bin out.bin\n"); return 2; }
    FILE* in = fopen(argv[1], "rb");
    FILE* out = fopen(argv[2], "wb");
    if (!in || !out) { fprintf(stderr, "eigen_probe: cannot open files\n"); return 2; }
    int ver[4] = {EIGEN_WORLD_VERSION, EIGEN_MAJOR_VERSION, EIGEN_MINOR_VERSION,
#ifdef EIGEN_VECTORIZE
                  1
#else
                  0
#endif
    };
    wr(out, ver, sizeof(ver));
    int ncase = 0;
    rd(in, &ncase, 4);
    for (int c = 0; c < ncase; ++c) {
        int hdr[3]; float scale;
        rd(in, hdr, sizeof(hdr)); rd(in, &scale, 4);
        if (hdr[0] == 0) case_gpou(in, out, hdr[1], hdr[2]);
        else case_ongpis(in, out, hdr[0], hdr[1], hdr[2], scale);
    }
    fclose(in); fclose(out);
    return 0;
}
