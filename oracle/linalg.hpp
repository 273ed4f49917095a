// ORACLE (test infrastructure only) -- fixed-order fp32 dense kernels.
//
// This directory is the CPU restatement of the GPisMap GP hot path.  It is NOT
// product code: only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may use it.  PARITY UNPINNED: the reference has no tests or
// golden vectors, and its arithmetic lives in Eigen (un-vendored, version not
// pinned, absent from this image), so the reference cannot be built here.
//
// The reference calls Eigen for (reference file:line)
//   K.llt().matrixL()                         ObsGP.cpp:41, OnGPIS.cpp:79,139
//   L.triangularView<Lower>().solveInPlace    ObsGP.cpp:43,56  OnGPIS.cpp:82,142,199
//   L^T.triangularView<Upper>().solveInPlace  ObsGP.cpp:44     OnGPIS.cpp:83,143
// Eigen fixes only the summation order of these textbook operations (last-bit
// rounding).  This restatement fixes its own order, chosen so that a blocked
// GPU implementation reproduces it bit for bit:
//   (O1) Cholesky / forward substitution: every element is one fmaf chain over
//        ascending k:   s = a_ij; s = fmaf(-l_ik, l_jk, s), k = 0..j-1
//   (O2) backward substitution: one fmaf chain over DEscending k.
//   (O3) long dot products / sums of squares in OnGPIS prediction: 2*W interleaved fmaf
//        chains (gp.hpp, OnGPIS::reduce_O3); (O4)/(O5) ObsGP mean butterfly / variance chain.
//   sqrt and divide are IEEE correctly rounded.
// Storage is column-major with leading dimension ld (as Eigen's MatrixXf).
#pragma once
#include <cmath>
#include <cstddef>
#include <vector>

namespace orc {

// In-place lower Cholesky of the n x n matrix whose lower triangle is stored
// column-major in A (ld >= n).  Row-by-row ("up-looking") so the inner loop is
// a contiguous axpy over a column of L; the per-element operation order is (O1).
// A non-positive pivot yields NaN exactly like sqrtf would (Eigen reports
// NumericalIssue but the reference never checks it).
static inline void chol_lower(float* A, int n, int ld) {
    std::vector<float> s(n > 0 ? n : 1);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < i; ++j) s[j] = A[i + (size_t)j * ld];
        float d = A[i + (size_t)i * ld];
        for (int k = 0; k < i; ++k) {
            const float* col = A + (size_t)k * ld;
            float xk = s[k] / col[k];
            s[k] = xk;
            float nxk = -xk;
            for (int j = k + 1; j < i; ++j) s[j] = fmaf(col[j], nxk, s[j]);
            d = fmaf(nxk, xk, d);
        }
        for (int j = 0; j < i; ++j) A[i + (size_t)j * ld] = s[j];
        A[i + (size_t)i * ld] = sqrtf(d);
    }
}

// b <- L^{-1} b for nrhs right-hand sides stored column-major (ldb).  Order (O1).
static inline void fwd_subst(const float* L, int n, int ld, float* B, int nrhs, int ldb) {
    for (int c = 0; c < nrhs; ++c) {
        float* b = B + (size_t)c * ldb;
        for (int k = 0; k < n; ++k) {
            const float* col = L + (size_t)k * ld;
            float xk = b[k] / col[k];
            b[k] = xk;
            float nxk = -xk;
            for (int j = k + 1; j < n; ++j) b[j] = fmaf(col[j], nxk, b[j]);
        }
    }
}

// B <- L^{-1} B for a MATRIX right-hand side (the K x (1+dim) cross-covariance block of one test
// point, OnGPIS.cpp:199).  For a matrix rhs Eigen dispatches to its blocked triangular-solve kernel,
// which scales the pivot row by a precomputed reciprocal (a = 1/l_kk; b_k *= a) instead of dividing;
// the vector overload above (used for alpha) divides.  Order (O1) otherwise.
static inline void fwd_subst_rcp(const float* L, int n, int ld, float* B, int nrhs, int ldb) {
    std::vector<float> rinv(n > 0 ? n : 1);
    for (int k = 0; k < n; ++k) rinv[k] = 1.0f / L[k + (size_t)k * ld];
    for (int c = 0; c < nrhs; ++c) {
        float* b = B + (size_t)c * ldb;
        for (int k = 0; k < n; ++k) {
            const float* col = L + (size_t)k * ld;
            float xk = b[k] * rinv[k];
            b[k] = xk;
            float nxk = -xk;
            for (int j = k + 1; j < n; ++j) b[j] = fmaf(col[j], nxk, b[j]);
        }
    }
}

// b <- L^{-T} b (single rhs).  Order (O2): for row j the chain runs k = n-1 .. j+1.
static inline void bwd_subst(const float* L, int n, int ld, float* b) {
    for (int j = n - 1; j >= 0; --j) {
        const float* col = L + (size_t)j * ld;
        float s = b[j];
        for (int k = n - 1; k > j; --k) s = fmaf(-col[k], b[k], s);
        b[j] = s / col[j];
    }
}

}  // namespace orc
