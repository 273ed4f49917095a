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
//        chains (gp.hpp, OnGPIS::reduce_O3); (O4)/(O5) ObsGP mean butterfly / variance chain;
//   (O6) the product with an inverted diagonal block in the blocked matrix solve (below).
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
// point, OnGPIS.cpp:199).  For a matrix rhs Eigen dispatches to its blocked triangular-solve kernel
// (panels + GEMM updates); this restatement is the blocked algorithm GPU BLAS libraries use, with
// 32 x 32 diagonal blocks applied through their explicit inverses:
//     for each block c:   V_c = inv(L_cc) U_c ;   U_b -= L_bc V_c  for the rows b below.
// inv(L_cc) (blocked_diag_inverses) is computed once per factor by forward substitution (O1) on the
// unit vectors.  Orders: the update of a row is the ascending-k fmaf chain (O1); the product with the
// inverse is one fmaf chain from zero over k in the order (O6)
//     k = 0,4,1,5,2,6,3,7, 8,12,9,13,10,14,11,15, 16,... (pairs (j, j+4) inside every group of 8)
// -- the order in which a 32x32x2 matrix instruction meets the rows of an accumulator tile.  Terms with
// k > i vanish (the inverse is lower triangular).  Measured against an fp64 solve with the same factor
// this is at least as accurate as plain substitution ("Numerical contract").
static inline int o6_k(int t) { return (t & ~7) + ((t & 7) >> 1) + 4 * (t & 1); }
// inv: [nb][32*32] row-major inverse blocks, identity-padded past n.
static inline void blocked_diag_inverses(const float* L, int n, int ld, std::vector<float>& inv) {
    const int nb = (n + 31) / 32;
    inv.assign((size_t)nb * 1024, 0.f);
    for (int c = 0; c < nb; ++c) {
        const int r0 = 32 * c, m = (n - r0 < 32) ? n - r0 : 32;
        float* I = &inv[(size_t)c * 1024];
        for (int j = 0; j < 32; ++j) {
            float e[32];
            for (int k = 0; k < 32; ++k) e[k] = (k == j) ? 1.f : 0.f;
            if (j < m) fwd_subst(L + r0 + (size_t)r0 * ld, m, ld, e, 1, 32);
            for (int i = 0; i < 32; ++i) I[i * 32 + j] = e[i];
        }
    }
}
static inline void fwd_subst_blocked(const float* L, const float* inv, int n, int ld, float* B, int nrhs, int ldb) {
    const int nb = (n + 31) / 32;
    for (int cidx = 0; cidx < nrhs; ++cidx) {
        float* b = B + (size_t)cidx * ldb;
        for (int c = 0; c < nb; ++c) {
            const int r0 = 32 * c, m = (n - r0 < 32) ? n - r0 : 32;
            const float* I = inv + (size_t)c * 1024;
            float u[32], v[32];
            for (int k = 0; k < 32; ++k) u[k] = (k < m) ? b[r0 + k] : 0.f;
            for (int i = 0; i < m; ++i) {
                float s = 0.f;
                for (int t = 0; t < 32; ++t) { const int k = o6_k(t); s = fmaf(I[i * 32 + k], u[k], s); }
                v[i] = s;
            }
            for (int i = 0; i < m; ++i) b[r0 + i] = v[i];
            for (int k = 0; k < m; ++k) {
                const float* col = L + (size_t)(r0 + k) * ld;
                const float nv = v[k];
                for (int j = r0 + m; j < n; ++j) b[j] = fmaf(-col[j], nv, b[j]);
            }
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


// ---------------------------------------------------------------------------------------------------------
// Arithmetic variants (SURVEY 8(c) "fp32 and fp64-accumulate variants").  The routines above
// fix the summation orders a tiled GPU implementation reproduces bit for bit ("tiled", mode 0 -- the default
// and the only mode the product is compared with bit-exactly).  Eigen's own orders are unknowable without
// Eigen, so two INDEPENDENT orders of the same textbook operations are provided to show that the results
// do not depend on the order beyond fp32 round-off:
//   mode 1 "natural": what a plain non-blocked implementation does -- every element is
//          (a - (p_1 + p_2 + ... )) / d with the products summed left to right, separate multiply and
//          add roundings (no fused multiply-add, as an x86 build without -mfma), plain substitution for
//          the matrix right-hand side of a prediction (no blocks, no explicit inverses), sequential
//          left-to-right dot products and sums of squares;
//   mode 2 "fp64acc": the same algorithm as mode 1 with every accumulator (and the final subtract /
//          divide / sqrt of an element) in double, stored values rounded once to float.
// The mode is process-global test state (orc_set_arith_mode); it is read, never written, while a map runs.
// ---------------------------------------------------------------------------------------------------------
enum { ARITH_TILED = 0, ARITH_NATURAL = 1, ARITH_FP64ACC = 2 };
inline int& arith_mode() { static int m = ARITH_TILED; return m; }

template <class T>
static inline void chol_lower_nat(float* A, int n, int ld) {
    std::vector<T> s(n > 0 ? n : 1);
    std::vector<float> xs(n > 0 ? n : 1);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < i; ++j) s[j] = (T)0;
        T d = (T)0;
        for (int k = 0; k < i; ++k) {
            const float* col = A + (size_t)k * ld;
            const float xk = (float)(((T)A[i + (size_t)k * ld] - s[k]) / (T)col[k]);
            xs[k] = xk;
            const T xt = (T)xk;
            for (int j = k + 1; j < i; ++j) s[j] += (T)col[j] * xt;
            d += xt * xt;
        }
        for (int j = 0; j < i; ++j) A[i + (size_t)j * ld] = xs[j];
        A[i + (size_t)i * ld] = (float)std::sqrt((T)A[i + (size_t)i * ld] - d);
    }
}
template <class T>
static inline void fwd_subst_nat(const float* L, int n, int ld, float* B, int nrhs, int ldb) {
    std::vector<T> s(n > 0 ? n : 1);
    for (int c = 0; c < nrhs; ++c) {
        float* b = B + (size_t)c * ldb;
        for (int j = 0; j < n; ++j) s[j] = (T)0;
        for (int k = 0; k < n; ++k) {
            const float* col = L + (size_t)k * ld;
            const float xk = (float)(((T)b[k] - s[k]) / (T)col[k]);
            b[k] = xk;
            const T xt = (T)xk;
            for (int j = k + 1; j < n; ++j) s[j] += (T)col[j] * xt;
        }
    }
}
template <class T>
static inline void bwd_subst_nat(const float* L, int n, int ld, float* b) {
    for (int j = n - 1; j >= 0; --j) {
        const float* col = L + (size_t)j * ld;
        T s = (T)0;
        for (int k = j + 1; k < n; ++k) s += (T)col[k] * (T)b[k];
        b[j] = (float)(((T)b[j] - s) / (T)col[j]);
    }
}
template <class T>
static inline float dot_nat(const float* a, const float* b, int n) {
    T s = (T)0;
    for (int i = 0; i < n; ++i) s += (T)a[i] * (T)b[i];
    return (float)s;
}

// mode-dispatching entry points used by gp.hpp
static inline void chol_lower_m(float* A, int n, int ld) {
    switch (arith_mode()) {
        case ARITH_NATURAL: chol_lower_nat<float>(A, n, ld); break;
        case ARITH_FP64ACC: chol_lower_nat<double>(A, n, ld); break;
        default: chol_lower(A, n, ld);
    }
}
static inline void fwd_subst_m(const float* L, int n, int ld, float* B, int nrhs, int ldb) {
    switch (arith_mode()) {
        case ARITH_NATURAL: fwd_subst_nat<float>(L, n, ld, B, nrhs, ldb); break;
        case ARITH_FP64ACC: fwd_subst_nat<double>(L, n, ld, B, nrhs, ldb); break;
        default: fwd_subst(L, n, ld, B, nrhs, ldb);
    }
}
static inline void bwd_subst_m(const float* L, int n, int ld, float* b) {
    switch (arith_mode()) {
        case ARITH_NATURAL: bwd_subst_nat<float>(L, n, ld, b); break;
        case ARITH_FP64ACC: bwd_subst_nat<double>(L, n, ld, b); break;
        default: bwd_subst(L, n, ld, b);
    }
}

}  // namespace orc
