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
#include <algorithm>
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
//   mode 3 "eigen33": the operation orders of Eigen 3.3's PUBLISHED algorithms for exactly the calls the reference makes
//          (restated from the Eigen 3.3.x sources from memory -- Eigen is not in this image, so this is a proxy that
//          nobody here can check against the real library; PARITY STAYS UNPINNED):
//            K.llt()                   LLT.h llt_inplace<Lower>::blocked: block size (n/8 rounded down to a multiple of 16,
//                                      clamped to [8, 128]; unblocked below 32); per block: unblocked factorisation of
//                                      the diagonal block (row squaredNorm summed left to right, column update by the
//                                      column-major GEMV kernel, true division by the pivot), right-side triangular
//                                      solve of the panel (TriangularSolverMatrix.h: small panels of 12 columns, the
//                                      earlier columns of the kc block subtracted as ONE sum, reciprocal multiply),
//                                      rank update of the trailing matrix (one sum per block, subtracted once);
//            solveInPlace(vector)      TriangularSolverVector.h: panels of 8; column-major (L): true division, in-panel
//                                      axpy, then the GEMV kernel on the rows below; row-major (L^T): dot-product GEMV
//                                      on the columns already solved, in-panel dots, true division;
//            solveInPlace(matrix)      TriangularSolverMatrix.h OnTheLeft: kc blocks, small panels of 12 rows solved with
//                                      reciprocal multiplies and axpys, 12-term sums subtracted from the rows below;
//            K^T alpha, dots           GeneralMatrixVector.h row-major kernel / Redux.h: four packet lanes of partial sums,
//                                      reduced (a0+a2)+(a1+a3) (SSE2 predux), scalar tail;
//            .pow(2).colwise().sum()   sequential.
//          Modelled build: x86-64 SSE2 (the reference's mex build: -O2, no -march), i.e. packets of 4 floats, multiply
//          and add rounded separately (no fma), gebp mr x nr = 12 x 4, L1 = 32 KB in the blocking heuristic.
//          Simplifications (stated, not hidden): every GEMV / reduction is taken in its "all aligned" configuration --
//          Eigen peels a few leading elements or columns depending on the run-time address of the data, which only moves
//          individual elements between the scalar and the packet association.
enum { ARITH_TILED = 0, ARITH_NATURAL = 1, ARITH_FP64ACC = 2, ARITH_EIGEN33 = 3 };
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

// ---- mode 3: Eigen 3.3 orders (see the header of this section) ---------------------------------------------------
namespace eig {
static inline float predux4(const float* p) { return (p[0] + p[2]) + (p[1] + p[3]); }   // SSE2 predux<Packet4f>
// Redux.h, LinearVectorizedTraversal, NoUnrolling: sum of v[0..n)
static inline float redux_sum(const float* v, int n) {
    if (n <= 0) return 0.f;
    const int ps = 4, a2 = (n / (2 * ps)) * (2 * ps), a1 = (n / ps) * ps;
    float res;
    if (a1) {
        float p0[4] = {v[0], v[1], v[2], v[3]};
        if (a1 > ps) {
            float p1[4] = {v[4], v[5], v[6], v[7]};
            for (int i = 2 * ps; i < a2; i += 2 * ps)
                for (int l = 0; l < 4; ++l) { p0[l] = p0[l] + v[i + l]; p1[l] = p1[l] + v[i + ps + l]; }
            for (int l = 0; l < 4; ++l) p0[l] = p0[l] + p1[l];
            if (a1 > a2) for (int l = 0; l < 4; ++l) p0[l] = p0[l] + v[a2 + l];
        }
        res = predux4(p0);
        for (int i = a1; i < n; ++i) res = res + v[i];
    } else {
        res = v[0];
        for (int i = 1; i < n; ++i) res = res + v[i];
    }
    return res;
}
static inline float dot(const float* a, const float* b, int n) {      // (a.cwiseProduct(b)).sum()
    std::vector<float> p(n > 0 ? n : 1);
    for (int i = 0; i < n; ++i) p[i] = a[i] * b[i];
    return redux_sum(p.data(), n);
}
// GeneralMatrixVector.h, column-major kernel: res += alpha * A x  (A: rows x cols, leading dimension lda)
static inline void gemv_col(int rows, int cols, const float* A, int lda, const float* x, float* res, float alpha) {
    const int vec = rows & ~3, cb = (cols / 4) * 4;
    for (int i = 0; i < cb; i += 4) {
        const float t0 = alpha * x[i], t1 = alpha * x[i + 1], t2 = alpha * x[i + 2], t3 = alpha * x[i + 3];
        const float *c0 = A + (size_t)i * lda, *c1 = c0 + lda, *c2 = c1 + lda, *c3 = c2 + lda;
        for (int j = 0; j < vec; ++j) res[j] = res[j] + ((c0[j] * t0 + c1[j] * t1) + (c2[j] * t2 + c3[j] * t3));
        for (int j = vec; j < rows; ++j) {
            res[j] = c0[j] * t0 + res[j]; res[j] = c1[j] * t1 + res[j];
            res[j] = c2[j] * t2 + res[j]; res[j] = c3[j] * t3 + res[j];
        }
    }
    for (int i = cb; i < cols; ++i) {
        const float t = alpha * x[i];
        const float* c = A + (size_t)i * lda;
        for (int j = 0; j < rows; ++j) res[j] = c[j] * t + res[j];
    }
}
// GeneralMatrixVector.h, row-major kernel: res[i] += alpha * (row i of A) . x ; row i = A + i * lda, contiguous
static inline void gemv_row(int rows, int cols, const float* A, int lda, const float* x, float* res, float alpha) {
    const int vec = cols & ~3;
    for (int i = 0; i < rows; ++i) {
        const float* r = A + (size_t)i * lda;
        float pt[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < vec; j += 4)
            for (int l = 0; l < 4; ++l) pt[l] = x[j + l] * r[j + l] + pt[l];
        float tmp = 0.f;
        tmp = tmp + predux4(pt);
        for (int j = vec; j < cols; ++j) tmp = tmp + r[j] * x[j];
        res[i] = res[i] + alpha * tmp;
    }
}
// LLT.h llt_inplace<Lower>::unblocked on the n x n block at A (leading dimension ld)
static inline void llt_unblocked(float* A, int n, int ld) {
    std::vector<float> row(n > 0 ? n : 1);
    for (int k = 0; k < n; ++k) {
        const int rs = n - k - 1;
        float x = A[k + (size_t)k * ld];
        if (k > 0) {
            float sq = A[k] * A[k];
            for (int c = 1; c < k; ++c) sq = sq + A[k + (size_t)c * ld] * A[k + (size_t)c * ld];
            x = x - sq;
        }
        x = std::sqrt(x);
        A[k + (size_t)k * ld] = x;
        if (k > 0 && rs > 0) {
            for (int c = 0; c < k; ++c) row[c] = A[k + (size_t)c * ld];
            gemv_col(rs, k, A + k + 1, ld, row.data(), A + (k + 1) + (size_t)k * ld, -1.f);
        }
        for (int i = 0; i < rs; ++i) A[(k + 1 + i) + (size_t)k * ld] = A[(k + 1 + i) + (size_t)k * ld] / x;
    }
}
// product blocking (GeneralBlockPanelKernel.h evaluateProductBlockingSizesHeuristic, one thread, L1 = 32 KB)
static inline int kc_of(int k, int m, int n, int kcfactor) {
    if (std::max(k, std::max(m, n)) < 48) return k;
    const int k_peeling = 8, k_div = kcfactor * (12 * 4 + 4 * 4), k_sub = 12 * 4 * 4;
    const int max_kc = std::max(((32768 - k_sub) / k_div) & ~(k_peeling - 1), 1);
    if (k > max_kc) k = (k % max_kc) == 0 ? max_kc : max_kc - k_peeling * ((max_kc - 1 - (k % max_kc)) / (k_peeling * (k / max_kc + 1)));
    return k;
}
// TriangularSolverMatrix.h, OnTheRight, Upper (= transposed lower factor): X (rows x n, ldx) <- X T^-T, T = lower n x n (ldt)
static inline void trsm_right(float* X, int rows, int ldx, const float* T, int n, int ldt) {
    const int kc = kc_of(n, rows, n, 4), SP = 12;
    std::vector<float> acc(rows > 0 ? rows : 1);
    for (int k2 = 0; k2 < n; k2 += kc) {
        const int akc = std::min(n - k2, kc);
        for (int j2 = 0; j2 < akc; j2 += SP) {
            const int pw = std::min(akc - j2, SP);
            if (j2 > 0)       // gebp: the earlier columns of this kc block, one sum per element
                for (int c = 0; c < pw; ++c) {
                    const int j = k2 + j2 + c;
                    for (int i = 0; i < rows; ++i) acc[i] = 0.f;
                    for (int k = 0; k < j2; ++k) {
                        const float b = T[j + (size_t)(k2 + k) * ldt];
                        const float* a = X + (size_t)(k2 + k) * ldx;
                        for (int i = 0; i < rows; ++i) acc[i] = a[i] * b + acc[i];
                    }
                    float* r = X + (size_t)j * ldx;
                    for (int i = 0; i < rows; ++i) r[i] = acc[i] * -1.f + r[i];
                }
            for (int k = 0; k < pw; ++k) {
                const int j = k2 + j2 + k;
                float* r = X + (size_t)j * ldx;
                for (int k3 = 0; k3 < k; ++k3) {
                    const float b = T[j + (size_t)(k2 + j2 + k3) * ldt];
                    const float* a = X + (size_t)(k2 + j2 + k3) * ldx;
                    for (int i = 0; i < rows; ++i) r[i] = r[i] - a[i] * b;
                }
                const float inv = 1.f / T[j + (size_t)j * ldt];
                for (int i = 0; i < rows; ++i) r[i] = r[i] * inv;
            }
        }
        for (int j = k2 + akc; j < n; ++j) {     // the columns to the right of the kc block
            for (int i = 0; i < rows; ++i) acc[i] = 0.f;
            for (int k = 0; k < akc; ++k) {
                const float b = T[j + (size_t)(k2 + k) * ldt];
                const float* a = X + (size_t)(k2 + k) * ldx;
                for (int i = 0; i < rows; ++i) acc[i] = a[i] * b + acc[i];
            }
            float* r = X + (size_t)j * ldx;
            for (int i = 0; i < rows; ++i) r[i] = acc[i] * -1.f + r[i];
        }
    }
}
// LLT.h llt_inplace<Lower>::blocked
static inline void chol_lower(float* A, int n, int ld) {
    if (n < 32) { llt_unblocked(A, n, ld); return; }
    int bsz = n / 8;
    bsz = (bsz / 16) * 16;
    bsz = std::min(std::max(bsz, 8), 128);
    std::vector<float> acc;
    for (int k = 0; k < n; k += bsz) {
        const int bs = std::min(bsz, n - k), rs = n - k - bs;
        float* A11 = A + k + (size_t)k * ld;
        float* A21 = A + (k + bs) + (size_t)k * ld;
        llt_unblocked(A11, bs, ld);
        if (rs > 0) {
            trsm_right(A21, rs, ld, A11, bs, ld);
            // selfadjointView<Lower>().rankUpdate(A21, -1): lower triangle of A22 -= A21 A21^T, one sum over the block per element
            // (depth blocking: kc_of(bs, ., ., 1) >= 128 >= bs, a single pass)
            for (int j = 0; j < rs; ++j) {
                float* c = A + (k + bs) + (size_t)(k + bs + j) * ld;
                acc.assign(rs, 0.f);
                for (int kk = 0; kk < bs; ++kk) {
                    const float b = A21[j + (size_t)kk * ld];
                    const float* a = A21 + (size_t)kk * ld;
                    for (int i = j; i < rs; ++i) acc[i] = a[i] * b + acc[i];
                }
                for (int i = j; i < rs; ++i) c[i] = acc[i] * -1.f + c[i];
            }
        }
    }
}
// TriangularSolverVector.h, OnTheLeft, Lower, ColMajor: b <- L^-1 b
static inline void fwd_vec(const float* L, int n, int ld, float* b) {
    const int PW = 8;
    for (int pi = 0; pi < n; pi += PW) {
        const int apw = std::min(n - pi, PW), endb = pi + apw;
        for (int k = 0; k < apw; ++k) {
            const int i = pi + k;
            if (b[i] != 0.f) {
                b[i] = b[i] / L[i + (size_t)i * ld];
                const float* col = L + (size_t)i * ld;
                for (int r = i + 1; r < endb; ++r) b[r] = b[r] - b[i] * col[r];
            }
        }
        const int r = n - endb;
        if (r > 0) gemv_col(r, apw, L + endb + (size_t)pi * ld, ld, b + pi, b + endb, -1.f);
    }
}
// TriangularSolverVector.h, OnTheLeft, Upper, RowMajor (L^T seen through the column-major L): b <- L^-T b
static inline void bwd_vec(const float* L, int n, int ld, float* b) {
    const int PW = 8;
    std::vector<float> prod(PW);
    for (int pi = n; pi > 0; pi -= PW) {
        const int apw = std::min(pi, PW), r = n - pi, startRow = pi - apw;
        // rows startRow .. pi-1 of L^T are columns of L: row i of L^T = L[i .. , i] contiguous from the diagonal down
        if (r > 0) gemv_row(apw, r, L + pi + (size_t)startRow * ld, ld, b + pi, b + startRow, -1.f);
        for (int k = 0; k < apw; ++k) {
            const int i = pi - k - 1, s = i + 1;
            if (k > 0) {
                for (int q = 0; q < k; ++q) prod[q] = L[(s + q) + (size_t)i * ld] * b[s + q];
                b[i] = b[i] - redux_sum(prod.data(), k);
            }
            b[i] = b[i] / L[i + (size_t)i * ld];
        }
    }
}
// TriangularSolverMatrix.h, OnTheLeft, Lower, ColMajor: B (n x nrhs, ldb) <- L^-1 B
static inline void fwd_mat(const float* L, int n, int ld, float* B, int nrhs, int ldb) {
    const int kc = kc_of(n, n, nrhs, 4), SP = 12;
    for (int k2 = 0; k2 < n; k2 += kc) {
        const int akc = std::min(n - k2, kc);
        for (int c = 0; c < nrhs; ++c) {
            float* b = B + (size_t)c * ldb;
            for (int k1 = 0; k1 < akc; k1 += SP) {
                const int apw = std::min(akc - k1, SP);
                for (int k = 0; k < apw; ++k) {
                    const int i = k2 + k1 + k;
                    const float a = 1.f / L[i + (size_t)i * ld];
                    const float bv = (b[i] = b[i] * a);
                    const float* col = L + (size_t)i * ld;
                    for (int r = i + 1; r < k2 + k1 + apw; ++r) b[r] = b[r] - bv * col[r];
                }
                // the rows of this kc block below the small panel: one 12-term sum per row, subtracted once (gebp, alpha = -1)
                for (int r = k2 + k1 + apw; r < k2 + akc; ++r) {
                    float acc = 0.f;
                    for (int k = 0; k < apw; ++k) acc = L[r + (size_t)(k2 + k1 + k) * ld] * b[k2 + k1 + k] + acc;
                    b[r] = acc * -1.f + b[r];
                }
            }
            // the rows below the kc block: one sum over the block per row
            for (int r = k2 + akc; r < n; ++r) {
                float acc = 0.f;
                for (int k = 0; k < akc; ++k) acc = L[r + (size_t)(k2 + k) * ld] * b[k2 + k] + acc;
                b[r] = acc * -1.f + b[r];
            }
        }
    }
}
static inline float sum_sq_seq(const float* v, int n) {   // .array().pow(2).colwise().sum(): no packet path for pow
    float s = 0.f;
    for (int i = 0; i < n; ++i) s = s + v[i] * v[i];
    return s;
}
}  // namespace eig

// mode-dispatching entry points used by gp.hpp
static inline void chol_lower_m(float* A, int n, int ld) {
    switch (arith_mode()) {
        case ARITH_NATURAL: chol_lower_nat<float>(A, n, ld); break;
        case ARITH_FP64ACC: chol_lower_nat<double>(A, n, ld); break;
        case ARITH_EIGEN33: eig::chol_lower(A, n, ld); break;
        default: chol_lower(A, n, ld);
    }
}
static inline void fwd_subst_m(const float* L, int n, int ld, float* B, int nrhs, int ldb) {
    switch (arith_mode()) {
        case ARITH_NATURAL: fwd_subst_nat<float>(L, n, ld, B, nrhs, ldb); break;
        case ARITH_FP64ACC: fwd_subst_nat<double>(L, n, ld, B, nrhs, ldb); break;
        case ARITH_EIGEN33: eig::fwd_mat(L, n, ld, B, nrhs, ldb); break;      // matrix right-hand side (predictions)
        default: fwd_subst(L, n, ld, B, nrhs, ldb);
    }
}
static inline void bwd_subst_m(const float* L, int n, int ld, float* b) {
    switch (arith_mode()) {
        case ARITH_NATURAL: bwd_subst_nat<float>(L, n, ld, b); break;
        case ARITH_FP64ACC: bwd_subst_nat<double>(L, n, ld, b); break;
        case ARITH_EIGEN33: eig::bwd_vec(L, n, ld, b); break;
        default: bwd_subst(L, n, ld, b);
    }
}

}  // namespace orc
