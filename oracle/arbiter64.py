"""ORACLE (test infrastructure only; parity unpinned): float64 numpy restatement of the GP
primitives (a1-a7 of SURVEY.md 8a) used as an arbiter for the fp32 oracle and the HIP kernels.
Formulas: reference cpp/src/covFnc.cpp:29-33 (kf, kf1, kf2), :47-109 (OU), :142-450 (Matern-3/2
with first-derivative blocks), cpp/src/OnGPIS.cpp:91-216, cpp/src/ObsGP.cpp:32-62."""
import numpy as np
from scipy.linalg import cholesky, solve_triangular


def ou_train(x, f, scale=0.5, noise=0.01):
    """x: [n, dim].  Returns (L, alpha)."""
    x = np.asarray(x, dtype=np.float64)
    d = np.linalg.norm(x[:, None, :] - x[None, :, :], axis=2)
    K = np.exp(-d / scale)
    np.fill_diagonal(K, 1.0 + noise)
    L = cholesky(K, lower=True)
    alpha = solve_triangular(L.T, solve_triangular(L, np.asarray(f, dtype=np.float64), lower=True), lower=False)
    return L, alpha


def ou_test(x, L, alpha, xq, scale=0.5, noise=0.01):
    x = np.asarray(x, dtype=np.float64)
    xq = np.asarray(xq, dtype=np.float64)
    k = np.exp(-np.linalg.norm(x[:, None, :] - xq[None, :, :], axis=2) / scale)   # [n, nq]
    mean = k.T @ alpha
    v = solve_triangular(L, k, lower=True)
    return mean, 1.0 + noise - np.sum(v * v, axis=0)


def _rows(N, gidx, dim):
    ng = int((gidx >= 0).sum())
    return ng, N + dim * ng


def matern_train_K(x, gidx, scale, sigx, sigg, quirk2d=True):
    """Full symmetric K for points x [N, dim]; row order [f; d/dx; d/dy; (d/dz)]."""
    x = np.asarray(x, dtype=np.float64)
    N, dim = x.shape
    ng, K = _rows(N, gidx, dim)
    a = np.sqrt(3.0) / scale
    M = np.zeros((K, K))
    for k in range(N):
        M[k, k] = 1.0 + sigx[k]
        if gidx[k] >= 0:
            for c in range(dim):
                r = N + c * ng + gidx[k]
                M[r, r] = a * a + sigg[k]
            if dim == 2 and quirk2d:
                r = N + gidx[k]
                M[r, r] = a * a + np.sqrt(sigx[k] * sigg[k])     # covFnc.cpp:352
        for j in range(k + 1, N):
            d = x[k] - x[j]
            r = np.linalg.norm(d)
            e = np.exp(-a * r)
            M[k, j] = M[j, k] = (1 + a * r) * e
            for c in range(dim):
                if gidx[k] >= 0:
                    rk = N + c * ng + gidx[k]
                    M[rk, j] = M[j, rk] = -a * a * d[c] * e
                if gidx[j] >= 0:
                    rj = N + c * ng + gidx[j]
                    M[k, rj] = M[rj, k] = a * a * d[c] * e
            if gidx[k] >= 0 and gidx[j] >= 0:
                for c1 in range(dim):
                    for c2 in range(dim):
                        rk = N + c1 * ng + gidx[k]
                        rj = N + c2 * ng + gidx[j]
                        v = a * a * ((1.0 if c1 == c2 else 0.0) - a * d[c1] * d[c2] / r) * e
                        M[rk, rj] = M[rj, rk] = v
    return M


def matern_cross(x, gidx, scale, xq):
    """k* for one query: [K, 1+dim]."""
    x = np.asarray(x, dtype=np.float64)
    xq = np.asarray(xq, dtype=np.float64)
    N, dim = x.shape
    ng, K = _rows(N, gidx, dim)
    a = np.sqrt(3.0) / scale
    out = np.zeros((K, 1 + dim))
    for k in range(N):
        d = x[k] - xq
        r = np.linalg.norm(d)
        e = np.exp(-a * r)
        out[k, 0] = (1 + a * r) * e
        for c in range(dim):
            out[k, 1 + c] = a * a * d[c] * e
        if gidx[k] >= 0:
            for c1 in range(dim):
                rk = N + c1 * ng + gidx[k]
                out[rk, 0] = -a * a * d[c1] * e
                for c2 in range(dim):
                    out[rk, 1 + c2] = a * a * ((1.0 if c1 == c2 else 0.0) - a * d[c1] * d[c2] / r) * e
    return out


def ongpis_train(pos, grad, val, sx, sg, scale):
    pos = np.asarray(pos, dtype=np.float64)
    grad = np.asarray(grad, dtype=np.float64)
    N, dim = pos.shape
    sigx = np.asarray(sx, dtype=np.float64).copy()
    sigg = np.asarray(sg, dtype=np.float64)
    gidx = np.full(N, -1, dtype=np.int64)
    g = 0
    for k in range(N):
        if np.float32(sg[k]) > 0.1001 or np.all(np.abs(grad[k]) < 1e-6):
            sigx[k] = 2.0
        else:
            gidx[k] = g
            g += 1
    ng = g
    K = N + dim * ng
    y = np.zeros(K)
    y[:N] = val
    for k in range(N):
        if gidx[k] >= 0:
            for c in range(dim):
                y[N + c * ng + gidx[k]] = grad[k, c]
    M = matern_train_K(pos, gidx, scale, sigx, sigg)
    L = cholesky(M, lower=True)
    alpha = solve_triangular(L.T, solve_triangular(L, y, lower=True), lower=False)
    return dict(gidx=gidx, K=K, L=L, alpha=alpha, x=pos, scale=scale, dim=dim)


def ongpis_test(m, xq):
    ks = matern_cross(m["x"], m["gidx"], m["scale"], xq)
    mean = ks.T @ m["alpha"]
    v = solve_triangular(m["L"], ks, lower=True)
    ss = np.sum(v * v, axis=0)
    tos = 3.0 / (m["scale"] ** 2)
    if m["dim"] == 3:
        prior = np.array([1.001] + [tos + 0.001] * 3)
    else:
        prior = np.array([1.01] + [tos + 0.1] * 2)
    return mean, prior - ss
