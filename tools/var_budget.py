#!/usr/bin/env python3
"""Error budget of the predicted variances: where does the gradient-variance difference between
the arithmetic orders come from?  CPU only (oracle + float64 arbiter).

For every trained cluster of data/3D after FRAME updates (default 40), with 40 queries near its points, the four
variances are computed six ways on the SAME training set:
   G  float64 everything (arbiter: the truth for the fp32 kernel matrix entries rounded as the reference rounds them)
   T  tiled      fp32 Cholesky (O1), explicit inverse (O6), fmaf chains           = the HIP path, bit for bit
   S  T's factor, plain fp32 forward substitution per query (no explicit inverse)
   D  T's factor, float64 substitution and sums                                   -> what the fp32 FACTOR alone costs
   N  natural    fp32, no fma, plain substitution                                 = independent order #1
   A  fp64acc    natural with double accumulators                                 = independent order #2
and reported as max / rms error against G, relative to the prior (1.001 value, 1875.001 gradients).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
from scipy.linalg import solve_triangular  # noqa: E402
import oracle_lib  # noqa: E402
import replay  # noqa: E402
import arbiter64  # noqa: E402
import ctypes as C  # noqa: E402


def main():
    nframes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    frames = replay.load_bigbird()
    om = oracle_lib.OracleMap3(frames[0]["cam"])
    for f in frames[:nframes]:
        om.set_camera(f["cam"])
        om.update(f["depth"], f["pose"])
    L_ = oracle_lib.lib()
    rng = np.random.default_rng(7)
    scale, dim, tos = 0.04, 3, 1875.0
    prior = np.array([1.001, tos + 0.001, tos + 0.001, tos + 0.001])
    names = ["T tiled (HIP path)", "S tiled factor + fp32 substitution", "D tiled factor + fp64 solve", "N natural", "A fp64acc"]
    err = {k: [] for k in names}
    Ks, conds = [], []
    i = 0
    while True:
        nd = om.cluster_samples(i)
        if nd is None:
            break
        i += 1
        pos, grad, val, sx, sg = nd[:, :3], nd[:, 3:6], nd[:, 6], nd[:, 7], nd[:, 8]
        n = nd.shape[0]
        xq = (pos[rng.integers(0, n, 40)] + rng.normal(0, 0.3 * scale, (40, 3))).astype(np.float32)
        g = arbiter64.ongpis_train(pos, grad, val, sx, sg, scale)
        # the arbiter's kernel matrix in float64 differs from the fp32-rounded entries the pipelines factor; use the
        # oracle's fp32 entries as THE matrix so that G measures arithmetic, not entry rounding
        ot = oracle_lib.ongpis_train(dim, scale, pos, grad, val, sx, sg)       # tiled: L (fp32), gidx
        K = ot["K"]
        gidx = ot["gidx"].astype(np.int32)
        ngr = int((gidx >= 0).sum())
        sigx2 = np.where(gidx >= 0, sx, 2.0).astype(np.float32)
        Kmat = np.zeros(K * K, dtype=np.float32)
        x32 = np.ascontiguousarray(pos, dtype=np.float32)
        L_.orc_matern32_train(dim, n, oracle_lib._p(x32), oracle_lib._p(gidx, C.c_int), ngr, C.c_float(scale),
                              oracle_lib._p(sigx2), oracle_lib._p(np.ascontiguousarray(sg, dtype=np.float32)), oracle_lib._p(Kmat))
        Kl = np.tril(Kmat.reshape(K, K).T.astype(np.float64))
        Kfull = Kl + np.tril(Kl, -1).T
        L64 = np.linalg.cholesky(Kfull)
        Lt = ot["L"].astype(np.float64)
        preds = {}
        for mode, key in (("tiled", names[0]), ("natural", names[3]), ("fp64acc", names[4])):
            oracle_lib.set_arith_mode(mode)
            preds[key] = oracle_lib.ongpis_predict(dim, scale, pos, grad, val, sx, sg, xq)[:, 4:8].astype(np.float64)
        oracle_lib.set_arith_mode("tiled")
        G = np.zeros((40, 4)); S = np.zeros((40, 4)); D = np.zeros((40, 4))
        Lt32 = np.ascontiguousarray(ot["L"].T, dtype=np.float32)     # column-major for the C entry
        for q in range(40):
            ks = np.zeros(K * 4, dtype=np.float32)
            L_.orc_matern32_cross(dim, n, oracle_lib._p(x32), oracle_lib._p(gidx, C.c_int), ngr, C.c_float(scale),
                                  oracle_lib._p(np.ascontiguousarray(xq[q])), oracle_lib._p(ks))
            ksm = ks.reshape(4, K).T.astype(np.float64)               # [K, 4]
            v = solve_triangular(L64, ksm, lower=True)
            G[q] = prior - np.sum(v * v, axis=0)
            v = solve_triangular(Lt, ksm, lower=True)
            D[q] = prior - np.sum(v * v, axis=0)
            b = ks.copy()
            L_.orc_fwd_subst(oracle_lib._p(Lt32), K, K, oracle_lib._p(b), 4, K)
            vb = b.reshape(4, K).T
            S[q] = prior - np.array([np.float32(np.sum(vb[:, c].astype(np.float32) ** 2, dtype=np.float32)) for c in range(4)])
        preds[names[1]] = S
        preds[names[2]] = D
        for k in names:
            err[k].append(np.abs(preds[k] - G) / prior)
        Ks.append(K)
        conds.append(float(np.max(np.diag(L64)) / np.min(np.diag(L64))))
    print("data/3D after %d frames: %d clusters, K %d..%d (mean %.0f), diag(L) ratio up to %.0f"
          % (nframes, len(Ks), min(Ks), max(Ks), np.mean(Ks), max(conds)))
    print("%-40s %-24s %-24s" % ("errors against float64, relative to the prior", "value variance max / rms", "gradient variances max / rms"))
    for k in names:
        e = np.concatenate(err[k])
        print("%-40s %.2e / %.2e      %.2e / %.2e" % (k, e[:, 0].max(), np.sqrt(np.mean(e[:, 0] ** 2)), e[:, 1:].max(), np.sqrt(np.mean(e[:, 1:] ** 2))))
    return 0


if __name__ == "__main__":
    sys.exit(main())
