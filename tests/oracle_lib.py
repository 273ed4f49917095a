"""ctypes binding of the CPU oracle (oracle/libgpis_oracle.so) -- test infrastructure only."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None

fp = C.POINTER(C.c_float)
dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)


def _p(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t))


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ORACLE_DIR, "libgpis_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc3_create.restype = C.c_void_p
        L.orc3_create.argtypes = [dp]
        for name in ("orc3_destroy", "orc3_reset"):
            getattr(L, name).argtypes = [C.c_void_p]
        L.orc3_set_threads.argtypes = [C.c_void_p, C.c_int]
        L.orc3_set_camera.argtypes = [C.c_void_p, dp]
        L.orc3_update.argtypes = [C.c_void_p, fp, C.c_int, fp]
        L.orc3_test.argtypes = [C.c_void_p, fp, C.c_int, C.c_int, fp]
        L.orc3_num_points.argtypes = [C.c_void_p]
        L.orc3_get_points.argtypes = [C.c_void_p, fp, C.c_int]
        L.orc3_get_nodes.argtypes = [C.c_void_p, fp, C.c_int]
        L.orc3_num_clusters.argtypes = [C.c_void_p]
        L.orc3_stats.argtypes = [C.c_void_p, C.POINTER(C.c_long)]
        L.orc3_obsgp_query.argtypes = [C.c_void_p, fp, C.c_int, fp, fp]
        L.orc3_test_flags.argtypes = [C.c_void_p, fp, C.c_int, ip]
        L.orc3_obs_dims.argtypes = [C.c_void_p, ip, ip]
        L.orc3_get_obs.argtypes = [C.c_void_p, fp, fp]
        L.orc3_obsgp_num_tiles.argtypes = [C.c_void_p]
        L.orc3_obsgp_tile.argtypes = [C.c_void_p, C.c_int, fp, fp, fp]
        L.orc2_create.restype = C.c_void_p
        for name in ("orc2_destroy", "orc2_reset"):
            getattr(L, name).argtypes = [C.c_void_p]
        L.orc2_set_threads.argtypes = [C.c_void_p, C.c_int]
        L.orc2_update.argtypes = [C.c_void_p, fp, fp, C.c_int, fp]
        L.orc2_test.argtypes = [C.c_void_p, fp, C.c_int, C.c_int, fp]
        L.orc2_test_flags.argtypes = [C.c_void_p, fp, C.c_int, ip]
        L.orc2_get_nodes.argtypes = [C.c_void_p, fp, C.c_int]
        L.orc2_stats.argtypes = [C.c_void_p, C.POINTER(C.c_long)]
        L.orc2_obsgp_sizes.argtypes = [C.c_void_p, ip, C.c_int]
        L.orc3_retrain_all.argtypes = [C.c_void_p]
        L.orc3_cluster_samples.argtypes = [C.c_void_p, C.c_int, fp, C.c_int]
        L.orc2_retrain_all.argtypes = [C.c_void_p]
        L.orc_set_arith_mode.argtypes = [C.c_int]
        L.orc_chol_lower.argtypes = [fp, C.c_int, C.c_int]
        L.orc_fwd_subst.argtypes = [fp, C.c_int, C.c_int, fp, C.c_int, C.c_int]
        L.orc_fwd_subst_blocked.argtypes = [fp, C.c_int, C.c_int, fp, C.c_int, C.c_int]
        L.orc_bwd_subst.argtypes = [fp, C.c_int, C.c_int, fp]
        L.orc_gpou_train.argtypes = [fp, fp, C.c_int, C.c_int, fp, fp]
        L.orc_gpou_test.argtypes = [fp, fp, C.c_int, C.c_int, fp, C.c_int, fp, fp]
        L.orc_ongpis_train.argtypes = [C.c_int, C.c_float, fp, fp, fp, fp, fp, C.c_int, fp, fp, ip]
        L.orc_matern32_train.argtypes = [C.c_int, C.c_int, fp, ip, C.c_int, C.c_float, fp, fp, fp]
        L.orc_matern32_cross.argtypes = [C.c_int, C.c_int, fp, ip, C.c_int, C.c_float, fp, fp]
        L.orc_ongpis_predict.argtypes = [C.c_int, C.c_float, fp, fp, fp, fp, fp, C.c_int, fp, C.c_int, fp]
        _LIB = L
    return _LIB


ARITH_MODES = {"tiled": 0, "natural": 1, "fp64acc": 2, "eigen33": 3}


def set_arith_mode(mode):
    """Arithmetic variant of every subsequent oracle training / prediction (oracle/linalg.hpp): "tiled" (default;
    the orders the HIP kernels reproduce bit for bit), "natural" (plain left-to-right fp32, no fma, plain
    substitution), "fp64acc" (natural order with double accumulators) or "eigen33" (the orders of Eigen 3.3's published
    blocked LLT / triangular solves / reductions, restated from memory: a proxy, parity stays unpinned).  Process-global; reset to "tiled" after use."""
    lib().orc_set_arith_mode(ARITH_MODES[mode] if isinstance(mode, str) else int(mode))


def ongpis_train(dim, scale, pos, grad, val, sx, sg):
    """Oracle OnGPIS::train: returns dict(K, L[r,c], alpha, gidx)."""
    L_ = lib()
    pos = np.ascontiguousarray(pos, dtype=np.float32); grad = np.ascontiguousarray(grad, dtype=np.float32)
    val = np.ascontiguousarray(val, dtype=np.float32); sx = np.ascontiguousarray(sx, dtype=np.float32)
    sg = np.ascontiguousarray(sg, dtype=np.float32)
    n = val.size
    gidx = np.zeros(n, dtype=np.int32)
    K = L_.orc_ongpis_train(dim, C.c_float(scale), _p(pos), _p(grad), _p(val), _p(sx), _p(sg), n, None, None, _p(gidx, C.c_int))
    Lm = np.zeros(K * K, dtype=np.float32)
    alpha = np.zeros(K, dtype=np.float32)
    L_.orc_ongpis_train(dim, C.c_float(scale), _p(pos), _p(grad), _p(val), _p(sx), _p(sg), n, _p(Lm), _p(alpha), _p(gidx, C.c_int))
    return dict(K=K, L=Lm.reshape(K, K).T.copy(), alpha=alpha, gidx=gidx)


def ongpis_predict(dim, scale, pos, grad, val, sx, sg, xq):
    """Oracle train + testSinglePoint for each query: returns [nq, 2(1+dim)] (mean, var)."""
    L_ = lib()
    pos = np.ascontiguousarray(pos, dtype=np.float32); grad = np.ascontiguousarray(grad, dtype=np.float32)
    val = np.ascontiguousarray(val, dtype=np.float32); sx = np.ascontiguousarray(sx, dtype=np.float32)
    sg = np.ascontiguousarray(sg, dtype=np.float32); xq = np.ascontiguousarray(xq, dtype=np.float32)
    nq = xq.shape[0]
    out = np.zeros((nq, 2 * (1 + dim)), dtype=np.float32)
    L_.orc_ongpis_predict(dim, C.c_float(scale), _p(pos), _p(grad), _p(val), _p(sx), _p(sg), val.size, _p(xq), nq, _p(out))
    return out


class OracleMap3:
    """Mirror of the reference's mexGPisMap3 command set on the CPU oracle."""

    def __init__(self, cam6=None, threads=None):
        self.L = lib()
        c = None if cam6 is None else _p(np.ascontiguousarray(cam6, dtype=np.float64), C.c_double)
        self.h = C.c_void_p(self.L.orc3_create(c))
        if threads:
            self.L.orc3_set_threads(self.h, threads)

    def close(self):
        if self.h:
            self.L.orc3_destroy(self.h)
            self.h = None

    __del__ = close

    def set_camera(self, cam6):
        self.L.orc3_set_camera(self.h, _p(np.ascontiguousarray(cam6, dtype=np.float64), C.c_double))

    def update(self, depth, pose):
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        pose = np.ascontiguousarray(pose, dtype=np.float32)
        self.L.orc3_update(self.h, _p(depth), depth.size, _p(pose))

    def test(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        res = np.zeros((x.shape[0], 8), dtype=np.float32)
        ok = self.L.orc3_test(self.h, _p(x), 3, x.shape[0], _p(res))
        return res if ok else None

    def test_flags(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        fl = np.zeros(x.shape[0], dtype=np.int32)
        self.L.orc3_test_flags(self.h, _p(x), x.shape[0], _p(fl, C.c_int))
        return fl

    def obs(self):
        ni, nj = C.c_int(0), C.c_int(0)
        self.L.orc3_obs_dims(self.h, C.byref(ni), C.byref(nj))
        vu = np.zeros(2 * ni.value * nj.value, dtype=np.float32)
        zinv = np.zeros(ni.value * nj.value, dtype=np.float32)
        self.L.orc3_get_obs(self.h, _p(vu), _p(zinv))
        return vu, zinv, ni.value, nj.value

    def obsgp_tile(self, t):
        x = np.zeros((64, 2), dtype=np.float32)
        alpha = np.zeros(64, dtype=np.float32)
        Lm = np.zeros(64 * 64, dtype=np.float32)
        n = self.L.orc3_obsgp_tile(self.h, t, _p(x), _p(alpha), _p(Lm))
        return n, x[:n], alpha[:n], Lm[:n * n].reshape(n, n).T.copy()  # L[r, c]

    def obsgp_num_tiles(self):
        return self.L.orc3_obsgp_num_tiles(self.h)

    def num_points(self):
        return self.L.orc3_num_points(self.h)

    def nodes(self):
        n = self.L.orc3_get_nodes(self.h, None, 0)
        out = np.zeros((n, 9), dtype=np.float32)
        if n:
            self.L.orc3_get_nodes(self.h, _p(out), n)
        return out

    def num_clusters(self):
        return self.L.orc3_num_clusters(self.h)

    def cluster_samples(self, i):
        """Training set of the i-th trained cluster (traversal order): [n, 9] pos3 grad3 val sigx sigg, or None."""
        n = self.L.orc3_cluster_samples(self.h, i, None, 0)
        if n <= 0:
            return None
        out = np.zeros((n, 9), dtype=np.float32)
        self.L.orc3_cluster_samples(self.h, i, _p(out), n)
        return out

    def retrain_all(self, mode):
        """Re-factorise every trained cluster on its stored training set in arithmetic `mode`; the map itself (points,
        tree, cluster sets) is untouched.  Leaves the process-global mode at "tiled".  Returns the clusters retrained."""
        set_arith_mode(mode)
        n = self.L.orc3_retrain_all(self.h)
        set_arith_mode("tiled")
        return n

    def stats(self):
        a = (C.c_long * 6)()
        self.L.orc3_stats(self.h, a)
        return dict(zip(("obsgp_tiles", "obsgp_queries", "clusters_trained", "sumK", "maxK", "gp_evals"), list(a)))

    def obsgp_query(self, vu):
        vu = np.ascontiguousarray(vu, dtype=np.float32)
        n = vu.shape[0]
        val = np.zeros(n, dtype=np.float32)
        var = np.zeros(n, dtype=np.float32)
        self.L.orc3_obsgp_query(self.h, _p(vu), n, _p(val), _p(var))
        return val, var


class OracleMap2:
    """Mirror of the reference's mexGPisMap command set ('update', 'test', 'reset') on the CPU oracle."""

    def __init__(self, threads=None):
        self.L = lib()
        self.h = C.c_void_p(self.L.orc2_create())
        if threads:
            self.L.orc2_set_threads(self.h, threads)

    def close(self):
        if self.h:
            self.L.orc2_destroy(self.h)
            self.h = None

    __del__ = close

    def reset(self):
        self.L.orc2_reset(self.h)

    def update(self, thetas, ranges, pose6):
        thetas = np.ascontiguousarray(thetas, dtype=np.float32)
        ranges = np.ascontiguousarray(ranges, dtype=np.float32)
        pose6 = np.ascontiguousarray(pose6, dtype=np.float32)
        self.L.orc2_update(self.h, _p(thetas), _p(ranges), ranges.size, _p(pose6))

    def test(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        res = np.zeros((x.shape[0], 6), dtype=np.float32)
        ok = self.L.orc2_test(self.h, _p(x), 2, x.shape[0], _p(res))
        return res if ok else None

    def test_flags(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        fl = np.zeros(x.shape[0], dtype=np.int32)
        self.L.orc2_test_flags(self.h, _p(x), x.shape[0], _p(fl, C.c_int))
        return fl

    def retrain_all(self, mode):
        set_arith_mode(mode)
        n = self.L.orc2_retrain_all(self.h)
        set_arith_mode("tiled")
        return n

    def nodes(self):
        n = self.L.orc2_get_nodes(self.h, None, 0)
        out = np.zeros((n, 7), dtype=np.float32)
        if n:
            self.L.orc2_get_nodes(self.h, _p(out), n)
        return out

    def stats(self):
        a = (C.c_long * 6)()
        self.L.orc2_stats(self.h, a)
        return dict(zip(("obsgp_tiles", "obsgp_queries", "clusters_trained", "sumK", "maxK", "gp_evals"), list(a)))

    def obsgp_sizes(self):
        a = (C.c_int * 64)()
        n = self.L.orc2_obsgp_sizes(self.h, a, 64)
        return list(a)[:n]
