"""Generates the arithmetic-variant goldens and the small fixtures F1-F3/F5 of SURVEY.md 8(c):

  variants_3d.npz   data/3D (40 updates): map-point counts per frame and test() on the demo grid after frames
                    {1, 3, 10, 40} for the three oracle arithmetic modes (oracle/linalg.hpp: tiled / natural /
                    fp64acc), the branch-ambiguity flags (F5) of the tiled run, var<0.5 counts.
  variants_2d.npz   data/2D (28 updates): counts per frame, test() on every 3rd point of the demo grid after
                    frames {101, 2801}, flags, per mode.
  variants_syn.npz  synthetic 640x480, F = 5 frames: counts per frame, test() on a 32^3 sample of the 256^3 grid.
  fixtures_gp.npz   F1 kernel matrices (3-D and 2-D, mixed gradient flags, train + cross), F2 two ObsGP tiles of
                    data/3D frame 1 (one full, one sparse) with 20 queries, F3 three 3-D clusters (small / medium /
                    large) of frame 1 and one 2-D cluster with 50 predictions each -- L, alpha, predictions per mode.

The reference itself cannot be built in this image (Eigen absent): these pin the ORACLE and its order variants,
not the reference ("parity unpinned").  Run:  python tests/golden/make_variants.py   (about 6 minutes on 8 cores)."""
import ctypes as C
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib  # noqa: E402
import replay  # noqa: E402

MODES = ("tiled", "natural", "fp64acc")
FR3 = (0, 2, 9, 39)
FR2 = (0, 27)
oracle_lib.build()
L_ = oracle_lib.lib()
_p = oracle_lib._p


def syn_sample(m=32, g=256):
    idx = np.linspace(0, g - 1, m).round().astype(np.int64)
    xs = np.linspace(-0.60, 0.60, g)[idx]; ys = np.linspace(-0.45, 0.45, g)[idx]; zs = np.linspace(0.85, 1.15, g)[idx]
    Z, Y, X = np.meshgrid(zs, ys, xs, indexing="ij")
    return np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1).astype(np.float32)


def run_3d():
    frames = replay.load_bigbird(); grid = replay.demo3_grid()
    out = {}
    for mode in MODES:
        oracle_lib.set_arith_mode(mode)
        om = oracle_lib.OracleMap3(frames[0]["cam"])
        counts = []
        for i, fr in enumerate(frames):
            if i:
                om.set_camera(fr["cam"])
            om.update(fr["depth"], fr["pose"])
            counts.append(om.num_points())
            if i in FR3:
                r = om.test(grid)
                out["%s_res_%d" % (mode, i + 1)] = r
                out["%s_varlt_%d" % (mode, i + 1)] = np.int64((r[:, 4] < 0.5).sum())
                if mode == "tiled":
                    out["flags_%d" % (i + 1)] = om.test_flags(grid).astype(np.uint8)
        out["%s_counts" % mode] = np.array(counts, dtype=np.int32)
        print("3d", mode, counts, flush=True)
    oracle_lib.set_arith_mode("tiled")
    np.savez_compressed(os.path.join(HERE, "variants_3d.npz"), **out)


def run_2d():
    frames = replay.load_gazebo(); grid = replay.demo2_grid()[::3]
    out = {"sub_stride": np.int32(3)}
    for mode in MODES:
        oracle_lib.set_arith_mode(mode)
        om = oracle_lib.OracleMap2()
        counts = []
        for i, fr in enumerate(frames):
            om.update(fr["thetas"], fr["ranges"], fr["pose"])
            counts.append(om.nodes().shape[0])
            if i in FR2:
                out["%s_res_%d" % (mode, i)] = om.test(grid)
                if mode == "tiled":
                    out["flags_%d" % i] = om.test_flags(grid).astype(np.uint8)
        out["%s_counts" % mode] = np.array(counts, dtype=np.int32)
        print("2d", mode, counts, flush=True)
    oracle_lib.set_arith_mode("tiled")
    np.savez_compressed(os.path.join(HERE, "variants_2d.npz"), **out)


def run_syn():
    x = syn_sample()
    out = {"x": x}
    for mode in MODES:
        oracle_lib.set_arith_mode(mode)
        om = oracle_lib.OracleMap3()
        counts = []
        for f in range(5):
            om.update(replay.synthetic_depth(f), replay.IDENTITY_POSE)
            counts.append(om.num_points())
            if f == 0:
                out["%s_res_f1" % mode] = om.test(x)     # after ONE frame the three maps still hold the same points
                if mode == "tiled":
                    out["flags_f1"] = om.test_flags(x).astype(np.uint8)
        out["%s_res" % mode] = om.test(x)
        if mode == "tiled":
            out["flags"] = om.test_flags(x).astype(np.uint8)
            st = om.stats()
            out["stats"] = np.array([st["obsgp_tiles"], st["clusters_trained"], st["maxK"]], dtype=np.int64)
        out["%s_counts" % mode] = np.array(counts, dtype=np.int32)
        print("syn", mode, counts, flush=True)
    oracle_lib.set_arith_mode("tiled")
    np.savez_compressed(os.path.join(HERE, "variants_syn.npz"), **out)


def run_fixtures():
    out = {}
    rng = np.random.default_rng(20190520)
    # ---- F1: kernel matrices, hand-made points with mixed gradient flags (mode independent) ----
    for dim, scale in ((3, 0.04), (2, 1.2)):
        n = 6
        x = (rng.normal(0, 0.6 * scale, (n, dim))).astype(np.float32)
        gidx = np.array([0, -1, 1, 2, -1, 3], dtype=np.int32); ng = 4
        sigx = rng.uniform(1e-3, 5e-3, n).astype(np.float32); sigg = rng.uniform(0.01, 0.1, n).astype(np.float32)
        K = n + dim * ng
        Kt = np.zeros(K * K, dtype=np.float32)
        L_.orc_matern32_train(dim, n, _p(x), _p(gidx, C.c_int), ng, C.c_float(scale), _p(sigx), _p(sigg), _p(Kt))
        xq = (x[[0, 3, 5]] + rng.normal(0, 0.3 * scale, (3, dim))).astype(np.float32)
        cross = []
        for q in xq:
            o = np.zeros(K * (1 + dim), dtype=np.float32)
            L_.orc_matern32_cross(dim, n, _p(x), _p(gidx, C.c_int), ng, C.c_float(scale), _p(np.ascontiguousarray(q)), _p(o))
            cross.append(o.reshape(1 + dim, K).T.copy())
        out.update({"f1_%dd_x" % dim: x, "f1_%dd_gidx" % dim: gidx, "f1_%dd_sigx" % dim: sigx, "f1_%dd_sigg" % dim: sigg,
                    "f1_%dd_K" % dim: np.tril(Kt.reshape(K, K).T), "f1_%dd_xq" % dim: xq, "f1_%dd_cross" % dim: np.stack(cross)})
    # ---- F2 / F3 inputs from data/3D frame 1 ----
    frames = replay.load_bigbird()
    oracle_lib.set_arith_mode("tiled")
    om = oracle_lib.OracleMap3(frames[0]["cam"])
    om.update(frames[0]["depth"], frames[0]["pose"])
    ntile = om.obsgp_num_tiles()
    sizes = []
    for t in range(ntile):
        sizes.append(om.obsgp_tile(t)[0])
    sizes = np.array(sizes)
    full = int(np.flatnonzero(sizes == 64)[0]); sparse = int(np.flatnonzero((sizes > 5) & (sizes < 30))[0])
    vu, zinv, _, _ = om.obs()
    tiles_in = {}
    for name, t in (("full", full), ("sparse", sparse)):
        n, x, alpha, Lt = om.obsgp_tile(t)
        x = np.ascontiguousarray(x[:n]).astype(np.float32)
        # training targets are the 1/z of the tile's pixels: recover them by matching (v,u) in the frame grid
        key = {(float(a), float(b)): float(c) for (a, b), c in zip(vu.reshape(-1, 2), zinv.reshape(-1))}
        f = np.array([key[(float(a), float(b))] for a, b in x], dtype=np.float32)
        q = (x[rng.integers(0, n, 20)] + rng.normal(0, 1e-3, (20, 2))).astype(np.float32)
        tiles_in[name] = (x, f, q)
        out["f2_%s_x" % name] = x; out["f2_%s_f" % name] = f; out["f2_%s_q" % name] = q
    nodes = om.nodes()
    cells = np.floor(nodes[:, :3] / 0.05).astype(np.int64)
    uniq, cnt = np.unique(cells, axis=0, return_counts=True)
    clusters = {}
    want = {"small": (8, 60), "medium": (90, 130), "large": (200, 400)}
    for name, (lo, hi) in want.items():
        for c in uniq[np.argsort(cnt)]:
            ctr = (c + 0.5) * 0.05
            sel = np.all(np.abs(nodes[:, :3] - ctr) <= 0.05, axis=1)
            if lo <= sel.sum() <= hi:
                clusters[name] = nodes[sel]
                break
    g2 = replay.load_gazebo()
    om2 = oracle_lib.OracleMap2()
    for i in range(3):
        om2.update(g2[i]["thetas"], g2[i]["ranges"], g2[i]["pose"])
    n2 = om2.nodes()
    ctr2 = n2[n2.shape[0] // 2, :2]
    sel2 = np.all(np.abs(n2[:, :2] - ctr2) <= 3.2, axis=1)
    clusters["2d"] = n2[sel2][:150]
    for name, nd in clusters.items():
        dim = 2 if name == "2d" else 3
        scale = 1.2 if dim == 2 else 0.04
        pos = np.ascontiguousarray(nd[:, :dim]); grad = np.ascontiguousarray(nd[:, dim:2 * dim])
        val = np.ascontiguousarray(nd[:, 2 * dim]); sx = np.ascontiguousarray(nd[:, 2 * dim + 1]); sg = np.ascontiguousarray(nd[:, 2 * dim + 2])
        xq = (pos[rng.integers(0, pos.shape[0], 50)] + rng.normal(0, 0.4 * scale, (50, dim))).astype(np.float32)
        out["f3_%s_nodes" % name] = nd; out["f3_%s_xq" % name] = xq
        clusters[name] = (dim, scale, pos, grad, val, sx, sg, xq)
    # ---- per-mode outputs ----
    for mode in MODES:
        oracle_lib.set_arith_mode(mode)
        for name, (x, f, q) in tiles_in.items():
            n = x.shape[0]
            Lo = np.zeros(n * n, dtype=np.float32); al = np.zeros(n, dtype=np.float32)
            L_.orc_gpou_train(_p(x), _p(f), 2, n, _p(Lo), _p(al))
            v = np.zeros(20, dtype=np.float32); r = np.zeros(20, dtype=np.float32)
            L_.orc_gpou_test(_p(x), _p(f), 2, n, _p(q), 20, _p(v), _p(r))
            out["f2_%s_%s_L" % (name, mode)] = np.tril(Lo.reshape(n, n).T); out["f2_%s_%s_alpha" % (name, mode)] = al
            out["f2_%s_%s_val" % (name, mode)] = v; out["f2_%s_%s_var" % (name, mode)] = r
        for name, (dim, scale, pos, grad, val, sx, sg, xq) in clusters.items():
            o = oracle_lib.ongpis_train(dim, scale, pos, grad, val, sx, sg)
            pr = oracle_lib.ongpis_predict(dim, scale, pos, grad, val, sx, sg, xq)
            out["f3_%s_%s_alpha" % (name, mode)] = o["alpha"]; out["f3_%s_%s_pred" % (name, mode)] = pr
            if mode == "tiled":
                out["f3_%s_K" % name] = np.int64(o["K"]); out["f3_%s_gidx" % name] = o["gidx"]
            print("f3", name, mode, "N", pos.shape[0], "K", o["K"], flush=True)
    oracle_lib.set_arith_mode("tiled")
    np.savez_compressed(os.path.join(HERE, "fixtures_gp.npz"), **out)


if __name__ == "__main__":
    t0 = time.time()
    what = sys.argv[1:] or ["fixtures", "3d", "2d", "syn"]
    if "fixtures" in what: run_fixtures()
    if "3d" in what: run_3d()
    if "2d" in what: run_2d()
    if "syn" in what: run_syn()
    print("done in %.0f s" % (time.time() - t0))
