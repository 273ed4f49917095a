"""Generates samemap.npz: SAME-MAP comparisons of the oracle's arithmetic orders.

variants_*.npz hold full independent replays per arithmetic mode: every mode builds its OWN map, and after a few frames
the maps differ (point positions / noises come from ObsGP results in that arithmetic, single threshold decisions flip),
so their grids mix arithmetic differences with map divergence.  Here the map is built ONCE (tiled mode = the HIP path's
map, bit for bit) and every trained cluster is then re-factorised on its stored training set in the other modes
(oracle retrain_all): what remains is arithmetic only.

  3d_res_{10,40}_{natural,fp64acc,eigen33}   data/3D, every 2nd point of the demo grid
  2d_res_27_{...}                            data/2D frame 2801, every 6th point of the demo grid
  syn_res_f5_{...}                           synthetic 640x480, F = 5, every 4th point of the 32^3 sample
  {3d,2d}_eigen33_counts                     map-point counts of a full replay in the eigen33 mode (its own map)
  f2_* / f3_*_eigen33_*                      the F2 / F3 fixtures of fixtures_gp.npz in the eigen33 mode

Run:  python tests/golden/make_samemap.py   (about 8 minutes on 8 cores)."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib  # noqa: E402
import replay  # noqa: E402

OTHER = ("natural", "fp64acc", "eigen33")
oracle_lib.build()
L_ = oracle_lib.lib()
_p = oracle_lib._p


def syn_sample(m=32, g=256):
    idx = np.linspace(0, g - 1, m).round().astype(np.int64)
    xs = np.linspace(-0.60, 0.60, g)[idx]; ys = np.linspace(-0.45, 0.45, g)[idx]; zs = np.linspace(0.85, 1.15, g)[idx]
    Z, Y, X = np.meshgrid(zs, ys, xs, indexing="ij")
    return np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1).astype(np.float32)


def main():
    out = {}
    # ---- data/3D ----
    frames = replay.load_bigbird(); grid = replay.demo3_grid()[::2]
    om = oracle_lib.OracleMap3(frames[0]["cam"])
    for i, fr in enumerate(frames):
        if i:
            om.set_camera(fr["cam"])
        om.update(fr["depth"], fr["pose"])
        if i + 1 in (10, 40):
            out["3d_res_%d_tiled" % (i + 1)] = om.test(grid)
            out["3d_flags_%d" % (i + 1)] = om.test_flags(grid).astype(np.uint8)
            for m in OTHER:
                om.retrain_all(m)
                out["3d_res_%d_%s" % (i + 1, m)] = om.test(grid)
            om.retrain_all("tiled")
            assert np.array_equal(om.test(grid), out["3d_res_%d_tiled" % (i + 1)])
            print("3d frame", i + 1, "done", flush=True)
    oracle_lib.set_arith_mode("eigen33")
    oe = oracle_lib.OracleMap3(frames[0]["cam"])
    cnt = []
    for i, fr in enumerate(frames):
        if i:
            oe.set_camera(fr["cam"])
        oe.update(fr["depth"], fr["pose"])
        cnt.append(oe.num_points())
    oracle_lib.set_arith_mode("tiled")
    out["3d_eigen33_counts"] = np.array(cnt, dtype=np.int32)
    print("3d eigen33 counts", cnt, flush=True)
    # ---- data/2D ----
    g2 = replay.load_gazebo(); grid2 = replay.demo2_grid()[::6]
    o2 = oracle_lib.OracleMap2()
    for fr in g2:
        o2.update(fr["thetas"], fr["ranges"], fr["pose"])
    out["2d_res_27_tiled"] = o2.test(grid2)
    out["2d_flags_27"] = o2.test_flags(grid2).astype(np.uint8)
    for m in OTHER:
        o2.retrain_all(m)
        out["2d_res_27_%s" % m] = o2.test(grid2)
    oracle_lib.set_arith_mode("eigen33")
    o2e = oracle_lib.OracleMap2()
    cnt = []
    for fr in g2:
        o2e.update(fr["thetas"], fr["ranges"], fr["pose"])
        cnt.append(o2e.nodes().shape[0])
    oracle_lib.set_arith_mode("tiled")
    out["2d_eigen33_counts"] = np.array(cnt, dtype=np.int32)
    print("2d done", cnt, flush=True)
    # ---- synthetic, F = 5 ----
    xs = syn_sample()[::4]
    out["syn_x"] = xs
    os_ = oracle_lib.OracleMap3()
    for f in range(5):
        os_.update(replay.synthetic_depth(f), replay.IDENTITY_POSE)
    out["syn_res_f5_tiled"] = os_.test(xs)
    out["syn_flags_f5"] = os_.test_flags(xs).astype(np.uint8)
    for m in OTHER:
        os_.retrain_all(m)
        out["syn_res_f5_%s" % m] = os_.test(xs)
        print("syn", m, flush=True)
    # ---- F2 / F3 in the eigen33 mode ----
    z = np.load(os.path.join(HERE, "fixtures_gp.npz"))
    oracle_lib.set_arith_mode("eigen33")
    for name in ("full", "sparse"):
        x = z["f2_%s_x" % name]; f = z["f2_%s_f" % name]; q = z["f2_%s_q" % name]; n = x.shape[0]
        Lo = np.zeros(n * n, dtype=np.float32); al = np.zeros(n, dtype=np.float32)
        L_.orc_gpou_train(_p(x), _p(f), 2, n, _p(Lo), _p(al))
        v = np.zeros(20, dtype=np.float32); r = np.zeros(20, dtype=np.float32)
        L_.orc_gpou_test(_p(x), _p(f), 2, n, _p(q), 20, _p(v), _p(r))
        out["f2_%s_eigen33_L" % name] = np.tril(Lo.reshape(n, n).T); out["f2_%s_eigen33_alpha" % name] = al
        out["f2_%s_eigen33_val" % name] = v; out["f2_%s_eigen33_var" % name] = r
    for name in ("small", "medium", "large", "2d"):
        nd = z["f3_%s_nodes" % name]; dim = 2 if name == "2d" else 3; scale = 1.2 if dim == 2 else 0.04
        pos = np.ascontiguousarray(nd[:, :dim]); grad = np.ascontiguousarray(nd[:, dim:2 * dim])
        val = np.ascontiguousarray(nd[:, 2 * dim]); sx = np.ascontiguousarray(nd[:, 2 * dim + 1]); sg = np.ascontiguousarray(nd[:, 2 * dim + 2])
        o = oracle_lib.ongpis_train(dim, scale, pos, grad, val, sx, sg)
        out["f3_%s_eigen33_alpha" % name] = o["alpha"]
        out["f3_%s_eigen33_pred" % name] = oracle_lib.ongpis_predict(dim, scale, pos, grad, val, sx, sg, z["f3_%s_xq" % name])
    oracle_lib.set_arith_mode("tiled")
    np.savez_compressed(os.path.join(HERE, "samemap.npz"), **out)
    print("written", os.path.getsize(os.path.join(HERE, "samemap.npz")), "bytes")


if __name__ == "__main__":
    main()
