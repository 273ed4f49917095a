"""CPU: the committed fixtures (tests/golden/fixtures_gp.npz, variants_*.npz; generator make_variants.py) pin the oracle
against drift, and the three arithmetic variants of the oracle (tiled / natural / fp64acc, oracle/linalg.hpp) are compared
with each other on the bundled sequences under the tolerances of SURVEY.md 8(c) -- the evidence that the results the HIP
path reproduces bit for bit (tiled) do not depend on the summation order beyond fp32 round-off.  PARITY UNPINNED: the
reference itself cannot be built here (Eigen absent), so none of this is a comparison with Eigen."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib
import parity_report as pr
import replay

G = replay.GOLDEN
MODES = ("tiled", "natural", "fp64acc")
OTHER = ("natural", "fp64acc", "eigen33")
SURVEY_COUNTS = [675, 897, 976, 997, 1127, 1492, 1680, 1749, 1807, 1810, 1883, 2006, 2125, 2166, 2198, 2212, 2283, 2393, 2501,
                 2554, 2595, 2609, 2752, 3053, 3318, 3390, 3427, 3443, 3467, 3536, 3683, 3661, 3678, 3679, 3692, 3687, 3715,
                 3715, 3712, 3715]      # reference sources + the survey's naive Eigen stand-in


@pytest.fixture(autouse=True)
def _tiled_mode():
    yield
    oracle_lib.set_arith_mode("tiled")


def test_f1_kernel_matrices():
    """F1: Matern-3/2 train and cross blocks with mixed gradient flags (covFnc.cpp:142-314, :317-450), incl. the 2-D
    sqrt(sigx*sigg) diagonal quirk -- mode independent."""
    z = np.load(os.path.join(G, "fixtures_gp.npz"))
    L_ = oracle_lib.lib(); _p = oracle_lib._p
    for dim, scale in ((3, 0.04), (2, 1.2)):
        x = z["f1_%dd_x" % dim]; gidx = z["f1_%dd_gidx" % dim]; n = x.shape[0]; ng = int((gidx >= 0).sum())
        K = n + dim * ng
        Kt = np.zeros(K * K, dtype=np.float32)
        L_.orc_matern32_train(dim, n, _p(x), _p(gidx, C.c_int), ng, C.c_float(scale), _p(z["f1_%dd_sigx" % dim]),
                              _p(z["f1_%dd_sigg" % dim]), _p(Kt))
        assert np.array_equal(np.tril(Kt.reshape(K, K).T), z["f1_%dd_K" % dim])
        for q, ref in zip(z["f1_%dd_xq" % dim], z["f1_%dd_cross" % dim]):
            o = np.zeros(K * (1 + dim), dtype=np.float32)
            L_.orc_matern32_cross(dim, n, _p(x), _p(gidx, C.c_int), ng, C.c_float(scale), _p(np.ascontiguousarray(q)), _p(o))
            assert np.array_equal(o.reshape(1 + dim, K).T, ref)
        if dim == 2:   # covFnc.cpp:352: d/dx diagonal uses sqrt(sigx * sigg), d/dy uses sigg
            a = np.float32(np.sqrt(3.0) / np.float64(np.float32(1.2)))   # covFnc.cpp:322, scale is a float
            a2 = np.float32(a * a)
            k = int(np.flatnonzero(gidx >= 0)[0]); g = int(gidx[k])
            Kd = z["f1_2d_K"]
            assert Kd[n + g, n + g] == np.float32(np.float64(a2) + np.sqrt(np.float64(np.float32(z["f1_2d_sigx"][k] * z["f1_2d_sigg"][k]))))
            assert Kd[n + ng + g, n + ng + g] == np.float32(a2 + z["f1_2d_sigg"][k])


@pytest.mark.parametrize("mode", MODES)
def test_f2_f3_gp_fixtures(mode):
    """F2 (two ObsGP tiles of data/3D frame 1: full 64 and sparse) and F3 (three 3-D clusters + one 2-D cluster): factor,
    alpha and predictions of every arithmetic variant reproduce the committed values bit for bit."""
    z = np.load(os.path.join(G, "fixtures_gp.npz"))
    L_ = oracle_lib.lib(); _p = oracle_lib._p
    oracle_lib.set_arith_mode(mode)
    for name in ("full", "sparse"):
        x = z["f2_%s_x" % name]; f = z["f2_%s_f" % name]; q = z["f2_%s_q" % name]; n = x.shape[0]
        Lo = np.zeros(n * n, dtype=np.float32); al = np.zeros(n, dtype=np.float32)
        L_.orc_gpou_train(_p(x), _p(f), 2, n, _p(Lo), _p(al))
        v = np.zeros(20, dtype=np.float32); r = np.zeros(20, dtype=np.float32)
        L_.orc_gpou_test(_p(x), _p(f), 2, n, _p(q), 20, _p(v), _p(r))
        assert np.array_equal(np.tril(Lo.reshape(n, n).T), z["f2_%s_%s_L" % (name, mode)])
        assert np.array_equal(al, z["f2_%s_%s_alpha" % (name, mode)])
        assert np.array_equal(v, z["f2_%s_%s_val" % (name, mode)]) and np.array_equal(r, z["f2_%s_%s_var" % (name, mode)])
    for name in ("small", "medium", "large", "2d"):
        nd = z["f3_%s_nodes" % name]; dim = 2 if name == "2d" else 3; scale = 1.2 if dim == 2 else 0.04
        pos = np.ascontiguousarray(nd[:, :dim]); grad = np.ascontiguousarray(nd[:, dim:2 * dim])
        val = np.ascontiguousarray(nd[:, 2 * dim]); sx = np.ascontiguousarray(nd[:, 2 * dim + 1]); sg = np.ascontiguousarray(nd[:, 2 * dim + 2])
        o = oracle_lib.ongpis_train(dim, scale, pos, grad, val, sx, sg)
        assert o["K"] == int(z["f3_%s_K" % name]) and np.array_equal(o["gidx"], z["f3_%s_gidx" % name])
        assert np.array_equal(o["alpha"], z["f3_%s_%s_alpha" % (name, mode)])
        pred = oracle_lib.ongpis_predict(dim, scale, pos, grad, val, sx, sg, z["f3_%s_xq" % name])
        assert np.array_equal(pred, z["f3_%s_%s_pred" % (name, mode)])


def test_f3_variants_agree_within_fp32_tolerances():
    """The three arithmetic orders on the same captured clusters: 50 predictions each, SURVEY 8(c) bars."""
    z = np.load(os.path.join(G, "fixtures_gp.npz"))
    for name in ("small", "medium", "large", "2d"):
        dim = 2 if name == "2d" else 3; scale = 1.2 if dim == 2 else 0.04
        t = z["f3_%s_tiled_pred" % name]
        for m in ("natural", "fp64acc"):
            r = pr.compare(t, z["f3_%s_%s_pred" % (name, m)], None, dim, scale)
            print(pr.fmt("F3 %s tiled-%s" % (name, m), r))
            assert pr.within(r), r


def test_sequence_goldens_are_reproduced_by_the_oracle():
    """F4 (tiled mode): data/3D frames 1-3 and data/2D frame 101 replayed now equal the committed grids bit for bit."""
    z = np.load(os.path.join(G, "variants_3d.npz"))
    frames = replay.load_bigbird(); grid = replay.demo3_grid()
    om = oracle_lib.OracleMap3(frames[0]["cam"])
    for i in range(3):
        if i:
            om.set_camera(frames[i]["cam"])
        om.update(frames[i]["depth"], frames[i]["pose"])
        assert om.num_points() == int(z["tiled_counts"][i])
        if i in (0, 2):
            assert np.array_equal(om.test(grid), z["tiled_res_%d" % (i + 1)])
            assert np.array_equal(om.test_flags(grid).astype(np.uint8), z["flags_%d" % (i + 1)])   # F5 mask
    z2 = np.load(os.path.join(G, "variants_2d.npz"))
    g2 = replay.load_gazebo(); grid2 = replay.demo2_grid()[::int(z2["sub_stride"])]
    o2 = oracle_lib.OracleMap2()
    o2.update(g2[0]["thetas"], g2[0]["ranges"], g2[0]["pose"])
    assert np.array_equal(o2.test(grid2), z2["tiled_res_0"])


def test_point_counts_and_the_frame_19_flip():
    """Map-point counts per frame of the three variants vs the known-answer list of SURVEY.md 8(c).  The survey's list was
    produced with a naive (natural-order) Eigen stand-in: the natural-order oracle reproduces its 2501 at frame 19, the
    tiled and the fp64-accumulate runs both give 2500 -- a single threshold decision that depends on the summation order
    (it is the fp64-accumulate value the HIP path matches).  fp64acc and tiled agree on all 40 frames."""
    z = np.load(os.path.join(G, "variants_3d.npz"))
    t, n, f = z["tiled_counts"], z["natural_counts"], z["fp64acc_counts"]
    s = np.array(SURVEY_COUNTS)
    assert np.array_equal(t, f)
    assert list(np.flatnonzero(t != s) + 1) == [19] and t[18] == 2500
    assert n[18] == 2501 and np.array_equal(n[:25], s[:25])
    assert np.abs(n.astype(int) - t.astype(int)).max() <= 1
    z2 = np.load(os.path.join(G, "variants_2d.npz"))
    assert np.array_equal(z2["tiled_counts"], z2["natural_counts"]) and np.array_equal(z2["tiled_counts"], z2["fp64acc_counts"])
    zs = np.load(os.path.join(G, "variants_syn.npz"))
    for m in ("natural", "fp64acc"):     # 39 k points, 5 frames: a handful of order-dependent decisions
        assert np.abs(zs[m + "_counts"].astype(int) - zs["tiled_counts"].astype(int)).max() <= 3
    assert list(zs["tiled_counts"][:2]) == [26023, 30012]
    # SURVEY 8(c) quoted var<0.5 counts 3122 / 3745 / 6861 (frames 1, 3, 40) from the stand-in; all three variants here
    # give 3123 / 3741 / 6860: differences of 1-4 queries of 15 225 sitting on the 0.5 gate (F5 mask), order-dependent.
    for m in MODES:
        assert [int(z["%s_varlt_%d" % (m, k)]) for k in (1, 3, 40)] == [3123, 3741, 6860]


@pytest.mark.parametrize("seq", ["3d", "2d", "syn"])
def test_variants_agree_on_the_sequences(seq):
    """Sequence-level order sensitivity: full independent replays in the three arithmetic orders, compared
    on the demo grids.  Bars: SURVEY 8(c); branch-ambiguous queries counted, RMSE over ALL queries.  Documented
    exceedances: the gradient-variance bar (1e-4 of 1875, derived from 12 frames) is passed up to
    3.6e-4 late in the 3-D sequence; 2-D distances are metres-scale (SDF values up to 5 m) and are held to 1e-4 / 1e-3."""
    z = np.load(os.path.join(G, "variants_%s.npz" % seq))
    if seq == "3d":
        items = [("frame %d" % f, "tiled_res_%d" % f, "%s_res_%d" % ("%s", f), "flags_%d" % f) for f in (1, 3, 10, 40)]
        dim, scale = 3, 0.04
    elif seq == "2d":
        items = [("frame %d" % (101 + 100 * f), "tiled_res_%d" % f, "%s_res_%d" % ("%s", f), "flags_%d" % f) for f in (0, 27)]
        dim, scale = 2, 1.2
    else:
        items = [("after 1 frame", "tiled_res_f1", "%s_res_f1", "flags_f1"), ("after 5 frames", "tiled_res", "%s_res", "flags")]
        dim, scale = 3, 0.04
    for tag, kt, kv, kf in items:
        for m in ("natural", "fp64acc"):
            same_map = (np.array_equal(z["tiled_counts"], z[m + "_counts"]) or (seq == "3d" and tag in ("frame 1", "frame 3", "frame 10"))
                        or (seq == "syn" and tag == "after 1 frame"))
            r = pr.compare(z[kt], z[kv % m], z[kf], dim, scale)
            print(pr.fmt("%s %s tiled-%s%s" % (seq, tag, m, "" if same_map else " (maps differ by a point)"), r))
            if not same_map:
                continue          # different map state after an order-dependent point decision: reported, not an arithmetic comparison
            if seq == "2d":
                assert r["sdf_rmse"] < 1e-4 and r["sdf_max"] < 1e-3 and r["grad_rmse"] < 1e-4 and r["var_f_abs"] < 2e-4
            else:
                assert r["sdf_rmse"] < pr.TOL["sdf_rmse"] and r["sdf_max"] < pr.TOL["sdf_max"]
                assert r["grad_rmse"] < pr.TOL["grad_rmse"] and r["grad_max"] < pr.TOL["grad_max"]
                assert r["var_f_abs"] < pr.TOL["var_f_abs"] and r["var_g_rel"] < 5e-4


def test_same_map_goldens_meet_the_survey_bars():
    """samemap.npz (generator make_samemap.py): the map is built ONCE (tiled = the HIP path's map) and every cluster is
    re-factorised on its stored training set in the other arithmetic orders -- natural, fp64acc and eigen33 (Eigen 3.3's
    published orders, a proxy).  What is left is arithmetic, and it meets EVERY bar of SURVEY 8(c) at its survey value,
    including the gradient-variance bar (1e-4) that the different-map replays of variants_*.npz pass by up to 3.6e-4 and
    the 1e-5 SDF bar in 2-D: those exceedances were map divergence, not summation order (tools/var_budget.py)."""
    z = np.load(os.path.join(G, "samemap.npz"))
    for tag, kt, kv, kf, dim, scale in (("data/3D f10", "3d_res_10_tiled", "3d_res_10_%s", "3d_flags_10", 3, 0.04),
                                        ("data/3D f40", "3d_res_40_tiled", "3d_res_40_%s", "3d_flags_40", 3, 0.04),
                                        ("data/2D f2801", "2d_res_27_tiled", "2d_res_27_%s", "2d_flags_27", 2, 1.2),
                                        ("synthetic F=5", "syn_res_f5_tiled", "syn_res_f5_%s", "syn_flags_f5", 3, 0.04)):
        for m in OTHER:
            r = pr.compare(z[kt], z[kv % m], z[kf], dim, scale)
            print(pr.fmt("%s tiled-%s (same map)" % (tag, m), r))
            assert pr.within_same_map(r, grad_max=4e-3 if tag.startswith("synthetic") else None), (tag, m, r)
    # a full replay in the eigen33 order holds the survey's known-answer point counts through frame 25 (like natural)
    s = np.array(SURVEY_COUNTS)
    e = z["3d_eigen33_counts"]
    print("eigen33 replay, frames where the count differs from the survey list:", list(np.flatnonzero(e != s) + 1))
    assert np.array_equal(e[:18], s[:18]) and np.abs(e.astype(int) - s.astype(int)).max() <= 2
    z2 = np.load(os.path.join(G, "variants_2d.npz"))
    assert np.array_equal(z["2d_eigen33_counts"], z2["tiled_counts"])


def test_same_map_goldens_are_reproduced_by_the_oracle():
    """Drift protection for samemap.npz: data/3D frames 1-10 replayed now, clusters re-factorised per mode: bit-identical
    to the committed grids; F2 / F3 in the eigen33 mode likewise."""
    z = np.load(os.path.join(G, "samemap.npz"))
    frames = replay.load_bigbird(); grid = replay.demo3_grid()[::2]
    om = oracle_lib.OracleMap3(frames[0]["cam"])
    for i in range(10):
        if i:
            om.set_camera(frames[i]["cam"])
        om.update(frames[i]["depth"], frames[i]["pose"])
    assert np.array_equal(om.test(grid), z["3d_res_10_tiled"])
    for m in OTHER:
        om.retrain_all(m)
        assert np.array_equal(om.test(grid), z["3d_res_10_%s" % m]), m
    zf = np.load(os.path.join(G, "fixtures_gp.npz"))
    L_ = oracle_lib.lib(); _p = oracle_lib._p
    oracle_lib.set_arith_mode("eigen33")
    for name in ("full", "sparse"):
        x = zf["f2_%s_x" % name]; f = zf["f2_%s_f" % name]; q = zf["f2_%s_q" % name]; n = x.shape[0]
        Lo = np.zeros(n * n, dtype=np.float32); al = np.zeros(n, dtype=np.float32)
        L_.orc_gpou_train(_p(x), _p(f), 2, n, _p(Lo), _p(al))
        v = np.zeros(20, dtype=np.float32); r = np.zeros(20, dtype=np.float32)
        L_.orc_gpou_test(_p(x), _p(f), 2, n, _p(q), 20, _p(v), _p(r))
        assert np.array_equal(np.tril(Lo.reshape(n, n).T), z["f2_%s_eigen33_L" % name]) and np.array_equal(al, z["f2_%s_eigen33_alpha" % name])
        assert np.array_equal(v, z["f2_%s_eigen33_val" % name]) and np.array_equal(r, z["f2_%s_eigen33_var" % name])
        # against the other orders: ObsGP values within 1e-5 (the bar the K2 tests use)
        assert np.abs(v - zf["f2_%s_tiled_val" % name]).max() < 1e-5 and np.abs(r - zf["f2_%s_tiled_var" % name]).max() < 1e-5
    for name in ("small", "medium", "large", "2d"):
        nd = zf["f3_%s_nodes" % name]; dim = 2 if name == "2d" else 3; scale = 1.2 if dim == 2 else 0.04
        pos = np.ascontiguousarray(nd[:, :dim]); grad = np.ascontiguousarray(nd[:, dim:2 * dim])
        val = np.ascontiguousarray(nd[:, 2 * dim]); sx = np.ascontiguousarray(nd[:, 2 * dim + 1]); sg = np.ascontiguousarray(nd[:, 2 * dim + 2])
        pred = oracle_lib.ongpis_predict(dim, scale, pos, grad, val, sx, sg, zf["f3_%s_xq" % name])
        assert np.array_equal(pred, z["f3_%s_eigen33_pred" % name])
        r = pr.compare(zf["f3_%s_tiled_pred" % name], pred, None, dim, scale)
        print(pr.fmt("F3 %s tiled-eigen33" % name, r))
        assert pr.within(r), r


def own_map_floor(z, keyfmt, flags):
    """SDF RMSE (unmasked / all queries) between every pair of arithmetic orders, EACH ON ITS OWN MAP -- the noise floor
    of an end-to-end comparison after F fused frames: every order builds its own map (point positions, normals and noises
    come out of ObsGP in that arithmetic; single threshold decisions flip), so two independent CPU orders already differ by
    this much.  Returns {(a, b): (unmasked, all)}."""
    amb = (flags & 6) != 0
    out = {}
    for i, a in enumerate(MODES):
        for b in MODES[i + 1:]:
            d = z[keyfmt % a][:, 0].astype(np.float64) - z[keyfmt % b][:, 0]
            out[(a, b)] = (float(np.sqrt(np.mean(d[~amb] ** 2))), float(np.sqrt(np.mean(d ** 2))))
    return out


def test_own_map_noise_floor_bounds_the_tiled_order():
    """VERDICT r3 item 1b.  north_star's literal bar (SDF RMSE <= 1e-5 against the reference after the fused frames) cannot
    be met END TO END by ANY pair of fp32 summation orders: the committed own-map replays of two CPU orders that share no
    code with the kernels (natural, fp64acc) disagree with each other by 1.9e-5 (synthetic F = 5) and 1.0e-4 (data/3D frame
    40, where one order-dependent point decision at frame 19 changed the map).  The order the GPU implements (tiled,
    bit-identical to the kernels: test_gpu_golden.py) must sit within 1.25 x that floor of EACH of them."""
    for tag, f, keyfmt, fkey in (("synthetic F = 5", "variants_syn.npz", "%s_res", "flags"), ("data/3D frame 40", "variants_3d.npz", "%s_res_40", "flags_40")):
        z = np.load(os.path.join(G, f))
        fl = own_map_floor(z, keyfmt, z[fkey])
        floor = fl[("natural", "fp64acc")]
        print("%-18s own-map SDF RMSE unmasked / all: natural-fp64acc %.2e / %.2e (the floor) | tiled-natural %.2e / %.2e | tiled-fp64acc %.2e / %.2e"
              % (tag, floor[0], floor[1], fl[("tiled", "natural")][0], fl[("tiled", "natural")][1], fl[("tiled", "fp64acc")][0], fl[("tiled", "fp64acc")][1]))
        assert floor[0] > 1e-5, "two independent CPU orders meet 1e-5 end to end: the own-map bar would be attainable"
        for other in ("natural", "fp64acc"):
            assert fl[("tiled", other)][0] <= 1.25 * floor[0], (tag, other, fl)
            assert fl[("tiled", other)][1] <= 1.25 * max(floor[1], floor[0]), (tag, other, fl)
