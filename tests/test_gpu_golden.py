"""GPU vs the COMMITTED goldens (tests/golden/variants_*.npz, fixtures_gp.npz): the HIP path through the C-ABI on the
bundled sequences, compared with the three arithmetic variants of the oracle -- bit-identical to "tiled" (the order the
kernels implement), within the fp32 tolerances of SURVEY.md 8(c) of the independent "natural" and "fp64acc" orders.
Unlike the live-oracle tests this protects against oracle drift and prints the table DESIGN.md section 2 quotes."""
import os

import numpy as np
import pytest

import parity_report as pr
import replay

pytestmark = pytest.mark.gpu
G = replay.GOLDEN


def _check(tag, rg, z, key_fmt, flags, dim, scale, modes, loose_vg=5e-4, two_d=False):
    """variants_*.npz: every arithmetic mode replayed the sequence on its OWN map, so beyond the first frames these
    comparisons mix arithmetic with map divergence (different point positions / noises / single flipped decisions):
    the bars here are the wide ones of round 2 and only document that.  The ARITHMETIC comparison at the survey's bars
    is test_same_map_* below (samemap.npz: same map, same training sets, clusters re-factorised per mode)."""
    rt = z[key_fmt % "tiled"]
    r = pr.compare(rg, rt, flags, dim, scale)
    print(pr.fmt("GPU-tiled %s" % tag, r))
    assert r["identical_rows"] >= 0.9995, r
    assert r["sdf_rmse"] < 1e-5
    for m in modes:
        r = pr.compare(rg, z[key_fmt % m], flags, dim, scale)
        print(pr.fmt("GPU-%s %s" % (m, tag), r))
        if two_d:
            assert r["sdf_rmse"] < 1e-4 and r["sdf_max"] < 1e-3 and r["grad_rmse"] < 1e-4 and r["var_f_abs"] < 2e-4
        else:
            assert r["sdf_rmse"] < pr.TOL["sdf_rmse"] and r["sdf_max"] < pr.TOL["sdf_max"]
            assert r["grad_rmse"] < pr.TOL["grad_rmse"] and r["grad_max"] < pr.TOL["grad_max"]
            assert r["var_f_abs"] < pr.TOL["var_f_abs"] and r["var_g_rel"] < loose_vg


def test_data3d_sequence_vs_committed_goldens():
    import gpismap_amd
    z = np.load(os.path.join(G, "variants_3d.npz"))
    frames = replay.load_bigbird(); grid = replay.demo3_grid()
    gm = gpismap_amd.GPisMap3(frames[0]["cam"])
    for i, fr in enumerate(frames):
        if i:
            gm.set_camera(fr["cam"])
        gm.update(fr["depth"], fr["pose"])
        assert gm.num_points() == int(z["tiled_counts"][i]), i          # map-point counts: exact, all 40 frames
        if (i + 1) in (1, 3, 10, 40):
            rg = gm.test(grid)
            assert int((rg[:, 4] < 0.5).sum()) == int(z["tiled_varlt_%d" % (i + 1)])
            # natural-order replay holds a different map from frame 19 on (one order-dependent point decision)
            modes = ("natural", "fp64acc") if i + 1 <= 10 else ("fp64acc",)
            _check("data/3D frame %d" % (i + 1), rg, z, "%s_res_" + str(i + 1), z["flags_%d" % (i + 1)], 3, 0.04, modes)


def test_data2d_sequence_vs_committed_goldens():
    import gpismap_amd
    z = np.load(os.path.join(G, "variants_2d.npz"))
    frames = replay.load_gazebo(); grid = replay.demo2_grid()[::int(z["sub_stride"])]
    g2 = gpismap_amd.GPisMap()
    for i, fr in enumerate(frames):
        g2.update(fr["thetas"], fr["ranges"], fr["pose"])
        assert g2.nodes().shape[0] == int(z["tiled_counts"][i]), i
        if i in (0, 27):
            _check("data/2D frame %d" % (101 + 100 * i), g2.test(grid), z, "%s_res_" + str(i), z["flags_%d" % i], 2, 1.2,
                   ("natural", "fp64acc"), two_d=True)


def test_synthetic_five_frames_vs_committed_goldens():
    """BASELINE config 4 with F = 5 frames, 32^3 sample of the 256^3 grid."""
    import gpismap_amd
    z = np.load(os.path.join(G, "variants_syn.npz"))
    gm = gpismap_amd.GPisMap3()
    for f in range(5):
        gm.update(replay.synthetic_depth(f), replay.IDENTITY_POSE)
        assert gm.num_points() == int(z["tiled_counts"][f]), f
        if f == 0:      # all three oracle variants still hold the same 26 023 points: arithmetic-only comparison
            _check("synthetic frame 1", gm.test(z["x"]), z, "%s_res_f1", z["flags_f1"], 3, 0.04, ("natural", "fp64acc"))
    # after 5 frames the variant maps differ by 1-3 of 39 k points (order-dependent decisions): compare with tiled only
    _check("synthetic frame 5", gm.test(z["x"]), z, "%s_res", z["flags"], 3, 0.04, ())


def test_f3_clusters_vs_committed_fixtures():
    """F3 through the kernel-level C-ABI: alpha and 50 predictions of the captured clusters, bit-identical to the
    committed tiled values, within tolerance of the other two orders."""
    import gpismap_amd
    from test_gpu_ongpis import soa9
    z = np.load(os.path.join(G, "fixtures_gp.npz"))
    for name in ("small", "medium", "large", "2d"):
        nd = z["f3_%s_nodes" % name]; dim = 2 if name == "2d" else 3; scale = 1.2 if dim == 2 else 0.04
        pos = nd[:, :dim]; grad = nd[:, dim:2 * dim]; val = nd[:, 2 * dim]; sx = nd[:, 2 * dim + 1]; sg = nd[:, 2 * dim + 2]
        st = gpismap_amd.OnGPIS(dim, scale, keep_factor=True)
        n = nd.shape[0]
        models = st.train(soa9(dim, pos, grad, val, sx, sg), np.array([0, n], dtype=np.int32), np.arange(n, dtype=np.int32))
        m = st.model(models[0])
        assert m["K"] == int(z["f3_%s_K" % name])
        assert np.array_equal(m["alpha"], z["f3_%s_tiled_alpha" % name])
        xq = z["f3_%s_xq" % name]
        out = st.eval(xq, np.arange(xq.shape[0], dtype=np.int32), np.full(xq.shape[0], models[0], dtype=np.int32))
        nc = 1 + dim
        got = np.concatenate([out[:, :nc], out[:, 4:4 + nc]], axis=1)
        assert np.array_equal(got, z["f3_%s_tiled_pred" % name])
        for mname in ("natural", "fp64acc"):
            assert pr.within(pr.compare(got, z["f3_%s_%s_pred" % (name, mname)], None, dim, scale))


def _same_map(tag, rg, z, kfmt, kflags, dim, scale, grad_max=None):
    rt = z[kfmt % "tiled"]
    r = pr.compare(rg, rt, z[kflags], dim, scale)
    print(pr.fmt("GPU-tiled %s" % tag, r))
    assert r["identical_rows"] >= 0.9995, r
    for m in ("natural", "fp64acc", "eigen33"):
        r = pr.compare(rg, z[kfmt % m], z[kflags], dim, scale)
        print(pr.fmt("GPU-%s %s (same map)" % (m, tag), r))
        assert pr.within_same_map(r, grad_max), (tag, m, r)


def test_same_map_data3d_vs_committed_orders():
    """Arithmetic only: the HIP path's grids after frames 10 and 40 against the committed results of the natural,
    fp64-accumulate and Eigen-3.3-order oracles re-factorised on the SAME (tiled) map -- every SURVEY 8(c) bar at its
    survey value, incl. gradient variances < 1e-4 of the prior (samemap.npz, generator make_samemap.py)."""
    import gpismap_amd
    z = np.load(os.path.join(G, "samemap.npz"))
    frames = replay.load_bigbird(); grid = replay.demo3_grid()[::2]
    gm = gpismap_amd.GPisMap3(frames[0]["cam"])
    for i, fr in enumerate(frames):
        if i:
            gm.set_camera(fr["cam"])
        gm.update(fr["depth"], fr["pose"])
        if i + 1 in (10, 40):
            _same_map("data/3D frame %d" % (i + 1), gm.test(grid), z, "3d_res_%d_" % (i + 1) + "%s", "3d_flags_%d" % (i + 1), 3, 0.04)


def test_same_map_data2d_and_synthetic_vs_committed_orders():
    import gpismap_amd
    z = np.load(os.path.join(G, "samemap.npz"))
    g2 = gpismap_amd.GPisMap()
    for fr in replay.load_gazebo():
        g2.update(fr["thetas"], fr["ranges"], fr["pose"])
    _same_map("data/2D frame 2801", g2.test(replay.demo2_grid()[::6]), z, "2d_res_27_%s", "2d_flags_27", 2, 1.2)
    gm = gpismap_amd.GPisMap3()
    for f in range(5):
        gm.update(replay.synthetic_depth(f), replay.IDENTITY_POSE)
    # (one query of 8192 shows a gradient difference of 2.2e-3 against the natural order: clusters of up to 2344 rows)
    _same_map("synthetic F=5", gm.test(z["syn_x"]), z, "syn_res_f5_%s", "syn_flags_f5", 3, 0.04, grad_max=4e-3)


def test_f1_kernel_matrices_vs_committed():
    """F1 on the GPU: the build kernel (ongpis_buildK_kernel; the fused training kernel uses the same entry formulas and is
    pinned through L) on the committed hand-made points with mixed gradient flags, 3-D and 2-D (sqrt(sigx sigg) quirk)."""
    import gpismap_amd
    z = np.load(os.path.join(G, "fixtures_gp.npz"))
    for dim, scale in ((3, 0.04), (2, 1.2)):
        st = gpismap_amd.OnGPIS(dim, scale)
        K = st.kernel_matrix(z["f1_%dd_x" % dim], z["f1_%dd_gidx" % dim], z["f1_%dd_sigx" % dim], z["f1_%dd_sigg" % dim])
        ref = z["f1_%dd_K" % dim]
        same = float(np.mean(np.tril(K) == ref))
        print("F1 %d-D kernel matrix: bit-identical entries %.4f, max abs diff %.3e" % (dim, same, float(np.abs(np.tril(K) - ref).max())))
        assert np.array_equal(np.tril(K), ref)


def test_f2_obsgp_tiles_vs_committed():
    """F2 on the GPU: K1 / K2 on data/3D frame 1; the two committed tiles (full 64 / sparse) are located by their points:
    factor and alpha bit-identical to the committed tiled values; the committed queries whose answering tile is that
    tile (the map-level lookup may hand a query in an overlap to the neighbour) bit-identical too, all of them within
    1e-5 of the three other orders."""
    import gpismap_amd
    import oracle_lib
    z = np.load(os.path.join(G, "fixtures_gp.npz"))
    zs = np.load(os.path.join(G, "samemap.npz"))
    frames = replay.load_bigbird()
    om = oracle_lib.OracleMap3(frames[0]["cam"])          # only for the frame's observation grid (inputs of K1)
    om.update(frames[0]["depth"], frames[0]["pose"])
    vu, zinv, ni, nj = om.obs()
    g = gpismap_amd.ObsGP()
    g.train2d(vu, zinv, ni, nj)
    for name in ("full", "sparse"):
        x = z["f2_%s_x" % name]; n = x.shape[0]
        hit = None
        for t in range(g.num_groups()):
            gn, gx, ga, gL = g.group(t)
            if gn == n and np.array_equal(gx[:n], x):
                hit = (ga, gL)
                break
        assert hit is not None, name
        ga, gL = hit
        assert np.array_equal(np.tril(gL[:n, :n]), z["f2_%s_tiled_L" % name])
        assert np.array_equal(ga[:n], z["f2_%s_tiled_alpha" % name])
        # queries: the map-level lookup hands a query to the tile whose (non-overlapping) range holds it -- usually a
        # neighbour of the committed tile, since the 8 x 8 tiles overlap by 3 pixels -- so the committed per-tile values
        # pin K2 only where the lookup picks this very tile; everywhere the answer must equal the oracle's lookup, bit for bit
        q = z["f2_%s_q" % name]
        val, var = g.query(q)
        oval, ovar = om.obsgp_query(q)
        assert np.array_equal(val, oval) and np.array_equal(var, ovar)
        tv, tr = z["f2_%s_tiled_val" % name], z["f2_%s_tiled_var" % name]
        same = (val == tv) & (var == tr)
        print("F2 %s: factor and alpha identical to the committed tile; %d of 20 committed queries are answered by this tile (identical)" % (name, int(same.sum())))
        for m, src in (("natural", z), ("fp64acc", z), ("eigen33", zs)):
            if same.any():
                assert np.abs(val[same] - src["f2_%s_%s_val" % (name, m)][same]).max() < 1e-5
                assert np.abs(var[same] - src["f2_%s_%s_var" % (name, m)][same]).max() < 1e-5
            # the factor itself against the other orders
            assert np.abs(np.tril(gL[:n, :n]) - src["f2_%s_%s_L" % (name, m)]).max() < 1e-5


def test_gpu_sits_inside_the_own_map_noise_floor():
    """end to end (every arithmetic order on its OWN map after the fused frames) two independent CPU
    orders disagree by 1.9e-5 (synthetic F = 5) / 1.0e-4 (data/3D frame 40) in SDF RMSE -- north_star's 1e-5 is below the
    floor of the comparison itself.  What CAN be asserted: the GPU differs from each CPU order by at most 1.25 x what the
    CPU orders differ from each other (the same-map tests below carry the arithmetic comparison at the survey's bars)."""
    import gpismap_amd
    from test_golden_variants import own_map_floor
    zs = np.load(os.path.join(G, "variants_syn.npz"))
    gm = gpismap_amd.GPisMap3()
    for f in range(5):
        gm.update(replay.synthetic_depth(f), replay.IDENTITY_POSE)
    rs = gm.test(zs["x"])
    z3 = np.load(os.path.join(G, "variants_3d.npz"))
    frames = replay.load_bigbird()
    g3 = gpismap_amd.GPisMap3(frames[0]["cam"])
    for i, fr in enumerate(frames):
        if i:
            g3.set_camera(fr["cam"])
        g3.update(fr["depth"], fr["pose"])
    r3 = g3.test(replay.demo3_grid())
    for tag, rg, z, keyfmt, fkey in (("synthetic F = 5", rs, zs, "%s_res", "flags"), ("data/3D frame 40", r3, z3, "%s_res_40", "flags_40")):
        amb = (z[fkey] & 6) != 0
        floor = own_map_floor(z, keyfmt, z[fkey])[("natural", "fp64acc")]
        for other in ("natural", "fp64acc"):
            d = rg[:, 0].astype(np.float64) - z[keyfmt % other][:, 0]
            e_un, e_all = float(np.sqrt(np.mean(d[~amb] ** 2))), float(np.sqrt(np.mean(d ** 2)))
            print("%-18s GPU-%-8s own-map SDF RMSE unmasked %.2e all %.2e | floor natural-fp64acc %.2e / %.2e" % (tag, other, e_un, e_all, floor[0], floor[1]))
            assert e_un <= 1.25 * floor[0] and e_all <= 1.25 * max(floor), (tag, other, e_un, e_all, floor)
