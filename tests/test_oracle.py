"""CPU tests of the oracle: fixed-order linear algebra vs float64, GP primitives vs the float64
numpy arbiter, the host tree's invariants, and the committed golden vectors."""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

import oracle_lib
import replay

sys.path.insert(0, os.path.join(oracle_lib.ROOT, "oracle"))
import arbiter64  # noqa: E402

from test_gpu_ongpis import make_cluster  # noqa: E402


def test_chol_and_substitutions_against_float64():
    rng = np.random.default_rng(0)
    L_ = oracle_lib.lib()
    for n in (1, 5, 33, 120):
        A = rng.normal(size=(n, n))
        K = (A @ A.T + n * np.eye(n)).astype(np.float32)
        ref = np.linalg.cholesky(K.astype(np.float64))
        flat = K.T.reshape(-1).copy()                 # column-major: element (r, c) at r + c*n
        L_.orc_chol_lower(oracle_lib._p(flat), n, n)
        Lm = np.tril(flat.reshape(n, n).T)
        assert np.abs(Lm - ref).max() < 5e-6 * np.abs(ref).max()
        b = rng.normal(size=n).astype(np.float32)
        x = b.copy()
        L_.orc_fwd_subst(oracle_lib._p(flat), n, n, oracle_lib._p(x), 1, n)
        assert np.abs(x - np.linalg.solve(np.tril(ref), b)).max() < 1e-4 * (1 + np.abs(x).max())
        y = b.copy()
        L_.orc_bwd_subst(oracle_lib._p(flat), n, n, oracle_lib._p(y))
        assert np.abs(y - np.linalg.solve(np.tril(ref).T, b)).max() < 1e-4 * (1 + np.abs(y).max())


def test_blocked_matrix_solve_is_equivalent_to_substitution():
    """The prediction solves L V = k* with the blocked algorithm (inverted 32x32 diagonal blocks, order O6).
    Against an fp64 solve with the same fp32 factor it must be as good as plain fp32 substitution (O1), on
    trained cluster factors of every size class; the two fp32 results agree to fp32 round-off."""
    from scipy.linalg import solve_triangular
    L_ = oracle_lib.lib()
    rng = np.random.default_rng(21)
    for dim, scale, n in ((3, 0.04, 40), (3, 0.04, 130), (3, 0.04, 330), (2, 1.2, 150)):
        pos, grad, val, sx, sg = make_cluster(rng, dim, n, scale)
        tr = oracle_lib.ongpis_train(dim, scale, pos, grad, val, sx, sg)
        K, Lm = tr["K"], tr["L"]
        B = rng.normal(size=(K, 8)).astype(np.float32) * np.exp(-rng.uniform(0, 6, (K, 1))).astype(np.float32)
        flat = np.ascontiguousarray(Lm.T).reshape(-1)          # column-major
        ref = solve_triangular(Lm.astype(np.float64), B.astype(np.float64), lower=True)
        plain = np.ascontiguousarray(B.T).copy(); blocked = plain.copy()
        L_.orc_fwd_subst(oracle_lib._p(flat), K, K, oracle_lib._p(plain), 8, K)
        L_.orc_fwd_subst_blocked(oracle_lib._p(flat), K, K, oracle_lib._p(blocked), 8, K)
        e_plain = np.abs(plain.T - ref).max() / np.abs(ref).max()
        e_blocked = np.abs(blocked.T - ref).max() / np.abs(ref).max()
        print("K=%d: rel err vs fp64  substitution %.2e  blocked %.2e" % (K, e_plain, e_blocked))
        assert e_blocked < 1e-5 and e_blocked < 2 * e_plain + 1e-7
        assert np.abs(blocked - plain).max() / np.abs(ref).max() < 1e-5


def test_gpou_against_arbiter():
    rng = np.random.default_rng(1)
    L_ = oracle_lib.lib()
    for dim, n in ((2, 64), (2, 9), (1, 26)):
        x = (rng.uniform(-0.05, 0.05, (n, dim)) + np.arange(n)[:, None] * 0.0035).astype(np.float32)
        f = (1.0 + 0.05 * rng.normal(size=n)).astype(np.float32)
        xq = x[rng.integers(0, n, 16)] + rng.normal(0, 1e-3, (16, dim)).astype(np.float32)
        val = np.zeros(16, dtype=np.float32); var = np.zeros(16, dtype=np.float32)
        L_.orc_gpou_test(oracle_lib._p(np.ascontiguousarray(x)), oracle_lib._p(f), dim, n,
                         oracle_lib._p(np.ascontiguousarray(xq.astype(np.float32))), 16, oracle_lib._p(val), oracle_lib._p(var))
        Lr, ar = arbiter64.ou_train(x, f)
        mr, vr = arbiter64.ou_test(x, Lr, ar, xq.astype(np.float32))
        # K is ill-conditioned (cond ~1e5): fp32 vs fp64 agree to ~1e-3 relative on the mean
        assert np.abs(val - mr).max() < 5e-3 * np.abs(mr).max()
        assert np.abs(var - vr).max() < 5e-3


@pytest.mark.parametrize("dim,scale,n", [(3, 0.04, 60), (3, 0.04, 150), (2, 1.2, 40)])
def test_ongpis_against_arbiter(dim, scale, n):
    rng = np.random.default_rng(10 + n)
    pos, grad, val, sx, sg = make_cluster(rng, dim, n, scale)
    o = oracle_lib.ongpis_train(dim, scale, pos, grad, val, sx, sg)
    a = arbiter64.ongpis_train(pos, grad, val, sx, sg, scale)
    assert o["K"] == a["K"]
    np.testing.assert_array_equal(o["gidx"], a["gidx"].astype(np.int32))
    # kernel matrix entries (incl. the 2-D sqrt(sigx*sigg) diagonal quirk) to float precision
    ng = int((a["gidx"] >= 0).sum())
    Kc = np.zeros(o["K"] * o["K"], dtype=np.float32)
    sigx = sx.copy(); sigx[a["gidx"] < 0] = 2.0
    oracle_lib.lib().orc_matern32_train(dim, n, oracle_lib._p(np.ascontiguousarray(pos)), oracle_lib._p(o["gidx"], C.c_int), ng,
                                        C.c_float(scale), oracle_lib._p(sigx), oracle_lib._p(sg), oracle_lib._p(Kc))
    Ko = np.tril(Kc.reshape(o["K"], o["K"]).T)
    Ka = np.tril(arbiter64.matern_train_K(pos, a["gidx"], scale, sigx.astype(np.float64), sg.astype(np.float64)))
    assert np.abs(Ko - Ka).max() < 2e-6 * np.abs(Ka).max()
    xq = pos[rng.integers(0, n, 20)] + rng.normal(0, 0.3 * scale, (20, dim)).astype(np.float32)
    got = oracle_lib.ongpis_predict(dim, scale, pos, grad, val, sx, sg, xq)
    nc = 1 + dim
    tos = 3.0 / scale ** 2
    for i in range(20):
        mr, vr = arbiter64.ongpis_test(a, xq[i].astype(np.float32))
        assert np.abs(got[i, 0] - mr[0]) < 2e-5
        assert np.abs(got[i, 1:nc] - mr[1:]).max() < 2e-3 * max(1.0, 1.0 / scale / 25)
        assert np.abs(got[i, nc] - vr[0]) < 1e-4
        assert np.abs(got[i, nc + 1:] - vr[1:]).max() / tos < 1e-4


def test_golden_bigbird_sequence():
    """Oracle replay of the bundled 3-D sequence against the committed golden vectors and the
    known-answer list of SURVEY.md 8(c) (reference sources + Eigen stand-in, survey session)."""
    gold = json.load(open(os.path.join(replay.GOLDEN, "oracle_bigbird.json")))
    gres = np.load(os.path.join(replay.GOLDEN, "oracle_bigbird_res.npz"))
    frames = replay.load_bigbird()
    grid = replay.demo3_grid()
    m = oracle_lib.OracleMap3(frames[0]["cam"])
    nfr = 8
    counts = []
    for i in range(nfr):
        if i:
            m.set_camera(frames[i]["cam"])
        m.update(frames[i]["depth"], frames[i]["pose"])
        counts.append(m.num_points())
        if i in (0, 2):
            res = m.test(grid)
            np.testing.assert_allclose(res[gres["sub"]], gres["res_%d" % (i + 1)], rtol=0, atol=0)
            assert int((res[:, 4] < 0.5).sum()) == gold["var_lt_half"][str(i + 1)]
    assert counts == gold["point_counts"][:nfr]
    survey = [675, 897, 976, 997, 1127, 1492, 1680, 1749, 1807, 1810, 1883, 2006, 2125, 2166, 2198, 2212, 2283, 2393, 2501,
              2554, 2595, 2609, 2752, 3053, 3318, 3390, 3427, 3443, 3467, 3536, 3683, 3661, 3678, 3679, 3692, 3687, 3715,
              3715, 3712, 3715]
    full = gold["point_counts"]
    assert len(full) == 40
    mism = [i for i in range(40) if full[i] != survey[i]]
    assert mism == [18], mism                       # one single-point flip on frame 19 (2500 vs 2501)
    st = m.stats()
    assert gold["frame1"]["obsgp_tiles"] == 154 and gold["frame1"]["clusters"] == 24 and gold["frame1"]["maxK"] == 697


def test_synthetic_frame_shape_matches_survey():
    """synthetic 640x480 frame 1 -> 3072 ObsGP tiles, 473 clusters, K max 936,
    26 023 map points."""
    m = oracle_lib.OracleMap3()
    m.update(replay.synthetic_depth(0), replay.IDENTITY_POSE)
    st = m.stats()
    assert (st["obsgp_tiles"], st["clusters_trained"], st["maxK"], m.num_points()) == (3072, 473, 936, 26023)


def test_test_before_update_returns_false():
    m = oracle_lib.OracleMap3()
    assert m.test(np.zeros((3, 3), dtype=np.float32)) is None
