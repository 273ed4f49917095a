"""K1/K2 parity: device ObsGP vs the CPU oracle on data/3D frames and a 1-D laser scan.
Called through the C-ABI (gpismap_amd.ObsGP -> gpis_obsgp_*)."""
import numpy as np
import pytest

import oracle_lib
import replay

pytestmark = pytest.mark.gpu


def _oracle_after(nframes, dev=None):
    """Replays nframes on the oracle; the device ObsGP (if given) is re-trained on every frame's
    grid so that it keeps the FIRST frame's tile boundaries like the reference (SURVEY B-15)."""
    frames = replay.load_bigbird()
    m = oracle_lib.OracleMap3(frames[0]["cam"])
    for i in range(nframes):
        if i:
            m.set_camera(frames[i]["cam"])
        m.update(frames[i]["depth"], frames[i]["pose"])
        if dev is not None:
            vu, zinv, ni, nj = m.obs()
            dev.train2d(vu, zinv, ni, nj)
    return m


@pytest.mark.parametrize("nframes", [1, 3])
def test_obsgp2d_train_and_query_match_oracle(nframes):
    import gpismap_amd
    g = gpismap_amd.ObsGP()
    om = _oracle_after(nframes, g)
    vu, zinv, ni, nj = om.obs()
    assert g.num_groups() == om.obsgp_num_tiles() == 48 * 64
    ntr = 0
    worst_L = worst_a = 0.0
    exact = total = 0
    for t in range(g.num_groups()):
        n, x, alpha, L = g.group(t)
        on, ox, oalpha, oL = om.obsgp_tile(t)
        assert n == on, (t, n, on)
        if n == 0:
            continue
        ntr += 1
        np.testing.assert_array_equal(x[:n], ox)
        Lg = np.tril(L[:n, :n])
        worst_L = max(worst_L, float(np.abs(Lg - np.tril(oL)).max()))
        worst_a = max(worst_a, float(np.abs(alpha[:n] - oalpha).max() / (np.abs(oalpha).max() + 1e-30)))
        exact += int((Lg == np.tril(oL)).sum()) + int((alpha[:n] == oalpha).sum())
        total += Lg.size + n
    print("tiles trained %d, L max abs diff %.3e, alpha max rel diff %.3e, bit-identical %.6f" % (ntr, worst_L, worst_a, exact / total))
    assert ntr > 50
    # fixed chain order => identical up to a rare 1-ulp difference of exp() between glibc and the device
    assert worst_L < 1e-6
    assert worst_a < 1e-4

    # queries: every valid pixel centre plus jittered copies (some fall outside / on untrained tiles)
    rng = np.random.default_rng(7)
    valid = np.nonzero(zinv > 0)[0]
    q = np.stack([vu[2 * valid], vu[2 * valid + 1]], axis=1)
    q = np.concatenate([q, q + rng.normal(0, 2e-3, q.shape).astype(np.float32),
                        rng.uniform(-0.6, 0.6, (2000, 2)).astype(np.float32)]).astype(np.float32)
    val, var = g.query(q)
    oval, ovar = om.obsgp_query(q)
    hit = ovar < 1e5
    assert np.array_equal(hit, var < 1e5)
    assert hit.sum() > 1000
    dv = np.abs(val[hit] - oval[hit]).max()
    dr = np.abs(var[hit] - ovar[hit]).max()
    same = float(np.mean((val[hit] == oval[hit]) & (var[hit] == ovar[hit])))
    print("queries %d answered %d, max |dval| %.3e max |dvar| %.3e bit-identical %.6f" % (q.shape[0], hit.sum(), dv, dr, same))
    assert dv < 1e-5 and dr < 1e-5
    assert np.all(val[~hit] == 0) and np.all(var[~hit] == np.float32(1e6))


def test_obsgp1d_matches_oracle():
    import gpismap_amd
    z = np.load(replay.GOLDEN + "/gazebo2d_seq.npz")
    theta = z["thetas"].astype(np.float32)
    f = (1.0 / np.sqrt(z["ranges"][0].astype(np.float32))).astype(np.float32)
    g = gpismap_amd.ObsGP()
    g.train1d(theta, f)
    assert g.num_groups() == 14
    sizes = [g.group(i)[0] for i in range(14)]
    assert sizes == [26] * 12 + [22, 15]
    L_ = oracle_lib.lib()
    import ctypes as C
    # oracle per group
    starts = [20 * i for i in range(12)] + [240, 255]
    for gi, (a, n) in enumerate(zip(starts, sizes)):
        x = np.ascontiguousarray(theta[a:a + n]); ff = np.ascontiguousarray(f[a:a + n])
        oL = np.zeros(n * n, dtype=np.float32); oa = np.zeros(n, dtype=np.float32)
        L_.orc_gpou_train(oracle_lib._p(x), oracle_lib._p(ff), 1, n, oracle_lib._p(oL), oracle_lib._p(oa))
        _, gx, ga, gL = g.group(gi)
        np.testing.assert_array_equal(gx[:n, 0], x)
        assert np.abs(np.tril(gL[:n, :n]) - np.tril(oL.reshape(n, n).T)).max() < 1e-6
        assert np.abs(ga[:n] - oa).max() <= 1e-4 * np.abs(oa).max()


def test_obsgp1d_query_kernel_matches_oracle():
    """1-D obsgp_query at kernel level (ObsGP1D::test ObsGP.cpp:145-187): group selection by the strict
    (range[j], range[j+1]) intervals, margin 0.0175 at both ends, untouched val / var = 1e6 elsewhere; per-group
    predictions bit-identical to the oracle's GPou::test."""
    import gpismap_amd
    z = np.load(replay.GOLDEN + "/gazebo2d_seq.npz")
    theta = z["thetas"].astype(np.float32)
    f = (1.0 / np.sqrt(z["ranges"][3].astype(np.float32))).astype(np.float32)
    g = gpismap_amd.ObsGP()
    g.train1d(theta, f)
    n = theta.shape[0]
    starts = [20 * i for i in range(12)] + [240, 255]
    sizes = [26] * 12 + [22, 15]
    rng_edges = [theta[0]] + [theta[20 * i + 23] for i in range(12)] + [theta[258], theta[n - 1]]
    rng = np.random.default_rng(11)
    q = np.concatenate([
        rng.uniform(theta[0] - 0.05, theta[-1] + 0.05, 3000).astype(np.float32),
        np.array(rng_edges, dtype=np.float32),                                  # exactly on a boundary: no group answers
        np.array([theta[0] + np.float32(0.0175), theta[-1] - np.float32(0.0175)], dtype=np.float32),
        theta[::7]]).astype(np.float32)
    val, var = g.query(q, val0=-7.0)
    L_ = oracle_lib.lib()
    liml = np.float32(rng_edges[0]) + np.float32(0.0175)
    limr = np.float32(rng_edges[-1]) - np.float32(0.0175)
    oval = np.full(q.shape, np.float32(-7.0), dtype=np.float32)
    ovar = np.full(q.shape, np.float32(1e6), dtype=np.float32)
    for k, x in enumerate(q):
        if x < liml or x > limr:
            continue
        for j in range(14):
            if x > rng_edges[j] and x < rng_edges[j + 1]:
                a, m = starts[j], sizes[j]
                xs = np.ascontiguousarray(theta[a:a + m]); fs = np.ascontiguousarray(f[a:a + m])
                v = np.zeros(1, dtype=np.float32); r = np.zeros(1, dtype=np.float32)
                xq = np.array([x], dtype=np.float32)
                L_.orc_gpou_test(oracle_lib._p(xs), oracle_lib._p(fs), 1, m, oracle_lib._p(xq), 1, oracle_lib._p(v), oracle_lib._p(r))
                oval[k], ovar[k] = v[0], r[0]
                break
    hit = ovar < 1e5
    assert hit.sum() > 2500 and (~hit).sum() > 20
    assert np.array_equal(hit, var < 1e5)
    assert np.array_equal(val[~hit], oval[~hit]) and np.all(var[~hit] == np.float32(1e6))
    same = float(np.mean((val[hit] == oval[hit]) & (var[hit] == ovar[hit])))
    print("1-D queries %d answered %d bit-identical %.6f" % (q.shape[0], int(hit.sum()), same))
    assert same >= 0.9995
    assert np.abs(val[hit] - oval[hit]).max() < 1e-5 and np.abs(var[hit] - ovar[hit]).max() < 1e-5


def test_query_batches_of_every_size_give_the_same_bits():
    """K2 has three launch shapes: unsorted (below 4096 queries), and sorted by group with one to four workgroups per group
    (by the batch's mean queries per group).  A query's answer must not depend on the batch it travels in."""
    import gpismap_amd
    g = gpismap_amd.ObsGP()
    om = _oracle_after(1, g)
    vu, zinv, ni, nj = om.obs()
    rng = np.random.default_rng(11)
    valid = np.nonzero(zinv > 0)[0]
    c = np.stack([vu[2 * valid], vu[2 * valid + 1]], axis=1)
    reps = 420000 // c.shape[0] + 1
    big = np.concatenate([c + rng.normal(0, 1e-3, c.shape).astype(np.float32) for _ in range(reps)]).astype(np.float32)   # > 128 queries per group on average: 4 workgroups per group
    assert big.shape[0] > 128 * g.num_groups()
    vb, rb = g.query(big)
    assert (rb < 1e5).sum() > 100000
    # the same queries in slices of 3000 (unsorted kernel), 5000 (sorted, one workgroup per group) and 150 000 (two)
    for step in (3000, 5000, 150000):
        idx = rng.permutation(big.shape[0])[: 2 * step]
        for k in range(0, idx.size, step):
            sel = idx[k:k + step]
            v, r = g.query(np.ascontiguousarray(big[sel]))
            assert np.array_equal(v, vb[sel]) and np.array_equal(r, rb[sel]), step
