"""BASELINE config 5 (stress) at reduced scale: clusters on a 0.05 m lattice sheet, 64 points each,
all with normals (K = 256), batched train (K6+K3) and 64 queries per cluster (K4).  SURVEY.md 8(d).
Full scale is 50 000 clusters; the test uses 1 500 and checks a sample of clusters against the oracle
bit for bit plus size-independent properties on all of them."""
import numpy as np
import pytest

import oracle_lib

pytestmark = pytest.mark.gpu


def make_stress(ncl, rng):
    side = int(np.ceil(np.sqrt(ncl)))
    pts, grads, sx, sg = [], [], [], []
    for c in range(ncl):
        cx, cy = (c % side) * 0.05 + 0.025, (c // side) * 0.05 + 0.025
        # 64 points on a jittered 8x8 grid inside the cell: spacing 6.25 mm keeps the min-distance rule
        gx, gy = np.meshgrid(np.arange(8), np.arange(8), indexing="ij")
        p = np.stack([cx - 0.025 + (gx.ravel() + 0.5) * 0.00625, cy - 0.025 + (gy.ravel() + 0.5) * 0.00625,
                      rng.uniform(-0.02, 0.02, 64)], axis=1)
        p[:, :2] += rng.uniform(-0.001, 0.001, (64, 2))
        n = np.array([0.0, 0.0, 1.0]) + rng.normal(0, 0.05, (64, 3))
        n /= np.linalg.norm(n, axis=1, keepdims=True)
        pts.append(p); grads.append(n)
        sx.append(rng.uniform(1e-3, 5e-3, 64)); sg.append(rng.uniform(0.01, 0.1, 64))
    pos = np.concatenate(pts).astype(np.float32); grad = np.concatenate(grads).astype(np.float32)
    val = np.full(pos.shape[0], -0.2, dtype=np.float32)
    return pos, grad, val, np.concatenate(sx).astype(np.float32), np.concatenate(sg).astype(np.float32)


def test_stress_batch_train_and_predict():
    import gpismap_amd
    from test_gpu_ongpis import soa9
    rng = np.random.default_rng(355)
    ncl = 1500
    pos, grad, val, sx, sg = make_stress(ncl, rng)
    off = (np.arange(ncl + 1) * 64).astype(np.int32)
    ids = np.arange(ncl * 64, dtype=np.int32)
    st = gpismap_amd.OnGPIS(3, 0.04, keep_factor=True)
    models = st.train(soa9(3, pos, grad, val, sx, sg), off, ids)
    tr_ms = st.last_ms()[0]
    # 64 queries per cluster, near its points
    nq = 64
    xq = (pos.reshape(ncl, 64, 3)[:, rng.integers(0, 64, nq), :] + rng.normal(0, 0.005, (ncl, nq, 3))).reshape(-1, 3).astype(np.float32)
    jq = np.arange(ncl * nq, dtype=np.int32)
    jm = np.repeat(models, nq).astype(np.int32)
    out = st.eval(xq, jq, jm)
    ev_ms = st.last_ms()[1]
    print("stress: %d clusters K=256: train %.2f ms (%.1f us/cluster), %d evaluations in %.2f ms" % (ncl, tr_ms, 1e3 * tr_ms / ncl, out.shape[0], ev_ms))
    assert np.all(np.isfinite(out))
    # size-independent properties: variances bounded by their priors, value variance small near data
    assert np.all(out[:, 4] <= 1.001 + 1e-6) and np.all(out[:, 5:8] <= 1875.001 + 1e-3)
    assert np.median(out[:, 4]) < 0.05
    # SDF value close to the stored -0.2 near the points
    assert np.median(np.abs(out[:, 0] + 0.2)) < 0.1
    # oracle, bit for bit, on a sample of clusters
    for c in rng.choice(ncl, 12, replace=False):
        s = slice(c * 64, (c + 1) * 64)
        g = st.model(models[c])
        o = oracle_lib.ongpis_train(3, 0.04, pos[s], grad[s], val[s], sx[s], sg[s])
        assert g["K"] == o["K"] == 256
        assert np.array_equal(np.tril(g["L"][:256, :256]), np.tril(o["L"]))
        assert np.array_equal(g["alpha"], o["alpha"])
        ref = oracle_lib.ongpis_predict(3, 0.04, pos[s], grad[s], val[s], sx[s], sg[s], xq[c * nq:(c + 1) * nq])
        got = out[c * nq:(c + 1) * nq]
        assert np.array_equal(got[:, :4], ref[:, :4]) and np.array_equal(got[:, 4:8], ref[:, 4:8])


def test_stress_full_scale_50k_clusters():
    """BASELINE config 5 at FULL scale on one GPU: 50 000 clusters x 64 points (K = 256), batched training and 64 queries
    per cluster; size-independent properties on everything, a random sample of clusters bit for bit against the oracle."""
    import gpismap_amd
    import replay
    from test_gpu_ongpis import soa9
    rng = np.random.default_rng(355)
    ncl = 50000
    pos, grad, val, sx, sg = replay.stress_clusters(ncl, rng)
    off = (np.arange(ncl + 1) * 64).astype(np.int32)
    ids = np.arange(ncl * 64, dtype=np.int32)
    st = gpismap_amd.OnGPIS(3, 0.04, keep_factor=True)
    models = st.train(soa9(3, pos, grad, val, sx, sg), off, ids)
    tr_ms = st.last_ms()[0]
    nq = 64
    xq = replay.stress_queries(pos, ncl, nq, rng)
    jq = np.arange(ncl * nq, dtype=np.int32)
    jm = np.repeat(models, nq).astype(np.int32)
    out = st.eval(xq, jq, jm)
    ev_ms = st.last_ms()[1]
    K = 256.0
    print("stress FULL: %d clusters K=256: train %.1f ms (%.2f us/cluster, %.1f TFLOP/s of K^3/3+2K^2), %d evaluations in %.1f ms (%.1f TFLOP/s)"
          % (ncl, tr_ms, 1e3 * tr_ms / ncl, ncl * (K ** 3 / 3 + 2 * K * K) / tr_ms / 1e9, out.shape[0], ev_ms,
             ncl * nq * (4 * K * K + 8 * K + 1600) / ev_ms / 1e9))
    assert out.shape == (ncl * nq, 8) and np.all(np.isfinite(out))
    assert np.all(out[:, 4] <= 1.001 + 1e-6) and np.all(out[:, 5:8] <= 1875.001 + 1e-3) and np.all(out[:, 4] > -1e-3)
    assert np.median(out[:, 4]) < 0.05 and np.median(np.abs(out[:, 0] + 0.2)) < 0.1
    for c in rng.choice(ncl, 8, replace=False):
        s = slice(c * 64, (c + 1) * 64)
        g = st.model(models[c])
        o = oracle_lib.ongpis_train(3, 0.04, pos[s], grad[s], val[s], sx[s], sg[s])
        assert g["K"] == o["K"] == 256
        assert np.array_equal(np.tril(g["L"][:256, :256]), np.tril(o["L"])) and np.array_equal(g["alpha"], o["alpha"])
        ref = oracle_lib.ongpis_predict(3, 0.04, pos[s], grad[s], val[s], sx[s], sg[s], xq[c * nq:(c + 1) * nq])
        got = out[c * nq:(c + 1) * nq]
        assert np.array_equal(got[:, :4], ref[:, :4]) and np.array_equal(got[:, 4:8], ref[:, 4:8])
