"""Parity tests of the archived experiments (tools/experiments/README.md).  Not collected by the product test run: they need a
library built with `make -C gpismap_amd/csrc EXTRA=-DGPIS_EXPERIMENTS` and select the kernels through the environment
(GPIS_SMALL_KERNEL=1, GPIS_ASYNC_CHOL=1 -- read when a store is created).  Run: pytest tools/experiments/test_experiments.py"""
import os
import sys
import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import oracle_lib
from test_gpu_ongpis import make_cluster, soa9

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dim,scale,sizes", [
    (3, 0.04, [1, 2, 8, 9, 31, 33, 40, 50, 63, 64, 71]),        # K = 4 N: 1 .. 9 block rows (K = 256: the mean row alone in row 8; K = 284)
    (3, 0.04, [5, 17, 40, 64, 90, 150, 256, 280]),               # mixed gradient flags / value-only points up to K = N = 280
    (2, 1.2, [3, 26, 60, 95]),
])
def test_small_cluster_predictor_equals_general_kernel(dim, scale, sizes):
    """Opt-in path: clusters of at most 9 block rows predicted by the resident-X kernel (ongpis_test_small.hip: X in registers across the
    tiles of a cluster, B double-buffered, reduction through LDS).  Same clusters, many queries per cluster (several tiles
    per workgroup, ragged last tiles, cluster changes inside a workgroup): bit-identical to the general kernel and the oracle."""
    import gpismap_amd
    rng = np.random.default_rng(777 + dim + len(sizes))
    frac = 0.0 if sizes[0] == 1 else 0.3
    clusters = [make_cluster(rng, dim, n, scale, frac_nograd=(1.0 if n >= 256 else frac)) for n in sizes]
    pos = np.concatenate([c[0] for c in clusters]); grad = np.concatenate([c[1] for c in clusters])
    val = np.concatenate([c[2] for c in clusters]); sx = np.concatenate([c[3] for c in clusters])
    sg = np.concatenate([c[4] for c in clusters])
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    ids = np.arange(off[-1], dtype=np.int32)
    st = gpismap_amd.OnGPIS(dim, scale)
    models = st.train(soa9(dim, pos, grad, val, sx, sg), off, ids)
    nqs = [3, 8, 9, 64, 77, 130, 5, 200, 16, 1, 41][:len(sizes)]
    xq, jm = [], []
    for i, n in enumerate(sizes):
        xq.append(pos[off[i]:off[i + 1]][rng.integers(0, n, nqs[i])] + rng.normal(0, 0.3 * scale, (nqs[i], dim)))
        jm += [models[i]] * nqs[i]
    xq = np.concatenate(xq).astype(np.float32)
    jq = np.arange(xq.shape[0], dtype=np.int32)
    jm = np.array(jm, dtype=np.int32)
    os.environ["GPIS_SMALL_KERNEL"] = "1"          # (experiment builds read the switches at launch time)
    small = st.eval(xq, jq, jm).copy()
    os.environ["GPIS_SMALL_KERNEL"] = "0"
    general = st.eval(xq, jq, jm)
    assert np.array_equal(small.view(np.uint32), general.view(np.uint32))
    nc = 1 + dim
    o = 0
    for i, n in enumerate(sizes):
        s_ = slice(off[i], off[i + 1])
        ref = oracle_lib.ongpis_predict(dim, scale, pos[s_], grad[s_], val[s_], sx[s_], sg[s_], xq[o:o + nqs[i]])
        got = np.concatenate([small[o:o + nqs[i], :nc], small[o:o + nqs[i], 4:4 + nc]], axis=1)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (n, i)
        o += nqs[i]


def test_async_factorisation_equals_barrier_kernel():
    """Opt-in K3 variant for the one-workgroup clusters of more than 256 rows (ongpis_chol_async_kernel: block rows owned by
    wavefronts, LDS counters instead of a barrier per block column, micro-blocked diagonal factorisation): factor, alpha and
    predictions bit-identical to the barrier kernel and to the oracle, on sizes with full, partial and extra last blocks."""
    import gpismap_amd
    dim, scale = 3, 0.04
    rng = np.random.default_rng(977)
    sizes = [90, 96, 128, 200, 257, 300]          # K ~ 290 ... 1000: 10 ... 32 block rows (incl. K % 32 == 0: N = 96, 128 with all normals)
    clusters = [make_cluster(rng, dim, n, scale, frac_nograd=(0.0 if n in (96, 128) else 0.25)) for n in sizes]
    pos = np.concatenate([c[0] for c in clusters]); grad = np.concatenate([c[1] for c in clusters])
    val = np.concatenate([c[2] for c in clusters]); sx = np.concatenate([c[3] for c in clusters])
    sg = np.concatenate([c[4] for c in clusters])
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    ids = np.arange(off[-1], dtype=np.int32)
    P = soa9(dim, pos, grad, val, sx, sg)
    nq = 17
    xq = np.concatenate([pos[off[i]:off[i + 1]][rng.integers(0, sizes[i], nq)] + rng.normal(0, 0.3 * scale, (nq, dim))
                         for i in range(len(sizes))]).astype(np.float32)
    jq = np.arange(xq.shape[0], dtype=np.int32)
    res = []
    for use_async in (False, True):
        st = gpismap_amd.OnGPIS(dim, scale, keep_factor=True)
        os.environ["GPIS_ASYNC_CHOL"] = "1" if use_async else "0"
        models = st.train(P, off, ids)
        out = st.eval(xq, jq, np.repeat(models, nq).astype(np.int32)).copy()
        fac = [st.model(mm) for mm in models]
        res.append((out, fac))
    assert np.array_equal(res[0][0].view(np.uint32), res[1][0].view(np.uint32))
    for ci, n in enumerate(sizes):
        a, b = res[0][1][ci], res[1][1][ci]
        K = a["K"]
        assert np.array_equal(np.tril(a["L"][:K, :K]).view(np.uint32), np.tril(b["L"][:K, :K]).view(np.uint32)), n
        assert np.array_equal(a["alpha"].view(np.uint32), b["alpha"].view(np.uint32)), n
    sel = ids[off[3]:off[4]]
    o = oracle_lib.ongpis_train(dim, scale, pos[sel], grad[sel], val[sel], sx[sel], sg[sel])
    g = res[1][1][3]
    assert np.array_equal(np.tril(g["L"][:o["K"], :o["K"]]).view(np.uint32), np.tril(o["L"]).view(np.uint32))


@pytest.mark.parametrize("dim,scale,sizes,nograd", [
    (3, 0.04, [41, 48, 50, 56, 57, 63, 64], 0.0),       # K = 4 N: 6 .. 8 block rows, every tile of one type (generated in registers)
    (3, 0.04, [75, 90, 107, 108, 112], 0.6),            # mixed gradient flags: tiles that straddle the type boundaries; 107 / 108: the pair-table limit
    (3, 0.04, [170, 200, 224, 256], 1.0),               # value-only, K = N: the three-pass pair walk (no pair tables)
    (2, 1.2, [60, 70, 85], 0.0),                        # 2-D: K = 3 N
])
def test_register_resident_training_equals_fused_kernel(dim, scale, sizes, nograd):
    """Opt-in fused K3 with the tiles in registers, four wavefronts per cluster, two clusters per CU (ongpis_fused_rp.inc): factor,
    alpha and predictions bit-identical to the shipped fused kernel and the oracle for clusters of 6 .. 8 block rows."""
    import gpismap_amd
    rng = np.random.default_rng(4242 + dim + len(sizes))
    clusters = [make_cluster(rng, dim, n, scale, frac_nograd=nograd) for n in sizes]
    pos = np.concatenate([c[0] for c in clusters]); grad = np.concatenate([c[1] for c in clusters])
    val = np.concatenate([c[2] for c in clusters]); sx = np.concatenate([c[3] for c in clusters])
    sg = np.concatenate([c[4] for c in clusters])
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    ids = np.arange(off[-1], dtype=np.int32)
    P = soa9(dim, pos, grad, val, sx, sg)
    nq = 19
    xq = np.concatenate([pos[off[i]:off[i + 1]][rng.integers(0, sizes[i], nq)] + rng.normal(0, 0.3 * scale, (nq, dim))
                         for i in range(len(sizes))]).astype(np.float32)
    jq = np.arange(xq.shape[0], dtype=np.int32)
    res = []
    for rp in (False, True):
        st = gpismap_amd.OnGPIS(dim, scale, keep_factor=True)
        os.environ["GPIS_FUSED_RP"] = "1" if rp else "0"
        models = st.train(P, off, ids)
        out = st.eval(xq, jq, np.repeat(models, nq).astype(np.int32)).copy()
        res.append((out, [st.model(mm) for mm in models]))
    os.environ["GPIS_FUSED_RP"] = "0"
    assert np.array_equal(res[0][0].view(np.uint32), res[1][0].view(np.uint32))
    for ci, n in enumerate(sizes):
        a, b = res[0][1][ci], res[1][1][ci]
        K = a["K"]
        assert K <= 256 and (K + 31) // 32 >= 6, (n, K)
        assert np.array_equal(np.tril(a["L"][:K, :K]).view(np.uint32), np.tril(b["L"][:K, :K]).view(np.uint32)), n
        assert np.array_equal(a["alpha"].view(np.uint32), b["alpha"].view(np.uint32)), n
    sel = ids[off[1]:off[2]]
    o = oracle_lib.ongpis_train(dim, scale, pos[sel], grad[sel], val[sel], sx[sel], sg[sel])
    g = res[1][1][1]
    assert np.array_equal(np.tril(g["L"][:o["K"], :o["K"]]).view(np.uint32), np.tril(o["L"]).view(np.uint32))
