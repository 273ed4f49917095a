"""K6/K3/K4 parity: batched OnGPIS train + predict vs the CPU oracle on synthetic clusters of
several sizes (all K4 workgroup classes) in 3-D and 2-D.  Through the C-ABI (gpis_ongpis_*)."""
import numpy as np
import pytest

import oracle_lib

pytestmark = pytest.mark.gpu


def make_cluster(rng, dim, n, scale, frac_nograd=0.2):
    """Points on a wavy surface patch with unit normals and the reference's noise ranges."""
    ext = scale * 2.0
    pos = rng.uniform(-ext, ext, (n, dim)).astype(np.float32)
    if dim == 3:
        pos[:, 2] = (0.2 * ext * np.sin(3 * pos[:, 0] / ext) * np.cos(2 * pos[:, 1] / ext)).astype(np.float32)
        nrm = np.stack([-0.3 * np.cos(3 * pos[:, 0] / ext), 0.2 * np.sin(2 * pos[:, 1] / ext), np.ones(n)], axis=1)
    else:
        pos[:, 1] = (0.2 * ext * np.sin(3 * pos[:, 0] / ext)).astype(np.float32)
        nrm = np.stack([-0.3 * np.cos(3 * pos[:, 0] / ext), np.ones(n)], axis=1)
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    val = np.full(n, -0.2, dtype=np.float32)
    sx = rng.uniform(1e-3, 5e-3, n).astype(np.float32)
    sg = rng.uniform(0.01, 0.1, n).astype(np.float32)
    k = int(frac_nograd * n)
    sg[:k // 2] = 0.5            # too uncertain -> value-only point
    nrm[k // 2:k] = 0.0          # no normal -> value-only point
    return pos, nrm, val, sx, sg


def soa9(dim, pos, grad, val, sx, sg):
    n = val.size
    P = np.zeros((9, n), dtype=np.float32)
    P[0:dim] = pos.T
    P[3:3 + dim] = grad.T
    P[6], P[7], P[8] = val, sx, sg
    return P


@pytest.mark.parametrize("dim,scale,sizes", [
    (3, 0.04, [1, 7, 40, 64, 150, 230]),        # K up to ~800: classes 0 and 1
    (3, 0.04, [420, 700]),                       # K ~1400 / ~2300: classes 2 and 3
    (2, 1.2, [3, 26, 90, 180]),
])
def test_train_and_predict_match_oracle(dim, scale, sizes):
    import gpismap_amd
    rng = np.random.default_rng(100 + dim + len(sizes))
    clusters = [make_cluster(rng, dim, n, scale) for n in sizes]
    pos = np.concatenate([c[0] for c in clusters]); grad = np.concatenate([c[1] for c in clusters])
    val = np.concatenate([c[2] for c in clusters]); sx = np.concatenate([c[3] for c in clusters])
    sg = np.concatenate([c[4] for c in clusters])
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    # training order inside a cluster is the id order given: shuffle to exercise the gather
    ids = np.concatenate([off[i] + rng.permutation(sizes[i]) for i in range(len(sizes))]).astype(np.int32)
    st = gpismap_amd.OnGPIS(dim, scale, keep_factor=True)
    models = st.train(soa9(dim, pos, grad, val, sx, sg), off, ids)
    print("train ms", st.last_ms()[0])
    jq, jm, xq_all, ref_all = [], [], [], []
    for ci, n in enumerate(sizes):
        sel = ids[off[ci]:off[ci + 1]]
        o = oracle_lib.ongpis_train(dim, scale, pos[sel], grad[sel], val[sel], sx[sel], sg[sel])
        g = st.model(models[ci])
        assert (g["N"], g["K"]) == (n, o["K"])
        np.testing.assert_array_equal(g["gidx"], o["gidx"])
        K = o["K"]
        Lg = np.tril(g["L"][:K, :K]); Lo = np.tril(o["L"])
        dL = float(np.abs(Lg - Lo).max())
        da = float(np.abs(g["alpha"] - o["alpha"]).max() / (np.abs(o["alpha"]).max() + 1e-30))
        same = float(np.mean(Lg == Lo))
        print("cluster N=%d K=%d: L max|d| %.3e (bit-identical %.5f), alpha rel %.3e" % (n, K, dL, same, da))
        assert dL < 2e-5 * max(1.0, float(np.abs(Lo).max()))
        assert da < 1e-3
        # padded square must be a valid triangular factor: identity outside K
        ld = g["ld"]
        pad = g["L"][K:ld, :]
        assert np.array_equal(np.tril(pad[:, K:ld]), np.eye(ld - K, dtype=np.float32))
        assert np.all(pad[:, :K] == 0)
        # queries: near the surface, off the surface, far away
        nq = 37
        xq = pos[sel][rng.integers(0, n, nq)] + rng.normal(0, 0.3 * scale, (nq, dim)).astype(np.float32)
        xq[-3:] += 5 * scale
        base = sum(x.shape[0] for x in xq_all)
        xq_all.append(xq.astype(np.float32))
        ref_all.append(oracle_lib.ongpis_predict(dim, scale, pos[sel], grad[sel], val[sel], sx[sel], sg[sel], xq))
        jq.extend(range(base, base + nq)); jm.extend([models[ci]] * nq)
    xq = np.concatenate(xq_all); ref = np.concatenate(ref_all)
    perm = rng.permutation(len(jq))          # unsorted jobs: the store sorts by model
    out = st.eval(xq, np.array(jq)[perm], np.array(jm)[perm])
    got = np.zeros_like(out); got[perm] = out
    nc = 1 + dim
    mean_g, var_g = got[:, :nc], got[:, 4:4 + nc]
    mean_o, var_o = ref[:, :nc], ref[:, nc:]
    tos = 3.0 / (scale * scale)
    d_f = np.abs(mean_g[:, 0] - mean_o[:, 0]).max()
    d_g = np.abs(mean_g[:, 1:] - mean_o[:, 1:]).max()
    d_vf = np.abs(var_g[:, 0] - var_o[:, 0]).max()
    d_vg = (np.abs(var_g[:, 1:] - var_o[:, 1:]) / tos).max()
    same = float(np.mean(np.all(got[:, list(range(nc)) + list(range(4, 4 + nc))] == ref, axis=1)))
    print("predict bit-identical rows %.5f" % same)
    assert same >= 0.999
    print("predict: max|df| %.3e max|dgrad| %.3e max|dvar_f| %.3e max rel dvar_g %.3e (eval ms %.3f)" % (d_f, d_g, d_vf, d_vg, st.last_ms()[1]))
    assert d_f < 2e-5          # SDF value
    assert d_g < 2e-3          # gradients are O(1/scale)
    assert d_vf < 1e-4         # abs < 1e-4
    assert d_vg < 1e-4         # relative to the prior 3/s^2


@pytest.mark.parametrize("n", [32, 33, 64, 65, 128, 129, 256, 257, 512, 513, 768])
def test_size_class_boundaries(n):
    """Clusters whose K = 4 N sits exactly on / just past a size-class boundary of K3 and K4 (K = 128, 256, 512,
    1024, 2048 and the maximum 3072 are multiples of 32: no padding rows in the last block, the y row lives in a block of its
    own) -- factor, alpha and predictions bit-identical to the oracle, including partial query tiles."""
    import gpismap_amd
    dim, scale = 3, 0.04
    rng = np.random.default_rng(1000 + n)
    pos, grad, val, sx, sg = make_cluster(rng, dim, n, scale, frac_nograd=0.0)
    st = gpismap_amd.OnGPIS(dim, scale, keep_factor=True)
    models = st.train(soa9(dim, pos, grad, val, sx, sg), np.array([0, n], dtype=np.int32), np.arange(n, dtype=np.int32))
    o = oracle_lib.ongpis_train(dim, scale, pos, grad, val, sx, sg)
    g = st.model(models[0])
    K = o["K"]
    assert g["K"] == K == 4 * n
    assert np.array_equal(np.tril(g["L"][:K, :K]), np.tril(o["L"]))
    assert np.array_equal(g["alpha"][:K], o["alpha"])
    nq = 21                                       # 2 full tiles + a partial one
    xq = (pos[rng.integers(0, n, nq)] + rng.normal(0, 0.3 * scale, (nq, dim))).astype(np.float32)
    ref = oracle_lib.ongpis_predict(dim, scale, pos, grad, val, sx, sg, xq)
    out = st.eval(xq, np.arange(nq, dtype=np.int32), np.full(nq, models[0], dtype=np.int32))
    got = np.concatenate([out[:, :4], out[:, 4:8]], axis=1)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("dim,scale,n", [(3, 0.04, 8), (3, 0.04, 16), (3, 0.04, 24), (2, 1.2, 32), (2, 1.2, 64), (2, 1.2, 96)])
def test_lone_mean_row_in_the_small_classes_and_in_2d(dim, scale, n):
    """K a multiple of 32 in the one- and two-wavefront classes of K4 (K = 32, 64, 96) and in 2-D (K = 3 N = 96, 192, 288): alpha sits
    alone in the last block row of the stored inverse, the mean is the vector-ALU chain of round 6 and the variance sums deal
    K / 32 block rows -- bit-identical to the oracle, full and partial query tiles."""
    import gpismap_amd
    rng = np.random.default_rng(4000 + 10 * dim + n)
    pos, grad, val, sx, sg = make_cluster(rng, dim, n, scale, frac_nograd=0.0)
    st = gpismap_amd.OnGPIS(dim, scale, keep_factor=True)
    models = st.train(soa9(dim, pos, grad, val, sx, sg), np.array([0, n], dtype=np.int32), np.arange(n, dtype=np.int32))
    g = st.model(models[0])
    assert g["K"] == (1 + dim) * n and g["K"] % 32 == 0
    nq = 19
    xq = (pos[rng.integers(0, n, nq)] + rng.normal(0, 0.3 * scale, (nq, dim))).astype(np.float32)
    ref = oracle_lib.ongpis_predict(dim, scale, pos, grad, val, sx, sg, xq)
    out = st.eval(xq, np.arange(nq, dtype=np.int32), np.full(nq, models[0], dtype=np.int32))
    nc = 1 + dim
    got = np.concatenate([out[:, :nc], out[:, 4:4 + nc]], axis=1)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_predict_without_exp_table_is_identical():
    """Clusters too large for the per-tile exp table in LDS take a path that evaluates exp() per kernel
    entry (gpis_ongpis_set_exp_table(0) forces it): results must not change by a bit."""
    import gpismap_amd
    dim, scale = 3, 0.04
    rng = np.random.default_rng(77)
    sizes = [30, 130, 260]
    clusters = [make_cluster(rng, dim, n, scale) for n in sizes]
    pos = np.concatenate([c[0] for c in clusters]); grad = np.concatenate([c[1] for c in clusters])
    val = np.concatenate([c[2] for c in clusters]); sx = np.concatenate([c[3] for c in clusters])
    sg = np.concatenate([c[4] for c in clusters])
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    ids = np.arange(off[-1], dtype=np.int32)
    st = gpismap_amd.OnGPIS(dim, scale, keep_factor=True)
    models = st.train(soa9(dim, pos, grad, val, sx, sg), off, ids)
    nq = 41
    xq = np.concatenate([pos[off[i]:off[i + 1]][rng.integers(0, sizes[i], nq)] + rng.normal(0, 0.3 * scale, (nq, dim))
                         for i in range(len(sizes))]).astype(np.float32)
    jq = np.arange(xq.shape[0], dtype=np.int32)
    jm = np.repeat(models, nq).astype(np.int32)
    with_table = st.eval(xq, jq, jm).copy()
    st.set_exp_table(False)
    without = st.eval(xq, jq, jm)
    assert np.array_equal(with_table.view(np.uint32), without.view(np.uint32))


def test_large_cluster_beyond_one_row_group():
    """K = 3600 (113 block rows): four row groups of K4's widest class, B chunks regenerated per group; the round-1
    build refused K > 3072 (the reference has no size limit, OnGPIS.cpp:91-149).  Factor, alpha and predictions
    bit-identical to the oracle."""
    import gpismap_amd
    rng = np.random.default_rng(3)
    n = 900
    pos, nrm, val, sx, sg = make_cluster(rng, 3, n, 0.04, frac_nograd=0.0)
    st = gpismap_amd.OnGPIS(3, 0.04, keep_factor=True)
    models = st.train(soa9(3, pos, nrm, val, sx, sg), np.array([0, n], dtype=np.int32), np.arange(n, dtype=np.int32))
    m = st.model(models[0])
    assert m["K"] == 3600
    o = oracle_lib.ongpis_train(3, 0.04, pos, nrm, val, sx, sg)
    assert np.array_equal(np.tril(m["L"][:3600, :3600]).view(np.uint32), np.tril(o["L"]).view(np.uint32))
    assert np.array_equal(m["alpha"].view(np.uint32), o["alpha"].view(np.uint32))
    nq = 19
    xq = (pos[rng.integers(0, n, nq)] + rng.normal(0, 0.012, (nq, 3))).astype(np.float32)
    ref = oracle_lib.ongpis_predict(3, 0.04, pos, nrm, val, sx, sg, xq)
    out = st.eval(xq, np.arange(nq, dtype=np.int32), np.full(nq, models[0], dtype=np.int32))
    got = np.concatenate([out[:, :4], out[:, 4:8]], axis=1)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("dim,scale,sizes", [
    (3, 0.04, [1, 2, 7, 8, 9, 31, 33, 40, 50, 63, 64]),     # K = 4 N (all with normals): 1 .. 8 block rows, K = 256 included
    (3, 0.04, [5, 17, 40, 64, 100, 150, 230, 256]),         # mixed gradient flags; N = 256 value-only points -> K = 256
    (2, 1.2, [3, 26, 60, 85]),
])
def test_fused_training_equals_separate_kernels(dim, scale, sizes):
    """Clusters of at most 256 rows are trained by ONE on-chip kernel (ongpis_fused.hip) that only writes what prediction
    reads.  Same batch through the fused kernel without the factor (the product configuration), through the fused kernel
    with the factor kept, and through the separate gather / build / factorise / invert kernels: factor and alpha
    bit-identical to the oracle, predictions bit-identical across the three."""
    import gpismap_amd
    rng = np.random.default_rng(4242 + dim + len(sizes))
    value_only = sizes[-1] == 256
    clusters = [make_cluster(rng, dim, n, scale, frac_nograd=(1.0 if (value_only and n == 256) else (0.0 if sizes[0] == 1 else 0.3)))
                for n in sizes]
    pos = np.concatenate([c[0] for c in clusters]); grad = np.concatenate([c[1] for c in clusters])
    val = np.concatenate([c[2] for c in clusters]); sx = np.concatenate([c[3] for c in clusters])
    sg = np.concatenate([c[4] for c in clusters])
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    ids = np.concatenate([off[i] + rng.permutation(sizes[i]) for i in range(len(sizes))]).astype(np.int32)
    P = soa9(dim, pos, grad, val, sx, sg)
    nq = 21
    xq = np.concatenate([pos[off[i]:off[i + 1]][rng.integers(0, sizes[i], nq)] + rng.normal(0, 0.3 * scale, (nq, dim))
                         for i in range(len(sizes))]).astype(np.float32)
    jq = np.arange(xq.shape[0], dtype=np.int32)
    outs = []
    for kw in (dict(), dict(keep_factor=True), dict(fused=False)):
        st = gpismap_amd.OnGPIS(dim, scale, **kw)
        models = st.train(P, off, ids)
        outs.append(st.eval(xq, jq, np.repeat(models, nq).astype(np.int32)).copy())
        if kw.get("keep_factor"):
            for ci, n in enumerate(sizes):
                sel = ids[off[ci]:off[ci + 1]]
                o = oracle_lib.ongpis_train(dim, scale, pos[sel], grad[sel], val[sel], sx[sel], sg[sel])
                g = st.model(models[ci])
                K = o["K"]
                assert g["K"] == K
                np.testing.assert_array_equal(g["gidx"], o["gidx"])
                assert np.array_equal(np.tril(g["L"][:K, :K]).view(np.uint32), np.tril(o["L"]).view(np.uint32)), (n, K)
                assert np.array_equal(g["alpha"].view(np.uint32), o["alpha"].view(np.uint32)), (n, K)
                ld = g["ld"]
                pad = g["L"][K:ld, :]
                assert np.array_equal(np.tril(pad[:, K:ld]), np.eye(ld - K, dtype=np.float32)) and np.all(pad[:, :K] == 0)
        elif not kw:
            with pytest.raises(gpismap_amd.GpisError):     # lean models carry no factor
                st.model(models[0])
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
    assert np.array_equal(outs[0].view(np.uint32), outs[2].view(np.uint32))


def test_cooperative_wait_expiry_is_reported():
    """The largest clusters are factorised by several workgroups handing block rows over through device-scope flags.  A
    hand-over that never arrives (injected here) must not hang the device or yield a garbage factor silently: the bounded
    wait expires, the batch is dropped and training reports GPIS_ERR_STATE; the store stays usable afterwards."""
    import gpismap_amd
    rng = np.random.default_rng(99)
    n = 300
    pos, grad, val, sx, sg = make_cluster(rng, 3, n, 0.04, frac_nograd=0.0)      # K = 1200: 38 block rows -> cooperative
    P = soa9(3, pos, grad, val, sx, sg)
    off = np.array([0, n], dtype=np.int32); ids = np.arange(n, dtype=np.int32)
    st = gpismap_amd.OnGPIS(3, 0.04)
    st.set_debug(inject=1, wait_limit_ms=20)
    with pytest.raises(gpismap_amd.GpisError) as e:
        st.train(P, off, ids)
    assert "-3" in str(e.value)                                                   # GPIS_ERR_STATE
    st.set_debug(inject=0, wait_limit_ms=0)
    models = st.train(P, off, ids)
    xq = (pos[:16] + rng.normal(0, 0.01, (16, 3))).astype(np.float32)
    out = st.eval(xq, np.arange(16, dtype=np.int32), np.full(16, models[0], dtype=np.int32))
    ref = oracle_lib.ongpis_predict(3, 0.04, pos, grad, val, sx, sg, xq)
    assert np.array_equal(np.concatenate([out[:, :4], out[:, 4:8]], axis=1).view(np.uint32), ref.view(np.uint32))


def test_cooperative_groups_fit_the_cus_a_masked_stream_leaves_on_one_xcd():
    """A cooperative cluster's workgroups sit on one XCD and must all be resident.  The pipelined update trains on CU-masked
    streams (64 CUs reserved: 24 of an XCD's 32 left); a cluster that the batch leaves room for gets up to four wavefronts per
    block row -- 32 workgroups on ordinary streams, and no more than the mask leaves of an XCD on masked ones (the schedule's cap;
    on the boxes seen the uncapped 32 did become resident all the same).  The factor, alpha and the predictions must not depend on
    how many workgroups / wavefronts per row shared the work: three different splits, the same bits."""
    import gpismap_amd
    rng = np.random.default_rng(77)
    n = 600
    pos, grad, val, sx, sg = make_cluster(rng, 3, n, 0.04, frac_nograd=0.0)      # K = 2400: 75 block rows -> wants 32 workgroups
    P = soa9(3, pos, grad, val, sx, sg)
    off = np.array([0, n], dtype=np.int32); ids = np.arange(n, dtype=np.int32)
    xq = (pos[:64] + rng.normal(0, 0.01, (64, 3))).astype(np.float32)
    outs = []
    for reserve in (0, 64, 128):
        st = gpismap_amd.OnGPIS(3, 0.04)
        st.set_debug(inject=0, wait_limit_ms=500)
        st.set_cu_reserve(reserve)
        models = st.train(P, off, ids)
        m = st.model(models[0])
        out = st.eval(xq, np.arange(64, dtype=np.int32), np.full(64, models[0], dtype=np.int32))
        outs.append((np.tril(m["L"][:m["K"], :m["K"]]).copy(), m["alpha"].copy(), out.copy()))      # (above the diagonal: scratch)
    for o in outs[1:]:
        assert np.array_equal(o[0].view(np.uint32), outs[0][0].view(np.uint32))
        assert np.array_equal(o[1].view(np.uint32), outs[0][1].view(np.uint32))
        assert np.array_equal(o[2].view(np.uint32), outs[0][2].view(np.uint32))


def test_cooperative_factorisation_is_the_same_every_time():
    """The cooperative kernel is a data flow: which wavefront gets how far before it has to wait depends on timing.  The result
    must not -- every element is one fmaf chain in a fixed order whoever computes it.  A mixed batch (clusters of 5 to 20 block rows
    one-workgroup, larger ones cooperative with different numbers of workgroups) trained six times: the same bits every time."""
    import gpismap_amd
    rng = np.random.default_rng(5)
    sizes = [120 + 26 * i for i in range(20)] + [650]
    cl = [make_cluster(rng, 3, n, 0.04) for n in sizes]
    P = soa9(3, *[np.concatenate([c[i] for c in cl]) for i in range(5)])
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    ids = np.arange(off[-1], dtype=np.int32)
    ref = None
    for rep in range(6):
        st = gpismap_amd.OnGPIS(3, 0.04)
        models = st.train(P, off, ids)
        got = []
        for slot in models:
            m = st.model(slot)
            got.append((np.tril(m["L"][:m["K"], :m["K"]]).view(np.uint32).copy(), m["alpha"].view(np.uint32).copy()))
        if ref is None:
            ref = got
        else:
            for a, b in zip(ref, got):
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_cluster_beyond_the_lds_buffer_of_the_back_substitution():
    """K > 9216: the target vector no longer fits the back substitution's LDS buffer, so alpha takes the barrier path through the
    model's global scratch -- the one branch of the training kernels no other test reaches (the oracle needs minutes for such a
    cluster).  Checked against float64 LAPACK on the GPU-built kernel matrix: K alpha = y to fp32 accuracy, alpha within 2e-3 of the
    float64 solution (1e-4 / 3e-4 measured; 2e-5 / 1e-4 at K = 1200, where the oracle comparison is bit-exact)."""
    import gpismap_amd
    rng = np.random.default_rng(3)
    n = 2330
    pos, grad, val, sx, sg = make_cluster(rng, 3, n, 0.11, frac_nograd=0.0)      # the patch of a 300-point cluster scaled to the same density
    P = soa9(3, pos, grad, val, sx, sg)
    off = np.array([0, n], dtype=np.int32); ids = np.arange(n, dtype=np.int32)
    st = gpismap_amd.OnGPIS(3, 0.04)
    models = st.train(P, off, ids)
    m = st.model(models[0])
    assert m["K"] == 4 * n and m["K"] > 9216
    alpha = m["alpha"].astype(np.float64)
    assert np.isfinite(alpha).all()
    Km = st.kernel_matrix(pos, m["gidx"], sx, sg).astype(np.float64)
    Kf = np.tril(Km) + np.tril(Km, -1).T
    del Km
    y = np.concatenate([val.astype(np.float64)] + [grad[:, c].astype(np.float64) for c in range(3)])
    assert np.abs(Kf @ alpha - y).max() < 1e-3 * np.abs(y).max()
    a64 = np.linalg.solve(Kf, y)
    assert np.abs(a64 - alpha).max() < 2e-3 * np.abs(a64).max()


def test_query_on_a_training_point_reproduces_the_references_nan():
    """kf2 divides by r, so a query that coincides with a gradient-bearing training point makes the
    reference's cross-covariance NaN (covFnc.cpp:31-33, no guard) and with it the whole prediction of that query.  The
    kernels reproduce it (the shared-reciprocal division of K4's generation included): NaN where the oracle has NaN, the same
    bits everywhere else."""
    import gpismap_amd
    dim, scale = 3, 0.04
    rng = np.random.default_rng(31)
    n = 90
    pos, grad, val, sx, sg = make_cluster(rng, dim, n, scale, frac_nograd=0.0)
    st = gpismap_amd.OnGPIS(dim, scale)
    models = st.train(soa9(dim, pos, grad, val, sx, sg), np.array([0, n], dtype=np.int32), np.arange(n, dtype=np.int32))
    xq = (pos[rng.integers(0, n, 24)] + rng.normal(0, 0.3 * scale, (24, dim))).astype(np.float32)
    xq[[3, 8, 17]] = pos[[5, 40, 77]]                # three queries ON training points
    out = st.eval(xq, np.arange(24, dtype=np.int32), np.full(24, models[0], dtype=np.int32))
    ref = oracle_lib.ongpis_predict(dim, scale, pos, grad, val, sx, sg, xq)
    got = np.concatenate([out[:, :4], out[:, 4:8]], axis=1)
    assert np.isnan(ref[[3, 8, 17]]).any(axis=1).all() and not np.isnan(ref[[0, 1, 2]]).any()
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert np.array_equal(got[ok].view(np.uint32), ref[ok].view(np.uint32))


def test_ranged_sqrt_and_division_equal_the_ieee_ones():
    """The pivot chains of the factorisations take sqrt and division through the compiler's own correctly rounded sequences
    WITHOUT their operand scaling / classification (csrc/tile_solve.h).  On 2 x 33 million operand pairs -- the documented
    ranges, and operands shaped like the factorisations' incl. zeros of both signs and unit quotients -- the device's IEEE
    sqrtf and `/` give the same bits."""
    import gpismap_amd
    for mode in (0, 1):
        bad_sqrt, bad_div = gpismap_amd.selftest_ranged_arith(seed=20261003 + mode, blocks=2048, per_thread=64, mode=mode)
        assert (bad_sqrt, bad_div) == (0, 0), (mode, bad_sqrt, bad_div)


def test_table_driven_exp_stays_within_an_ulp_of_the_device_librarys():
    """exp_tab.h replaces the device library's double-precision exp in K4's and K2's generation.  On 16 M arguments in [-12, 0]
    (plus the underflow range) no result is more than one ulp from the library's.  The two disagree in the last bit for ~6 % of
    the arguments -- the library's 11-term polynomial is the less accurate of the two: against 80-digit arithmetic the table-driven
    one misses the correctly rounded result for 0.2 % of the arguments (tools/exp_table.py; glibc's, which the oracle uses, 0.1 %)."""
    import gpismap_amd
    n = 1024 * 256 * 64
    far, any_ = gpismap_amd.selftest_ranged_arith(seed=7, blocks=1024, per_thread=64, mode=2)
    assert far == 0
    assert any_ < 0.10 * n, any_ / n


def test_ring_wait_expiry_of_the_predictor_is_reported():
    """K4 hands its B chunks over through LDS counters with BOUNDED waits.  A signal that never arrives (injected: the first
    workgroup of every launch withholds one) must neither hang the queue nor produce plausible numbers: the wait expires,
    the tile's results are NaN, the launch's error word makes the call fail with GPIS_ERR_STATE, every other tile is
    untouched, and the next call is clean."""
    import time
    import gpismap_amd
    dim, scale = 3, 0.04
    rng = np.random.default_rng(4242)
    sizes = [70, 300]                              # K ~ 240 (two wavefronts per workgroup) and K ~ 1000 (eight): two launches
    clusters = [make_cluster(rng, dim, n, scale) for n in sizes]
    pos = np.concatenate([c[0] for c in clusters]); grad = np.concatenate([c[1] for c in clusters])
    val = np.concatenate([c[2] for c in clusters]); sx = np.concatenate([c[3] for c in clusters])
    sg = np.concatenate([c[4] for c in clusters])
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    st = gpismap_amd.OnGPIS(dim, scale)
    models = st.train(soa9(dim, pos, grad, val, sx, sg), off, np.arange(off[-1], dtype=np.int32))
    nq = 24 * 8                                    # 24 tiles per cluster
    xq = np.concatenate([pos[off[i]:off[i + 1]][rng.integers(0, sizes[i], nq)] + rng.normal(0, 0.3 * scale, (nq, dim))
                         for i in range(2)]).astype(np.float32)
    jq = np.arange(2 * nq, dtype=np.int32)
    jm = np.repeat(models, nq).astype(np.int32)
    clean = st.eval(xq, jq, jm).copy()
    assert np.isfinite(clean).all()
    st.set_debug(inject=16)
    t0 = time.time()
    rc, out = st.eval(xq, jq, jm, return_status=True)
    assert time.time() - t0 < 20.0                 # bounded
    assert rc == -3                                # GPIS_ERR_STATE
    bad = np.isnan(out[:, :4]).any(axis=1)
    # the first workgroup of each of the two launches: one tile of 8 queries per cluster, all of its results NaN
    for i in range(2):
        b = bad[i * nq:(i + 1) * nq]
        assert b.sum() == 8, b.sum()
        rows = out[i * nq:(i + 1) * nq][b]
        assert np.isnan(rows[:, :4]).all() and np.isnan(rows[:, 4:8]).all()
    assert np.array_equal(out[~bad].view(np.uint32), clean[~bad].view(np.uint32))
    st.set_debug(inject=0)
    again = st.eval(xq, jq, jm)
    assert np.array_equal(again.view(np.uint32), clean.view(np.uint32))


def test_cluster_the_predictor_cannot_hold_is_refused_at_training():
    """K4 stages a cluster's row table and points in LDS; a cluster of very many value-only points fits the factorisation
    but not that staging.  It is refused when it is TRAINED (GPIS_ERR_LIMIT, nothing factorised) instead of making every
    later test() of the map fail."""
    import gpismap_amd
    rng = np.random.default_rng(5)
    n = 8300                                      # 4 (K + pad) + 16 N bytes > 158 KB with K = N
    pos = rng.uniform(-0.1, 0.1, (n, 3)).astype(np.float32)
    P = soa9(3, pos, np.zeros((n, 3), np.float32), np.full(n, -0.2, np.float32), np.full(n, 2e-3, np.float32), np.full(n, 0.05, np.float32))
    st = gpismap_amd.OnGPIS(3, 0.04)
    with pytest.raises(gpismap_amd.GpisError) as e:
        st.train(P, np.array([0, n], dtype=np.int32), np.arange(n, dtype=np.int32))
    assert "-4" in str(e.value)                   # GPIS_ERR_LIMIT
