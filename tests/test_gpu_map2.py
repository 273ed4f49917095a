"""End-to-end parity on the bundled 2-D laser sequence (BASELINE configs 1-2): GPisMap
update()/test() on the HIP path vs the CPU oracle, through the C-ABI (gpis2_*)."""
import numpy as np
import pytest

import oracle_lib
import replay

pytestmark = pytest.mark.gpu


def test_2d_sequence_matches_oracle():
    import gpismap_amd
    frames = replay.load_gazebo()
    grid = replay.demo2_grid()
    gm = gpismap_amd.GPisMap()
    om = oracle_lib.OracleMap2()
    assert gm.test(grid[:10]) is None
    tos = 3.0 / 1.2 ** 2
    for i, fr in enumerate(frames):
        gm.update(fr["thetas"], fr["ranges"], fr["pose"])
        om.update(fr["thetas"], fr["ranges"], fr["pose"])
        ng, no = gm.nodes(), om.nodes()
        assert ng.shape == no.shape, (i, ng.shape, no.shape)
        assert np.array_equal(ng, no), (i, float(np.abs(ng - no).max()))
        if True:      # test() on the demo grid after every frame (the 2-D oracle is fast)
            rg, ro = gm.test(grid), om.test(grid)
            fl = om.test_flags(grid)
            ok = np.ones(grid.shape[0], dtype=bool)       # nothing masked
            same = float(np.mean(np.all(rg == ro, axis=1)))
            assert same >= 0.9995, same
            e_f = rg[ok, 0] - ro[ok, 0]
            e_g = rg[ok, 1:3] - ro[ok, 1:3]
            rmse = float(np.sqrt(np.mean(e_f ** 2)))
            print("bit-identical rows %.5f" % same)
            print("frame %d: %d pts, %d clusters, %d flagged(not masked), SDF rmse %.3e max %.3e grad max %.3e var_f max %.3e var_g rel %.3e"
                  % (fr and i + 1, ng.shape[0], gm.stats()["clusters"], int(((fl & 6) != 0).sum()), rmse, float(np.abs(e_f).max()),
                     float(np.abs(e_g).max()), float(np.abs(rg[ok, 3] - ro[ok, 3]).max()),
                     float(np.abs(rg[ok, 4:6] - ro[ok, 4:6]).max() / tos)))
            assert rmse < 1e-5 and np.abs(e_f).max() < 1e-4
            assert np.abs(e_g).max() < 1e-3
            assert np.abs(rg[ok, 3] - ro[ok, 3]).max() < 1e-4
            assert np.abs(rg[ok, 4:6] - ro[ok, 4:6]).max() / tos < 1e-4
    # known answers of SURVEY.md 8(c) for the final frame
    ro = om.test(grid)
    assert int((ro[:, 3] < 0.4).sum()) == 18720
    rg = gm.test(grid)
    assert abs(int((rg[:, 3] < 0.4).sum()) - 18720) <= 3
    assert abs(float(rg[:, 0].mean()) - (-0.0838)) < 1e-4


def test_2d_reset_gives_a_fresh_map():
    """reset() (GPisMap.cpp:90) leaves nothing behind: replaying frames after it equals a fresh map bit for bit."""
    import gpismap_amd
    frames = replay.load_gazebo()
    grid = replay.demo2_grid()[::7]
    a = gpismap_amd.GPisMap()
    for fr in frames[:4]:
        a.update(fr["thetas"], fr["ranges"], fr["pose"])
    ra, na = a.test(grid), a.nodes()
    b = gpismap_amd.GPisMap()
    for fr in frames[10:13]:
        b.update(fr["thetas"], fr["ranges"], fr["pose"])
    b.reset()
    assert b.test(grid) is None
    for fr in frames[:4]:
        b.update(fr["thetas"], fr["ranges"], fr["pose"])
    assert np.array_equal(b.nodes(), na)
    assert np.array_equal(b.test(grid).view(np.uint32), ra.view(np.uint32))


def test_2d_pipelined_update_equals_synchronous():
    """Round 6: the 2-D update() is pipelined like the 3-D one (returns once its training is enqueued; test(), the next update,
    stats and sync join it).  Same nodes and the same test() bits as the synchronous mode on the bundled sequence, with and
    without a test() between the frames; gpis2_sync reports the joined training's status."""
    import gpismap_amd
    frames = replay.load_gazebo()[:8]
    grid = replay.demo2_grid()[::5]
    ref = gpismap_amd.GPisMap(); ref.set_pipeline(False)
    a = gpismap_amd.GPisMap()                      # default: pipelined
    b = gpismap_amd.GPisMap()
    for i, fr in enumerate(frames):
        ref.update(fr["thetas"], fr["ranges"], fr["pose"])
        a.update(fr["thetas"], fr["ranges"], fr["pose"])
        b.update(fr["thetas"], fr["ranges"], fr["pose"])
        r = ref.test(grid)
        assert np.array_equal(a.test(grid).view(np.uint32), r.view(np.uint32)), i      # test() after every update
    b.sync()                                        # updates back to back, joined once
    assert np.array_equal(b.nodes(), ref.nodes()) and np.array_equal(a.nodes(), ref.nodes())
    assert np.array_equal(b.test(grid).view(np.uint32), ref.test(grid).view(np.uint32))
