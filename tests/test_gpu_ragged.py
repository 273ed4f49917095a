"""Ragged and invalid inputs of GPisMap3::update (cpp/src/GPisMap3.cpp:125-216 preprocData: the range test `r < MAX && r > MIN`
drops NaN, infinities, zero, negative and far pixels; a frame with at most one valid pixel returns silently :212-215; a first
frame whose size does not match the camera is refused :150-153) and of test(): the HIP map against the CPU oracle -- same
points in the same traversal order after every call, bit-identical rows where the oracle answers."""
import numpy as np
import pytest

import oracle_lib
import replay

pytestmark = pytest.mark.gpu


def _same_state(gm, om, grid):
    assert gm.num_points() == om.num_points()
    assert np.array_equal(gm.nodes(), om.nodes())
    a, b = gm.test(grid), om.test(grid)
    if b is None:
        assert a is None
        return
    same = float(np.mean(np.all(a == b, axis=1)))
    assert same >= 0.9995, same


def test_invalid_and_sparse_depth_frames_match_the_oracle():
    import gpismap_amd
    rng = np.random.default_rng(505)
    grid = replay.synthetic_grid(20)
    gm = gpismap_amd.GPisMap3()
    om = oracle_lib.OracleMap3()
    W, H = 640, 480

    def both(depth):
        gm.update(depth, replay.IDENTITY_POSE)
        om.update(depth, replay.IDENTITY_POSE)

    # wrong size on the FIRST frame: refused, the map stays empty (test() is refused too)
    short = replay.synthetic_depth(0)[: W * H - 480]
    both(short)
    assert gm.num_points() == 0 and gm.test(grid) is None
    # one valid pixel only (on the sub-sampled grid): obs_numdata > 1 fails, silent return
    d = np.zeros(W * H, dtype=np.float32)
    d[0] = 1.0
    both(d)
    assert gm.num_points() == 0
    # a frame with NaN / inf / negative / zero / far patches and a band of valid pixels
    d = replay.synthetic_depth(0).copy().reshape(W, H)             # column-major: index = col * 480 + row
    d[:80, :] = np.nan
    d[80:120, :] = np.inf
    d[120:160, :] = -1.0
    d[160:200, :] = 0.0
    d[200:240, :] = 1e6
    d[240:, ::7] = np.nan                                          # ragged holes inside the valid band
    both(d.reshape(-1))
    assert gm.num_points() > 0
    _same_state(gm, om, grid)
    # sparse frame: 3 % of the pixels valid at random
    d2 = replay.synthetic_depth(1).copy()
    d2[rng.random(d2.size) > 0.03] = 0.0
    both(d2)
    _same_state(gm, om, grid)
    # an all-invalid frame between two good ones changes nothing
    n0 = gm.num_points()
    both(np.full(W * H, np.nan, dtype=np.float32))
    assert gm.num_points() == n0
    both(replay.synthetic_depth(2))
    _same_state(gm, om, grid)
    # test(): a single query, and queries with NaN coordinates do not poison their neighbours' rows
    x = grid[:9].copy()
    r0 = gm.test(x)
    x2 = x.copy(); x2[4] = np.nan
    r2 = gm.test(x2)
    keep = [0, 1, 2, 3, 5, 6, 7, 8]
    assert np.array_equal(r0[keep].view(np.uint32), r2[keep].view(np.uint32))


def test_invalid_ranges_in_laser_scans_match_the_oracle():
    """2-D twin (cpp/src/GPisMap.cpp:105-149 preproData): beams with NaN / inf / zero / negative / too-far ranges are dropped one by
    one (the 1-D ObsGP groups then hold fewer than 26 points: ragged groups), a scan with a single valid beam returns silently."""
    import gpismap_amd
    frames = replay.load_gazebo()
    grid = replay.demo2_grid()[::7]
    gm = gpismap_amd.GPisMap()
    om = oracle_lib.OracleMap2()
    rng = np.random.default_rng(77)

    def both(th, ra, pose):
        gm.update(th, ra, pose)
        om.update(th, ra, pose)

    fr = frames[0]
    one = np.full_like(fr["ranges"], np.nan); one[100] = fr["ranges"][100]
    both(fr["thetas"], one, fr["pose"])                       # one valid beam: nothing happens
    assert gm.nodes().shape[0] == om.nodes().shape[0] == 0
    for i in range(4):
        fr = frames[i]
        ra = fr["ranges"].copy()
        bad = rng.random(ra.size)
        ra[bad < 0.10] = np.nan
        ra[(bad >= 0.10) & (bad < 0.15)] = np.inf
        ra[(bad >= 0.15) & (bad < 0.20)] = 0.0
        ra[(bad >= 0.20) & (bad < 0.25)] = -2.0
        ra[(bad >= 0.25) & (bad < 0.30)] = 1e4
        both(fr["thetas"], ra, fr["pose"])
        assert np.array_equal(gm.nodes(), om.nodes()), i
        a, b = gm.test(grid), om.test(grid)
        assert (a is None) == (b is None)
        if b is not None:
            assert float(np.mean(np.all(a == b, axis=1))) >= 0.9995
