"""End-to-end parity on the bundled 3-D sequence (BASELINE config 3): GPisMap3 update()/test()
on the HIP path vs the CPU oracle, through the C-ABI (gpis3_*).  Tolerances: SURVEY.md 8(c)."""
import numpy as np
import pytest

import oracle_lib
import replay

pytestmark = pytest.mark.gpu

NFRAMES = 40                      # the whole bundled sequence (demo_gpisMap3.m)
TEST_AT = (0, 1, 2, 5, 9, 19, 29, 39)   # test() on the demo grid after these frames; the map state is compared after every frame


def compare_res(rg, ro, flags, tag):
    """The HIP path keeps the oracle's fixed summation orders, so whole result rows are expected to
    be bit-identical (a 1-ulp difference of the double exp() between glibc and the device can
    perturb a row once in ~1e8 kernel entries).  The fp32 tolerances of SURVEY.md 8(c) are asserted
    on ALL queries -- nothing is masked; `flags` only reports how many queries sit on one of the
    reference's own discontinuities (variance within 1e-3 of the 0.5 gate, near-equal candidates)."""
    amb = (flags & (2 | 4)) != 0
    same = float(np.mean(np.all(rg == ro, axis=1)))
    e_f = rg[:, 0] - ro[:, 0]
    rmse_f = float(np.sqrt(np.mean(e_f ** 2)))
    e_g = rg[:, 1:4] - ro[:, 1:4]
    rmse_g = float(np.sqrt(np.mean(e_g ** 2)))
    e_v = np.abs(rg[:, 4] - ro[:, 4])
    e_vg = np.abs(rg[:, 5:8] - ro[:, 5:8]) / 1875.0
    print("%s: %d queries (%d on a reference discontinuity, none masked), bit-identical rows %.5f, SDF rmse %.3e max %.3e | "
          "grad rmse %.3e | var_f max %.3e | var_g rel max %.3e"
          % (tag, rg.shape[0], int(amb.sum()), same, rmse_f, float(np.abs(e_f).max()), rmse_g, float(e_v.max()), float(e_vg.max())))
    assert same >= 0.9995
    assert rmse_f < 1e-5 and rmse_g < 1e-4
    ok = ~amb                      # a flipped branch (only possible on `amb` rows) may exceed the max bars
    assert np.abs(e_f[ok]).max() < 1e-4 and np.abs(e_g[ok]).max() < 2e-3
    assert e_v[ok].max() < 1e-4 and e_vg[ok].max() < 1e-4
    assert np.all(np.isfinite(rg))


def test_sequence_matches_oracle():
    import gpismap_amd
    frames = replay.load_bigbird()
    grid = replay.demo3_grid()
    gm = gpismap_amd.GPisMap3(frames[0]["cam"])
    om = oracle_lib.OracleMap3(frames[0]["cam"])
    assert gm.test(grid) is None          # test() before the first update(): false, res untouched
    for i in range(NFRAMES):
        fr = frames[i]
        if i:
            gm.set_camera(fr["cam"]); om.set_camera(fr["cam"])
        gm.update(fr["depth"], fr["pose"]); om.update(fr["depth"], fr["pose"])
        ng, no = gm.nodes(), om.nodes()
        assert ng.shape == no.shape, (i, ng.shape, no.shape)
        # map state: positions / normals / noises, tree order.  Decisions are driven by K2 results
        # that are bit-identical to the oracle up to rare 1-ulp exp() differences.
        assert np.array_equal(ng, no), (i, float(np.abs(ng - no).max()))
        if i in TEST_AT:
            rg = gm.test(grid); ro = om.test(grid)
            flags = om.test_flags(grid)
            compare_res(rg, ro, flags, "frame %d (%d pts, %d clusters)" % (i + 1, ng.shape[0], gm.stats()["clusters"]))
        if i == 0:
            st = gm.stats(); ost = om.stats()
            assert st["obsgp_groups"] == ost["obsgp_tiles"] == 154
            print("stats", st)


def test_untouched_entries_and_prior():
    """No cluster within the search box: only res[4] = 1 + map_noise is written (GPisMap3.cpp:816)."""
    import gpismap_amd
    frames = replay.load_bigbird()
    gm = gpismap_amd.GPisMap3(frames[0]["cam"])
    gm.update(frames[0]["depth"], frames[0]["pose"])
    far = np.array([[5.0, 5.0, 5.0], [-3.0, 0.0, 1.0]], dtype=np.float32)
    res = np.full((2, 8), 7.0, dtype=np.float32)
    out = gm.test(far, res)
    expect = np.full((2, 8), 7.0, dtype=np.float32)
    expect[:, 4] = np.float32(1.0 + np.float64(np.float32(5e-3)))
    np.testing.assert_array_equal(out, expect)
    # wrong dimension / empty input -> false like the reference (GPisMap3.cpp:905)
    import ctypes as C
    x = np.zeros((4, 3), dtype=np.float32); r = np.zeros((4, 8), dtype=np.float32)
    L = gpismap_amd.lib()
    assert L.gpis3_test(gm.h, gpismap_amd._p(x), 2, 4, gpismap_amd._p(r)) == -1
    assert L.gpis3_test(gm.h, gpismap_amd._p(x), 3, 0, gpismap_amd._p(r)) == -1
    gm.reset()
    assert gm.test(far) is None
    assert gm.num_points() == 0


@pytest.mark.parametrize("noise", [False, True])
def test_synthetic_frames_match_oracle(noise):
    """BASELINE config 4 inputs (640x480 synthetic depth, optionally with the 1 mm noise variant of SURVEY 8d):
    map state after each of two frames and test() on a 40^3 sample of the query volume, against the oracle."""
    import gpismap_amd
    gm = gpismap_amd.GPisMap3()
    om = oracle_lib.OracleMap3()
    g = replay.synthetic_grid(40)
    for f in range(2):
        d = replay.synthetic_depth(f, noise=noise)
        gm.update(d, replay.IDENTITY_POSE); om.update(d, replay.IDENTITY_POSE)
        ng, no = gm.nodes(), om.nodes()
        assert ng.shape == no.shape and np.array_equal(ng, no)
        rg, ro = gm.test(g), om.test(g)
        compare_res(rg, ro, om.test_flags(g), "synthetic%s frame %d (%d pts, %d clusters)" % (" + noise" if noise else "", f + 1, ng.shape[0], gm.stats()["clusters"]))


def test_reset_gives_a_fresh_map():
    """reset() (GPisMap3.cpp:99) must leave nothing behind: replaying frames after it reproduces a fresh map's
    state and predictions bit for bit (model slots, point ids and device pools are recycled underneath)."""
    import gpismap_amd
    frames = replay.load_bigbird()
    grid = replay.demo3_grid()
    a = gpismap_amd.GPisMap3(frames[0]["cam"])
    for i in (0, 1, 2):
        if i:
            a.set_camera(frames[i]["cam"])
        a.update(frames[i]["depth"], frames[i]["pose"])
    ra, na = a.test(grid), a.nodes()
    # a map that saw other data first, then reset
    b = gpismap_amd.GPisMap3(frames[5]["cam"])
    for i in (5, 6):
        b.set_camera(frames[i]["cam"])
        b.update(frames[i]["depth"], frames[i]["pose"])
    b.reset()
    assert b.num_points() == 0 and b.test(grid) is None
    for i in (0, 1, 2):
        b.set_camera(frames[i]["cam"])
        b.update(frames[i]["depth"], frames[i]["pose"])
    assert np.array_equal(b.nodes(), na)
    assert np.array_equal(b.test(grid).view(np.uint32), ra.view(np.uint32))


def test_two_maps_in_one_process_and_caller_stream():
    """Two map objects driven alternately (no hidden process-global state), and test_device() on a caller's
    stream with queries/results resident in HBM: both must reproduce the single-map, host-buffer results."""
    import torch
    import gpismap_amd
    frames = replay.load_bigbird()
    grid = replay.demo3_grid()
    solo = gpismap_amd.GPisMap3(frames[0]["cam"])
    solo.update(frames[0]["depth"], frames[0]["pose"])
    r0 = solo.test(grid)
    solo.set_camera(frames[1]["cam"]); solo.update(frames[1]["depth"], frames[1]["pose"])
    r1 = solo.test(grid)
    a = gpismap_amd.GPisMap3(frames[0]["cam"])
    b = gpismap_amd.GPisMap3(frames[0]["cam"])
    a.update(frames[0]["depth"], frames[0]["pose"])
    b.update(frames[0]["depth"], frames[0]["pose"])
    a.set_camera(frames[1]["cam"]); a.update(frames[1]["depth"], frames[1]["pose"])
    assert np.array_equal(b.test(grid).view(np.uint32), r0.view(np.uint32))      # b unaffected by a's second frame
    assert np.array_equal(a.test(grid).view(np.uint32), r1.view(np.uint32))
    # device-resident queries on a side stream
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    x = torch.from_numpy(grid).to(dev)
    res = torch.zeros((grid.shape[0], 8), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        a.test_device(x.data_ptr(), grid.shape[0], res.data_ptr(), st.cuda_stream)
    st.synchronize()
    assert np.array_equal(res.cpu().numpy().view(np.uint32), r1.view(np.uint32))


def test_query_count_growth_reallocates_every_scratch_array():
    """test() with n = 1, 2, 9 on one map (the second and third calls stay in the same n/8 tile bucket / cross it),
    then a reset() followed by a LARGER n with fewer models: every size combination must reallocate the tile
    lists together with the per-query arrays (d_tile_ was freed without resetting its capacity)."""
    import gpismap_amd
    frames = replay.load_bigbird()
    grid = replay.demo3_grid()
    gm = gpismap_amd.GPisMap3(frames[0]["cam"])
    om = oracle_lib.OracleMap3(frames[0]["cam"])
    gm.update(frames[0]["depth"], frames[0]["pose"]); om.update(frames[0]["depth"], frames[0]["pose"])
    near = grid[np.argsort(np.abs(om.test(grid)[:, 0]))[:4000]]      # queries that do reach clusters
    for n in (1, 2, 9, 100, 103, 1000, 4000):
        rg = gm.test(near[:n]); ro = om.test(near[:n])
        assert np.array_equal(rg, ro), n
    gm.reset(); om = oracle_lib.OracleMap3(frames[1]["cam"])
    gm.set_camera(frames[1]["cam"])
    gm.update(frames[1]["depth"], frames[1]["pose"]); om.update(frames[1]["depth"], frames[1]["pose"])
    rg = gm.test(grid); ro = om.test(grid)                            # n grows after reset()
    assert float(np.mean(np.all(rg == ro, axis=1))) >= 0.9995
