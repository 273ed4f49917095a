"""Several devices behind ONE map object (gpis3_create_multi / GPIS_DEVICES): update() trains every rank's K^3-balanced
share of the frame's clusters on its device, the packed models travel device to device, test() deals the queries to the
ranks in blocks.  A device may be listed more than once (logical shards on one GPU), which is how this single-GPU box
exercises the whole path; per-query arithmetic does not depend on the cut, so everything must be bit-identical to a
one-device map.  Reference: the fan-out over host threads inside the call, GPisMap3.cpp:759-784 and :904-949."""
import os

import numpy as np
import pytest

import replay

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]])
def test_multi_device_map_equals_single_device_map(devices):
    import gpismap_amd
    frames = replay.load_bigbird()
    grid = replay.demo3_grid()
    one = gpismap_amd.GPisMap3(frames[0]["cam"])
    many = gpismap_amd.GPisMap3(frames[0]["cam"], devices=devices)
    assert one.num_devices() == 1 and many.num_devices() == len(devices)
    assert many.test(grid) is None                      # before the first update()
    for i in range(6):
        fr = frames[i]
        if i:
            one.set_camera(fr["cam"]); many.set_camera(fr["cam"])
        one.update(fr["depth"], fr["pose"]); many.update(fr["depth"], fr["pose"])
        assert np.array_equal(one.nodes(), many.nodes())
        # the model exchange moved the other ranks' records at their own sizes and nothing else
        assert many.stats()["exchange_bytes"] == sum(many.shard_bytes(r) for r in range(1, len(devices)))
        # the host logic (preprocessing, ObsGP, tree replay) ran ONCE for the whole map, on the lead device
        assert many.stats()["host_replays"] == 1 and one.stats()["host_replays"] == 1
        a, b = one.test(grid), many.test(grid)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), i
    many.reset()
    assert many.test(grid) is None
    many.set_camera(frames[0]["cam"])
    many.update(frames[0]["depth"], frames[0]["pose"])
    fresh = gpismap_amd.GPisMap3(frames[0]["cam"])
    fresh.update(frames[0]["depth"], frames[0]["pose"])
    assert np.array_equal(many.test(grid).view(np.uint32), fresh.test(grid).view(np.uint32))


def test_multi_device_query_blocks_and_prefilled_result():
    """More queries than one block per rank (block = 65 536 rows, ragged last block), result rows pre-filled by the
    caller as the mex gateway does: untouched entries survive on every rank."""
    import gpismap_amd
    one = gpismap_amd.GPisMap3()
    many = gpismap_amd.GPisMap3(devices=[0, 0, 0])
    for f in range(2):
        d = replay.synthetic_depth(f)
        one.update(d, replay.IDENTITY_POSE); many.update(d, replay.IDENTITY_POSE)
    x = replay.synthetic_grid(64)[: 3 * 65536 + 1234]
    ra = np.full((x.shape[0], 8), 7.5, dtype=np.float32)
    rb = ra.copy()
    one.test(x, ra); many.test(x, rb)
    assert np.array_equal(ra.view(np.uint32), rb.view(np.uint32))
    assert (ra == 7.5).any() and not (ra == 7.5).all()


def test_gateway_uses_every_listed_device(monkeypatch):
    """GPIS_DEVICES in the environment: the reference's UNCHANGED mex gateway then drives a multi-device map."""
    import mexdrive
    import oracle_lib
    mexdrive.build()
    if not os.path.exists(mexdrive.gateway_path("mexGPisMap3")):
        pytest.skip("gateway object not built (reference tree absent at build time)")
    monkeypatch.setenv("GPIS_DEVICES", "0,0")
    g = mexdrive.Gateway("mexGPisMap3")
    frames = replay.load_bigbird(); seq = replay.demo3_sequence(); grid = replay.demo3_grid()
    X = np.ascontiguousarray(grid.T)
    om = None
    for i in range(2):
        fr = frames[i]
        g.call(0, "setCamera", np.array([[float(seq[i][1])]]), "bigbird")
        g.call(0, "update", np.asfortranarray(fr["depth"].reshape(640, 480).T), fr["pose"].reshape(1, 12))
        if om is None:
            om = oracle_lib.OracleMap3(fr["cam"])
        else:
            om.set_camera(fr["cam"])
        om.update(fr["depth"], fr["pose"])
        res = g.call(1, "test", X)
        ro = om.test(grid)
        assert float(np.mean(np.all(res[0].T == ro, axis=1))) >= 0.9995
        assert np.array_equal(g.call(1, "getAllPoints")[0].T, om.nodes()[:, :3])
    g.call(0, "reset")


def test_factor_records_keep_the_lazy_inverse_across_the_exchange():
    """with the lazy inverse the exchange ships FACTORS (model_pack.hip kind 1, the same bytes) and every
    receiver inverts when it first predicts -- several updates without a test() in between never invert on any rank; forcing
    prediction records gives the same bits."""
    import gpismap_amd
    grid = replay.synthetic_grid(24)
    one = gpismap_amd.GPisMap3()
    fac = gpismap_amd.GPisMap3(devices=[0, 0])                       # default: factor records (lazy inverse)
    xrec = gpismap_amd.GPisMap3(devices=[0, 0]); xrec.set_shard_factors(0)
    for f in range(3):
        d = replay.synthetic_depth(f)
        for m in (one, fac, xrec):
            m.update(d, replay.IDENTITY_POSE)
        sf, sx = fac.stats(), xrec.stats()
        assert sf["exchange_bytes"] == sx["exchange_bytes"] == fac.shard_bytes(1)     # same record sizes either way
        assert sf["deferred_inverses"] > 0 and sx["deferred_inverses"] == 0
        assert sf["last_inverse_jobs"] == 0 or f == 0                  # no inverse pass ran inside the sharded update()
    a, b, c = one.test(grid), fac.test(grid), xrec.test(grid)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.array_equal(a.view(np.uint32), c.view(np.uint32))
    assert np.array_equal(one.nodes(), fac.nodes())
