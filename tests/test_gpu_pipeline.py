"""Pipelined update() (include/gpismap_amd.h: gpis3_sync / gpis3_set_pipeline): the frame's OnGPIS training is enqueued
and joined by the next update / test.  Map state and test() results must not depend on the mode, and a training failure
(the cooperative kernel's bounded wait, fault injection) must surface at the call that joins."""
import numpy as np
import pytest

import gpismap_amd
import replay

pytestmark = pytest.mark.gpu


def _grid(n=20):
    g = np.linspace(-0.9, 0.9, n, dtype=np.float32)
    return np.stack(np.meshgrid(0.5 * g, 0.5 * g, 1.0 + 0.2 * g, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)


def _run(pipeline, frames=3, test_every_frame=False):
    gm = gpismap_amd.GPisMap3()
    gm.set_pipeline(pipeline)
    X = _grid()
    outs = []
    for f in range(frames):
        gm.update(replay.synthetic_depth(f), replay.IDENTITY_POSE)
        if test_every_frame:
            outs.append(gm.test(X).copy())
    gm.sync()
    outs.append(gm.test(X).copy())
    return gm.num_points(), gm.nodes().copy(), outs


def test_pipelined_update_equals_synchronous():
    n1, nodes1, o1 = _run(True)
    n0, nodes0, o0 = _run(False)
    assert n1 == n0 and np.array_equal(nodes1, nodes0)
    assert np.array_equal(o1[-1], o0[-1])
    assert np.isfinite(o1[-1][:, 0]).all() and np.abs(o1[-1][:, 0]).max() > 0


def test_test_joins_the_training_in_flight():
    # test() right after every update(), no explicit sync: it must see the models of THAT frame
    _, _, o1 = _run(True, test_every_frame=True)
    _, _, o0 = _run(False, test_every_frame=True)
    assert len(o1) == len(o0)
    for a, b in zip(o1, o0):
        assert np.array_equal(a, b)


def test_stats_join_and_report_training_time():
    gm = gpismap_amd.GPisMap3()
    gm.set_profile(True)
    gm.update(replay.synthetic_depth(0), replay.IDENTITY_POSE)
    s = gm.stats()            # joins: the event pair of the batch is complete
    assert s["last_train_ms"] > 0 and s["last_train_jobs"] > 0


def test_device_range_gather_equals_host_walk():
    # K6's range filter on the device against the host walk it replaced: same training sets -> same models -> same bits
    X = _grid()
    outs = []
    for host in (False, True):
        gm = gpismap_amd.GPisMap3()
        gm.set_host_gather(host)
        for f in range(3):
            gm.update(replay.synthetic_depth(f), replay.IDENTITY_POSE)
        outs.append((gm.stats()["clusters_trained"], gm.test(X).copy()))
    assert outs[0][0] == outs[1][0] and outs[0][0] > 0
    assert np.array_equal(outs[0][1], outs[1][1])


def test_lazy_inverse_equals_eager():
    # default: update() trains factors and alpha, the first test() computes the explicit inverses.  Eager mode (the inverse
    # behind every factorisation) must answer with the same bits, also when a test() sits between the updates.
    X = _grid()
    res = {}
    for lazy in (True, False):
        gm = gpismap_amd.GPisMap3()
        gm.set_lazy_inverse(lazy)
        gm.set_profile(True)
        outs = []
        for f in range(4):
            gm.update(replay.synthetic_depth(f), replay.IDENTITY_POSE)
            if f == 1:
                outs.append(gm.test(X).copy())      # inverts what frames 0-1 left; frames 2-3 retrain and go stale again
        gm.prepare_test()
        s = gm.stats()
        outs.append(gm.test(X).copy())
        res[lazy] = (outs, s["last_inverse_jobs"])
    for a, b in zip(res[True][0], res[False][0]):
        assert np.array_equal(a, b)
    assert res[True][1] > 0 and res[False][1] == 0       # the deferred pass had work only in lazy mode


def test_reset_and_destroy_join_the_training_in_flight():
    # a pipelined update() leaves the factorisations running: reset() and the destructor must join them before any memory goes
    gm = gpismap_amd.GPisMap3()
    gm.set_pipeline(True)
    gm.update(replay.synthetic_depth(0), replay.IDENTITY_POSE)
    gm.reset()                                  # training of frame 0 possibly still in flight
    assert gm.num_points() == 0
    gm.update(replay.synthetic_depth(0), replay.IDENTITY_POSE)
    n = gm.num_points()
    assert n > 0
    gm.update(replay.synthetic_depth(1), replay.IDENTITY_POSE)
    del gm                                      # destructor with a batch in flight
    g2 = gpismap_amd.GPisMap3()
    g2.update(replay.synthetic_depth(0), replay.IDENTITY_POSE)
    assert g2.num_points() == n


def test_default_mode_and_cu_reserve(monkeypatch):
    """update() is pipelined by default and its training streams then leave CUs to the ObsGP batches of the next frame
    (GPIS_PIPELINE_RESERVE_CUS, default 64); the synchronous mode gives the reserve back; the environment can turn the
    default off (the unchanged mex gateway has no other switch)."""
    monkeypatch.delenv("GPIS_PIPELINE_UPDATE", raising=False)
    monkeypatch.delenv("GPIS_PIPELINE_RESERVE_CUS", raising=False)
    gm = gpismap_amd.GPisMap3()
    s = gm.stats()
    assert s["pipelined"] == 1 and s["train_cu_reserve"] == 64
    gm.update(replay.synthetic_depth(0), replay.IDENTITY_POSE)
    gm.set_pipeline(False)                       # joins the training in flight, recreates the training streams unmasked
    s = gm.stats()
    assert s["pipelined"] == 0 and s["train_cu_reserve"] == 0
    gm.update(replay.synthetic_depth(1), replay.IDENTITY_POSE)
    gm.set_pipeline(True)
    gm.update(replay.synthetic_depth(2), replay.IDENTITY_POSE)
    ref = _run(False, frames=3)
    assert gm.num_points() == ref[0]
    monkeypatch.setenv("GPIS_PIPELINE_UPDATE", "0")
    g0 = gpismap_amd.GPisMap3()
    assert g0.stats()["pipelined"] == 0
    monkeypatch.setenv("GPIS_PIPELINE_UPDATE", "1")
    monkeypatch.setenv("GPIS_PIPELINE_RESERVE_CUS", "16")
    g1 = gpismap_amd.GPisMap3()
    s = g1.stats()
    assert s["pipelined"] == 1 and s["train_cu_reserve"] == 16


def test_pool_cache_trim_returns_the_chunks_of_destroyed_maps():
    """the chunks of destroyed pools are cached per device for the next map of the process; gpis_pool_cache_trim()
    hands them back to the driver (bytes released > 0 after a map was destroyed, 0 when called again) and a map created
    afterwards works as before."""
    import gc
    import gpismap_amd
    grid = replay.synthetic_grid(16)
    a = gpismap_amd.GPisMap3()
    a.update(replay.synthetic_depth(0), replay.IDENTITY_POSE)
    ref = a.test(grid).copy()
    a.close()
    del a
    gc.collect()
    released = gpismap_amd.pool_cache_trim()
    assert released > 0 and released % (512 << 20) == 0
    assert gpismap_amd.pool_cache_trim() == 0
    b = gpismap_amd.GPisMap3()
    b.update(replay.synthetic_depth(0), replay.IDENTITY_POSE)
    assert np.array_equal(b.test(grid).view(np.uint32), ref.view(np.uint32))
