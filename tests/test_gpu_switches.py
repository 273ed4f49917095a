"""The three environment switches a user of the reference's UNCHANGED gateway can reach -- GPIS_DEVICES, GPIS_PIPELINE_UPDATE,
GPIS_EAGER_INVERSE (INTEGRATION.md "Environment switches") -- in every combination on the reference's own sequence: the map
state after every update() and the rows of every test() must be bit-identical to the default configuration.  The cross-check
paths that used to be environment switches (host gather, kept factors) are setters now and are exercised here too.
Reference: GPisMap3::update / test, cpp/src/GPisMap3.cpp:218-237, :904-949."""
import itertools

import numpy as np
import pytest

import replay

pytestmark = pytest.mark.gpu

NFRAMES = 8


def _run(frames, grid, setup=None):
    import gpismap_amd
    m = gpismap_amd.GPisMap3(frames[0]["cam"])
    if setup:
        setup(m)
    nodes, rows = [], []
    for i in range(NFRAMES):
        fr = frames[i]
        if i:
            m.set_camera(fr["cam"])
        m.update(fr["depth"], fr["pose"])
        if i % 2 == 1 or i == NFRAMES - 1:      # several updates per test(): the lazy inverse skips intermediate factors
            rows.append(m.test(grid).view(np.uint32).copy())
            nodes.append(m.nodes().copy())
    return m, nodes, rows


@pytest.fixture(scope="module")
def reference_rows():
    frames = replay.load_bigbird()
    grid = replay.demo3_grid()
    _, nodes, rows = _run(frames, grid)
    return frames, grid, nodes, rows


@pytest.mark.parametrize("devices,pipeline,eager", list(itertools.product(["", "0,0"], ["0", "1"], ["0", "1"])))
def test_every_gateway_switch_combination_gives_identical_rows(reference_rows, monkeypatch, devices, pipeline, eager):
    frames, grid, ref_nodes, ref_rows = reference_rows
    if devices:
        monkeypatch.setenv("GPIS_DEVICES", devices)
    monkeypatch.setenv("GPIS_PIPELINE_UPDATE", pipeline)
    monkeypatch.setenv("GPIS_EAGER_INVERSE", eager)
    m, nodes, rows = _run(frames, grid)
    assert m.num_devices() == (2 if devices else 1)
    for a, b in zip(ref_nodes, nodes):
        assert np.array_equal(a, b)
    for k, (a, b) in enumerate(zip(ref_rows, rows)):
        assert np.array_equal(a, b), (devices, pipeline, eager, k)


def test_cross_check_setters_give_identical_rows(reference_rows):
    frames, grid, ref_nodes, ref_rows = reference_rows

    def setup(m):
        m.set_host_gather(True)
        m.set_keep_factors(True)
    _, nodes, rows = _run(frames, grid, setup)
    for a, b in zip(ref_nodes, nodes):
        assert np.array_equal(a, b)
    for a, b in zip(ref_rows, rows):
        assert np.array_equal(a, b)


def test_pipeline_wish_survives_sharding_round_trip(monkeypatch):
    """gpis3_set_shard(world > 1) switched the pipeline off for good; the caller's wish is remembered now and
    applies again at world 1 (stats()['pipeline'])."""
    import gpismap_amd
    monkeypatch.delenv("GPIS_PIPELINE_UPDATE", raising=False)
    monkeypatch.delenv("GPIS_DEVICES", raising=False)
    m = gpismap_amd.GPisMap3()
    st = m.stats()
    if "pipelined" not in st:
        pytest.skip("stats() does not report the pipeline state")
    assert st["pipelined"] == 1
    m.set_shard(0, 2)
    assert m.stats()["pipelined"] == 0
    m.set_shard(0, 1)
    assert m.stats()["pipelined"] == 1
    m.set_pipeline(False)
    m.set_shard(0, 2); m.set_shard(0, 1)
    assert m.stats()["pipelined"] == 0
