"""Map checkpoint (include/gpismap_amd.h: gpis3_save / gpis3_load; SURVEY 8(f)4, optional -- the reference keeps its map only in
the mex singleton): a reloaded map must answer test() with the same bits as the saved one, hold the same points in the same
traversal order, and continue the sequence exactly like the map that was never saved."""
import numpy as np
import pytest

import gpismap_amd
import replay

pytestmark = pytest.mark.gpu


def _grid(n=20):
    g = np.linspace(-0.9, 0.9, n, dtype=np.float32)
    return np.stack(np.meshgrid(0.5 * g, 0.5 * g, 1.0 + 0.2 * g, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)


def test_checkpoint_round_trip_and_continuation(tmp_path):
    X = _grid()
    a = gpismap_amd.GPisMap3()
    for f in range(3):
        a.update(replay.synthetic_depth(f), replay.IDENTITY_POSE)
    ref = a.test(X).copy()
    path = str(tmp_path / "map.ckpt")
    a.save(path)
    assert np.array_equal(a.test(X), ref, equal_nan=True)          # saving changes nothing
    b = gpismap_amd.GPisMap3()
    b.update(replay.synthetic_depth(4), replay.IDENTITY_POSE)        # some other state, replaced by the load
    b.load(path)
    assert b.num_points() == a.num_points()
    assert np.array_equal(b.nodes(), a.nodes())
    assert np.array_equal(b.test(X), ref, equal_nan=True)
    assert b.stats()["clusters"] == a.stats()["clusters"]
    # both continue with the next frame: same map, same answers
    a.update(replay.synthetic_depth(3), replay.IDENTITY_POSE)
    b.update(replay.synthetic_depth(3), replay.IDENTITY_POSE)
    assert np.array_equal(b.nodes(), a.nodes())
    assert np.array_equal(b.test(X), a.test(X), equal_nan=True)


def test_checkpoint_of_an_empty_map_and_bad_files(tmp_path):
    X = _grid(8)
    e = gpismap_amd.GPisMap3()
    p0 = str(tmp_path / "empty.ckpt")
    e.save(p0)
    a = gpismap_amd.GPisMap3()
    a.update(replay.synthetic_depth(0), replay.IDENTITY_POSE)
    ref = a.test(X).copy()
    n = a.num_points()
    # refused files leave the map as it was
    bad = tmp_path / "bad.ckpt"
    bad.write_bytes(b"not a checkpoint" * 64)
    with pytest.raises(gpismap_amd.GpisError):
        a.load(str(bad))
    good = tmp_path / "good.ckpt"
    a.save(str(good))
    cut = tmp_path / "cut.ckpt"
    cut.write_bytes(good.read_bytes()[: good.stat().st_size // 2])
    with pytest.raises(gpismap_amd.GpisError):
        a.load(str(cut))
    with pytest.raises(gpismap_amd.GpisError):
        a.load(str(tmp_path / "missing.ckpt"))
    assert a.num_points() == n and np.array_equal(a.test(X), ref, equal_nan=True)
    # the empty map's checkpoint empties a map
    a.load(p0)
    assert a.num_points() == 0
    a.update(replay.synthetic_depth(0), replay.IDENTITY_POSE)
    assert a.num_points() == n and np.array_equal(a.test(X), ref, equal_nan=True)


def test_checkpoint_into_a_map_over_two_logical_devices(tmp_path):
    X = _grid(12)
    a = gpismap_amd.GPisMap3()
    for f in range(2):
        a.update(replay.synthetic_depth(f), replay.IDENTITY_POSE)
    ref = a.test(X).copy()
    path = str(tmp_path / "map.ckpt")
    a.save(path)
    m = gpismap_amd.GPisMap3(devices=[0, 0])
    m.load(path)
    assert m.num_points() == a.num_points()
    assert np.array_equal(m.test(X), ref, equal_nan=True)
    a.update(replay.synthetic_depth(2), replay.IDENTITY_POSE)
    m.update(replay.synthetic_depth(2), replay.IDENTITY_POSE)
    assert np.array_equal(m.test(X), a.test(X), equal_nan=True)


def test_damaged_payload_of_the_right_size_is_refused_and_files_are_deterministic(tmp_path):
    """a checkpoint of the right size whose payload was overwritten must be refused (checksum + index validation),
    never walked; two saves of the same map are the same bytes (no uninitialised padding in the file)."""
    X = _grid(8)
    a = gpismap_amd.GPisMap3()
    for f in range(2):
        a.update(replay.synthetic_depth(f), replay.IDENTITY_POSE)
    ref = a.test(X).copy()
    nodes = a.nodes().copy()
    p1, p2 = tmp_path / "a.ckpt", tmp_path / "b.ckpt"
    a.save(str(p1)); a.save(str(p2))
    raw = p1.read_bytes()
    assert raw == p2.read_bytes()
    rng = np.random.default_rng(5)
    for k in range(6):
        b = bytearray(raw)
        # damage the tree images (first part of the payload) and, in the later trials, anywhere in the file
        lo, hi = (200, min(len(b), 200000)) if k < 3 else (128, len(b))
        for pos in rng.integers(lo, hi, size=1 + 7 * (k % 3)):
            b[int(pos)] ^= 0x5A
        bad = tmp_path / ("bad%d.ckpt" % k)
        bad.write_bytes(bytes(b))
        with pytest.raises(gpismap_amd.GpisError):
            a.load(str(bad))
        assert np.array_equal(a.nodes(), nodes) and np.array_equal(a.test(X), ref, equal_nan=True)
    a.load(str(p1))
    assert np.array_equal(a.nodes(), nodes) and np.array_equal(a.test(X), ref, equal_nan=True)
