"""BASELINE config 4 at FULL size (synthetic 640x480 depth, 256^3 query grid) through the C-ABI, checked with
size-independent properties (the CPU oracle needs hours for this grid; test_gpu_map3 / bench.py compare
samples with it):
  * determinism: the same pass twice gives bit-identical results;
  * partition invariance (1-, 2-, 4-, 8-GPU runs must agree bit for bit): any slab of the query
    array tested on its own reproduces the rows of the full pass -- per-query arithmetic must not depend on
    how queries are grouped into chunks, bins or 8-query tiles;
  * order invariance: a permuted query array gives the permuted result;
  * checksum of checksums over the pass stays finite and matches between the runs;
and the edge cases of the interface: empty / single / non-multiple-of-8 query counts, queries with no
cluster in range (only the prior variance is written, GPisMap3.cpp:816), a frame without valid pixels."""
import numpy as np
import pytest

import replay

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gm():
    import gpismap_amd
    g = gpismap_amd.GPisMap3()
    for f in range(5):                        # F = 5 frames: the map bench.py measures (BASELINE config 4)
        g.update(replay.synthetic_depth(f), replay.IDENTITY_POSE)
    return g


def _rows_equal(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_full_grid_determinism_and_partition_invariance(gm):
    x = replay.synthetic_grid(256)
    n = x.shape[0]
    assert n == 256 ** 3
    r1 = gm.test(x)
    r2 = gm.test(x)
    assert np.all(np.isfinite(r1))
    assert _rows_equal(r1, r2), "two passes over the same grid differ"
    touched = r1[:, 4] != 0.0
    assert touched.all()                      # every query gets at least the prior variance
    ev = gm.stats()
    assert ev["last_test_evals"] > n          # more than one GP evaluation per query on average (fallback / blending)
    # slabs as the multi-GPU path cuts them (8 ranks), plus ragged cuts that do not align with chunks or tiles of 8
    rng = np.random.default_rng(4)
    cuts = [(k * n // 8, (k + 1) * n // 8) for k in (0, 3, 7)]
    for _ in range(3):
        a = int(rng.integers(0, n - 300000))
        cuts.append((a, a + int(rng.integers(1, 300000))))
    for a, b in cuts:
        rs = gm.test(x[a:b])
        assert _rows_equal(rs, r1[a:b]), "slab [%d,%d) differs from the full pass" % (a, b)
    # checksum of checksums
    c1 = np.bitwise_xor.reduce(r1.view(np.uint32).reshape(-1, 8), axis=0)
    c2 = np.bitwise_xor.reduce(r2.view(np.uint32).reshape(-1, 8), axis=0)
    assert np.array_equal(c1, c2)


def test_order_invariance(gm):
    x = replay.synthetic_grid(96)
    r = gm.test(x)
    perm = np.random.default_rng(9).permutation(x.shape[0])
    rp = gm.test(x[perm])
    assert _rows_equal(rp, r[perm])


def test_edge_cases(gm):
    import gpismap_amd
    x = replay.synthetic_grid(32)
    r = gm.test(x)
    # empty query set: the reference's test() returns false for leng < 1 (GPisMap3.cpp:905)
    assert gm.test(np.zeros((0, 3), dtype=np.float32)) is None
    # 1, 7, 9 queries: partial tiles
    for k in (1, 7, 9, 8 * 37 + 5):
        assert _rows_equal(gm.test(x[:k]), r[:k])
    # queries far from every cluster: only res[4] = 1 + map noise is written, the caller's pre-fill survives
    far = np.array([[5.0, 5.0, 5.0], [-3.0, 0.0, 9.0]], dtype=np.float32)
    res = np.full((2, 8), 7.5, dtype=np.float32)
    out = gm.test(far, res)
    assert np.all(out[:, [0, 1, 2, 3, 5, 6, 7]] == 7.5)
    assert np.all(out[:, 4] > 1.0) and np.all(out[:, 4] < 1.1)
    # a frame with no valid pixel leaves the map as it is (update() returns silently, GPisMap3.cpp:212-215)
    n0 = gm.num_points()
    gm.update(np.zeros(640 * 480, dtype=np.float32), replay.IDENTITY_POSE)
    assert gm.num_points() == n0
    assert _rows_equal(gm.test(x), r)
    # a fresh map: test() before the first update() is refused
    g2 = gpismap_amd.GPisMap3()
    assert g2.test(x[:4]) is None
