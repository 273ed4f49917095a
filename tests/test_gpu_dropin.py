"""The C++ surface end to end on the GPU: tests/cpp/dropin_demo.cpp (built with the mex gateways' flags against
include/GPisMap3.h / GPisMap.h) must produce bit-identical results to the ctypes path on the same inputs."""
import os
import subprocess

import numpy as np
import pytest

import replay
from test_host import _build_dropin

pytestmark = pytest.mark.gpu


def _fnv(a):
    c = np.uint32(2166136261)
    for u in np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).ravel():
        c = np.uint32((int(c) ^ int(u)) * 16777619 & 0xFFFFFFFF)
    return int(c)


def test_cpp_classes_match_ctypes_path(tmp_path):
    import gpismap_amd
    exe = _build_dropin(tmp_path)
    d = tmp_path / "in"
    d.mkdir()
    depth = [replay.synthetic_depth(f) for f in range(2)]
    G = 24
    xs = np.linspace(-0.60, 0.60, G); ys = np.linspace(-0.45, 0.45, G); zs = np.linspace(0.85, 1.15, G)
    Z, Y, X = np.meshgrid(zs, ys, xs, indexing="ij")
    x = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1).astype(np.float32)
    th = ((-135.0 + np.arange(270)) * np.pi / 180.0).astype(np.float32)
    rg = (4.0 + 0.5 * np.sin(0.07 * np.arange(270))).astype(np.float32)
    jj, ii = np.meshgrid(np.arange(40), np.arange(40), indexing="ij")
    x2 = np.stack([-5.0 + 0.25 * ii.ravel(), -5.0 + 0.25 * jj.ravel()], axis=1).astype(np.float32)
    for name, arr in (("depth0.bin", depth[0]), ("depth1.bin", depth[1]), ("x.bin", x), ("th.bin", th), ("rg.bin", rg), ("x2.bin", x2)):
        np.ascontiguousarray(arr, dtype=np.float32).tofile(str(d / name))
    r = subprocess.run([exe, str(d)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    out = dict((l.split()[0], l.split()[1:]) for l in r.stdout.strip().splitlines())
    assert out["test_before_update"] == ["0"] and out["test_wrong_dim"] == ["0"] and out["map_dimension"] == ["2"]
    # the same calls through ctypes
    gm = gpismap_amd.GPisMap3(np.array([568.0, 568.0, 310.0, 224.0, 640, 480], dtype=np.float32))
    for f in range(2):
        gm.update(depth[f], replay.IDENTITY_POSE)
    res = gm.test(x)
    assert out["map3"][0:2] == ["ok", "1"]
    assert int(out["map3"][3]) == gm.num_points()
    assert int(out["map3"][5], 16) == _fnv(res)
    g2 = gpismap_amd.GPisMap()
    g2.update(th, rg, np.array([0, 0, 1, 0, 0, 1], dtype=np.float32))
    res2 = g2.test(x2)
    assert out["map2"][0:2] == ["ok", "1"]
    assert int(out["map2"][3], 16) == _fnv(res2)
