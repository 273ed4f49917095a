"""The reference's UNCHANGED mex gateways (compiled by path into oracle/_ref/*.so, see mexdrive.py) driving the
HIP library on the GPU exactly as matlab/demo_gpisMap3.m / demo_gpisMap.m drive them: 'setCamera', 'update',
'test', 'getAllPoints', 'reset'.  Results are compared with the CPU oracle (bit-identical rows, SURVEY 8(c)
tolerances) and with what the visualisation scripts consume (visualize_gpisMap3.m:27 `res(1,:)+0.2`, row 5)."""
import os

import numpy as np
import pytest

import mexdrive
import oracle_lib
import replay

pytestmark = pytest.mark.gpu


def _gateway(name):
    mexdrive.build()
    if not os.path.exists(mexdrive.gateway_path(name)):
        pytest.skip("gateway object not built (reference tree absent at build time)")
    return mexdrive.Gateway(name)


def test_mexGPisMap3_gateway_replays_demo_frames():
    g = _gateway("mexGPisMap3")
    frames = replay.load_bigbird()
    seq = replay.demo3_sequence()
    grid = replay.demo3_grid()
    X = np.ascontiguousarray(grid.T)                     # 3 x N single, as demo_gpisMap3.m:38 builds xtest
    om = None
    for i in range(3):
        fr = frames[i]
        cam_id = seq[i][1]
        g.call(0, "setCamera", np.array([[float(cam_id)]]), "bigbird")       # demo_gpisMap3.m:53
        D = fr["depth"].reshape(640, 480).T                                  # 480 x 640 single (column-major = col*480+row)
        assert g.call(0, "update", np.asfortranarray(D), fr["pose"].reshape(1, 12)) == []   # :54
        if om is None:
            om = oracle_lib.OracleMap3(fr["cam"])
        else:
            om.set_camera(fr["cam"])
        om.update(fr["depth"], fr["pose"])
        pts = g.call(1, "getAllPoints")                                      # visualize_gpisMap3.m:59
        assert len(pts) == 1 and pts[0].dtype == np.float32 and pts[0].shape[0] == 3
        no = om.nodes()
        assert pts[0].shape[1] == no.shape[0] == om.num_points()
        assert np.array_equal(pts[0].T, no[:, :3])       # contents AND order (octree traversal order)
        res = g.call(1, "test", X)                                           # demo_gpisMap3.m:68
        assert len(res) == 1 and res[0].shape == (8, grid.shape[0]) and res[0].dtype == np.float32
        ro = om.test(grid)
        rg = res[0].T
        same = float(np.mean(np.all(rg == ro, axis=1)))
        assert same >= 0.9995, same
        assert float(np.sqrt(np.mean((rg[:, 0] - ro[:, 0]) ** 2))) < 1e-5
        # what the plotting scripts consume
        sdf = res[0][0, :] + np.float32(0.2)             # visualize_gpisMap3.m:27
        var = res[0][4, :]                               # :28
        assert np.all(np.isfinite(sdf)) and np.all(var > -1e-3) and np.all(var <= np.float32(1.005) + 1e-6)
        untouched = (ro[:, 4] == np.float32(1.005)) & (ro[:, 0] == 0)
        assert np.array_equal(res[0][0, untouched], np.zeros(int(untouched.sum()), dtype=np.float32))   # mex zero-fill survives
    g.call(0, "reset")                                                       # demo_gpisMap3.m:25
    assert g.call(1, "getAllPoints") == []               # gpm == 0 after 'reset'
    assert g.call(1, "test", X) == []                    # "the map is not initialized"
    # a fresh map after 'reset' behaves like the first one
    g.call(0, "setCamera", np.array([[float(seq[0][1])]]), "bigbird")
    g.call(0, "update", np.asfortranarray(frames[0]["depth"].reshape(640, 480).T), frames[0]["pose"].reshape(1, 12))
    om2 = oracle_lib.OracleMap3(frames[0]["cam"])
    om2.update(frames[0]["depth"], frames[0]["pose"])
    assert np.array_equal(g.call(1, "getAllPoints")[0].T, om2.nodes()[:, :3])
    g.call(0, "reset")


def test_mexGPisMap_gateway_replays_demo_frames():
    g = _gateway("mexGPisMap")
    frames = replay.load_gazebo()
    grid = replay.demo2_grid()
    X = np.ascontiguousarray(grid.T)                     # 2 x N single (demo_gpisMap.m:35)
    om = oracle_lib.OracleMap2()
    for i in range(3):
        fr = frames[i]
        t = g.call(1, "update", fr["thetas"].reshape(1, -1), fr["ranges"].reshape(-1, 1), fr["pose"].reshape(-1, 1))   # :49-51
        assert len(t) == 1 and t[0].shape == (1, 1) and t[0][0, 0] > 0       # seconds, double
        om.update(fr["thetas"], fr["ranges"], fr["pose"])
        out = g.call(2, "test", X)                                           # :65 [res, time]
        assert len(out) == 2 and out[0].shape == (6, grid.shape[0]) and out[1].shape == (1, 1)
        ro = om.test(grid)
        rg = out[0].T
        assert float(np.mean(np.all(rg == ro, axis=1))) >= 0.9995
        assert float(np.sqrt(np.mean((rg[:, 0] - ro[:, 0]) ** 2))) < 1e-5
    # wrong input class: message, nothing created (mexGPisMap.cpp:92-95)
    assert g.call(1, "test", X.astype(np.float64)) == []
    g.call(0, "reset")
    assert g.call(1, "test", X) == []
