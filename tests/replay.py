"""Replay harness for the bundled sequences: the build's counterpart of the reference's
matlab/demo_gpisMap3.m and matlab/demo_gpisMap.m (frame order, pose packing, camera
switching and query grids).  Inputs come from the committed
fixtures under tests/golden/ (never from /root/reference at run time)."""
import os
import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# mex/mexGPisMap3.cpp:30-35 (bigbird intrinsics, 1-based camera ids in the demo)
BIGBIRD_FX = [570.9361, 572.3318, 568.9403, 567.9881, 572.7638]
BIGBIRD_FY = [570.9376, 572.3316, 568.9419, 567.9995, 572.7567]
BIGBIRD_CX = [306.8789, 309.9968, 308.4583, 310.5243, 310.4192]
BIGBIRD_CY = [238.8476, 230.6296, 225.8232, 223.9443, 214.8762]


def bigbird_cam(cam_id):
    n = cam_id - 1
    return np.array([BIGBIRD_FX[n], BIGBIRD_FY[n], BIGBIRD_CX[n], BIGBIRD_CY[n], 640, 480], dtype=np.float64)


def demo3_sequence():
    """(frame number, camera id) for the 40 updates of demo_gpisMap3.m:33-45."""
    frame_nums = list(range(93, 360, 3)) + list(range(3, 91, 3))
    cam_ids = [1, 2, 3, 4, 3, 2] * 30
    out = []
    count = 0
    for k in range(0, len(frame_nums), 3):
        out.append((frame_nums[k], cam_ids[count]))
        count += 1
    return out


def demo3_grid():
    """meshgrid(-0.07:0.01:0.13, -0.1:0.01:0.14, 0:0.01:0.28) flattened MATLAB-style
    (y fastest, then x, then z) -> float32 [N,3] (demo_gpisMap3.m:37-38)."""
    xs = np.arange(21) * 0.01 - 0.07
    ys = np.arange(25) * 0.01 - 0.10
    zs = np.arange(29) * 0.01
    Z, X, Y = np.meshgrid(zs, xs, ys, indexing="ij")
    return np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1).astype(np.float32)


def pose_from_row(row16):
    """poses.txt row -> [t' R(:)'] as demo_gpisMap3.m:49-54 packs it."""
    T = np.asarray(row16, dtype=np.float32).reshape(4, 4, order="F")
    R = T[0:3, 0:3]
    t = T[3, 0:3]
    return np.concatenate([t, R.ravel(order="F")]).astype(np.float32)


def load_bigbird():
    """Returns list of dicts {depth(float32[307200], column-major), pose(float32[12]), cam(float64[6])}."""
    z = np.load(os.path.join(GOLDEN, "bigbird_seq.npz"))
    frames = []
    for i in range(int(z["nframes"])):
        d = np.zeros(640 * 480, dtype=np.uint16)
        idx = z["idx_%02d" % i]
        d[idx] = z["val_%02d" % i]
        depth = d.astype(np.float32) * np.float32(0.0001)  # demo_gpisMap3.m:46-47
        frames.append(dict(depth=depth, pose=z["poses"][i].astype(np.float32), cam=bigbird_cam(int(z["cams"][i]))))
    return frames


def synthetic_depth(f, width=640, height=480, fx=568.0, fy=568.0, cx=310.0, cy=224.0, noise=False):
    """z(col,row) = 1 + 0.05 sin(6(u + 0.01 f)) cos(5 v), column-major."""
    col = np.arange(width, dtype=np.float64)[:, None]
    row = np.arange(height, dtype=np.float64)[None, :]
    u = (col - cx) / fx
    v = (row - cy) / fy
    z = 1.0 + 0.05 * np.sin(6.0 * (u + 0.01 * f)) * np.cos(5.0 * v)
    if noise:
        rng = np.random.default_rng(20190520 + f)
        z = z + rng.normal(0.0, 1e-3, size=z.shape)
    return z.astype(np.float32).ravel()  # index = col*height + row


IDENTITY_POSE = np.array([0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1], dtype=np.float32)


def synthetic_grid(n=256):
    """n^3 points over [-0.60,0.60]x[-0.45,0.45]x[0.85,1.15], x fastest."""
    xs = np.linspace(-0.60, 0.60, n)
    ys = np.linspace(-0.45, 0.45, n)
    zs = np.linspace(0.85, 1.15, n)
    Z, Y, X = np.meshgrid(zs, ys, xs, indexing="ij")
    return np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1).astype(np.float32)


def load_gazebo():
    """data/2D frames 101:100:2801 as demo_gpisMap.m:37-51 feeds them: list of dicts
    {thetas float32[270], ranges float32[270], pose float32[6] = [x y cos(phi) sin(phi) -sin(phi) cos(phi)]}."""
    z = np.load(os.path.join(GOLDEN, "gazebo2d_seq.npz"))
    thetas = z["thetas"].astype(np.float32)
    frames = []
    for i in range(z["ranges"].shape[0]):
        x, y, phi = z["poses"][i]
        rot = np.array([[np.cos(phi), -np.sin(phi)], [np.sin(phi), np.cos(phi)]])
        pose = np.concatenate([[x, y], rot.ravel(order="F")]).astype(np.float32)   # single([tr; Rot(:)])
        frames.append(dict(thetas=thetas, ranges=z["ranges"][i].astype(np.float32), pose=pose))
    return frames


def demo2_grid():
    """meshgrid(-4.9:0.1:19.9, -14.9:0.1:4.9) flattened MATLAB-style (y fastest): 249 x 199 = 49 551 points
    (demo_gpisMap.m:29-35)."""
    xs = -5.0 + 0.1 * np.arange(1, 250)
    ys = -15.0 + 0.1 * np.arange(1, 200)
    X, Y = np.meshgrid(xs, ys, indexing="ij")
    return np.stack([X.ravel(), Y.ravel()], axis=1).astype(np.float32)


def stress_clusters(ncl, rng):
    """BASELINE config 5: `ncl` clusters on a 0.05 m lattice sheet, exactly 64 points each (jittered 8x8
    grid: the 6.25 mm spacing keeps the min-distance rule), unit normals = sheet normal + N(0, 0.05^2) renormalised,
    val = -0.2, sigx = U(1e-3, 5e-3), sigg = U(0.01, 0.1): every point carries a gradient, K = 256 for every cluster.
    Returns pos [ncl*64, 3], grad, val, sigx, sigg (float32), cluster c = rows 64c .. 64c+63."""
    side = int(np.ceil(np.sqrt(ncl)))
    c = np.arange(ncl)
    cx = (c % side) * 0.05 + 0.025
    cy = (c // side) * 0.05 + 0.025
    gx, gy = np.meshgrid(np.arange(8), np.arange(8), indexing="ij")
    px = cx[:, None] - 0.025 + (gx.ravel()[None, :] + 0.5) * 0.00625 + rng.uniform(-0.001, 0.001, (ncl, 64))
    py = cy[:, None] - 0.025 + (gy.ravel()[None, :] + 0.5) * 0.00625 + rng.uniform(-0.001, 0.001, (ncl, 64))
    pz = rng.uniform(-0.02, 0.02, (ncl, 64))
    pos = np.stack([px, py, pz], axis=2).reshape(-1, 3).astype(np.float32)
    n = np.array([0.0, 0.0, 1.0]) + rng.normal(0, 0.05, (ncl * 64, 3))
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    val = np.full(ncl * 64, -0.2, dtype=np.float32)
    return (pos, n.astype(np.float32), val, rng.uniform(1e-3, 5e-3, ncl * 64).astype(np.float32),
            rng.uniform(0.01, 0.1, ncl * 64).astype(np.float32))


def stress_queries(pos, ncl, nq, rng):
    """nq queries per cluster near its points: [ncl*nq, 3]."""
    return (pos.reshape(ncl, 64, 3)[:, rng.integers(0, 64, nq), :] + rng.normal(0, 0.005, (ncl, nq, 3))).reshape(-1, 3).astype(np.float32)
