"""Generates tests/golden/bigbird_seq.npz and gazebo2d_seq.npz from the reference's
bundled DATA files (data/3D/bigbird_detergent, data/2D/gazebo1.mat).  Run once in the
build container (needs /root/reference); the outputs are committed so nothing reads
/root/reference at test/bench time.  Only data is taken: 16-bit depth pixels (sparse,
masked frames), the pose table and the laser scans of the frames the demos use."""
import os
import sys
import numpy as np
from PIL import Image

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import replay  # noqa: E402

REF = "/root/reference/data"
OUT = os.path.dirname(os.path.abspath(__file__))


def bigbird():
    seq = replay.demo3_sequence()
    poses = np.loadtxt(os.path.join(REF, "3D/bigbird_detergent/pose/poses.txt")).astype(np.float32)
    d = dict(nframes=len(seq), cams=np.array([c for _, c in seq], dtype=np.int32),
             frame_nums=np.array([f for f, _ in seq], dtype=np.int32))
    P = []
    for i, (frm, cam) in enumerate(seq):
        img = Image.open(os.path.join(REF, "3D/bigbird_detergent/masked_depth", "frame%d_cam%d.png" % (frm, cam)))
        a = np.array(img).astype(np.uint16)          # [480, 640]
        flat = a.ravel(order="F")                      # MATLAB column-major: index = col*480 + row
        idx = np.nonzero(flat)[0].astype(np.int32)
        d["idx_%02d" % i] = idx
        d["val_%02d" % i] = flat[idx]
        P.append(replay.pose_from_row(poses[i]))
    d["poses"] = np.stack(P)
    np.savez_compressed(os.path.join(OUT, "bigbird_seq.npz"), **d)
    print("bigbird:", len(seq), "frames")


def gazebo():
    from scipy.io import loadmat
    m = loadmat(os.path.join(REF, "2D/gazebo1.mat"))
    frames = np.arange(101, 2802, 100)  # MATLAB 1-based, demo_gpisMap.m:37
    np.savez_compressed(os.path.join(OUT, "gazebo2d_seq.npz"),
                        thetas=m["thetas"].astype(np.float64).ravel(),
                        ranges=m["ranges"][frames - 1].astype(np.float64),
                        poses=m["poses"][frames - 1].astype(np.float64),
                        frames=frames)
    print("gazebo:", len(frames), "frames")


if __name__ == "__main__":
    bigbird()
    gazebo()
