#!/usr/bin/env python3
"""A small test() call (24^3 points, the size of the demo grids) on the F = 5 synthetic map: wall time per call; run under
rocprofv3 --kernel-trace (tools/small_test_timeline.sh) for the kernels behind it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, gpismap_amd, replay
gm = gpismap_amd.GPisMap3()
for f in range(5):
    gm.update(replay.synthetic_depth(f), replay.IDENTITY_POSE)
g = np.linspace(-0.9, 0.9, 24, dtype=np.float32)
X = np.stack(np.meshgrid(0.5 * g, 0.5 * g, 1.0 + 0.2 * g, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
gm.test(X)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); gm.test(X); ts.append((time.perf_counter() - t0) * 1e3)
print("test() of %d points: %s ms" % (X.shape[0], " ".join("%.2f" % t for t in ts)))
