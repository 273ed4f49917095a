#!/usr/bin/env python3
"""update() wall time per frame with the pipelined training on and off (synthetic 640x480 frames), the drain of the last
frame's training timed separately; the two maps must hold the same points and answer a query grid with the same bits."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, gpismap_amd, replay
F = int(sys.argv[1]) if len(sys.argv) > 1 else 6
res = {}
for mode in (1, 0, 1, 0):
    gm = gpismap_amd.GPisMap3()
    gm.set_pipeline(bool(mode))
    ts = []
    for f in range(F):
        d = replay.synthetic_depth(f)
        t0 = time.perf_counter(); gm.update(d, replay.IDENTITY_POSE); ts.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter(); gm.sync(); drain = (time.perf_counter() - t0) * 1e3
    tot = sum(ts[1:]) + drain
    print("pipeline %d: frames %s drain %.1f | frames 1..%d + drain: %.1f ms = %.2f ms/frame" %
          (mode, " ".join("%.1f" % t for t in ts), drain, F - 1, tot, tot / (F - 1)))
    g = np.linspace(-0.9, 0.9, 24, dtype=np.float32)
    X = np.stack(np.meshgrid(0.5 * g, 0.5 * g, 1.0 + 0.2 * g, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    out = gm.test(X)
    res[mode] = (gm.num_points(), np.asarray(out).copy())
print("same points:", res[0][0] == res[1][0], " test() bit-identical:", np.array_equal(res[0][1], res[1][1], equal_nan=True))
