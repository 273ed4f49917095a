#!/usr/bin/env python3
"""Soak: F synthetic frames with a test() every third frame, in the default mode (synchronous, lazy inverse) and in the
pipelined mode with the eager inverse; the maps and every test() result must agree bit for bit."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, gpismap_amd, replay
F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = np.linspace(-0.9, 0.9, 28, dtype=np.float32)
X = np.stack(np.meshgrid(0.5 * g, 0.5 * g, 1.0 + 0.2 * g, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
outs = {}
for mode in ("synchronous", "pipelined+eager"):
    gm = gpismap_amd.GPisMap3()
    gm.set_pipeline(mode != "synchronous")
    if mode != "synchronous":
        gm.set_lazy_inverse(False)
    res, t0 = [], time.perf_counter()
    for f in range(F):
        gm.update(replay.synthetic_depth(f), replay.IDENTITY_POSE)
        if f % 3 == 2:
            res.append(gm.test(X).copy())
    gm.sync()
    res.append(gm.test(X).copy())
    s = gm.stats()
    print("%-16s %d frames %.0f ms, points %d, clusters %d, largest K of the last batch %d, pool %.1f GB" %
          (mode, F, (time.perf_counter() - t0) * 1e3, gm.num_points(), s["clusters"], s["last_train_maxK"], s["device_bytes"] / 1e9))
    outs[mode] = (gm.num_points(), res)
a, b = outs["synchronous"], outs["pipelined+eager"]
print("same points:", a[0] == b[0], " all test() results bit-identical:", all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a[1], b[1])))
