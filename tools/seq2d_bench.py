#!/usr/bin/env python3
"""data/2D through the HIP path alone: update / test ms per frame of the 2-D map (28 lidar frames, 49 551-point demo grid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, gpismap_amd, replay
frames = replay.load_gazebo(); grid = replay.demo2_grid()
for rep in range(3):
    g2 = gpismap_amd.GPisMap()
    if hasattr(g2, "set_pipeline") and hasattr(g2.L, "gpis2_set_pipeline"):
        g2.set_pipeline(rep != 1)           # pass 2: synchronous update (the reference's split); passes 1, 3: the default (pipelined)
    up, te = [], []
    for fr in frames:
        t0 = time.perf_counter(); g2.update(fr["thetas"], fr["ranges"], fr["pose"]); up.append((time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter(); g2.test(grid); te.append((time.perf_counter() - t0) * 1e3)
    print("2-D pass %d: update median %.2f ms (max %.2f), test median %.2f ms | update per frame %s" %
          (rep + 1, float(np.median(up[1:])), max(up[1:]), float(np.median(te[1:])), " ".join("%.1f" % v for v in up)))
