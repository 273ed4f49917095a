#!/usr/bin/env python3
"""Per-phase wall time of GPisMap3.update() on the synthetic 640x480 frames (host + device).  Synchronous update() unless
GPIS_PIPELINE_UPDATE=1 is set (the library default is pipelined: the frame time then excludes the training it left in flight)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, gpismap_amd, replay
gm = gpismap_amd.GPisMap3()
gm.set_pipeline(os.environ.get("GPIS_PIPELINE_UPDATE", "0") not in ("", "0"))
gm.set_profile(True)
for f in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    d = replay.synthetic_depth(f)
    t0 = time.perf_counter(); gm.update(d, replay.IDENTITY_POSE); dt = (time.perf_counter() - t0) * 1e3
    s = gm.stats()
    print("frame %d: %.1f ms | preproc %.1f obsgp_train %.1f reeval %.1f eval %.1f gps %.1f (K3 device %.1f) | pts %d clusters %d late %d | model pool %.2f GB"
          % (f, dt, s["upd_preproc_ms"], s["upd_obsgp_train_ms"], s["upd_reeval_ms"], s["upd_eval_ms"], s["upd_gps_ms"], s["last_train_ms"],
             gm.num_points(), s["clusters"], s["late_reevals"], s["device_bytes"] / 1e9))
