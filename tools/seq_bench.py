#!/usr/bin/env python3
"""Per-frame wall time of update() and test() on the bundled sequences (BASELINE configs 2 and 3, the frames
held in tests/golden): HIP path vs the CPU oracle on the same box."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import gpismap_amd  # noqa: E402
import oracle_lib  # noqa: E402
import replay  # noqa: E402


def ms(f, *a):
    t0 = time.perf_counter(); r = f(*a); return (time.perf_counter() - t0) * 1e3, r


def gpu_alone():
    """data/3D through the HIP path ALONE (no CPU oracle running its 256-thread update between the frames): VERDICT r3 asked
    whether the 12-15 ms update() outliers of the interleaved run are the oracle's doing."""
    frames = replay.load_bigbird(); grid = replay.demo3_grid()
    for rep in range(3):
        gm = gpismap_amd.GPisMap3(frames[0]["cam"])
        pipelined = rep == 2        # passes 1, 2: per-call split as the reference's (update() includes its training); pass 3: the default mode
        gm.set_pipeline(pipelined)
        up, te = [], []
        for i, fr in enumerate(frames):
            if i:
                gm.set_camera(fr["cam"])
            ug, _ = ms(gm.update, fr["depth"], fr["pose"]); tg, _ = ms(gm.test, grid)
            up.append(ug); te.append(tg)
        print("3-D GPU alone, pass %d (%s): update ms per frame %s" % (rep + 1, "pipelined update, the default: test() joins the training" if pipelined else "synchronous update", " ".join("%.1f" % v for v in up)))
        print("                        test ms per frame   %s" % " ".join("%.1f" % v for v in te))
        print("   update median of frames 2..40 %.1f ms, max %.1f ms; test median %.1f ms; update + test median %.1f ms" %
              (float(np.median(up[1:])), max(up[1:]), float(np.median(te[1:])), float(np.median(np.asarray(up[1:]) + np.asarray(te[1:])))))


def main():
    if "--gpu-alone" in sys.argv:
        return gpu_alone()
    frames = replay.load_bigbird(); grid = replay.demo3_grid()
    gm = gpismap_amd.GPisMap3(frames[0]["cam"]); om = oracle_lib.OracleMap3(frames[0]["cam"])
    gm.set_pipeline(False)
    print("3-D (data/3D, %d frames in the fixture, %d-point demo grid): frame | points | update ms gpu/cpu | test ms gpu/cpu" % (len(frames), grid.shape[0]))
    for i, fr in enumerate(frames):
        if i:
            gm.set_camera(fr["cam"]); om.set_camera(fr["cam"])
        ug, _ = ms(gm.update, fr["depth"], fr["pose"]); uc, _ = ms(om.update, fr["depth"], fr["pose"])
        tg, rg = ms(gm.test, grid); tc, ro = ms(om.test, grid)
        print("  %2d | %5d | %7.1f / %7.1f | %7.1f / %7.1f | rows identical %.5f" % (i + 1, gm.num_points(), ug, uc, tg, tc, float(np.mean(np.all(rg == ro, axis=1)))))
    frames = replay.load_gazebo(); grid = replay.demo2_grid()
    g2 = gpismap_amd.GPisMap(); o2 = oracle_lib.OracleMap2()
    print("2-D (data/2D, %d frames, %d-point demo grid): frame | update ms gpu/cpu | test ms gpu/cpu" % (len(frames), grid.shape[0]))
    for i, fr in enumerate(frames):
        ug, _ = ms(g2.update, fr["thetas"], fr["ranges"], fr["pose"]); uc, _ = ms(o2.update, fr["thetas"], fr["ranges"], fr["pose"])
        if i in (0, 9, 18, 27):
            tg, rg = ms(g2.test, grid); tc, ro = ms(o2.test, grid)
            print("  %2d | %7.1f / %7.1f | %7.1f / %7.1f | rows identical %.5f" % (i + 1, ug, uc, tg, tc, float(np.mean(np.all(rg == ro, axis=1)))))


if __name__ == "__main__":
    main()
