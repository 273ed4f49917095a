#!/usr/bin/env python3
"""Per-rank work of the 8-way query cut: builds the F-frame synthetic map once and evaluates, on one
GPU, the queries each rank of an N-way block-cyclic cut (and of a contiguous-slab cut) would own; prints GP evaluations
and K4 time per rank."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import gpismap_amd  # noqa: E402
from gpismap_amd import sharding  # noqa: E402
import replay  # noqa: E402


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    grid_n = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    gm = gpismap_amd.GPisMap3(); gm.set_profile(True)
    for f in range(frames):
        gm.update(replay.synthetic_depth(f), replay.IDENTITY_POSE)
    grid = replay.synthetic_grid(grid_n)
    dev = torch.device("cuda", 0)
    for name in ("block-cyclic 64K", "contiguous slabs"):
        ev, ms = [], []
        for r in range(world):
            if name.startswith("block"):
                x = sharding.take_blocks(grid, world, r)
            else:
                n = grid.shape[0]; x = grid[(n * r) // world:(n * (r + 1)) // world]
            xd = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
            res = torch.zeros((xd.shape[0], 8), dtype=torch.float32, device=dev)
            gm.test_device(xd.data_ptr(), xd.shape[0], res.data_ptr(), torch.cuda.current_stream().cuda_stream)
            s = gm.stats(); ev.append(s["last_test_evals"]); ms.append(s["last_test_k4_ms"])
        ev = np.array(ev); ms = np.array(ms)
        print("%-18s %d ranks: GP evaluations per rank (M) %s | max/mean %.3f | K4 ms per rank %s | max/mean %.3f"
              % (name, world, np.round(ev / 1e6, 2).tolist(), ev.max() / ev.mean(), np.round(ms, 1).tolist(), ms.max() / ms.mean()))


if __name__ == "__main__":
    main()
