"""Generates the oracle golden vectors (tests/golden/oracle_bigbird.json, oracle_bigbird_res.npz)
by running the CPU oracle over the committed input fixtures.  The reference itself cannot be
built in this image (Eigen absent), so these pin the ORACLE, not the reference: "parity unpinned"."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib  # noqa: E402
import replay  # noqa: E402

oracle_lib.build()
frames = replay.load_bigbird()
grid = replay.demo3_grid()
m = oracle_lib.OracleMap3(frames[0]["cam"])
counts, varlt = [], {}
sub = np.arange(0, grid.shape[0], 7)
res_keep = {}
frame1 = None
for i, fr in enumerate(frames):
    if i:
        m.set_camera(fr["cam"])
    m.update(fr["depth"], fr["pose"])
    counts.append(m.num_points())
    if i == 0:
        st = m.stats()
        frame1 = dict(obsgp_tiles=st["obsgp_tiles"], clusters=st["clusters_trained"], maxK=st["maxK"], sumK=st["sumK"])
    if i in (0, 2, 39):
        res = m.test(grid)
        varlt[str(i + 1)] = int((res[:, 4] < 0.5).sum())
        res_keep["res_%d" % (i + 1)] = res[sub]
    print(i + 1, counts[-1], flush=True)
json.dump(dict(point_counts=counts, var_lt_half=varlt, frame1=frame1), open(os.path.join(HERE, "oracle_bigbird.json"), "w"), indent=1)
np.savez_compressed(os.path.join(HERE, "oracle_bigbird_res.npz"), sub=sub, **res_keep)
