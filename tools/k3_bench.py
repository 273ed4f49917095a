#!/usr/bin/env python3
"""Micro-benchmark of the K > 256 training chain (gather, build, chol / chol_flow, inverse) through the kernel-level C-ABI:
M clusters of N points (K ~ 3.4 N).  Prints the ms of the timed second batch and the tile products per microsecond."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import gpismap_amd  # noqa: E402
from test_gpu_ongpis import make_cluster, soa9  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 350
M = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dim, scale = 3, 0.04
rng = np.random.default_rng(1)
cl = [make_cluster(rng, dim, N, scale) for _ in range(M)]
pos = np.concatenate([c[0] for c in cl]); grad = np.concatenate([c[1] for c in cl])
val = np.concatenate([c[2] for c in cl]); sx = np.concatenate([c[3] for c in cl]); sg = np.concatenate([c[4] for c in cl])
off = (np.arange(M + 1) * N).astype(np.int32)
ids = np.arange(M * N, dtype=np.int32)
st = gpismap_amd.OnGPIS(dim, scale)
pts = soa9(dim, pos, grad, val, sx, sg)
models = st.train(pts, off, ids)
K = st.model(models[0])["K"]
best = 1e9
for _ in range(3):
    st.train(pts, off, ids)
    best = min(best, st.last_ms()[0])
nb = (K + 31) // 32
prod = M * (nb ** 3 / 6.0) * 2      # factorisation + inverse
print("N=%d K=%d (nb %d) clusters=%d: train %.2f ms -> %.2f tile products/us (chol+inverse), %.1f TFLOP/s of K^3/3" %
      (N, K, nb, M, best, prod / best / 1e3, M * K ** 3 / 3.0 / best / 1e9))
