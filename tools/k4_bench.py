#!/usr/bin/env python3
"""Micro-benchmark of K4 (ongpis_eval_kernel) through the kernel-level C-ABI: M models of N points
(K ~ 3.4 N with 20% value-only points), Q queries each.  Prints ms and algorithmic TFLOP/s
((1+d) K^2 + 2(1+d) K + 25 N per evaluation)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import gpismap_amd  # noqa: E402
from test_gpu_ongpis import make_cluster, soa9  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 240
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    Q = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    dim, scale = 3, 0.04
    rng = np.random.default_rng(1)
    cl = [make_cluster(rng, dim, N, scale) for _ in range(M)]
    pos = np.concatenate([c[0] for c in cl]); grad = np.concatenate([c[1] for c in cl])
    val = np.concatenate([c[2] for c in cl]); sx = np.concatenate([c[3] for c in cl]); sg = np.concatenate([c[4] for c in cl])
    off = (np.arange(M + 1) * N).astype(np.int32)
    ids = np.arange(M * N, dtype=np.int32)
    st = gpismap_amd.OnGPIS(dim, scale, keep_factor=True)
    models = st.train(soa9(dim, pos, grad, val, sx, sg), off, ids)
    K = st.model(models[0])["K"]
    xq = (pos[rng.integers(0, M * N, M * Q)] + rng.normal(0, 0.3 * scale, (M * Q, dim))).astype(np.float32)
    jq = np.arange(M * Q, dtype=np.int32)
    jm = np.repeat(models, Q).astype(np.int32)
    flops = M * Q * (4.0 * K * K + 8.0 * K + 25.0 * N)
    best = 1e9
    for _ in range(reps):
        st.eval(xq, jq, jm)
        best = min(best, st.last_ms()[1])
    print("N=%d K=%d models=%d queries/model=%d: train %.2f ms, eval %.3f ms -> %.2f TFLOP/s (%.1f%% of 157.3)"
          % (N, K, M, Q, st.last_ms()[0], best, flops / best / 1e9, 100 * flops / best / 1e9 / 157.3))


if __name__ == "__main__":
    main()
