#!/usr/bin/env python3
"""bench.py -- BASELINE metric on MI355X: SDF test points/sec (+ per-frame update ms) for 3-D
640x480 synthetic depth and a 256^3 query grid.

A "step" is one GPisMap3 test() pass over the query grid, inputs already resident in HBM
(gpis3_test_device).  update() of the synthetic frames is timed separately during set-up and
reported as update_ms_per_frame.  With N > 1 ranks (torchrun, one process per GPU, RCCL) the grid is
cut into N contiguous slabs (strong scaling); every rank builds the same map, evaluates its slab,
and the slabs are gathered on rank 0 with one RCCL gather inside the timed region.

Prints ONE JSON line on rank 0 carrying `roofline` (dominant kernel = K4 ongpis_eval_kernel,
MFMA/FLOP-bound; achieved = algorithmic flops / time inside the K4 launches measured with HIP
events on the launch stream) and `cpu_baseline` (CPU oracle timed on the host cores, rank 0, on a
bounded subsample of the same grid).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=2, help="synthetic depth frames fused before testing")
    ap.add_argument("--grid", type=int, default=256, help="query grid is grid^3 points")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, the measured path) or gloo (rehearsal of the multi-rank logic on fewer GPUs: results staged through host memory)")
    ap.add_argument("--cpu-sample", type=int, default=64, help="CPU baseline runs on a sample^3 subgrid (0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import gpismap_amd
    from gpismap_amd import sharding
    import replay

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if not torch.cuda.is_available() or gpismap_amd.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: gpismap_amd has no CPU fallback")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()      # rehearsal: several ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # ---- set-up (untimed): fuse the synthetic frames; time update() per frame ----
    gm = gpismap_amd.GPisMap3()          # default camera 640x480, fx=fy=568, cx=310, cy=224
    upd_ms = []
    for f in range(args.frames):
        depth = replay.synthetic_depth(f)
        t0 = time.perf_counter()
        gm.update(depth, replay.IDENTITY_POSE)
        upd_ms.append((time.perf_counter() - t0) * 1e3)
    st0 = gm.stats()

    n_total = args.grid ** 3
    grid = replay.synthetic_grid(args.grid)                    # [n,3] float32, x fastest
    lo, hi = sharding.slab_bounds(n_total, world, rank)
    x = torch.from_numpy(grid[lo:hi]).to(dev)                  # resident in HBM before timing
    full = None
    if world > 1 and rank == 0:
        full = torch.zeros((n_total, 8), dtype=torch.float32, device=dev)   # the assembled map
        res = full[lo:hi]
    else:
        res = torch.zeros((hi - lo, 8), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    gm.set_profile(True)                                       # hipEvents around the K4 launches

    def step():
        gm.test_device(x.data_ptr(), hi - lo, res.data_ptr(), stream)
        if world > 1:
            if args.backend == "gloo":     # rehearsal path: host staging (gloo has no device P2P)
                torch.cuda.synchronize()
                got = sharding.gather_slabs(res.cpu(), n_total, world, rank, dst=0)
                if rank == 0:
                    full.copy_(got)
            else:
                sharding.gather_slabs(res, n_total, world, rank, dst=0, out=full)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    k4_ms = 0.0
    flops = 0.0
    launches = 0
    evals = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        s = gm.stats()
        k4_ms += s["last_test_k4_ms"]; flops += s["last_test_flops"]; launches += s["last_test_k4_launches"]
        evals += s["last_test_evals"]
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend != "gloo" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if args.backend == "gloo" and world > 1 and rank == 0:
        # rehearsal only: the assembled map must equal a single-rank pass over the whole grid, bit for bit
        xf = torch.from_numpy(grid).to(dev)
        ref = torch.zeros((n_total, 8), dtype=torch.float32, device=dev)
        gm.test_device(xf.data_ptr(), n_total, ref.data_ptr(), stream)
        torch.cuda.synchronize()
        same = bool(torch.equal(ref.view(torch.int32), full.view(torch.int32)))
        print("rehearsal (%d ranks, gloo staging): assembled map identical to a single-rank pass: %s" % (world, same), file=sys.stderr)
        if not same:
            raise SystemExit("multi-rank assembly differs from the single-rank result")
        del xf, ref

    # ---- CPU baseline: the oracle on the host cores, bounded subsample of the same grid ----
    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:   # N = 1 only (bench contract)
        import oracle_lib
        om = oracle_lib.OracleMap3()
        t0 = time.perf_counter()
        for f in range(args.frames):
            om.update(replay.synthetic_depth(f), replay.IDENTITY_POSE)
        cpu_upd_ms = (time.perf_counter() - t0) * 1e3 / args.frames
        m = args.cpu_sample
        idx = np.linspace(0, args.grid - 1, m).round().astype(np.int64)
        sub = grid.reshape(args.grid, args.grid, args.grid, 3)[np.ix_(idx, idx, idx)].reshape(-1, 3)
        t0 = time.perf_counter()
        ro = om.test(sub)
        cpu_s = time.perf_counter() - t0
        cores = os.cpu_count() or 1
        cpu = {"value": sub.shape[0] / cpu_s, "unit": "points/s", "cores": cores, "kind": "port",
               "sample": "%d^3 subsample of the same %d^3 grid after the same %d frames (CPU oracle, %d threads); "
                         "update %.0f ms/frame" % (m, args.grid, args.frames, cores, cpu_upd_ms),
               "update_ms_per_frame": cpu_upd_ms}
        # sanity: the GPU result on the sample must match the oracle
        if world == 1:
            flat = (idx[:, None, None] * args.grid + idx[None, :, None]) * args.grid + idx[None, None, :]
            rg = res[torch.from_numpy(flat.reshape(-1)).to(dev)].cpu().numpy()
            fl = om.test_flags(sub)
            okm = (fl & 6) == 0
            cpu["sdf_rmse_vs_oracle"] = float(np.sqrt(np.mean((rg[okm, 0] - ro[okm, 0]) ** 2)))

    if rank == 0:
        n_pts = n_total * args.steps
        value = n_pts / elapsed
        tflops = (flops / 1e12) / (k4_ms / 1e3) if k4_ms > 0 else None
        # HBM bytes per K4 launch: measured with rocprofv3 PMC counters in separate passes of this command
        # (profiles/README.md); a committed measurement, not collected live
        traffic = None
        tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_k4_traffic.json")
        if os.path.exists(tpath) and args.grid == 256 and args.frames == 2:
            with open(tpath) as fh:
                traffic = json.load(fh).get("hbm_bytes_per_launch")
        out = {
            "metric": "sdf_test_points_per_sec",
            "value": value,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "synthetic 640x480 depth z=1+0.05 sin(6(u+0.01f)) cos(5v), %d frames fused, "
                                   "%d^3 SDF query grid over [-0.6,0.6]x[-0.45,0.45]x[0.85,1.15] m" % (args.frames, args.grid),
                       "frames": args.frames, "grid": args.grid,
                       "clusters": int(st0["clusters"]), "map_points": gm.num_points(),
                       "parallelism": "replicated map, query slabs per rank, RCCL gather" if world > 1 else "single GPU"},
            "update_ms_per_frame": float(np.mean(upd_ms)),
            "update_ms_frames": upd_ms,
            "gp_evals_per_point": evals / (hi - lo) / args.steps,
            "roofline": {"bound": "mfma", "achieved": tflops, "peak": 157.3, "unit": "TFLOP/s",
                         "frac": (tflops / 157.3) if tflops else None, "traffic": traffic,
                         "kernel": "ongpis_eval_kernel (K4)", "k4_ms_per_step": k4_ms / args.steps,
                         "k4_launches_per_step": launches / args.steps,
                         "algorithmic_flops_per_step": flops / args.steps,
                         "model_bytes": st0["model_bytes"]},
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
