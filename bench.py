#!/usr/bin/env python3
"""bench.py -- BASELINE metric on MI355X: SDF test points/sec (+ per-frame update ms) for 3-D
640x480 synthetic depth and a 256^3 query grid (SURVEY.md 8d, config 4, F = 5 frames).

A "step" is one GPisMap3 test() pass over the query grid, inputs already resident in HBM
(gpis3_test_device).  update() of the synthetic frames is timed during set-up and reported beside it: in the library's
default pipelined mode (`update_ms_per_frame` = (frames 2..F + the drain of the last frame's training) / (F - 1), median
repeat), synchronous as the reference's (`update_ms_per_frame_synchronous`, median of frames 2..F, with the per-phase
split and the K3 chain's own time) and synchronous with the eager inverse (`update_ms_per_frame_with_inverse`).

`python bench.py --gpus N` works as typed: for N > 1 the parent starts N ranks with
torch.distributed.run BEFORE it touches the GPU and relays rank 0's JSON line; launched by torchrun
itself (RANK/WORLD_SIZE set) it is one rank.  One process per GPU over RCCL: the grid is dealt to the
ranks in 64 K-query blocks round-robin (strong scaling), every rank evaluates its blocks, the results
are assembled on rank 0 with one transfer per rank inside the timed region.  Training is replicated
(--train replicated, default: no exchange) or sharded (--train sharded: each rank factorises its
share of the frame's clusters, the packed models are all-gathered).

Prints ONE JSON line on rank 0 carrying `roofline` (dominant kernel = K4 ongpis_eval_kernel,
MFMA/FLOP-bound; achieved = algorithmic flops / time inside the K4 launches measured with HIP events
on the launch stream), `cpu_baseline` (CPU oracle on the host cores, rank 0, N = 1, one pass over a
bounded 64^3 subsample, plus the parity records: own-map and SAME-MAP comparisons) and the sub-records `update_roofline` (K3 on the frame's clusters), `stress`
(BASELINE config 5: 50 000 clusters x 64 points, train + predict) and `value_host_api` (the same pass
through the host-pointer API the mex gateway uses, PCIe included).
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=5, help="synthetic depth frames fused before testing (SURVEY 8d: F = 5)")
    ap.add_argument("--grid", type=int, default=256, help="query grid is grid^3 points")
    ap.add_argument("--update-repeats", type=int, default=5, help="fuse the frame sequence this many times (fresh maps); headline = median of frames 2..F of the MEDIAN repeat (the minimum over the repeats is a side field)")
    ap.add_argument("--train", default="replicated", choices=["replicated", "sharded", "lead"], help="multi-rank update(): every rank trains everything, or its K^3-balanced share + all-gather of the models; lead = sharded with the host logic run once on rank 0 (frame records broadcast, the other ranks apply them)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, the measured path) or gloo (rehearsal of the multi-rank logic on fewer GPUs: transfers staged through host memory)")
    ap.add_argument("--block", type=int, default=65536, help="queries per block of the block-cyclic cut")
    ap.add_argument("--cpu-sample", type=int, default=64, help="CPU baseline runs on a sample^3 subgrid (0 = skip); SURVEY 8(d): 64^3")
    ap.add_argument("--stress", type=int, default=50000, help="clusters of the stress sub-record (0 = skip)")
    ap.add_argument("--no-host-api", action="store_true", help="skip the value_host_api pass")
    ap.add_argument("--verify", action="store_true", help="multi-rank runs: after the timed region, check the assembled map bit for bit against a single-rank pass of rank 0 over a 64^3 sub-grid (any backend), and report per-rank pass / gather / exchange ms")
    return ap.parse_args()


def self_launch(args):
    """N > 1 without a launcher: start the ranks as children of a process that has not touched the GPU."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env)
    sys.exit(r.returncode)


def file_sha(path):
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)

    import numpy as np
    import torch
    import torch.distributed as dist
    import gpismap_amd
    from gpismap_amd import sharding
    import replay

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available() or gpismap_amd.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: gpismap_amd has no CPU fallback")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()      # rehearsal: several ranks may share a GPU
    torch.cuda.set_device(local_rank)
    gpismap_amd.set_device(local_rank)                           # the library selects its device explicitly
    dev = torch.device("cuda", local_rank)
    host_staged = args.backend == "gloo"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if host_staged:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    sharded = world > 1 and args.train in ("sharded", "lead")
    lead_mode = world > 1 and args.train == "lead"       # host logic of update() once, on rank 0 (gpismap_amd.sharding.update_lead_worker)
    host_replays = frame_record_bytes = 0

    # ---- set-up (untimed): fuse the synthetic frames; time update() per frame ----
    # The host side of update() (single-threaded tree replay) varies by +-30 % from box to box and run to run, so the
    # sequence is fused --update-repeats times into fresh maps and every frame is reported at its MINIMUM over the repeats
    # (`update_repeats`, `update_ms_frames_all`); the last map is the one the test passes run on.
    reps = max(1, args.update_repeats)
    upd_all, ph_all, k3_all = [], [], []
    exch_bytes = exch_padded = exch_records = 0
    exch_ms = []
    gm = None
    # Two update() accountings, reps fusions each: first with the EAGER inverse (K3b behind K3 inside every
    # update(): what a caller that tests after every update pays per frame), then the default LAZY inverse (update() stops at
    # the factor and alpha like OnGPIS::train; the inverse runs once at the first test()).  The last (lazy) map is the one tested.
    for rep in range(2 * reps):
        eager = rep < reps
        if gm is not None:
            del gm
        gm = gpismap_amd.GPisMap3()          # default camera 640x480, fx=fy=568, cx=310, cy=224
        assert gm.device() == local_rank
        gm.set_profile(True)                 # hipEvents around the K4 / K3 launches
        gm.set_pipeline(False)               # synchronous update(), as the reference's: per-frame times, phases and the K3 chain's own time are read after every frame
        gm.set_lazy_inverse(not eager)
        if sharded:
            gm.set_shard(rank, world)
        if lead_mode and rank == 0:
            gm.set_frame_export(True)
        upd_ms, phases, k3 = [], [], []
        host_replays = frame_record_bytes = 0
        exch_bytes = exch_padded = exch_records = 0
        for f in range(args.frames):
            depth = replay.synthetic_depth(f)
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            if lead_mode:
                frame_record_bytes += sharding.update_lead_worker(gm, depth, replay.IDENTITY_POSE, world, rank, dev, host_staged)
            else:
                gm.update(depth, replay.IDENTITY_POSE)
            if sharded:
                te = time.perf_counter()
                _, nb, npad, nrec = sharding.exchange_models(gm, world, rank, dev, host_staged)
                exch_bytes += nb; exch_padded += npad; exch_records += nrec
                exch_ms.append((time.perf_counter() - te) * 1e3)
            upd_ms.append((time.perf_counter() - t0) * 1e3)
            s = gm.stats()
            host_replays += int(s["host_replays"])
            phases.append([s["upd_preproc_ms"], s["upd_obsgp_train_ms"], s["upd_reeval_ms"], s["upd_eval_ms"], s["upd_gps_ms"]])
            k3.append(dict(ms=s["last_train_ms"], flops=s["last_train_flops"], bytes=s["last_train_bytes"],
                           clusters=int(s["last_train_jobs"]), maxK=int(s["last_train_maxK"])))
        upd_all.append(upd_ms)
        ph_all.append(phases)
        k3_all.append(k3)
    med_tail = lambda a: float(np.median(a[1:] if len(a) > 1 else a))

    def median_repeat(lo, hi):
        """index of the repeat in [lo, hi) whose median over frames 2..F is the median among the repeats (upper median)"""
        order = sorted(range(lo, hi), key=lambda r: med_tail(upd_all[r]))
        return order[len(order) // 2]
    r_eager, r_lazy = median_repeat(0, reps), median_repeat(reps, 2 * reps)
    upd_ms, phases, k3 = upd_all[r_lazy], ph_all[r_lazy], k3_all[r_lazy]
    upd_ms_eager, k3_eager = upd_all[r_eager], k3_all[r_eager]
    upd_min = [min(u[f] for u in upd_all[reps:]) for f in range(args.frames)]
    # Lazy inverse (the default): update() trained factors and alpha; the explicit inverses of the clusters retrained since the
    # last prediction are computed by the first test() -- timed here on its own and reported (`deferred_inverse_ms`), so that
    # the timed passes below start from the same state as with the eager inverse.
    t0 = time.perf_counter()
    gm.prepare_test()
    deferred_ms = (time.perf_counter() - t0) * 1e3
    st0 = gm.stats()
    # Pipelined mode (the library's default): update() returns once the frame's training is enqueued and the next update() / test() joins it
    # (include/gpismap_amd.h, gpis3_set_pipeline / gpis3_sync).  A second map fuses the same frames that way, nothing is read
    # between the frames, and the drain of the last frame's training is timed and CHARGED: per frame = (frames 2..F + drain) / (F-1).
    upd_pipe, drain_ms = [], 0.0
    pipe_all = []
    if not sharded:
        for rep in range(reps):
            gp = gpismap_amd.GPisMap3()
            gp.set_pipeline(True)
            up = []
            for f in range(args.frames):
                depth = replay.synthetic_depth(f)
                if world > 1:
                    dist.barrier()
                t0 = time.perf_counter()
                gp.update(depth, replay.IDENTITY_POSE)
                up.append((time.perf_counter() - t0) * 1e3)
            t0 = time.perf_counter()
            gp.sync()
            dr = (time.perf_counter() - t0) * 1e3
            assert gp.num_points() == gm.num_points(), "pipelined and synchronous update() disagree on the map"
            del gp
            pipe_all.append(((sum(up[1:]) + dr) / max(1, len(up) - 1), up, dr))
        order = sorted(range(len(pipe_all)), key=lambda r: pipe_all[r][0])
        _, upd_pipe, drain_ms = pipe_all[order[len(order) // 2]]          # the MEDIAN repeat (upper median)
    upd_pipe_mean = ((sum(upd_pipe[1:]) + drain_ms) / (len(upd_pipe) - 1)) if len(upd_pipe) > 1 else None

    n_total = args.grid ** 3
    grid = replay.synthetic_grid(args.grid)                    # [n,3] float32, x fastest
    x = torch.from_numpy(sharding.take_blocks(grid, world, rank, args.block)).to(dev)   # resident in HBM before timing
    n_loc = x.shape[0]
    res = torch.zeros((n_loc, 8), dtype=torch.float32, device=dev)
    full = torch.zeros((n_total, 8), dtype=torch.float32, device=dev) if (world > 1 and rank == 0) else None
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        gm.test_device(x.data_ptr(), n_loc, res.data_ptr(), stream)
        if world > 1:
            if host_staged:                # rehearsal path: host staging (gloo has no device P2P)
                torch.cuda.synchronize()
                got = sharding.gather_blocks(res.cpu(), n_total, world, rank, dst=0, block=args.block)
                if rank == 0:
                    full.copy_(got)
            else:
                sharding.gather_blocks(res, n_total, world, rank, dst=0, out=full, block=args.block)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    k4_ms = flops = launches = evals = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        s = gm.stats()
        k4_ms += s["last_test_k4_ms"]; flops += s["last_test_flops"]; launches += s["last_test_k4_launches"]
        evals += s["last_test_evals"]
    barrier()
    elapsed = time.perf_counter() - t0
    evals_rank = evals / max(1, args.steps)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if host_staged else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        ev = torch.tensor([evals_rank, k4_ms / max(1, args.steps), float(host_replays)], dtype=torch.float64, device="cpu" if host_staged else dev)
        evs = [torch.zeros_like(ev) for _ in range(world)]
        dist.all_gather(evs, ev)
        per_rank = [[float(v) for v in e.cpu()] for e in evs]
    else:
        per_rank = [[evals_rank, k4_ms / max(1, args.steps), float(host_replays)]]

    # ---- --verify (any backend, outside the timed region): per-rank phase times of one extra step, and the assembled map
    # against a single-rank pass of rank 0 over a 64^3 sub-grid, bit for bit
    detail = None
    if args.verify and world > 1:
        barrier()
        t0 = time.perf_counter()
        gm.test_device(x.data_ptr(), n_loc, res.data_ptr(), stream)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if host_staged:
            got = sharding.gather_blocks(res.cpu(), n_total, world, rank, dst=0, block=args.block)
            if rank == 0:
                full.copy_(got)
        else:
            sharding.gather_blocks(res, n_total, world, rank, dst=0, out=full, block=args.block)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ph = torch.tensor([(t1 - t0) * 1e3, (t2 - t1) * 1e3, float(np.mean(exch_ms)) if exch_ms else 0.0], dtype=torch.float64,
                          device="cpu" if host_staged else dev)
        phs = [torch.zeros_like(ph) for _ in range(world)]
        dist.all_gather(phs, ph)
        detail = {"pass_ms": [float(p[0]) for p in phs], "gather_ms": [float(p[1]) for p in phs], "exchange_ms_per_frame": [float(p[2]) for p in phs]}
        if rank == 0:
            m = 64
            idx = np.linspace(0, args.grid - 1, m).round().astype(np.int64)
            flat = ((idx[:, None, None] * args.grid + idx[None, :, None]) * args.grid + idx[None, None, :]).reshape(-1)
            xs = torch.from_numpy(grid[flat]).to(dev)
            ref = torch.zeros((flat.size, 8), dtype=torch.float32, device=dev)
            gm.test_device(xs.data_ptr(), flat.size, ref.data_ptr(), stream)
            torch.cuda.synchronize()
            same = bool(torch.equal(ref.view(torch.int32), full[torch.from_numpy(flat).to(dev)].view(torch.int32)))
            detail["assembled_equals_single_rank_on_64cubed"] = same
            print("verify (%d ranks, %s training, %s): assembled map identical to rank 0's single-rank pass on a 64^3 sub-grid: %s"
                  % (world, args.train, args.backend, same), file=sys.stderr)
            if not same:
                raise SystemExit("multi-rank assembly differs from the single-rank result")

    if host_staged and world > 1 and rank == 0:
        # rehearsal only: the assembled map must equal a single-rank pass over the whole grid, bit for bit
        xf = torch.from_numpy(grid).to(dev)
        ref = torch.zeros((n_total, 8), dtype=torch.float32, device=dev)
        gm.test_device(xf.data_ptr(), n_total, ref.data_ptr(), stream)
        torch.cuda.synchronize()
        same = bool(torch.equal(ref.view(torch.int32), full.view(torch.int32)))
        print("rehearsal (%d ranks, %s training, gloo staging): assembled map identical to a single-rank pass: %s"
              % (world, args.train, same), file=sys.stderr)
        if not same:
            raise SystemExit("multi-rank assembly differs from the single-rank result")
        del xf, ref

    # ---- value_host_api: the same pass through gpis3_test with HOST pointers (what the mex gateway calls) ----
    host_api = None
    if rank == 0 and world == 1 and not args.no_host_api:
        hres = np.zeros((n_total, 8), dtype=np.float32)
        gm.test(grid, hres)                                    # warm the staging buffers
        t0 = time.perf_counter()
        gm.test(grid, hres)
        host_api = n_total / (time.perf_counter() - t0)
        del hres

    # ---- stress sub-record: BASELINE config 5 through the kernel-level C-ABI ----
    stress = None
    if args.stress > 0:
        stress = run_stress(args, world, rank, dev, host_staged)

    # ---- CPU baseline: the oracle on the host cores, bounded subsample of the same grid ----
    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:   # N = 1 only (bench contract)
        import oracle_lib
        # timed in the oracle's "natural" arithmetic = the reference's own algorithm (Cholesky + substitution per query,
        # OnGPIS.cpp:177-216); the "tiled" mode carries an explicit inverse that only exists for the GPU formulation
        oracle_lib.set_arith_mode("natural")
        om = oracle_lib.OracleMap3()
        cu = []
        for f in range(args.frames):
            t0 = time.perf_counter()
            om.update(replay.synthetic_depth(f), replay.IDENTITY_POSE)
            cu.append((time.perf_counter() - t0) * 1e3)
        m = args.cpu_sample
        idx = np.linspace(0, args.grid - 1, m).round().astype(np.int64)
        sub = grid.reshape(args.grid, args.grid, args.grid, 3)[np.ix_(idx, idx, idx)].reshape(-1, 3)
        t0 = time.perf_counter()
        ro = om.test(sub)                                     # ONE pass (64^3 = 262 144 points: ~0.5 minute of host work)
        cpu_s = time.perf_counter() - t0
        cores = os.cpu_count() or 1
        fl = om.test_flags(sub)
        oracle_lib.set_arith_mode("tiled")
        cpu = {"value": sub.shape[0] / cpu_s, "unit": "points/s", "cores": cores, "kind": "port",
               "sample": "%d^3 subsample of the same %d^3 grid after the same %d frames (CPU oracle in its natural-order arithmetic = the "
                         "reference's substitution algorithm, %d threads), one pass; update median of frames 2..%d"
                         % (m, args.grid, args.frames, cores, args.frames),
               "update_ms_per_frame": float(np.median(cu[1:] if len(cu) > 1 else cu)), "update_ms_frames": cu,
               "test_s": cpu_s}
        flat = (idx[:, None, None] * args.grid + idx[None, :, None]) * args.grid + idx[None, None, :]
        rg = res[torch.from_numpy(flat.reshape(-1)).to(dev)].cpu().numpy()
        amb = (fl & 6) != 0
        # (a) against the natural-order run just timed: an independent summation order ON ITS OWN MAP -- after F frames the two
        # maps differ (point positions / noises come from ObsGP results in that arithmetic; single decisions flip), so this
        # figure mixes arithmetic with map divergence and is reported for continuity only
        d0 = rg[:, 0].astype(np.float64) - ro[:, 0]
        cpu["own_map_natural"] = {"sdf_rmse_all": float(np.sqrt(np.mean(d0 ** 2))), "sdf_rmse_unmasked": float(np.sqrt(np.mean(d0[~amb] ** 2))),
                                  "map_points": om.num_points(), "map_points_gpu": gm.num_points(),
                                  "map_point_count_difference": int(om.num_points() - gm.num_points()),
                                  "branch_ambiguous_in_sample": int(amb.sum())}
        # (a') the NOISE FLOOR of that figure: two more independent CPU orders, each building its OWN map
        # over the same frames, compared with each other and with the GPU on the 24^3 sub-sample.  If X<->Y is not below 1e-5,
        # no fp32 implementation can meet 1e-5 end to end; what is asserted (tests/test_gpu_golden.py) is GPU<->X <= 1.25 max(X<->Y).
        i3 = np.linspace(0, m - 1, 24).round().astype(np.int64)
        pk3 = ((i3[:, None, None] * m + i3[None, :, None]) * m + i3[None, None, :]).reshape(-1)
        own = {"natural": ro[pk3]}
        amb_f = amb[pk3]
        for mode in ("fp64acc", "eigen33"):
            oracle_lib.set_arith_mode(mode)
            o2 = oracle_lib.OracleMap3()
            for f in range(args.frames):
                o2.update(replay.synthetic_depth(f), replay.IDENTITY_POSE)
            own[mode] = o2.test(sub[pk3])
            own[mode + "_points"] = o2.num_points()
            o2.close()
        oracle_lib.set_arith_mode("tiled")
        rms = lambda a, b: [float(np.sqrt(np.mean((a[~amb_f, 0].astype(np.float64) - b[~amb_f, 0]) ** 2))), float(np.sqrt(np.mean((a[:, 0].astype(np.float64) - b[:, 0]) ** 2)))]
        fl_ = {"floor_natural_vs_fp64acc": rms(own["natural"], own["fp64acc"]), "floor_natural_vs_eigen33": rms(own["natural"], own["eigen33"]),
               "floor_fp64acc_vs_eigen33": rms(own["fp64acc"], own["eigen33"]),
               "gpu_vs_natural": rms(rg[pk3], own["natural"]), "gpu_vs_fp64acc": rms(rg[pk3], own["fp64acc"]), "gpu_vs_eigen33": rms(rg[pk3], own["eigen33"]),
               "map_points": {"gpu": gm.num_points(), "natural": om.num_points(), "fp64acc": own["fp64acc_points"], "eigen33": own["eigen33_points"]},
               "sample": "24^3 sub-sample (%d queries, %d branch-ambiguous); pairs are SDF RMSE [unmasked, all], every order on its OWN map" % (pk3.size, int(amb_f.sum()))}
        fl_["max_floor_unmasked"] = max(fl_[k][0] for k in ("floor_natural_vs_fp64acc", "floor_natural_vs_eigen33", "floor_fp64acc_vs_eigen33"))
        fl_["max_gpu_unmasked"] = max(fl_[k][0] for k in ("gpu_vs_natural", "gpu_vs_fp64acc", "gpu_vs_eigen33"))
        fl_["gpu_within_1p25_of_floor"] = bool(fl_["max_gpu_unmasked"] <= 1.25 * fl_["max_floor_unmasked"])
        cpu["own_map_natural"].update({"floor_natural_vs_fp64acc": fl_["floor_natural_vs_fp64acc"][0], "floor_natural_vs_eigen33": fl_["floor_natural_vs_eigen33"][0]})
        cpu["own_map_noise_floor"] = fl_
        import eigen_probe
        einc, ever = eigen_probe.find_eigen()
        cpu["eigen_on_box"] = bool(einc)
        cpu["eigen_note"] = ("real Eigen %s at %s: run tests/test_eigen_probe.py" % (ever, einc)) if einc else ("no real Eigen on this box (%s): the eigen33 order stays a restatement from memory, parity unpinned" % ever)
        # (b) the tiled-order oracle holds the GPU's map exactly: bit-exactness on a 16^3 sub-sample, then the ARITHMETIC
        # comparison: every cluster of that same map re-factorised in the natural order (and the Eigen-3.3 order) on its
        # stored training set, test() on a 24^3 sub-sample
        ot = oracle_lib.OracleMap3()
        for f in range(args.frames):
            ot.update(replay.synthetic_depth(f), replay.IDENTITY_POSE)
        i2 = np.linspace(0, m - 1, 16).round().astype(np.int64)
        pick = ((i2[:, None, None] * m + i2[None, :, None]) * m + i2[None, None, :]).reshape(-1)
        rt = ot.test(sub[pick])
        cpu["sdf_rmse_vs_oracle"] = float(np.sqrt(np.mean((rg[pick, 0].astype(np.float64) - rt[:, 0]) ** 2)))
        cpu["identical_rows_vs_oracle"] = float(np.mean(np.all(rg[pick] == rt, axis=1)))
        cpu["map_points_oracle"] = ot.num_points()
        cpu["map_nodes_identical_to_gpu"] = bool(np.array_equal(ot.nodes(), gm.nodes()))
        fl3 = ot.test_flags(sub[pk3]); amb3 = (fl3 & 6) != 0
        for mode in ("natural", "eigen33"):
            ot.retrain_all(mode)
            rn = ot.test(sub[pk3])
            dn = rg[pk3, 0].astype(np.float64) - rn[:, 0]
            cpu["sdf_rmse_same_map_%s" % mode] = float(np.sqrt(np.mean(dn ** 2)))
            cpu["sdf_rmse_same_map_%s_unmasked" % mode] = float(np.sqrt(np.mean(dn[~amb3] ** 2)))
            cpu["sdf_max_same_map_%s_unmasked" % mode] = float(np.abs(dn[~amb3]).max())
            cpu["var_g_rel_same_map_%s" % mode] = float((np.abs(rg[pk3, 5:8] - rn[:, 5:8])[~amb3] / 1875.0).max())
        cpu["same_map_sample"] = "24^3 sub-sample (%d queries, %d branch-ambiguous)" % (pk3.size, int(amb3.sum()))

    if rank == 0:
        n_pts = n_total * args.steps
        value = n_pts / elapsed
        tflops = (flops / 1e12) / (k4_ms / 1e3) if k4_ms > 0 else None
        # HBM bytes per K4 launch: rocprofv3 PMC passes of this command (tools/measure_traffic.sh), valid only for the
        # kernel source they were measured on
        traffic = traffic_step = None
        tpath = next((q for q in (os.path.join(ROOT, "profiles", "%s_k4_traffic.json" % r) for r in ("r06", "r05", "r04", "r03", "r02")) if os.path.exists(q)), "")
        if tpath:
            with open(tpath) as fh:
                tj = json.load(fh)
            if (tj.get("ongpis_test_sha") == file_sha(os.path.join(ROOT, "gpismap_amd", "csrc", "ongpis_test.hip"))
                    and tj.get("grid") == args.grid and tj.get("frames") == args.frames):
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_step = tj.get("hbm_bytes_per_pass")
        med = med_tail
        ph = np.array(phases)
        ksel = k3[1:] if len(k3) > 1 else k3
        k3_ms = float(np.median([k["ms"] for k in ksel]))
        k3_fl = float(np.median([k["flops"] for k in ksel]))
        ksel_e = k3_eager[1:] if len(k3_eager) > 1 else k3_eager
        k3e_ms = float(np.median([k["ms"] for k in ksel_e]))
        k3e_fl = float(np.median([k["flops"] for k in ksel_e]))
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
                       "parallelism": ("%s training, query blocks of %d dealt round-robin to %d ranks, RCCL point-to-point gather"
                                       % (args.train, args.block, world)) if world > 1 else "single GPU"},
            # headline = the library's default mode (pipelined since round 4); sharded multi-rank runs are synchronous (every frame ends with the exchange)
            "update_ms_per_frame": upd_pipe_mean if upd_pipe_mean is not None else med(upd_ms),
            "update_mode": ("default mode: pipelined update() (returns once the frame's training is enqueued, the next training / test() / gpis3_sync joins it), lazy inverse; "
                            "= (wall time of update() of frames 2..F + the drain of the last frame's training, gpis3_sync) / (F - 1), MEDIAN repeat of update_repeats fusions; "
                            "nothing is skipped: the drain is charged") if upd_pipe_mean is not None else "synchronous (sharded training: every frame ends with the model exchange)",
            "update_ms_per_frame_synchronous": med(upd_ms),
            "update_synchronous_mode": "gpis3_set_pipeline(map, 0) / GPIS_PIPELINE_UPDATE=0: every update() joins its own training before it returns (the reference's behaviour), lazy inverse: median of frames 2..F of the MEDIAN repeat",
            "update_ms_per_frame_with_inverse": med(upd_ms_eager),
            "update_with_inverse_mode": "synchronous with the eager inverse (gpis3_set_lazy_inverse(map, 0) / GPIS_EAGER_INVERSE=1: K3b inside every update()) -- what a test-after-every-update caller pays per frame; median repeat",
            "update_ms_per_frame_min_over_repeats": med(upd_min),
            "update_ms_frames": upd_ms,
            "update_ms_frames_with_inverse": upd_ms_eager,
            "update_repeats": reps,
            "update_ms_frames_all": {"eager_inverse": upd_all[:reps], "lazy_inverse": upd_all[reps:]},
            "update_ms_per_frame_pipelined": upd_pipe_mean,
            "update_pipelined": {"note": "the default mode: (frames 2..F + drain) / (F - 1) of the median repeat; all repeats listed",
                                 "ms_frames": upd_pipe, "drain_ms": drain_ms,
                                 "all_repeats": [{"ms_per_frame": a, "ms_frames": u, "drain_ms": d} for a, u, d in pipe_all]},
            "deferred_inverse_ms": deferred_ms,
            "deferred_inverse": {"clusters": int(st0["last_inverse_jobs"]), "device_ms": st0["last_inverse_ms"],
                                 "note": "lazy inverse (default): the explicit inverses K4 multiplies with are computed once, at the first "
                                         "test() after the updates, for the clusters retrained since the last prediction; with a test() after "
                                         "every update() add this to the per-frame update time (GPIS_EAGER_INVERSE=1 restores that split)"},
            "update_phases_ms": dict(zip(["preproc", "obsgp_train", "reeval_points", "new_points", "update_gps"],
                                         [float(v) for v in (np.median(ph[1:], axis=0) if len(ph) > 1 else ph[0])])),
            "gp_evals_per_point": sum(p[0] for p in per_rank) / n_total,
            # host_replays: executions of update()'s host logic per rank over the last fusion (F per rank when every rank replays; --train lead: F on rank 0, 0 elsewhere)
            "per_rank": dict({"gp_evals": [p[0] for p in per_rank], "k4_ms_per_step": [p[1] for p in per_rank], "host_replays": [int(p[2]) for p in per_rank]}, **(detail or {})),
            "frame_record_bytes_per_frame": (frame_record_bytes / max(1, args.frames)) if lead_mode else 0,
            "roofline": {"bound": "mfma", "achieved": tflops, "peak": 157.3, "unit": "TFLOP/s",
                         "frac": (tflops / 157.3) if tflops else None, "traffic": traffic,
                         # `traffic` = HBM-side bytes per K4 LAUNCH (the unit the contract names; a step is k4_launches_per_step launches),
                         # `traffic_per_step` the same per 256^3 pass -- the unit of everything else in this line
                         "traffic_unit": "HBM-side bytes per K4 launch (rocprofv3 PMC passes, FETCH_SIZE x 2 + WRITE_SIZE); traffic_per_step = per test() pass",
                         "traffic_per_step": traffic_step,
                         # over the algorithmic bytes of the same unit.  bench's denominator: 44 B per query + 32 B per evaluated (query, cluster)
                         # pair + the models once; SURVEY 8(d)'s: 44 B per query + the models once
                         "traffic_ratio": (traffic / ((44.0 * n_loc + 32.0 * evals / max(1, args.steps) + st0["model_bytes"]) / max(1.0, launches / args.steps))) if traffic else None,
                         "traffic_ratio_survey_8d": (traffic_step / (44.0 * n_loc + st0["model_bytes"])) if traffic_step else None,
                         "algorithmic_bytes_per_step": 44.0 * n_loc + 32.0 * evals / max(1, args.steps) + st0["model_bytes"],
                         "algorithmic_bytes_per_step_survey_8d": 44.0 * n_loc + st0["model_bytes"],
                         "kernel": "ongpis_eval_kernel (K4)", "k4_ms_per_step": k4_ms / args.steps,
                         "k4_launches_per_step": launches / args.steps,
                         "algorithmic_flops_per_step": flops / args.steps,
                         "model_bytes": st0["model_bytes"]},
            "update_roofline": {"kernel": "K6 gather + kernel build + K3 Cholesky (K3b inverse: deferred to the first test(), see deferred_inverse)", "bound": "mfma",
                                "ms_per_frame": k3_ms, "algorithmic_flops_per_frame": k3_fl,
                                "achieved": (k3_fl / 1e12) / (k3_ms / 1e3) if k3_ms > 0 else None, "peak": 157.3, "unit": "TFLOP/s",
                                "with_inverse": {"ms_per_frame": k3e_ms, "achieved": (k3e_fl / 1e12) / (k3e_ms / 1e3) if k3e_ms > 0 else None,
                                                 "note": "eager mode: K6 + build + K3 + K3b per frame; same flop count (the inverse's K^3/3 is NOT counted)"},
                                "clusters": ksel[-1]["clusters"], "max_K": max(k["maxK"] for k in ksel),
                                "note": "flops = sum K^3/3 + 2 K^2 over the clusters retrained per frame (SURVEY 8d); the explicit inverse (another K^3/3, never counted) runs once at the first test() after the updates: deferred_inverse"},
            # sharded training: bytes of model records rank 0 receives per frame (records at their own sizes, back to back), what the
            # one all_gather_into_tensor delivers (slots padded to the largest RANK total), and the sum of all records of a frame
            "exchange_bytes_per_frame": (exch_bytes / max(1, args.frames)) if sharded else 0,
            "exchange_bytes_per_frame_delivered": (exch_padded / max(1, args.frames)) if sharded else 0,
            "exchange_record_bytes_per_frame": (exch_records / max(1, args.frames)) if sharded else 0,
            "value_host_api": host_api,
            "stress": stress,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_stress(args, world, rank, dev, host_staged):
    """BASELINE config 5: NCL clusters x 64 points (K = 256).  Training is sharded over the ranks (contiguous cluster
    ranges: all clusters cost the same), the packed models are all-gathered, every rank then predicts 64 queries per
    cluster for the clusters of its RIGHT neighbour (so predictions run on exchanged models).  Returns the record on rank 0."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import gpismap_amd
    from gpismap_amd import sharding
    import replay
    from test_gpu_ongpis import soa9
    ncl = args.stress
    rng = np.random.default_rng(355)
    pos, grad, val, sx, sg = replay.stress_clusters(ncl, rng)
    xq_all = replay.stress_queries(pos, ncl, 64, rng)
    lo, hi = (ncl * rank) // world, (ncl * (rank + 1)) // world
    st = gpismap_amd.OnGPIS(3, 0.04)
    sl = slice(lo * 64, hi * 64)
    off = (np.arange(hi - lo + 1) * 64).astype(np.int32)
    ids = np.arange((hi - lo) * 64, dtype=np.int32)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    models = st.train(soa9(3, pos[sl], grad[sl], val[sl], sx[sl], sg[sl]), off, ids)
    train_wall = (time.perf_counter() - t0) * 1e3
    train_ms = st.last_ms()[0]
    t0 = time.perf_counter()
    ids_by_rank, nbytes = sharding.exchange_store_models(st, models, world, rank, dev, host_staged)
    torch.cuda.synchronize()
    exch_ms = (time.perf_counter() - t0) * 1e3 if world > 1 else 0.0
    nb = (rank + 1) % world
    nlo, nhi = (ncl * nb) // world, (ncl * (nb + 1)) // world
    xq = xq_all[nlo * 64:nhi * 64]
    jq = np.arange(xq.shape[0], dtype=np.int32)
    jm = np.repeat(ids_by_rank[nb], 64).astype(np.int32)
    out = st.eval(xq, jq, jm)
    out = st.eval(xq, jq, jm)
    pred_ms = st.last_ms()[1]
    finite = bool(np.all(np.isfinite(out)))
    K = 256.0
    t = torch.tensor([train_ms, exch_ms, pred_ms, train_wall], dtype=torch.float64)
    if world > 1:
        tt = t.clone() if host_staged else t.to(dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t = tt.cpu()
    train_ms, exch_ms, pred_ms, train_wall = [float(v) for v in t]
    if rank != 0:
        return None
    tr_flops = ncl * (K ** 3 / 3 + 2 * K * K)
    tr_bytes = ncl * (36.0 * 64 + 4.0 * (K * (K + 1) / 2 + K))
    pr_flops = ncl * 64 * (4.0 * K * K + 8.0 * K + 25.0 * 64)
    return {"workload": "BASELINE config 5: %d clusters x 64 points, K = 256, default_rng(355); %d rank(s), training sharded by cluster range, "
                        "packed models all-gathered, 64 queries per cluster" % (ncl, world),
            "clusters": ncl, "train_ms": train_ms, "train_tflops": tr_flops / 1e12 / (train_ms / 1e3),
            "train_gbs_algorithmic": tr_bytes / 1e9 / (train_ms / 1e3), "train_wall_ms_incl_allocation": train_wall,
            "exchange_ms": exch_ms, "exchange_bytes_received_per_rank": nbytes,
            "predict_ms": pred_ms, "predict_evaluations": ncl * 64, "predict_tflops": pr_flops / 1e12 / (pred_ms / 1e3),
            "finite": finite}


if __name__ == "__main__":
    main()
