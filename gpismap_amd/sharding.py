"""Multi-GPU layout of the hot path: one process per GPU (torch.distributed, backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in the CPU tests and the one-GPU rehearsal).

test(): queries are independent given the trained models.  The query array is dealt to the ranks in
BLOCKS of `block` consecutive queries, round-robin (`cyclic_blocks`): a 256^3 grid stored x-fastest is
a stack of z-slabs and the surface lives in a thin z range, so contiguous slabs would give the middle
ranks several GP evaluations per query and the outer ranks none; 64 K-query blocks (one x-y sheet each)
spread every z range over all ranks.  The 32 B/query results are assembled on rank 0 with ONE
point-to-point transfer per rank (`gather_blocks`; no all-reduce: xGMI is point-to-point, every rank
sends its results over its own link into rank 0 exactly once).  Per-query arithmetic does not depend
on the partition, so 1, 2, 4 and 8-rank results are bit-identical.

update(): two modes.  *Replicated* training: every rank runs the same update() and holds identical
models -- no exchange at all.  *Sharded* training (`GPisMap3.set_shard` + `exchange_models`): every
rank runs the same host logic but factorises only its share of the frame's clusters (greedy
longest-processing-time partition by K^3, computed identically on every rank), then the packed
models (what K4 needs: 2 K^2 + 20 K bytes each, records back to back at their own sizes) are
all-gathered in one collective over slots padded to the largest RANK total.
*Lead / worker* (`update_lead_worker`, round 6): sharded training with the host logic run ONCE -- rank 0 replays the
frame and broadcasts what it decided (the frame record), the other ranks apply the record instead of replaying.
`shard_clusters` is the same partition for harnesses that drive the kernel-level C-ABI themselves
(BASELINE config 5)."""
import heapq


# ---------------------------------------------------------------------------------- queries ----
def cyclic_blocks(n, world, rank, block=65536):
    """Block-cyclic cut: list of (lo, hi) query ranges owned by `rank` -- blocks rank, rank + world, ... of
    `block` consecutive queries (the last block may be short)."""
    nb = (n + block - 1) // block
    return [(b * block, min(n, (b + 1) * block)) for b in range(rank, nb, world)]


def local_count(n, world, rank, block=65536):
    return sum(hi - lo for lo, hi in cyclic_blocks(n, world, rank, block))


def take_blocks(x, world, rank, block=65536):
    """This rank's queries, concatenated in block order (x: [n, d] tensor or array)."""
    import numpy as np
    parts = [x[lo:hi] for lo, hi in cyclic_blocks(x.shape[0], world, rank, block)]
    if not parts:
        return x[:0]
    if isinstance(x, np.ndarray):
        return np.concatenate(parts, axis=0)
    import torch
    return torch.cat(parts, dim=0)


def gather_blocks(res_local, n, world, rank, dst=0, out=None, block=65536, staging=None):
    """Assemble the per-rank results of a block-cyclic cut on `dst`.  Every rank sends its whole local result
    ([local_count, C], contiguous) in ONE message; `dst` receives each into a staging buffer and scatters the blocks
    into the full [n, C] result with device-side copies.  Returns the full tensor on dst, None elsewhere.
    `staging`: optional preallocated [max local_count, C] buffer on dst (reused across calls)."""
    import torch
    import torch.distributed as dist
    if world == 1:
        if out is None:
            return res_local
        if out.data_ptr() != res_local.data_ptr():
            out.copy_(res_local)
        return out
    if rank != dst:
        if res_local.shape[0] > 0:          # a rank without blocks (fewer blocks than ranks) sends nothing
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, res_local, dst)]):
                w.wait()
        return None
    if out is None:
        out = torch.empty((n,) + tuple(res_local.shape[1:]), dtype=res_local.dtype, device=res_local.device)

    def scatter(src, r):
        o = 0
        for lo, hi in cyclic_blocks(n, world, r, block):
            out[lo:hi].copy_(src[o:o + hi - lo])
            o += hi - lo

    scatter(res_local, dst)
    counts = [local_count(n, world, r, block) for r in range(world)]
    bufs = {}
    ops = []
    for r in range(world):
        if r == dst or counts[r] == 0:
            continue
        bufs[r] = torch.empty((counts[r],) + tuple(res_local.shape[1:]), dtype=res_local.dtype, device=res_local.device)
        ops.append(dist.P2POp(dist.irecv, bufs[r], r))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for r, b in bufs.items():
        scatter(b, r)
    return out


# --------------------------------------------------------------------------------- training ----
def shard_clusters(costs, world):
    """Greedy LPT partition of cluster indices by cost (e.g. K^3), ties by index: returns `world` index lists.
    The same rule GPisMap3 applies internally after set_shard()."""
    heap = [(0.0, r) for r in range(world)]
    heapq.heapify(heap)
    out = [[] for _ in range(world)]
    for i in sorted(range(len(costs)), key=lambda k: (-costs[k], k)):
        load, r = heapq.heappop(heap)
        out[r].append(i)
        heapq.heappush(heap, (load + float(costs[i]), r))
    return out


def _all_gather_bytes(send, nbytes, world, rank, device, host_staged):
    """All-gather of per-rank byte buffers of DIFFERENT lengths nbytes[r] (uint8): one collective over slots padded to the
    largest rank total.  Returns a [world, max(nbytes)] uint8 tensor on `device` (row r valid up to nbytes[r])."""
    import torch
    import torch.distributed as dist
    mx = max(max(nbytes), 256)
    pad = torch.empty(mx, dtype=torch.uint8, device=device)
    if nbytes[rank]:
        pad[:nbytes[rank]].copy_(send[:nbytes[rank]])
    if host_staged:                               # gloo rehearsal: no device collectives
        torch.cuda.synchronize()
        hp = pad.cpu()
        outs = [torch.empty_like(hp) for _ in range(world)]
        dist.all_gather(outs, hp)
        return torch.stack(outs).to(device)
    out = torch.empty((world, mx), dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(out.view(-1), pad)
    return out


def _all_gather_records(send, counts, stride, world, rank, device, host_staged):
    """Equal-size records (kernel-level store API): counts[r] records of `stride` bytes per rank."""
    return _all_gather_bytes(send, [c * stride for c in counts], world, rank, device, host_staged)


def exchange_models(gm, world, rank, device, host_staged=False):
    """Complete a sharded GPisMap3.update(): pack the locally trained models (records back to back at their OWN sizes),
    all-gather the buffers (ONE RCCL all_gather_into_tensor over slots padded to the largest RANK total -- the K^3-balanced
    partition makes the totals near-equal; host-staged over gloo in the rehearsal), unpack the other ranks' models and
    build the cluster table.  Returns (clusters of the frame, record bytes received, bytes the collective delivered
    including the padding of the shorter ranks, sum of all ranks' record bytes)."""
    import torch
    if world == 1:
        gm.shard_finish()
        return 0, 0, 0, 0
    total, nloc, counts = gm.shard_info()
    if total == 0:
        gm.shard_finish()
        return 0, 0, 0, 0
    nbytes = [gm.shard_bytes(r) for r in range(world)]          # identical on every rank (sizes of all jobs are known everywhere)
    send = torch.empty(max(256, nbytes[rank]), dtype=torch.uint8, device=device)
    gm.shard_pack(send.data_ptr(), torch.cuda.current_stream().cuda_stream)
    allrec = _all_gather_bytes(send, nbytes, world, rank, device, host_staged)
    torch.cuda.synchronize()
    for r in range(world):
        if r != rank and nbytes[r]:
            gm.shard_unpack(r, allrec[r].data_ptr(), torch.cuda.current_stream().cuda_stream)
    gm.shard_finish()
    return total, sum(nbytes) - nbytes[rank], (world - 1) * max(nbytes), sum(nbytes)


def update_lead_worker(gm, depth, pose, world, rank, device, host_staged=False):
    """One frame of a sharded run with the HOST LOGIC OF update() RUN ONCE: rank 0 (the lead; `gm.set_frame_export()` done)
    replays the frame, broadcasts the frame record -- slot operations, point mirror, cell lists, jobs with owners, cluster
    table entries: a few hundred KB to a few MB -- and only then trains its own share; every other rank applies the
    record (`gm.apply_frame`: K6 on its own device + its share of the training) instead of replaying the frame.  Follow
    with `exchange_models`.  The record is host data on both ends: over nccl it makes one hop through device memory."""
    import numpy as np
    import torch
    import torch.distributed as dist
    if world == 1:
        gm.update(depth, pose)
        return 0
    dev = "cpu" if host_staged else device
    if rank == 0:
        gm.update(depth, pose)                      # host logic + record; the own share waits
        rec = gm.frame_record()
        n = torch.tensor([int(rec.size)], dtype=torch.int64, device=dev)
        dist.broadcast(n, src=0)
        payload = torch.from_numpy(rec if rec.size else np.zeros(1, dtype=np.uint8)).to(dev)
        dist.broadcast(payload, src=0)
        gm.train_deferred()
        return int(rec.size)
    n = torch.zeros(1, dtype=torch.int64, device=dev)
    dist.broadcast(n, src=0)
    payload = torch.empty(max(1, int(n.item())), dtype=torch.uint8, device=dev)
    dist.broadcast(payload, src=0)
    gm.apply_frame(payload.cpu().numpy()[:int(n.item())])
    return int(n.item())


def exchange_store_models(st, local_models, world, rank, device, host_staged=False):
    """Kernel-level variant (gpismap_amd.OnGPIS store): all-gather the packed records of `local_models` and unpack the
    other ranks' into new predict-only models.  Returns (list per rank of model ids valid on THIS rank, bytes received)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    n = len(local_models)
    if world == 1:
        return [np.asarray(local_models, dtype=np.int32)], 0
    meta = torch.tensor([n, st.packed_bytes(local_models) if n else 256], dtype=torch.int64)
    metas = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    if host_staged:
        dist.all_gather(metas, meta)
    else:
        m = meta.to(device)
        ms = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(world)]
        dist.all_gather(ms, m)
        metas = [t.cpu() for t in ms]
    counts = [int(t[0]) for t in metas]
    stride = max(int(t[1]) for t in metas)
    send = torch.empty(max(1, n) * stride, dtype=torch.uint8, device=device)
    if n:
        st.pack(local_models, send.data_ptr(), stride, torch.cuda.current_stream().cuda_stream)
    allrec = _all_gather_records(send, counts, stride, world, rank, device, host_staged)
    torch.cuda.synchronize()
    ids = []
    for r in range(world):
        if r == rank:
            ids.append(np.asarray(local_models, dtype=np.int32))
        elif counts[r]:
            ids.append(st.unpack(allrec[r].data_ptr(), counts[r], stride, None, torch.cuda.current_stream().cuda_stream))
        else:
            ids.append(np.zeros(0, dtype=np.int32))
    return ids, (sum(counts) - n) * stride
