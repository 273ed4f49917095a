"""Multi-GPU layout of the hot path: one process per GPU (torch.distributed, backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

test(): queries are independent given the trained models, so the query array is cut into
contiguous slabs, one per rank, and the 32 B/query results are assembled on rank 0 with ONE gather
step (no all-reduce: xGMI is point-to-point, every rank sends its slab over its own link into
rank 0 exactly once).
train: clusters are independent given the point set; `shard_clusters` balances them by their
K^3 cost (greedy longest-processing-time) -- used when training is sharded."""
import heapq


def slab_bounds(n, world, rank):
    """Contiguous slab [lo, hi) of n queries owned by `rank`."""
    return (n * rank) // world, (n * (rank + 1)) // world


def gather_slabs(res_local, n, world, rank, dst=0, out=None):
    """Assemble the per-rank result slabs on `dst` with point-to-point transfers straight into
    the destination buffer (slabs may differ in length by one, so a fixed-size gather does not
    fit; on RCCL this is one send per rank over its own xGMI link into rank `dst`).
    res_local: [hi-lo, C] tensor on this rank's device.  `out` (dst only, optional): preallocated
    [n, C] buffer -- if res_local already is out[lo:hi] nothing is copied locally.
    Returns the full [n, C] tensor on dst, None elsewhere."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return res_local
    if rank == dst:
        if out is None:
            out = torch.empty((n,) + tuple(res_local.shape[1:]), dtype=res_local.dtype, device=res_local.device)
        lo, hi = slab_bounds(n, world, rank)
        if out[lo:hi].data_ptr() != res_local.data_ptr():
            out[lo:hi].copy_(res_local)
        ops = []
        for r in range(world):
            if r == dst:
                continue
            lo, hi = slab_bounds(n, world, r)
            ops.append(dist.P2POp(dist.irecv, out[lo:hi], r))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        return out
    for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, res_local, dst)]):
        w.wait()
    return None


def shard_clusters(costs, world):
    """Greedy LPT partition of cluster indices by cost (e.g. K^3): returns `world` index lists."""
    heap = [(0.0, r) for r in range(world)]
    heapq.heapify(heap)
    out = [[] for _ in range(world)]
    for i in sorted(range(len(costs)), key=lambda k: -costs[k]):
        load, r = heapq.heappop(heap)
        out[r].append(i)
        heapq.heappush(heap, (load + float(costs[i]), r))
    return out
