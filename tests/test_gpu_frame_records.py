"""One process per GPU with update()'s host logic run once (gpis3_set_frame_export / gpis3_frame_record / gpis3_apply_frame):
lead and worker as two maps of ONE process on one GPU, the broadcast replaced by handing the record over, the model exchange
done by hand through two device buffers.  tests/test_gpu_multirank.py runs the same thing as 2 and 4 real processes."""
import numpy as np
import pytest

import replay

pytestmark = pytest.mark.gpu


def _exchange(maps):
    """What sharding.exchange_models does over a collective, for maps of one process: every rank packs its records, the others unpack."""
    import torch
    dev = torch.device("cuda", 0)
    world = len(maps)
    nbytes = [maps[0].shard_bytes(r) for r in range(world)]
    for m in maps[1:]:
        assert [m.shard_bytes(r) for r in range(world)] == nbytes          # every rank lays out every rank's buffer alike
    bufs = [torch.empty(max(256, nbytes[r]), dtype=torch.uint8, device=dev) for r in range(world)]
    s = torch.cuda.current_stream().cuda_stream
    for r, m in enumerate(maps):
        m.shard_pack(bufs[r].data_ptr(), s)
    torch.cuda.synchronize()
    for r, m in enumerate(maps):
        for o in range(world):
            if o != r and nbytes[o]:
                m.shard_unpack(o, bufs[o].data_ptr(), s)
    torch.cuda.synchronize()
    for m in maps:
        m.shard_finish()


def test_worker_applies_the_leads_records_and_answers_with_the_same_bits():
    import gpismap_amd
    ref = gpismap_amd.GPisMap3()
    lead = gpismap_amd.GPisMap3(); lead.set_shard(0, 2); lead.set_frame_export(True)
    work = gpismap_amd.GPisMap3(); work.set_shard(1, 2)
    grid = replay.synthetic_grid(24)
    sizes = []
    for f in range(3):
        d = replay.synthetic_depth(f)
        ref.update(d, replay.IDENTITY_POSE)
        lead.update(d, replay.IDENTITY_POSE)               # host logic + record, own share deferred
        rec = lead.frame_record()
        sizes.append(rec.size)
        work.apply_frame(rec)                              # no replay: K6 + its share on the record
        lead.train_deferred()
        _exchange([lead, work])
        assert lead.stats()["host_replays"] == 1 and work.stats()["host_replays"] == 0
        a = ref.test(grid)
        assert np.array_equal(lead.test(grid).view(np.uint32), a.view(np.uint32))
        assert np.array_equal(work.test(grid).view(np.uint32), a.view(np.uint32))
    assert lead.num_points() == ref.num_points() and work.num_points() == 0       # a worker holds no tree
    assert all(64 < s < (16 << 20) for s in sizes)
    # a frame that returns before it touches the map (no valid pixel) still leaves a record, and the worker's map survives it
    lead.update(np.zeros(640 * 480, dtype=np.float32), replay.IDENTITY_POSE)
    work.apply_frame(lead.frame_record())
    lead.train_deferred()
    assert np.array_equal(work.test(grid).view(np.uint32), ref.test(grid).view(np.uint32))


def test_damaged_or_foreign_records_are_refused():
    import gpismap_amd
    lead = gpismap_amd.GPisMap3(); lead.set_shard(0, 2); lead.set_frame_export(True)
    work = gpismap_amd.GPisMap3(); work.set_shard(1, 2)
    lead.update(replay.synthetic_depth(0), replay.IDENTITY_POSE)
    rec = lead.frame_record().copy()
    lead.train_deferred()
    with pytest.raises(gpismap_amd.GpisError):
        work.apply_frame(rec[:len(rec) // 2])              # truncated
    bad = rec.copy(); bad[:4] = 0
    with pytest.raises(gpismap_amd.GpisError):
        work.apply_frame(bad)                              # not a frame record
    # (header: 2 x u32, 4 x i32, then ten u64 counts -- the first is the point count)
    bad = rec.copy(); bad.view(np.uint64)[3] = 7           # point count shrunk: the cell lists point past it
    with pytest.raises(gpismap_amd.GpisError):
        work.apply_frame(bad)
    # a map that replays frames itself is not a worker, and a worker does not replay
    own = gpismap_amd.GPisMap3()
    own.update(replay.synthetic_depth(0), replay.IDENTITY_POSE)
    with pytest.raises(gpismap_amd.GpisError):
        own.apply_frame(rec)
    work.apply_frame(rec)                                  # the intact record is still welcome after the refusals ...
    with pytest.raises(gpismap_amd.GpisError):
        work.update(replay.synthetic_depth(1), replay.IDENTITY_POSE)  # ... and update() on a worker is refused (map untouched)
