"""CPU, world_size 2 (gloo): the bucketed gradient all-reduce used for data-parallel training."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import odtk  # noqa: F401
    from odtk.dist import BucketAllReducer
    sizes = [64, 1000, 64, 50000, 128, 30000, 64, 7000, 192]
    segs, off = [], 0
    for i, n in enumerate(sizes):
        segs.append((f'l{i}', off, off + n)); off += n
    flat = torch.zeros(off)
    red = BucketAllReducer(flat, segs, None, bucket_bytes=100_000)
    # buckets tile the buffer exactly once, suffix first
    cover = sorted((s, e) for s, e, _ in red.buckets)
    assert cover[0][0] == 0 and cover[-1][1] == off and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
    assert red.buckets[0][1] == off
    ok = True
    for step in range(3):
        flat.zero_()
        red.begin_step()
        for i in reversed(range(len(segs))):          # backward order: last layer first
            name, s, e = segs[i]
            flat[s:e] = torch.arange(s, e, dtype=torch.float32) * (rank + 1) + step
            red.segment_ready(name)
        red.finish_step()
        exp = torch.arange(0, off, dtype=torch.float32) * sum(r + 1 for r in range(world)) + step * world
        ok = ok and torch.allclose(flat, exp)
    q.put((rank, ok, len(red.buckets)))
    dist.destroy_process_group()


def test_bucketed_allreduce_world2():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert res[0][2] >= 3          # several buckets -> overlap opportunities


def test_single_process_is_a_noop():
    import odtk  # noqa: F401
    from odtk.dist import BucketAllReducer
    flat = torch.arange(100, dtype=torch.float32)
    red = BucketAllReducer(flat, [('a', 0, 40), ('b', 40, 100)], None, bucket_bytes=64)
    red.begin_step(); red.segment_ready('b'); red.segment_ready('a'); red.finish_step()
    assert torch.equal(flat, torch.arange(100, dtype=torch.float32))
