"""world_size-2 gloo run (CPU) of the sharding + single all-gather used by the multi-GPU path."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from dlpm_amd.dist import shard_range, all_gather_samples, sample_sharded


def test_shard_ranges_cover_everything():
    for total in (1, 7, 8, 1024, 8192, 1023):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


class FakeMethod:
    """sample() returns a pure function of the GLOBAL sample index, like the Philox-keyed sampler."""

    def __init__(self, offset):
        self.offset = offset

    def sample(self, models, shape, reverse_steps, **kw):
        idx = torch.arange(self.offset, self.offset + shape[0], dtype=torch.float32)
        return idx.view(-1, 1, 1, 1) * torch.ones(shape) + reverse_steps


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        full = sample_sharded(lambda off: FakeMethod(off), None, [total, 3, 2, 2], 10)
        lo, hi = shard_range(total, rank, world)
        same = all_gather_samples(full[lo:hi].clone(), total)
        q.put((rank, full.clone(), torch.equal(same, full)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('total', [8, 7])
def test_two_rank_gather_equals_unsharded(total):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + total
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = FakeMethod(0).sample(None, [total, 3, 2, 2], 10)
    for rank, full, ok in got:
        assert ok and torch.equal(full, want), rank
