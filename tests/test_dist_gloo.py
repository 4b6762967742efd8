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


def _run_bench(extra_env, *argv, timeout=300):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.update(extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(argv), env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    return r, [json.loads(l) for l in lines]


def test_bench_spawns_its_own_ranks_and_propagates_failure():
    """`python bench.py --gpus 2` started plainly launches its two ranks itself (torch.distributed.run, 127.0.0.1),
    runs the barrier / max-over-ranks / all-gather control flow and prints ONE JSON line; a failing rank makes the
    parent exit non-zero.  DLPM_BENCH_DRY_RUN swaps the sampler for a CPU stub (gloo): control flow only."""
    r, out = _run_bench({'DLPM_BENCH_DRY_RUN': '1'}, '--gpus', '2', '--steps', '5', '--warmup', '2', '--batch', '6')
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(out) == 1, r.stdout
    j = out[0]
    assert j['n_gpus'] == 2 and j['steps'] == 5 and j['warmup'] == 2 and j['scaling'] == 'weak'
    assert j['config']['global_batch'] == 12 and j['config']['allgather_ms'] is not None
    assert j['full_trajectory_s'] is not None and j['value_source'].startswith('measured') and 'dry_run' in j
    # single rank: no collective, no allgather time
    r, out = _run_bench({'DLPM_BENCH_DRY_RUN': '1'}, '--gpus', '1', '--steps', '3', '--warmup', '1', '--batch', '4')
    assert r.returncode == 0 and out[0]['n_gpus'] == 1 and out[0]['config']['allgather_ms'] is None
    # a rank that dies (more steps than exist -> assertion in every rank) must surface as a non-zero exit code
    r, out = _run_bench({'DLPM_BENCH_DRY_RUN': '1'}, '--gpus', '2', '--steps', '5000', '--warmup', '0')
    assert r.returncode != 0 and not out


@pytest.mark.parametrize('workload,per_gpu', [('cifar10_unet_b1024_T1000', 1024), ('celeba64_unet_b256_T1000', 256)])
def test_bench_eight_rank_control_flow_of_the_driver_commands(workload, per_gpu):
    """BASELINE configs 4 and 5 as the driver would launch them (`bench.py --gpus 8 [--workload ...]`), unattended: eight
    spawned ranks, gloo, stub sampler (DLPM_BENCH_DRY_RUN) -- rendezvous, barriers, max-over-ranks, ONE all-gather of the
    per-GPU shards, one JSON line whose batch is 8 x the per-GPU batch of the workload."""
    r, out = _run_bench({'DLPM_BENCH_DRY_RUN': '1', 'OMP_NUM_THREADS': '1'}, '--gpus', '8', '--steps', '4', '--warmup', '1',
                        '--workload', workload, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(out) == 1, r.stdout
    j = out[0]
    assert j['n_gpus'] == 8 and j['scaling'] == 'weak' and j['config']['global_batch'] == 8 * per_gpu
    assert j['config']['workload'] == workload and j['config']['allgather_ms'] is not None and j['samples_finite']
    assert j['full_trajectory_s'] is not None
    # round 4: the line validates itself -- one record per rank, the backend that carried the gather, its size, and the proof
    # that every rank's shard arrived (checksums of the shards == checksums of the same rows of the gathered batch)
    assert j['backend'] == 'gloo' and j['rccl_world_size'] == 8
    ranks = j['ranks']
    assert [r['rank'] for r in ranks] == list(range(8)) and len({r['pid'] for r in ranks}) == 8
    for r in ranks:
        assert set(r) >= {'rank', 'local_rank', 'device_index', 'device_uuid', 'pci_bus_id', 'device_name', 'ms_per_step',
                          'shard_checksum', 'gathered_rows_checksum'}
        assert r['ms_per_step'] > 0 and r['ms_per_step'] <= j['ms_per_step'] + 1e-3
        assert r['shard_checksum'] == r['gathered_rows_checksum']
    assert len({r['shard_checksum'] for r in ranks}) == 8          # the stub's shards differ by rank
    shape = j['config']['state_shape_per_gpu']
    assert j['allgather_bytes'] == 8 * per_gpu * shape[1] * shape[2] * shape[3] * 4
    assert j['distributed']['gather_verified'] is True and j['distributed']['allgather_bytes_per_rank_sent'] * 8 == j['allgather_bytes']


def test_bench_rank_dying_mid_trajectory_fails_the_parent_within_the_deadline():
    """A rank that exits in the middle of the trajectory (its peers are left in a barrier) makes the parent exit non-zero
    -- promptly, through torch.distributed.run -- and a rank that HANGS is ended by the parent's deadline (exit 124);
    only the children's own process group is ever signalled."""
    import time
    env = {'DLPM_BENCH_DRY_RUN': '1', 'DLPM_BENCH_DIE_RANK': '1', 'DLPM_BENCH_DIE_AT': '40', 'OMP_NUM_THREADS': '1'}
    t0 = time.time()
    r, out = _run_bench(env, '--gpus', '2', '--steps', '5', '--warmup', '2', '--batch', '4', '--rank-timeout', '200', timeout=400)
    assert r.returncode not in (0, 124) and not out, (r.returncode, r.stdout)
    assert time.time() - t0 < 150
    env['DLPM_BENCH_DIE_MODE'] = 'hang'
    t0 = time.time()
    r, out = _run_bench(env, '--gpus', '2', '--steps', '5', '--warmup', '2', '--batch', '4', '--rank-timeout', '25', timeout=400)
    assert r.returncode == 124 and not out, (r.returncode, r.stdout)
    assert time.time() - t0 < 120


def test_bench_under_the_drivers_own_launcher_command():
    """The driver's N > 1 command verbatim: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W` -- every process is ONE rank (RANK / LOCAL_RANK / WORLD_SIZE from the
    environment, no second spawn), rank 0 prints the single JSON line."""
    import json
    import subprocess
    env = dict(os.environ, DLPM_BENCH_DRY_RUN='1', OMP_NUM_THREADS='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    port = 29600 + os.getpid() % 300
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                        '--batch', '4'], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(out) == 1, r.stdout
    j = out[0]
    assert j['n_gpus'] == 2 and j['steps'] == 3 and j['warmup'] == 1 and j['config']['global_batch'] == 8
    assert 'cpu_baseline' not in j or j['cpu_baseline'] is None      # rank 0 at N = 1 only
