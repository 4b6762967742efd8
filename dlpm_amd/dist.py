"""Batch sharding across the GPUs of one node (one process per GPU, torch.distributed / RCCL).

The reverse loop has no cross-sample operation (GroupNorm and attention are per sample), so the
batch is cut into contiguous shards, each rank runs the whole T-step loop on its shard with Philox
noise keyed by the GLOBAL sample index (`sample_offset`), and ONE all-gather of the finished
[B/W, C, H, W] fp32 shards assembles the batch.  Results are bit-identical for every
world size: Philox is keyed by the global sample index and no kernel choice depends on the shard size
(dlpm_unet_set_conv_policy).
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous shard [lo, hi) of `total` samples for `rank`; sizes differ by at most one."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_samples(local, total, group=None, always_collective=False):
    """Gather contiguous shards (possibly of unequal length) into the full [total, ...] tensor on
    every rank with a single collective.  A one-rank group needs none and returns `local`, unless
    `always_collective` asks for the collective to be issued anyway (exercises the RCCL path on one GPU)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not always_collective):
        assert local.shape[0] == total
        return local
    world = dist.get_world_size(group)
    per = -(-total // world)
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    pieces = []
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        pieces.append(out[r * per:r * per + (hi - lo)])
    return torch.cat(pieces, dim=0)


def sample_sharded(make_method, models, shape, reverse_steps, group=None, **sample_kwargs):
    """Every rank samples its shard of `shape[0]` and all ranks return the full batch.

    `make_method(sample_offset)` builds the rank-local GenerativeLevyProcess (same seed on every
    rank, shard-specific offset)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    lo, hi = shard_range(shape[0], rank, world)
    method = make_method(lo)
    local = method.sample(models, [hi - lo] + list(shape[1:]), reverse_steps, **sample_kwargs)
    return all_gather_samples(local, shape[0], group)
