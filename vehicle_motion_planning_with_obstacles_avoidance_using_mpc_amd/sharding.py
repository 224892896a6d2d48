"""Multi-GPU layout: the batch of independent instances is cut into contiguous shards, one process per GPU;
there is no exchange on the data path, only one all_gather of the outputs at the end.
``torch.distributed`` backend "nccl" is RCCL over xGMI on MI355X; the same code runs on gloo for CPU tests."""
import torch


def shard_bounds(total, world, rank):
    """contiguous [lo, hi) of instances owned by ``rank``; sizes differ by at most one"""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_outputs(dist, local, total, world):
    """all_gather of per-shard output tensors (dict name -> tensor with leading shard dimension), unequal
    shards padded to the largest; returns the full-batch tensors in instance order on every rank."""
    sizes = [shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world)]
    big = max(sizes)
    out = {}
    for name, t in local.items():
        pad = torch.zeros((big,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[:t.shape[0]] = t
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
        out[name] = torch.cat([p[:n] for p, n in zip(parts, sizes)], dim=0)
    return out


def shard_worlds(worlds, world, rank):
    """the contiguous slice of a ``rollouts.PackedWorlds`` batch owned by ``rank``: every rank runs the WHOLE closed
    loop (harness + solves, obca_rollouts_run) for its own rollouts; nothing is exchanged until the histories are
    gathered."""
    lo, hi = shard_bounds(worlds.batch, world, rank)
    return worlds.slice(lo, hi)


def gather_rollouts(dist, local, total, world):
    """all_gather of the history dict of ``DeviceRollouts.read()`` (every entry has the rollout dimension first)"""
    return gather_outputs(dist, local, total, world)
