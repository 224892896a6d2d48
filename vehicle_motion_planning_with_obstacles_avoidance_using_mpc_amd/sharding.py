"""Multi-GPU layout: the batch of independent instances is cut into contiguous shards, one process per GPU;
there is no exchange on the data path, only ONE all_gather (one packed buffer per rank) of the outputs at the end.
``torch.distributed`` backend "nccl" is RCCL over xGMI on MI355X; the same code runs on gloo for CPU tests."""
import torch


def shard_bounds(total, world, rank):
    """contiguous [lo, hi) of instances owned by ``rank``; sizes differ by at most one"""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_outputs(dist, local, total, world):
    """ONE all_gather of per-shard output tensors (dict name -> tensor with leading shard dimension): every rank packs its
    tensors, padded to the largest shard, back to back into one byte buffer (each section 16-byte aligned), the buffers are
    exchanged in a single collective and cut up again.  Returns the full-batch tensors in instance order on every rank."""
    sizes = [shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world)]
    big = max(sizes)
    names = list(local)
    dev = local[names[0]].device
    sect, off = [], 0
    for name in names:
        t = local[name]
        nbytes = big * t.element_size()
        for d in t.shape[1:]:
            nbytes *= int(d)
        sect.append((off, nbytes))
        off += (nbytes + 15) // 16 * 16
    buf = torch.zeros(max(off, 16), dtype=torch.uint8, device=dev)
    for name, (o, nb) in zip(names, sect):
        t = local[name].contiguous()
        mine = t.numel() * t.element_size()
        buf[o:o + mine] = t.view(-1).view(torch.uint8)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    out = {}
    for name, (o, nb) in zip(names, sect):
        t = local[name]
        shape = (big,) + tuple(t.shape[1:])
        out[name] = torch.cat([p[o:o + nb].view(t.dtype).view(shape)[:n] for p, n in zip(parts, sizes)], dim=0)
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
