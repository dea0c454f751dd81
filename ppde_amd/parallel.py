"""Chain sharding across the GPUs of a node.

Chains are independent Markov chains, so the population splits into contiguous blocks, one per rank, with NO
per-step communication: the device RNG is keyed by the GLOBAL chain index, so a chain's trajectory does not
depend on how many ranks there are. The only collective is the final population collect (all_gather over
RCCL/xGMI on GPUs; gloo in the CPU tests).
"""
import os

import torch
import torch.distributed as dist

# PPDE_COLLECTIVES_AT_WORLD_1=1: run the collectives below even in a process group of ONE rank (where they are the
# identity). It lets a one-GPU box execute the RCCL path end to end -- library load, communicator set-up, host -> device
# staging, all_gather / broadcast, the way back -- before an 8-GPU node ever sees it (tests/test_host_gpu.py).
def force_at_world_1():
    """Read at every call (not at import), '' and '0' mean off -- as the library's atoi-style knobs."""
    return os.environ.get("PPDE_COLLECTIVES_AT_WORLD_1", "0") not in ("", "0")


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def active():
    """True when the collectives below actually communicate: more than one rank, or forced at world size 1."""
    _, ws = world()
    return ws > 1 or (force_at_world_1() and dist.is_available() and dist.is_initialized())


def shard_range(n, rank, world_size):
    """Contiguous block [lo, hi) of rank `rank`; the first n % world_size ranks hold one extra chain."""
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_rows(t, n_global, dim=0):
    """Concatenate per-rank shards (unequal sizes allowed) along `dim` on every rank."""
    rank, ws = world()
    if not active():
        return t
    sizes = [shard_range(n_global, r, ws) for r in range(ws)]
    mx = max(hi - lo for lo, hi in sizes)
    home = t.device
    if dist.get_backend() == "nccl" and t.device.type != "cuda":   # RCCL moves device memory only
        t = t.cuda()
    t = t.movedim(dim, 0).contiguous()
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(bufs, pad)
    out = torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], 0)
    return out.movedim(0, dim).to(home)


def broadcast_from(t, src):
    if active():
        home = t.device
        if dist.get_backend() == "nccl" and t.device.type != "cuda":
            t = t.cuda()
        dist.broadcast(t, src)
        t = t.to(home)
    return t


def agree_from_rank0(values):
    """Rank 0's integers on every rank (random chain index, default seed): ranks that were seeded differently must
    still record the same chain and draw the same Philox stream."""
    if not active():
        return [int(v) for v in values]
    t = torch.tensor([int(v) for v in values], dtype=torch.int64)
    return [int(v) for v in broadcast_from(t, 0).tolist()]
