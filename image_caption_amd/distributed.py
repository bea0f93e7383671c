"""Multi-GPU layout: images are independent, so the batch is split contiguously over
ranks (weights replicated, no per-step communication) and the decoded sequences are
all-gathered once at the end (RCCL over xGMI when the backend is nccl)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(total, world_size, rank):
    """Contiguous [lo, hi) slice of `total` images owned by `rank`; the first
    total % world_size ranks get one extra image."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, extra = divmod(total, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_sequences(tokens, total=None, group=None):
    """tokens [b_local, T] int32 on every rank -> [sum b_local, T] on every rank, in
    rank order.  Ragged shards (total not divisible by the world size) are padded to
    the largest shard for the collective and trimmed afterwards."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return tokens
    world = dist.get_world_size(group)
    if total is None:
        sizes = [tokens.shape[0]] * world
    else:
        sizes = [shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0]
                 for r in range(world)]
    cap = max(sizes)
    T = tokens.shape[1]
    send = tokens
    if tokens.shape[0] != cap:
        send = torch.zeros(cap, T, dtype=tokens.dtype, device=tokens.device)
        send[:tokens.shape[0]] = tokens
    if dist.get_backend(group) == "gloo" and send.is_cuda:
        # rehearsal of the multi-rank path without RCCL (ranks sharing one GPU, CPU-only
        # boxes): gloo gathers host tensors
        host = torch.empty(world * cap, T, dtype=tokens.dtype)
        dist.all_gather_into_tensor(host, send.contiguous().cpu(), group=group)
        out = host.to(tokens.device)
    else:
        out = torch.empty(world * cap, T, dtype=tokens.dtype, device=tokens.device)
        dist.all_gather_into_tensor(out, send.contiguous(), group=group)
    if all(s == cap for s in sizes):
        return out
    parts = [out[r * cap:r * cap + sizes[r]] for r in range(world)]
    return torch.cat(parts, dim=0)


def decode_sharded(make_engine, img_global, group=None):
    """Decodes this rank's contiguous slice of img_global [B,L,F] with an engine built by
    make_engine(b_local) and returns the all-gathered tokens [B,T].  Only meaningful for
    tile_mode='repeat' (in 'reference' mode beams borrow features across images)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(img_global.shape[0], world, rank)
    eng = make_engine(hi - lo)
    tokens, _ = eng.decode(img_global[lo:hi].contiguous())
    return all_gather_sequences(tokens, total=img_global.shape[0], group=group)
