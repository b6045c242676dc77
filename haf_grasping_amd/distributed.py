"""One-process-per-GPU helpers (torch.distributed: backend "nccl" is RCCL on ROCm, "gloo" in CPU tests).

The path shards without a data-path collective.  What is exchanged is tiny and latency-bound:
  * roll sharding of ONE request: an all-gather of the 16-byte haf_roll_record of every roll (the all-gather form,
    not a max-reduce, because the reference's early exit with show_only_best_grasp makes the cross-roll reduction
    order dependent -- server.cpp:362-365);
  * cloud sharding of a batch: nothing, or one 8-byte all-reduce(max) to elect the best grasp of the batch.
"""
import numpy as np

from . import capi


def roll_shard(n_rolls, world, rank):
    """Contiguous roll range of `rank`: 36 rolls over 8 ranks -> 5,5,5,5,4,4,4,4."""
    q, r = divmod(n_rolls, world)
    first = rank * q + min(rank, r)
    return first, q + (1 if rank < r else 0)


def gather_roll_records(local, n_rolls, device=None):
    """all-gather variable-length shards of roll records -> [n_clouds, n_rolls] on every rank (rank order = roll order)."""
    import torch
    import torch.distributed as dist
    local = np.ascontiguousarray(local, dtype=capi.ROLL_RECORD_DTYPE)
    n_clouds = local.shape[0]
    world = dist.get_world_size()
    cap = -(-n_rolls // world)                     # every shard padded to the largest shard
    buf = np.zeros((n_clouds, cap), dtype=capi.ROLL_RECORD_DTYPE)
    buf[:, :local.shape[1]] = local
    t = torch.from_numpy(buf.view(np.uint8).reshape(-1).copy())
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    full = np.zeros((n_clouds, n_rolls), dtype=capi.ROLL_RECORD_DTYPE)
    for rk, o in enumerate(out):
        first, count = roll_shard(n_rolls, world, rk)
        shard = o.cpu().numpy().view(capi.ROLL_RECORD_DTYPE).reshape(n_clouds, cap)
        full[:, first:first + count] = shard[:, :count]
    return full


def best_of_batch(best_vote, tag, device=None):
    """Elects the best grasp over all ranks with ONE 8-byte all-reduce(max): key = (vote + 1000) << 20 | (2^20-1 - tag);
    larger vote wins, then the smaller tag (e.g. global cloud index).  Returns (vote, tag)."""
    import torch
    import torch.distributed as dist
    key = torch.tensor([((int(best_vote) + 1000) << 20) | ((1 << 20) - 1 - int(tag))], dtype=torch.int64)
    if device is not None:
        key = key.to(device)
    dist.all_reduce(key, op=dist.ReduceOp.MAX)
    k = int(key.item())
    return (k >> 20) - 1000, (1 << 20) - 1 - (k & ((1 << 20) - 1))
