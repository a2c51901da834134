"""Multi-GPU layout of the batched path: instances are independent, so rank r of
W simulates the contiguous block shard_range(r, W, total) with model constants replicated and NO
per-step communication; the only collective is one all-gather of the final {dis, vel} per
rollout (RCCL over xGMI on GPUs, gloo in the CPU tests)."""


def shard_range(rank, world, total):
    """contiguous block [lo, hi) of instances owned by `rank`; remainders go to the low ranks"""
    if not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard request")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_final_states(dist, local, total):
    """all-gather of the per-rank final states [n_local, width] into [total, width] in instance
    order.  `local` is a torch tensor on the backend's device (cuda for nccl/RCCL, cpu for gloo)."""
    import torch
    world = dist.get_world_size()
    sizes = [shard_range(r, world, total)[1] - shard_range(r, world, total)[0] for r in range(world)]
    if len(set(sizes)) == 1:
        out = torch.empty((total, local.shape[1]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    # uneven shards: pad every rank's block to the largest one (collectives want equal sizes)
    mx = max(sizes)
    pad = torch.zeros((mx, local.shape[1]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty((world * mx, local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * mx:r * mx + sizes[r]] for r in range(world)], dim=0)
