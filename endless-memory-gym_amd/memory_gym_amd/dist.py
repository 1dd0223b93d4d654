"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" == RCCL over xGMI on ROCm, "gloo" in
the CPU tests).  Environment instances are independent, so the path shards with NO data-path collective:

    rank r of W owns global instances [lo, hi) = shard_range(N_total, r, W), and instance i is always seeded
    base_seed + i whatever W is, so results do not depend on the world size.

The only (optional) exchange is BASELINE config 5's gather of observations/rewards/dones to rank 0 for a
single-learner rollout.
"""
import torch
import torch.distributed as dist


def shard_range(n_total, rank, world):
    """Contiguous block of instance indices owned by `rank` (blocks differ by at most one instance)."""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_seeds(n_total, rank, world, base_seed=0, device=None):
    lo, hi = shard_range(n_total, rank, world)
    return torch.arange(lo, hi, dtype=torch.int64, device=device) + int(base_seed)


def gather_to_rank0(tensor, dst=0, group=None):
    """Gather equally-shaped per-rank tensors to `dst`; returns the concatenation on dst, None elsewhere.
    Ragged shards (N_total % W != 0) are padded to the largest shard and trimmed on dst."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_local = torch.tensor([tensor.shape[0]], device=tensor.device, dtype=torch.int64)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(s.item()) for s in sizes]
    n_max = max(sizes)
    if tensor.shape[0] < n_max:
        pad = torch.zeros((n_max - tensor.shape[0],) + tuple(tensor.shape[1:]), dtype=tensor.dtype, device=tensor.device)
        tensor = torch.cat([tensor, pad], 0)
    bufs = [torch.empty_like(tensor) for _ in range(world)] if rank == dst else None
    dist.gather(tensor.contiguous(), bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], 0)
