"""Replica-per-GPU sharding of independent image streams (reference advice:
tutorials/multi_GPU_processing.md:13-30 — one engine + context + stream per device, no cross-device
traffic on the data path).  torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used only to
agree on the partition, to bracket the timed region, and for the optional result gather."""
import os
from dataclasses import dataclass


@dataclass
class Rank:
    rank: int
    world: int
    local: int


def env_rank() -> Rank:
    return Rank(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
                int(os.environ.get("LOCAL_RANK", "0")))


def partition(n_items: int, world: int, rank: int):
    """Contiguous-by-rank split of `n_items` images (the remainder goes to the lowest ranks)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def max_over_ranks(seconds: float, dist, device=None) -> float:
    """The timed region of a multi-rank run is as long as its slowest rank."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_counts(local_counts, dist):
    """Optional batch-splitter epilogue: all ranks learn every image's kept-detection count
    (fixed-size, a few bytes per image — bandwidth-trivial next to 7x153 GB/s of xGMI)."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [local_counts]
    out = [torch.empty_like(local_counts) for _ in range(dist.get_world_size())]
    dist.all_gather(out, local_counts)
    return out
