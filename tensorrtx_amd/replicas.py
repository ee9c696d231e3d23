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


class DeviceReplicas:
    """One process driving several GPUs: a replica (engine + context + stream + bindings) per device, images dealt to the
    replicas by `partition`, no cross-device traffic (the reference's own recipe, tutorials/multi_GPU_processing.md:13-30:
    cudaSetDevice(i), then one Plan {engine, context, stream, buffers} per device).  bench.py uses one PROCESS per GPU instead
    (the driver's launch contract); this class is the in-process form for hosts that own all GPUs from one address space.

    make_engine(device_index) -> engine is called with that device current; engines refuse to run on any other device
    (trtx_engine_device / TRTX_ERR_STATE)."""

    def __init__(self, devices, make_engine):
        import torch
        self.devices = list(devices)
        if not self.devices:
            raise ValueError("DeviceReplicas needs at least one device")
        self.engines, self.streams = [], []
        for d in self.devices:
            with torch.cuda.device(d):
                self.engines.append(make_engine(d))
                self.streams.append(torch.cuda.Stream(device=d))

    def shards(self, n_items: int):
        """Image index range of every replica for a global batch of n_items."""
        return [partition(n_items, len(self.devices), r) for r in range(len(self.devices))]

    def enqueue(self, batches, bindings):
        """batches[r] / bindings[r]: batch size and binding list (tensors on devices[r]) of replica r.  Asynchronous: every
        replica runs on its own stream; call synchronize() (or record events on .streams) before reading results."""
        import torch
        for r, d in enumerate(self.devices):
            if batches[r] == 0:
                continue
            with torch.cuda.device(d):
                self.engines[r].enqueue(batches[r], bindings[r], stream=self.streams[r].cuda_stream)

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def close(self):
        import torch
        for d, e in zip(self.devices, self.engines):
            with torch.cuda.device(d):
                e.close()
        self.engines = []
