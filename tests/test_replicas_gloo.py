"""N>1 path on CPU: world_size-2 gloo processes exercise the image partition, the timing reduce and the
optional result gather used by bench.py (no data-path collective exists to test)."""
import os

import pytest
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tensorrtx_amd import replicas


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    r = replicas.env_rank()
    mine = list(replicas.partition(67, r.world, r.rank))
    # every rank "processes" its shard: kept count per image is a deterministic function of the image id
    counts = torch.tensor([(7 * i) % 13 for i in mine] + [-1] * (34 - len(mine)), dtype=torch.int32)
    allc = replicas.gather_counts(counts, dist)
    slow = replicas.max_over_ranks(1.0 + rank, dist)
    dist.barrier()
    q.put((rank, mine, [c.tolist() for c in allc], slow))
    dist.destroy_process_group()


def test_two_rank_partition_and_timing():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    shards = [r[1] for r in res]
    assert sorted(shards[0] + shards[1]) == list(range(67)) and not set(shards[0]) & set(shards[1])
    assert len(shards[0]) == 34 and len(shards[1]) == 33
    for r in res:
        assert r[3] == 2.0  # max over ranks
        gathered = r[2]
        flat = [c for part, sh in zip(gathered, shards) for c in part[:len(sh)]]
        assert flat == [(7 * i) % 13 for i in shards[0] + shards[1]]


def test_partition_edge_cases():
    assert list(replicas.partition(8, 8, 3)) == [3]
    assert list(replicas.partition(3, 8, 5)) == []
    assert sum(len(replicas.partition(1000, 7, r)) for r in range(7)) == 1000
    assert replicas.max_over_ranks(0.5, None) == 0.5


def test_device_replicas_shards_without_gpu():
    """The in-process multi-device helper deals images exactly like the process-per-GPU partition."""
    from tensorrtx_amd import replicas
    r = replicas.DeviceReplicas.__new__(replicas.DeviceReplicas)
    r.devices = [0, 1, 2]
    assert [list(s) for s in r.shards(8)] == [[0, 1, 2], [3, 4, 5], [6, 7]]
    with pytest.raises(ValueError):
        replicas.DeviceReplicas([], lambda d: None)
