"""N>1 path on CPU: world_size-2 gloo run of the sharding + timing-reduction helpers bench.py uses.
The data path has no collective (frames are independent); only barrier + MAX-reduce exist."""
import os
import socket
import sys

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from icer_compression_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.shard_range(7, rank, world)
    seeds = shard.frame_seeds(12345, rank, world, 3)
    mine = torch.tensor([lo, hi] + seeds, dtype=torch.int64)
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    dist.barrier()
    tmax = shard.max_over_ranks(1.0 + rank)
    if rank == 0:
        q.put(([g.tolist() for g in gathered], tmax))
    dist.destroy_process_group()


def test_two_rank_sharding_and_max_reduce():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered, tmax = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert tmax == 2.0
    (lo0, hi0, *seeds0), (lo1, hi1, *seeds1) = gathered
    assert (lo0, hi0, lo1, hi1) == (0, 4, 4, 7)                       # contiguous, disjoint, complete
    assert seeds0 + seeds1 == list(range(12345, 12351))               # frame k -> seed 12345 + k


def test_shard_range_properties():
    sys.path.insert(0, ROOT)
    from icer_compression_amd import shard
    for n in range(0, 70):
        for world in (1, 2, 3, 4, 8):
            covered = []
            for r in range(world):
                lo, hi = shard.shard_range(n, r, world)
                covered += list(range(lo, hi))
                assert 0 <= hi - lo <= -(-n // world)
            assert covered == list(range(n))
