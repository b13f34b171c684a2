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


def test_batch_configurations_cover_every_frame_once_and_have_goldens():
    """bench.py's frame -> rank map for the two 8-GPU batch configurations of BASELINE.json: with 8 ranks every frame of
    the batch is encoded exactly once, and every frame has a reference golden (tests/golden/batch_golden.json, which
    agrees with the single-frame goldens of tests/golden/golden.json)."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, "tests", "golden", "batch_golden.json")) as fh:
        bg = json.load(fh)
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as fh:
        g = json.load(fh)
    assert tuple(bg["C4"]["frames"][0]) == (g["C4_2048_frame0"]["size"], g["C4_2048_frame0"]["crc32"])
    assert tuple(bg["C4"]["frames"][1]) == (g["C4_2048_frame1"]["size"], g["C4_2048_frame1"]["crc32"])
    assert tuple(bg["C5"]["frames"][0]) == (g["C5_8192_frame0"]["size"], g["C5_8192_frame0"]["crc32"])
    for name in ("C4", "C5"):
        c = bench.CONFIGS[name]
        assert c["per_gpu"] * 8 == c["total"] == len(bg[name]["frames"])
        seen = []
        for rank in range(8):
            gold = bench.frame_goldens(name, rank)
            assert len(gold) == c["per_gpu"]
            lo = rank * c["per_gpu"]
            assert gold == [tuple(f) for f in bg[name]["frames"][lo: lo + c["per_gpu"]]]
            seen += list(range(lo, lo + c["per_gpu"]))
        assert seen == list(range(c["total"]))
    assert bench.frame_goldens("C2", 5) == [(g["C2_4096_gray_5st_10seg"]["size"], g["C2_4096_gray_5st_10seg"]["crc32"])]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` started plainly (no torch.distributed.run around it, WORLD_SIZE unset) becomes the launcher
    of N ranks: --launch-probe makes the ranks rendezvous over gloo on 127.0.0.1, reduce their shares of the timed
    configuration and print one line -- the N > 1 default is the whole C4 batch split over the ranks."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    for n in (2, 3):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--launch-probe"], env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
        assert len(lines) == 1                                            # rank 0 alone reports
        assert lines[0] == {"launch_probe": True, "n_gpus": n, "config": "C4", "scaling": "strong", "frames_over_ranks": 256,
                            "rank_sum": n * (n - 1) // 2}
    # one GPU: no launcher, C2, weak
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--launch-probe"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["config"] == "C2"
