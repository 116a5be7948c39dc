"""N > 1 path on CPU: world_size-2 gloo processes exercise the sharding / aggregation code bench.py uses
under RCCL.  The data path has no collective, so what must hold is: shards are disjoint and cover the batch,
per-rank batches are distinct, and the whole-job rate is (sum of rays) / (max of elapsed)."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    D = importlib.import_module("embree-compressed_amd.dist")
    rg = importlib.import_module("embree-compressed_amd.raygen")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    r, lr, w = D.env_rank()
    assert (r, w) == (rank, world)
    total = 100_001
    b, e = D.shard_range(total, rank, world)
    owned = np.zeros(total, np.int32)
    owned[b:e] = 1
    import torch
    t = torch.from_numpy(owned)
    dist.all_reduce(t)  # test-only collective: every ray is owned exactly once
    assert int(t.min()) == 1 and int(t.max()) == 1
    rays = rg.make_random_rays(1000, [0, 0, 0], [1, 1, 1], seed=D.batch_seed(rank, 0))
    digest = torch.tensor([int(rays.view(np.uint32).astype(np.uint64).sum() % (1 << 31))], dtype=torch.int64)
    gathered = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(gathered, digest)
    assert len({int(g.item()) for g in gathered}) == world  # distinct batches per rank
    D.barrier(world)
    rate, worst = D.whole_job_rate(rays_per_rank=1000 * (rank + 1), elapsed_local=0.5 * (rank + 1), world=world)
    assert worst == 0.5 * world and abs(rate - sum(1000 * (k + 1) for k in range(world)) / worst) < 1e-9
    np.save(os.path.join(out_dir, f"ok{rank}.npy"), np.array([rate]))
    dist.destroy_process_group()


def test_two_rank_sharding(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0.npy") and os.path.exists(tmp_path / "ok1.npy")


def test_shard_ranges_cover_and_are_disjoint():
    D = importlib.import_module("embree-compressed_amd.dist")
    for total in (0, 1, 7, 1_000_000):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
