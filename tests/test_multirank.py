"""N>1 path of bench.py on CPU: two gloo ranks shard the envs, draw their PD targets and all-gather their
observation blocks; the gathered tensor must be in global env order and the sharded targets must equal the
single-process ones (no data-path collective besides the observation gather)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def _worker(rank, world, port, n, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    ids = bench.shard_env_ids(rank, world, n)
    tg = bench.pd_targets(ids, 3)
    # observation block: row e carries its global env id and its first PD target so the order can be checked
    obs = torch.zeros((n, 96), dtype=torch.float64)
    obs[:, 0] = torch.from_numpy(ids.astype(np.float64))
    obs[:, 1] = torch.from_numpy(tg[0, :, 0])
    allobs = bench.gather_observations(obs, world)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)          # the max-over-ranks timing reduction of bench.py
    ret[rank] = (allobs.numpy().copy(), float(t.item()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_observation_allgather():
    import bench
    world, n = 2, 16
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, n, ret), nprocs=world, join=True)
    glob = bench.pd_targets(np.arange(world * n), 3)
    for r in range(world):
        allobs, tmax = ret[r]
        assert allobs.shape == (world * n, 96)
        assert np.array_equal(allobs[:, 0], np.arange(world * n))          # global env order, rank-major
        assert np.array_equal(allobs[:, 1], glob[0, :, 0])                  # per-env seeds do not depend on the sharding
        assert tmax == float(world)


def test_shards_partition_the_env_range():
    import bench
    ids = np.concatenate([bench.shard_env_ids(r, 8, 8192) for r in range(8)])
    assert np.array_equal(ids, np.arange(65536))                            # BASELINE config 3: 8 x 8192
