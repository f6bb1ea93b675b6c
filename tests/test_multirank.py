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


class _FakeBatch:
    """Stands in for the GPU batch: a step adds (global env id + 1) * nsub to column 0 of the observation block and
    counts the steps in column 1; targets of the bound policy step go to column 2."""

    def __init__(self, ids, tg):
        self.ids, self.tg = ids, tg
        self.obs = torch.zeros((len(ids), 96), dtype=torch.float64)
        self.bound = None
        self.seen = []

    def step(self, nsub):
        self.obs[:, 0] += torch.from_numpy((self.ids + 1.0) * nsub)
        self.obs[:, 1] += nsub
        self.obs[:, 2] = torch.from_numpy(self.tg[self.bound][:, 0])

    def bind(self, p):
        self.bound = p

    def restart(self, g):
        import bench
        rows = np.nonzero(self.ids % bench.NGROUP == g)[0]
        self.obs[rows, :2] = 0


def _schedule_worker(rank, world, port, n, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    ids = bench.shard_env_ids(rank, world, n)
    tg = bench.pd_targets(ids, 8)
    fb = _FakeBatch(ids, tg)
    allobs = torch.zeros((world * n, 96), dtype=torch.float64)
    snaps = []

    def gather():
        bench.gather_observations(fb.obs, world, allobs)
        snaps.append(allobs.numpy().copy())
    sch = bench.Schedule(step=fb.step, bind_targets=fb.bind, restart=fb.restart, gather=gather, substeps_per_launch=bench.HOLD)
    sch.run(0, 70)                                                    # "pre-roll + warm-up": ends inside policy step 1
    ticks = iter([10.0, 10.0 + (rank + 1)])                           # this rank "takes" rank + 1 seconds
    dt = bench.timed_region(sch, 70, 180, dist.barrier, clock=lambda: next(ticks))
    ret[rank] = (snaps, fb.obs.numpy().copy(), bench.max_over_ranks(dt, world), sch.launches, sch.gathers)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_schedule_gathers_every_policy_step_and_reduces_the_time():
    """bench.main()'s control flow for N > 1 (launch / restart / gather schedule, fences, max-over-ranks time) with a
    stand-in for the GPU batch: every gather must deliver all ranks' blocks of the SAME policy step in global env
    order, and the reported time is the slowest rank's."""
    import bench
    world, n = 2, 8
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + os.getpid() % 2000
    mp.spawn(_schedule_worker, args=(world, port, n, ret), nprocs=world, join=True)
    glob_ids = np.arange(world * n)
    glob_tg = bench.pd_targets(glob_ids, 8)
    for r in range(world):
        snaps, own, tmax, launches, gathers = ret[r]
        assert tmax == float(world)                                   # the slowest rank's elapsed time
        assert gathers == 4 and len(snaps) == 4                       # at steps 50, 100, 150, 200 (not at 0, not at the end)
        assert launches == 4                                          # of the timed region: 30 + 50 + 50 + 50
        for i, snap in enumerate(snaps):
            steps = 50.0 * (i + 1)
            # restart groups 0..i have restarted at policy steps 0..i: env e (group e % NGROUP <= i) has run 50 * (i - group) steps
            grp = glob_ids % bench.NGROUP
            ran = np.where(grp <= i, steps - 50.0 * grp, steps)
            assert np.array_equal(snap[:, 1], ran)
            assert np.array_equal(snap[:, 0], (glob_ids + 1.0) * ran)
            assert np.array_equal(snap[:, 2], glob_tg[i][:, 0])       # targets of the policy step that produced the block
        assert np.array_equal(own[:, 1], np.where(grp[r * n:(r + 1) * n] <= 4, 250.0 - 50.0 * grp[r * n:(r + 1) * n], 250.0))


def _parity_worker(rank, world, port, n, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    ids = bench.shard_env_ids(rank, world, n)
    rows = bench.parity_rows(n, world, 8)
    q = torch.from_numpy(np.outer(ids[rows] + 1.0, np.arange(1.0, 4.0)))     # a rank's "qpos" rows carry its global env ids
    gathered_ids = bench.gather_rows(torch.from_numpy(ids[rows].astype(np.int64)), world)
    gathered_q = bench.gather_rows(q, world)
    ret[rank] = (gathered_ids.numpy().copy(), gathered_q.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_parity_rows_of_every_rank_reach_rank_zero():
    """max_qpos_err must cover ranks 1..N-1: every rank contributes sampled rows of its own shard, and what rank 0 receives
    is labelled with the right global env ids (a sharding bug on another rank cannot hide)."""
    import bench
    world, n = 2, 16
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 33500 + os.getpid() % 2000
    mp.spawn(_parity_worker, args=(world, port, n, ret), nprocs=world, join=True)
    ids0, q0 = ret[0]
    rows = bench.parity_rows(n, world, 8)
    assert len(rows) == 4 and rows[0] == 0 and rows[-1] == n - 1            # 8 envs over 2 ranks, spread over the shard
    assert np.array_equal(ids0, np.concatenate([rows, n + rows]))           # rank-major, both ranks present
    assert np.array_equal(q0, np.outer(ids0 + 1.0, np.arange(1.0, 4.0)))
    assert set(ids0 // n) == {0, 1}


def test_env_count_defaults_follow_the_baseline_configs():
    import bench
    assert bench.resolve_envs(1) == (4096, "weak", "BASELINE configs[1]")                        # the headline, agrees with BENCH
    assert bench.resolve_envs(8)[:2] == (8192, "weak") and bench.resolve_envs(8)[2] == "BASELINE configs[2]"   # 65536 at 8 GPUs
    for w in (2, 4):
        assert bench.resolve_envs(w)[0] == 8192                                                   # configs[2]'s per-GPU shard
    for w in (1, 2, 4, 8):
        n, scaling, what = bench.resolve_envs(w, total_envs=65536)                                # the same 65536 envs at every N
        assert (n * w, scaling) == (65536, "strong") and "configs[2]" in what
    assert bench.resolve_envs(2, envs_per_gpu=4096)[0] == 4096
    with pytest.raises(SystemExit):
        bench.resolve_envs(3, total_envs=65536)
    with pytest.raises(SystemExit):
        bench.resolve_envs(2, envs_per_gpu=10, total_envs=20)


def test_parity_sample_and_snapshot_region():
    import bench
    assert len(bench.parity_rows(4096, 1, 64)) == 64
    assert len(bench.parity_rows(8192, 8, 64)) == 8 and len(bench.parity_rows(8192, 64, 64)) == 2   # never fewer than two per rank
    assert len(bench.parity_rows(3, 1, 64)) == 3
    # the driver's run (--steps 20 --warmup 5, 10 regions): the replay reaches the end of the last region
    assert bench.snapshot_region(20, 5, 10) == 9
    # the default run (--steps 1000 --warmup 100): the replay stops after the first region (2100 steps)
    assert bench.snapshot_region(1000, 100, 10) == 0
    assert bench.snapshot_region(5000, 100, 3) == 0 and bench.snapshot_region(100, 0, 10) == 9


def test_shards_partition_the_env_range():
    import bench
    ids = np.concatenate([bench.shard_env_ids(r, 8, 8192) for r in range(8)])
    assert np.array_equal(ids, np.arange(65536))                            # BASELINE config 3: 8 x 8192


def test_env_ranges_and_restart_rows_per_stream():
    """bench.half_ranges / rows_of_group_in_range: the ranges partition a rank's envs, and the rows a range restarts at a
    policy step are exactly the rows of that range whose GLOBAL env id is in the step's phase group."""
    import bench
    assert bench.half_ranges(4096, 2) == [(0, 2048), (2048, 2048)] and bench.half_ranges(4096, 1) == [(0, 4096)]
    assert bench.half_ranges(10, 3) == [(0, 3), (3, 3), (6, 4)] and bench.half_ranges(2, 5) == [(0, 1), (1, 1)]
    for rank, n, k in ((0, 4096, 2), (3, 8192, 2), (1, 100, 3)):
        ids = bench.shard_env_ids(rank, 8, n)
        for g in range(bench.NGROUP):
            want = np.nonzero(ids % bench.NGROUP == g)[0]
            got = []
            for first, cnt in bench.half_ranges(n, k):
                r0, c = bench.rows_of_group_in_range(g, int(ids[0]), first, cnt)
                rows = r0 + bench.NGROUP * np.arange(c)
                assert c == 0 or (rows[0] >= first and rows[-1] < first + cnt)
                got.append(rows)
            assert np.array_equal(np.concatenate(got), want), (rank, n, k, g)


def test_bench_main_starts_its_own_ranks_and_prints_one_line_for_all_of_them():
    """`python3 bench.py --gpus 2` the way the driver starts N = 1 -- no launcher, WORLD_SIZE unset: main() must start the two
    ranks itself (torch.distributed.run on 127.0.0.1, a port picked free), shard the envs, run the per-range launch /
    restart / gather schedule and the fenced regions on every rank, and rank 0 must print ONE JSON line with n_gpus = 2, the
    gathered observation block verified and the parity rows of BOTH ranks labelled with their rank.  Runs on CPU ranks
    (gloo) with the stand-in for the GPU batch (tests/bench_standin.py, `--dry-run-cpu`): no physics, the control flow is
    what is under test -- a mislabelled shard or a gather out of global env order shows as a non-zero stand-in error."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "60", "--warmup", "20", "--repeats", "2",
           "--envs-per-gpu", "48", "--parity-envs", "12", "--dry-run-cpu"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 60 and d["warmup"] == 20 and d["scaling"] == "weak"
    assert d["config"]["envs_total"] == 96 and d["config"]["parallelism"] == "env-sharded x2"
    assert d["obs_allgather_ok"] is True
    assert d["config"]["obs_allgather_every_steps"] == 50
    per_rank = d["parity"]["max_qpos_err_per_rank"]
    assert sorted(per_rank) == ["0", "1"] and d["parity"]["ranks_compared"] == 2
    assert d["max_qpos_err"] == 0.0 and all(v == 0.0 for v in per_rank.values())
    assert d["value"] > 0 and d["value_min"] <= d["value"] <= d["value_max"]
    assert "stand-in" in d["data"]                     # nobody can mistake this line for a measurement
    # an N > 1 line is complete: rank 0 times the CPU oracle behind the regions (the other ranks wait), and the roofline object says
    # that its figures are per GPU
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and "sample" in cb
    rf = d["roofline"]
    assert rf["scope"] == "per GPU" and rf["peak_node"] == 2 * rf["peak"] and abs(rf["achieved_node"] - 2 * rf["achieved"]) < 1e-9 * rf["achieved_node"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(rf)


def test_bench_refuses_a_rank_count_that_is_not_the_launchers():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-run-cpu"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stderr + out.stdout)


def _product_worker(rank, world, port, n, ret):
    """cassie_amd.distributed itself on two gloo ranks: shard ids, the observation block bound with a row stride, the
    overlapped gather of two env ranges per rank over three policy steps."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, os.path.join(REPO, "cassie-mujoco-sim_amd"))
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from cassie_amd import distributed as cd
    import bench_standin
    r, w, _ = cd.init_ranks("gloo")
    assert (r, w) == (rank, world)
    ids = cd.shard_env_ids(rank, world, n)

    class Pod:
        nq, nv, nsensordata = 35, 32, 29

    class FakeBatch:
        nenv = n
        bound = {}

        def bind(self, field, ptr, row_stride=None):
            self.bound[field] = (ptr, row_stride)

    fb = FakeBatch()
    init = torch.arange(96, dtype=torch.float64)
    ob = cd.ObservationBlock(fb, Pod, torch.device("cpu"), init)
    from cassie_amd import phys as P
    esz = ob.tensor.element_size()
    assert fb.bound[P.F_QPOS] == (ob.tensor.data_ptr(), 96) and fb.bound[P.F_QVEL] == (ob.tensor.data_ptr() + 35 * esz, 96)
    assert fb.bound[P.F_SENSORDATA] == (ob.tensor.data_ptr() + 67 * esz, 96)
    assert ob.qpos.shape == (n, 35) and ob.qvel.shape == (n, 32) and ob.sensordata.shape == (n, 29) and torch.equal(ob.tensor[3], init)
    rt = bench_standin.CpuRuntime(rank)
    ranges = cd.env_ranges(n, 2)
    streams = [rt.Stream() for _ in ranges]
    og = cd.OverlappedGather(ob.tensor, ranges, streams, world, rt)
    seen = []
    for p in range(3):
        ob.tensor[:, 0] = torch.from_numpy(ids.astype(np.float64)) * 1000 + p      # "a launch wrote the block"
        og.gather()
        seen.append([res[:, 0].clone().numpy() for res in og.result])
    ret[rank] = (seen, og.holds_own_rows(rank), og.count)
    dist.barrier()
    dist.destroy_process_group()


def test_product_module_shards_and_gathers_on_two_ranks():
    """cassie_amd.distributed (the multi-GPU layer is product code, not bench code): every gather delivers both ranks' rows of
    the SAME policy step, range by range, in global env order."""
    from cassie_amd import distributed as cd
    world, n = 2, 10
    ret = mp.Manager().dict()
    mp.spawn(_product_worker, args=(world, cd.free_port(), n, ret), nprocs=world, join=True)
    ranges = cd.env_ranges(n, 2)
    for r in range(world):
        seen, own, count = ret[r]
        assert own and count == 3
        for p in range(3):
            for (first, cnt), col in zip(ranges, seen[p]):
                want = np.concatenate([(cd.shard_env_ids(k, world, n)[first:first + cnt]) * 1000.0 + p for k in range(world)])
                assert np.array_equal(col, want), (r, p, first)
    assert cd.rows_of_group_in_range(3, 100, 0, 50, 20) == (3, 3)     # global ids 103, 123, 143 -> rows 3, 23, 43
