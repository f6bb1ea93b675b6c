"""Multi-GPU layer of the batched Cassie step: one process per GPU, envs sharded in contiguous blocks, no
collective on the data path, ONE exchange per policy step -- an all-gather of the observation block -- over RCCL / xGMI
(torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests).

The reference has no counterpart: its users run one ``cassie_sim_t`` per OS process (every simulator owns its model and data,
reference src/cassiemujoco.c:255-265; nothing is shared but the read-only initial model), which is exactly why the envs shard
without an exchange step.  What an RL loop does need from all ranks, once per policy step, is the observations; this module
holds that path:

  shard_env_ids / env_ranges / rows_of_group_in_range   who owns which env, and how a rank's shard is split into the ranges
                                                        that are stepped on their own streams
  ObservationBlock                                      the [n, nq + nv + nsensordata] tensor the step kernel reads and
                                                        writes IN PLACE (qpos | qvel | sensordata as strided column blocks
                                                        of one allocation: phys_batch_bind_strided) -- the very buffer the
                                                        all-gather sends, no staging copy into an observation layout
  OverlappedGather                                      the all-gather of a range's block BESIDE the range's next launch:
                                                        snapshot on the launch stream (device to device), collective on a
                                                        second stream, the next snapshot waits for the gather to have read
  init_ranks / launch_ranks / free_port                 rank start-up: under an external launcher (torch.distributed.run:
                                                        RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or
                                                        started by this process itself, rendezvous on 127.0.0.1

bench.py is a user of this module (its N > 1 path), tests/test_multirank.py drives it under gloo with world_size 2.
"""
import os
import sys

import numpy as np


# ----------------------------------------------------------------------------------------------- partition ----
def shard_env_ids(rank, world, envs_per_rank):
    """Contiguous block of global env ids owned by `rank` (weak scaling: the per-rank count is fixed)."""
    return np.arange(rank * envs_per_rank, (rank + 1) * envs_per_rank)


def env_ranges(n, nstreams):
    """Contiguous env ranges [(first, count)] a rank's batch is stepped in, one per stream (the last takes the remainder).
    Ranges of one batch may be in flight on different streams at once (phys_batch_step_range): one range's workgroups fill the
    slots the other leaves idle at the ends of its launches."""
    k = max(1, min(int(nstreams), n))
    base = n // k
    return [(i * base, base if i < k - 1 else n - i * base) for i in range(k)]


def rows_of_group_in_range(group, global_first, first, count, ngroup):
    """Rows of the range [first, first + count) whose GLOBAL env id (global_first + row) is in phase group `group` of `ngroup`:
    (first row, number of rows), the rows being `ngroup` apart -- what phys_batch_reset_envs takes (first, stride, count)."""
    r0 = first + (group - (global_first + first)) % ngroup
    return r0, len(range(r0, first + count, ngroup))


# --------------------------------------------------------------------------------------------- collectives ----
def gather_observations(obs, world, out=None):
    """All-gather of the per-rank observation block [n, nobs] into [world * n, nobs], rank-major = global env order."""
    import torch
    import torch.distributed as dist
    if out is None:
        out = torch.empty((world * obs.shape[0], obs.shape[1]), dtype=obs.dtype, device=obs.device)
    dist.all_gather_into_tensor(out, obs)
    return out


def gather_rows(x, world):
    """All-gather of equally shaped per-rank row blocks, rank-major (e.g. the sampled parity rows of every rank)."""
    import torch
    import torch.distributed as dist
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x.contiguous())
    return out


class ObservationBlock:
    """[n, nq + nv + nsensordata] fp64 on `device`, bound to `batch` so that the step kernel's qpos / qvel / sensordata ARE its
    column blocks (row stride = the block's width).  `init_row` ([nobs]) fills every row."""

    def __init__(self, batch, pod, device, init_row=None):
        import torch
        from . import phys as P
        self.nq, self.nv, self.nsd = pod.nq, pod.nv, pod.nsensordata
        self.width = self.nq + self.nv + self.nsd
        n = batch.nenv
        self.tensor = (torch.zeros((n, self.width), dtype=torch.float64, device=device) if init_row is None
                       else init_row.to(device=device, dtype=torch.float64).repeat(n, 1).contiguous())
        esz = self.tensor.element_size()
        batch.bind(P.F_QPOS, self.tensor.data_ptr(), row_stride=self.width)
        batch.bind(P.F_QVEL, self.tensor.data_ptr() + self.nq * esz, row_stride=self.width)
        batch.bind(P.F_SENSORDATA, self.tensor.data_ptr() + (self.nq + self.nv) * esz, row_stride=self.width)

    qpos = property(lambda s: s.tensor[:, : s.nq])
    qvel = property(lambda s: s.tensor[:, s.nq: s.nq + s.nv])
    sensordata = property(lambda s: s.tensor[:, s.nq + s.nv:])


class OverlappedGather:
    """The observation all-gather of every env range of a rank, beside the range's next launch.

    gather() is called between two launches of the ranges: range i's rows are copied on ITS launch stream into a snapshot
    (so the copy is ordered behind the launch that wrote them and ahead of the next), a second stream waits for that copy and
    runs the collective, and range i's NEXT snapshot waits for the collective to have read the previous one.  Nothing here
    blocks the host; `result[i]` ([world * count_i, nobs], rank-major) is valid once `done[i]` has completed.
    `runtime` supplies Stream() / Event() / use(stream) (torch.cuda on a GPU; a stand-in in the CPU tests)."""

    def __init__(self, obs, ranges, streams, world, runtime, collective=gather_observations):
        import torch
        self.obs, self.ranges, self.streams, self.world, self.rt, self.collective = obs, ranges, streams, world, runtime, collective
        self.snap = [torch.empty((cnt, obs.shape[1]), dtype=obs.dtype, device=obs.device) for _, cnt in ranges]
        self.result = [torch.empty((world * cnt, obs.shape[1]), dtype=obs.dtype, device=obs.device) for _, cnt in ranges]
        self.comm_streams = [runtime.Stream() for _ in ranges]
        self.done = [None] * len(ranges)
        self.count = 0

    def gather(self):
        for i, ((first, cnt), st) in enumerate(zip(self.ranges, self.streams)):
            if self.done[i] is not None:
                st.wait_event(self.done[i])
            with self.rt.use(st):
                self.snap[i].copy_(self.obs[first:first + cnt], non_blocking=True)
            ready = self.rt.Event()
            ready.record(st)
            with self.rt.use(self.comm_streams[i]):
                self.comm_streams[i].wait_event(ready)
                self.collective(self.snap[i], self.world, self.result[i])
                done = self.rt.Event()
                done.record(self.comm_streams[i])
            self.done[i] = done
        self.count += 1

    def holds_own_rows(self, rank):
        """Validation aid: the gathered block of the last gather holds this rank's snapshot where its rows belong."""
        import torch
        return bool(self.count > 0 and all(torch.equal(res[rank * cnt:(rank + 1) * cnt], sn)
                                           for res, sn, (_, cnt) in zip(self.result, self.snap, self.ranges)))


# ------------------------------------------------------------------------------------------- rank start-up ----
def free_port():
    """A TCP port that is free on 127.0.0.1 right now (the rendezvous of ranks this process starts itself)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(ngpus, script, argv):
    """Starts `script` as `ngpus` ranks of ONE node under torch.distributed.run (rank r binds GPU r through LOCAL_RANK),
    rendezvous on 127.0.0.1 at a port picked free, dmabuf IPC for RCCL between the ranks' processes; passes the ranks' output
    through and returns the launcher's exit code."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(script)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // ngpus)))
    return subprocess.call(cmd, env=env)


def init_ranks(backend="nccl", device=None):
    """(rank, world, local_rank) of this process from the launcher's environment; initialises the process group when
    WORLD_SIZE > 1 (or `force`d by passing a backend with WORLD_SIZE unset = single rank: MASTER_* default to 127.0.0.1 and a
    free port).  `device`: a torch.device to bind the group to (RCCL communicator on that GPU)."""
    import torch.distributed as dist
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        kw = {"device_id": device} if (device is not None and backend == "nccl") else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local
