#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched Cassie physics step on MI355X, and its error against the CPU reference.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1 is launched by
torch.distributed.run, one rank per GPU).  A *step* is one pass of the hot path over
one batch: every env of the batch advances by one 0.5 ms physics step
(= one cassie_sim_step_pd-equivalent; reference src/cassiemujoco.c:1130-1134).

Workload (BASELINE.json configs[1]): 4096 envs per GPU, cassie model, EPISODES of 1000 steps from the
cassie_sim_init pose (reference src/cassiemujoco.c:1023-1029); per-env random PD targets (offset + U(-0.3, 0.3)
rad, gains of reference example/cassietest_jac.py:51-52, :68) re-drawn every 50 steps from a table that is resident
in HBM.  Everything of a cassie_sim_step_pd that is not a closed Agility block runs on the device inside the step
kernel (--mode drive-pd, the default: pd_input's motor PD on the encoder measurements, the motor model with its
torque delay, the encoder models, the physics; --mode exact-pd: PD on the exact joint state, no delay), so no host
buffer is touched in the timed region.  Episodes are STAGGERED: env e restarts from the init pose every 1000 steps at phase
50 * (e mod 20), so at any moment the batch holds every phase of the episode in equal parts and the measured rate is
the episode average whatever --steps / --warmup are (synchronised episodes would make a short run measure whichever
phase it happened to land in -- e.g. the contact-free first 3.7 mm of the drop).  Before the warm-up the schedule is
advanced 1000 untimed steps so that the mix is the stationary one.
With N>1 envs shard across GPUs with no data-path exchange; the only collective is one RCCL all-gather of the
observation block (qpos|qvel|sensordata, 96 doubles/env, ONE tensor the kernel writes in place) per 50 steps.
Env counts: one GPU runs BASELINE configs[1] (4096 envs); N > 1 GPUs run configs[2]'s shard, 8192 envs per GPU = 65536
at 8 GPUs (weak scaling); `--total-envs T` shards a fixed total instead (strong scaling).
Timing: `--repeats` (10) fenced regions of exactly `--steps` steps each, one after the other on the stationary mix; every
region is bracketed by barrier + torch.cuda.synchronize on both sides and counted at its slowest rank; `value` is the
MEDIAN region, `value_min` / `value_max` the slowest / fastest.

The JSON line carries, beside the contract's fields:
  max_qpos_err    BASELINE.json's second half of the metric: sampled envs of EVERY rank's shard of the timed batch are
                  replayed on the CPU reference (oracle/, the fp64 restatement of mj_step1 + mj_step2) through the same
                  schedule (pre-roll, warm-up, timed steps, restarts) and the qpos at a timed region's end compared
  value_exact_pd  (or value_drive_pd) a short run of the other device mode, with its own parity figure
  value_all_outputs_every_substep
                  `value` when every substep of a launch also forms the outputs nobody can read: a launch of 50 fused substeps
                  returns its LAST substep's sensordata / measurement block / xquat, so by default the IMU sensor words and
                  body quaternions of the 48 substeps whose values are neither returned nor fed back are not evaluated
                  (state trajectory and returned outputs are bit for bit the same either way)
  value_step_pd   the same workload through the drop-in API itself (cassie_batch_step_pd = cassie_sim_step_pd for every
                  env: Agility blocks + encoder / motor models, see include/cassie_batch.h) -- PCIe- and host-inclusive
  roofline        algorithmic HBM bytes of one launch (1976 B per env-step) divided by the mean kernel
                  duration measured with HIP events on the launch stream
  cpu_baseline    the fp64 CPU oracle ("port") on the host cores, OpenMP over envs, same workload, bounded sample
  true_reference  genuine MuJoCo (libmujoco210 / `mujoco` wheel) timed and compared on the same inputs when one is
                  discoverable at run time, otherwise the string "unavailable"
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "cassie-mujoco-sim_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))

ALGO_BYTES_PER_ENV_STEP = 1976      # read 109 + write 138 doubles
HBM_PEAK_GBS = 8000.0               # MI355X spec (MI355X_MICROARCH.md)
HOLD = 50                           # substeps per policy step (reference example/test_hfield.c:108)
EPISODE = 1000                      # steps per episode ("run 1000 substeps")
NGROUP = EPISODE // HOLD            # restart phases: env e restarts at policy steps p with p % NGROUP == e % NGROUP
PREROLL = EPISODE                   # untimed steps before the warm-up: reaches the stationary mix of episode phases
PD_OFFSET = np.array([0.0045, 0, 0.4973, -1.1997, -1.5968] * 2)
PD_KP = np.array([70, 70, 100, 100, 50] * 2, dtype=np.float64)
PD_KD = np.array([7, 7, 8, 8, 5] * 2, dtype=np.float64)


# the multi-GPU layer is part of the product: sharding, the observation block the kernel writes in place, its
# all-gather beside the next launch, rank start-up
from cassie_amd.distributed import (ObservationBlock, OverlappedGather, env_ranges as half_ranges, free_port, gather_observations,  # noqa: E402,F401
                                    gather_rows, shard_env_ids)
from cassie_amd import distributed as cdist  # noqa: E402


def rows_of_group_in_range(group, global_first, first, count):
    """Rows of the range [first, first + count) whose global env id is in restart phase group `group` (NGROUP apart)."""
    return cdist.rows_of_group_in_range(group, global_first, first, count, NGROUP)


def launch_ranks(ngpus, argv):
    """`python bench.py --gpus N` without a launcher around it (WORLD_SIZE unset): start the N ranks ourselves -- this very file
    under torch.distributed.run, one rank per GPU -- and pass rank 0's ONE JSON line through (cassie_amd.distributed.launch_ranks)."""
    return cdist.launch_ranks(ngpus, __file__, argv)


TARGET_SPREAD = 0.3                 # rad; --target-spread 10 is the stress workload (reference example/cassietest_jac.py:106)


def pd_targets(env_ids, npolicy):
    """[npolicy][len(env_ids)][10] targets; env e uses numpy.random.default_rng(1234 + e)."""
    out = np.empty((npolicy, len(env_ids), 10))
    for i, e in enumerate(env_ids):
        out[:, i, :] = PD_OFFSET + np.random.default_rng(1234 + int(e)).uniform(-TARGET_SPREAD, TARGET_SPREAD, (npolicy, 10))
    return out


ENVS_PER_GPU_SINGLE = 4096          # BASELINE configs[1]: the headline, one GPU
ENVS_PER_GPU_SHARDED = 8192         # BASELINE configs[2]: 65536 envs sharded over 8 GPUs
STREAMS = 2                         # env ranges per rank, each stepped on its own stream (see device_rollout)
REPEATS = 10                        # fenced timed regions of exactly --steps steps each; `value` is their median
REPLAY_BUDGET_STEPS = 2200          # the CPU replay follows the schedule up to the last region boundary within this many steps


def resolve_envs(world, envs_per_gpu=None, total_envs=None):
    """(envs per rank, scaling, which BASELINE config the shape is).  One GPU runs configs[1] (4096 envs); N > 1 GPUs run
    configs[2]'s shard, 8192 envs per GPU = 65536 at 8 GPUs (weak scaling); --total-envs T shards a FIXED total T over
    the ranks instead (strong scaling; 65536 envs fit one GPU, 0.3 GB); --envs-per-gpu overrides the per-rank count."""
    if total_envs is not None:
        if envs_per_gpu is not None:
            raise SystemExit("--total-envs and --envs-per-gpu exclude each other")
        if total_envs <= 0 or total_envs % world:
            raise SystemExit("--total-envs must be a positive multiple of the number of GPUs")
        n = total_envs // world
        return n, "strong", ("BASELINE configs[2] (65536 envs)" if total_envs == 65536 else "%d envs in total" % total_envs)
    if envs_per_gpu is not None:
        n = envs_per_gpu
    else:
        n = ENVS_PER_GPU_SINGLE if world == 1 else ENVS_PER_GPU_SHARDED
    what = ("BASELINE configs[1]" if (world, n) == (1, ENVS_PER_GPU_SINGLE) else
            "BASELINE configs[2]" if world * n == 65536 else
            "BASELINE configs[2]'s per-GPU shard" if n == ENVS_PER_GPU_SHARDED else "custom")
    return n, "weak", what


def parity_rows(n, world, parity_envs):
    """Local rows of a rank's shard that are replayed on the CPU reference: the budget of `parity_envs` envs is spread
    over ALL ranks (at least two per rank), evenly spaced over the shard, so a fault on any rank shows in max_qpos_err."""
    k = max(1, min(n, max(2, parity_envs // world) if world > 1 else parity_envs))
    return np.unique(np.linspace(0, n - 1, k).astype(int))


def snapshot_region(steps, warmup, repeats):
    """Index of the timed region after which the sampled envs are snapshotted for the CPU replay: the last one the replay
    reaches within REPLAY_BUDGET_STEPS (the first one at least)."""
    r = (REPLAY_BUDGET_STEPS - PREROLL - warmup) // max(1, steps) - 1
    return int(min(repeats - 1, max(0, r)))


def restart_group(policy_step):
    """The phase group whose envs restart from the init pose at this policy step."""
    return policy_step % NGROUP


class Schedule:
    """The launch / restart / gather schedule of the benchmark, independent of what executes it (the GPU batch here, a
    stand-in in tests/test_multirank.py).  Physics steps are counted from 0; at every multiple of HOLD the PD targets
    of that policy step are bound, the envs of the due phase group restart, and (world > 1) the observation block
    the previous launches left is all-gathered.  One launch advances every env up to the next policy step at most."""

    def __init__(self, step, bind_targets, restart, gather=None, substeps_per_launch=HOLD):
        self.step, self.bind_targets, self.restart, self.gather = step, bind_targets, restart, gather
        self.substeps_per_launch = substeps_per_launch
        self.launches = 0
        self.gathers = 0

    def run(self, first, count):
        s, end = first, first + count
        while s < end:
            if s % HOLD == 0:
                p = s // HOLD
                if self.gather is not None and s > 0:
                    self.gather()
                    self.gathers += 1
                self.bind_targets(p)
                self.restart(restart_group(p))
            nsub = min(self.substeps_per_launch, HOLD - s % HOLD, end - s)
            self.step(nsub)
            self.launches += 1
            s += nsub
        return s


def timed_region(schedule, first, steps, fence, clock=time.perf_counter, mark=None):
    """fence, time exactly `steps` steps of the schedule, fence: wall seconds of this rank.  `mark(i)` (i = 0 at the
    start, 1 at the end) lets the caller drop stream events around the same region."""
    fence()
    schedule.launches = 0
    if mark:
        mark(0)
    t0 = clock()
    schedule.run(first, steps)
    if mark:
        mark(1)
    fence()
    return clock() - t0


def _reduce_max(seconds, device):
    """max-over-ranks through the collective even when there is a single rank (validation runs)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def max_over_ranks(seconds, world, device=None):
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def has_fast_kernel(model_name):
    """Models whose stepping launches start the row-capped fast instantiation (phys_batch.hip, launch)."""
    return model_name in ("cassie", "cassie_hfield", "cassie_tray_box")


def handed_over_in_last_launch(progress, nsub_of_last_launch):
    """Envs the fast kernel handed over to the full kernel in the last launch: PhysIO::progress holds the substeps the fast
    kernel completed, so an env was handed over iff that is short of the launch's OWN substep count (not of HOLD: the
    driver's `--steps 20` regions end with a 20-substep launch)."""
    return float(np.count_nonzero(np.asarray(progress) < int(nsub_of_last_launch)))


def launch_io_bytes_per_env(pod, drive=True):
    """What ONE launch moves per env whatever its substep count: the state it loads and the state + last-substep outputs it
    stores (fields of csrc/physics_kernel.h env_step's load / store blocks, in doubles)."""
    nq, nv, nu, nsd, nb = pod.nq, pod.nv, pod.nu, pod.nsensordata, pod.nbody
    drv = 1240 // 8 if drive else 0                                   # cm_drive_state_t
    load = nq + 2 * nv + nu + 1 + (nsd + nu + drv + 2 * 10 + 5 * 10 if drive else 3 * nu)
    store = nq + 2 * nv + 1 + nv + nsd + nu + 7 * nb + 2 + (drv + 56 + nu if drive else 0)
    return 8 * (load + store)


def launch_chunks(envs_per_launch, substeps_per_launch, whole_batch):
    """Workgroups per env a stepping launch of this shape is dispatched as (phys_batch.hip: launches in chunks -- 4 for a launch
    over the whole batch (7 since round 6), 2 for one over an env range -- 3 when that launch is at most 25 substeps long --, chunks of at least 5
    substeps, launches of at least 2048 envs)."""
    asked = int(os.environ.get("CASSIE_CHUNKS") or (7 if whole_batch else (3 if substeps_per_launch <= 25 else 2)))
    if envs_per_launch < 2048 or envs_per_launch % 8 or substeps_per_launch < 10:
        return 1
    return max(1, min(asked, int(substeps_per_launch) // 5))


def pmc_traffic(env_steps_per_launch, model_name="cassie", envs_per_launch=None, pod=None, chunks=1):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (tools/gpu_pmc_all.sh: FETCH_SIZE and WRITE_SIZE in
    separate --pmc runs, corrected as MI355X_MICROARCH.md prescribes), brought to this run's launch shape: the passes
    profile 50-substep launches, and a launch's traffic is a per-env part that does not depend on the substep count (state
    in, state and last-substep outputs out: launch_io_bytes_per_env) plus a per-env-step part (model constants, terrain,
    spills) -- only the second scales with the substeps.  The counters cannot be collected from inside the timed process, so
    this is the last measured figure, or None."""
    import glob
    import re

    def version(path):   # profiles/roundR/vN_pmc_summary.json -> (R, N)
        m = re.search(r"round(\d+).*?v(\d+)_pmc_summary", path)
        return (int(m.group(1)), int(m.group(2))) if m else (0, 0)
    suffix = {"cassie": "", "cassie_hfield": "_hfield", "cassie_tray_box": "_tray"}.get(model_name, "")   # the passes of this model's kernel
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "round*", "*pmc_summary%s.json" % suffix)), key=version)
    for path in reversed(files):
        try:
            d = json.load(open(path))["derived"]
            total, es = d["hbm_read_bytes_per_launch"] + d["hbm_write_bytes_per_launch"], d["env_steps_per_launch"]
            if envs_per_launch is None or pod is None:
                return total / es * env_steps_per_launch, os.path.relpath(path, REPO)
            envs_pmc = d.get("envs_per_launch", 4096)
            fixed = launch_io_bytes_per_env(pod)      # per env and CHUNK: every chunk of a launch loads and stores like a launch
            per_env_step = max(0.0, total - fixed * envs_pmc * d.get("chunks_per_env_launch", 1)) / es
            return fixed * envs_per_launch * chunks + per_env_step * env_steps_per_launch, os.path.relpath(path, REPO)
        except (KeyError, ValueError, OSError):
            continue
    return None, None


class OracleEnvs:
    """A set of envs on the CPU reference (oracle/cassie_oracle.c), driven through the benchmark's schedule."""

    def __init__(self, model, env_ids, hfield=None, pods=None):
        import oracle_py
        self.op, self.model, self.ids = oracle_py, model, np.asarray(env_ids)
        self.pods = pods            # per-env compiled models (domain randomisation: one cm_model_t per replayed env) or None
        self.L = oracle_py.lib()
        if hfield is not None:
            oracle_py.set_hfield(hfield)
        self.buf = (oracle_py.CoData * len(self.ids))()
        self.q0 = model.qpos_init()
        for i in range(len(self.ids)):
            self.reset(i)
        self.kp = np.tile(PD_KP, (len(self.ids), 1))
        self.kd = np.tile(PD_KD, (len(self.ids), 1))

    def pod(self, i):
        return self.pods[i] if self.pods is not None else self.model.pod

    def reset(self, i):
        self.L.co_reset(ctypes.byref(self.pod(i)), ctypes.byref(self.buf[i]))
        self.op.arr(self.buf[i].qpos)[: self.model.pod.nq] = self.q0

    def restart(self, group):
        for i, e in enumerate(self.ids):
            if int(e) % NGROUP == group:
                self.reset(i)

    def step(self, nsub, targets, threads):
        pt = np.ascontiguousarray(targets)
        if self.pods is not None:   # (one model per env: env by env)
            for i in range(len(self.ids)):
                self.L.co_step_batch(ctypes.byref(self.pods[i]), ctypes.byref(self.buf[i]), 1, nsub, pt[i:i + 1].ctypes.data, self.kp.ctypes.data, self.kd.ctypes.data, 1)
            return
        self.L.co_step_batch(ctypes.byref(self.model.pod), ctypes.byref(self.buf), len(self.ids), nsub, pt.ctypes.data,
                             self.kp.ctypes.data, self.kd.ctypes.data, threads)

    def qpos(self):
        return np.array([self.op.arr(b.qpos)[: self.model.pod.nq].copy() for b in self.buf])

    def counts(self):
        return np.array([[b.ncon, b.nefc, b.solver_iter] for b in self.buf])


def replay_on_oracle(model, env_ids, targets_of, total_steps, hfield=None, threads=1, envs=None, pods=None):
    """The schedule of the timed batch (restarts, PD targets, step count) for the envs `env_ids` on the CPU reference.
    targets_of(p) -> [len(env_ids)][10] targets of policy step p.  `envs`: OracleEnvs (exact-state PD) or HostChainEnvs."""
    o = (envs or OracleEnvs)(model, env_ids, hfield, pods=pods) if pods is not None else (envs or OracleEnvs)(model, env_ids, hfield)
    cur = {}
    sch = Schedule(step=lambda nsub: o.step(nsub, cur["t"], threads),
                   bind_targets=lambda p: cur.__setitem__("t", targets_of(p)),
                   restart=o.restart)
    sch.run(0, total_steps)
    return o


def cpu_baseline(model, budget_s=12.0, hfield=None):
    """Times the CPU oracle on a bounded sample of the same workload (staggered 1000-step episodes under the PD
    workload), all usable host cores (OpenMP over envs)."""
    from cassie_amd._lib import lib
    cores = lib().cassie_host_cpu_count()      # affinity- and cgroup-quota-aware
    nenv = 16 * cores
    ids = np.arange(nenv)
    o = OracleEnvs(model, ids, hfield)
    npol = 400
    tg = pd_targets(ids, npol)
    done, t_used, pol = 0, 0.0, 0
    while t_used < budget_s and pol < npol:
        o.restart(restart_group(pol))
        t0 = time.perf_counter()
        o.step(HOLD, tg[pol], cores)
        t_used += time.perf_counter() - t0
        done += nenv * HOLD
        pol += 1
    return {"value": done / t_used, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d envs x %d steps, same PD workload and episode schedule, oracle/cassie_oracle.c, OpenMP over envs" % (nenv, pol * HOLD)}


def step_pd_host_api(n, steps=200, warmup=50, device_drives=False):
    """The full cassie_sim_step_pd semantics for n envs (include/cassie_batch.h): Agility blocks + encoder / motor
    models on the host thread pool, ctrl / sensordata over PCIe every step, physics on the GPU.  Host-bound by the
    closed Agility code."""
    from cassie_amd._lib import MODEL_DIR, lib
    L = lib()
    L.cassie_batch_create.restype = ctypes.c_void_p
    L.cassie_batch_create.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.cassie_batch_step_pd.argtypes = [ctypes.c_void_p] * 3
    L.cassie_batch_free.argtypes = [ctypes.c_void_p]
    L.cassie_batch_nthreads.argtypes = [ctypes.c_void_p]
    L.cassie_batch_set_device_drives.argtypes = [ctypes.c_void_p, ctypes.c_int]
    b = L.cassie_batch_create(os.path.join(MODEL_DIR, "cassie.cmodel").encode(), n, 0, 0)
    if not b:
        return None
    if device_drives and L.cassie_batch_set_device_drives(b, 1) != 0:
        L.cassie_batch_free(b)
        return None
    u = np.zeros((n, 119))                      # pd_in_t as 119 doubles: [left task 30 | left motor 25 | right 55 | telemetry 9]
    y = np.zeros((n, 124))                      # state_out_t is 992 bytes
    for base in (30, 85):
        u[:, base + 15: base + 20] = PD_KP[:5]
        u[:, base + 20: base + 25] = PD_KD[:5]
    tg = pd_targets(range(n), (warmup + steps) // HOLD + 2)
    t0 = None
    for s in range(warmup + steps):
        if s == warmup:
            t0 = time.perf_counter()
        if s % HOLD == 0:
            u[:, 35:40] = tg[s // HOLD][:, :5]
            u[:, 90:95] = tg[s // HOLD][:, 5:]
        L.cassie_batch_step_pd(b, u.ctypes.data, y.ctypes.data)
    dt = time.perf_counter() - t0
    nthreads = L.cassie_batch_nthreads(b)
    L.cassie_batch_free(b)
    return {"value": n * steps / dt, "unit": "env-steps/s", "host_threads": nthreads, "host_cores_usable": L.cassie_host_cpu_count(), "host_cores_online": os.cpu_count(),
            "ms_per_step": 1e3 * dt / steps, "drive_level_models": "device" if device_drives else "host",
            "what": "cassie_batch_step_pd: pd_input + cassie_core_sim + state_output on host threads, motor / encoder models on the %s, "
                    "PCIe copies (%s) and the physics kernel every step"
                    % (("device (stand-alone drive pass ahead of the physics kernel)", "11 command doubles up, 56 measurement doubles down per env")
                       if device_drives else ("host threads", "10 ctrl doubles up, 39 sensordata / actuator_velocity doubles down per env"))}


class HostChainEnvs:
    """Envs on the CPU with the drive-level semantics of CM_DRIVE_PD: oracle physics + the host chain of
    csrc/cassie_hostpath.c (the reference's own encoder / motor arithmetic) + pd_input's motor PD on the measurements."""

    safe = False                # True: cassie_core_sim_step of the REAL Agility block between the PD law and the motor model (CM_DRIVE_PD_SAFE)

    def __init__(self, model, env_ids, hfield=None, pods=None):
        import oracle_py
        from cassie_amd import phys as P
        from hostchain_py import HostChain
        self.pods = pods            # per-env compiled models (domain randomisation) or None
        self.msgs = [[0, 0, 0, 0] for _ in env_ids]   # (safe: the block's message queue after the last step)
        if hfield is not None:
            oracle_py.set_hfield(hfield)
        self.model, self.ids, self.P = model, np.asarray(env_ids), P
        self.orcs = [None] * len(self.ids)
        self.chains = [HostChain(model) for _ in self.ids]
        self.meas = np.zeros((len(self.ids), P.MEAS_DIM))
        # Flip watch.  The encoder models truncate sensor / (2 pi) * 2^bits to an integer count; a device trajectory that
        # differs from this replay in the last bits can only truncate differently if the replay's own value sits that close
        # to a count boundary (a non-zero integer: truncation is continuous at zero).  flip_margin[i] is the smallest such
        # distance, in counts, env i has seen: an env whose margin stayed above ~1e-9 cannot have flipped a count.
        self.enc_slots = np.array([0, 1, 2, 3, 4, 8, 9, 10, 11, 12, 5, 6, 7, 13, 14, 15])
        self.enc_scale = np.array([float(1 << model.pod.sensor_bits[int(k)]) for k in self.enc_slots]) / (2 * np.pi)
        self.flip_margin = np.full(len(self.ids), np.inf)
        for i in range(len(self.ids)):
            self.reset(i, fresh_chain=False)

    def reset(self, i, fresh_chain=True):
        from oracle_py import Oracle
        self.orcs[i] = Oracle(self.pods[i] if self.pods is not None else self.model.pod, self.model.qpos_init())
        self.orcs[i].forward()                      # what cassie_sim_init leaves: the init pose's sensordata
        if fresh_chain:
            self.chains[i].reset()                  # a fresh cassie_sim_t: zero filter histories and delay lines
        self.meas[i] = 0

    def restart(self, group):
        for i, e in enumerate(self.ids):
            if int(e) % NGROUP == group:
                self.reset(i)

    def step(self, nsub, targets, threads=1):
        from hostchain_py import pd_command
        for i, (o, hc) in enumerate(zip(self.orcs, self.chains)):
            for _ in range(nsub):
                v = o.sensordata[self.enc_slots] * self.enc_scale
                k = np.rint(v)
                near = np.abs(v - k)[k != 0]
                if near.size:
                    self.flip_margin[i] = min(self.flip_margin[i], float(near.min()))
                cmd = pd_command(self.meas[i], targets[i], PD_KP, PD_KD)
                if self.safe:
                    cmd, self.msgs[i] = hc.core_sim(cmd)
                ctrl, self.meas[i], _y = hc.ethercat(cmd, False, o.sensordata.copy(), o.actuator_velocity.copy())
                o.ctrl[:] = ctrl
                o.step()

    def qpos(self):
        return np.array([o.qpos.copy() for o in self.orcs])

    def counts(self):
        return np.array([[o.d.ncon, o.d.nefc, o.d.solver_iter] for o in self.orcs])

    def init_sensordata(self):
        return self.orcs[0].sensordata.copy()


class SafeHostChainEnvs(HostChainEnvs):
    """HostChainEnvs with the REAL cassie_core_sim_step (libagilitycassie.a, linked into the product library) between pd_input's PD
    law and the motor model: the reference of CM_DRIVE_PD_SAFE."""
    safe = True


class GpuRuntime:
    """What device_rollout asks of the machine under it: the device, streams / events, the batch, the CPU reference for the
    replay.  This is the real one (HIP through torch, RCCL); tests/bench_standin.py has a CPU stand-in (gloo, no physics)
    through which tests/test_multirank.py drives main()'s N > 1 path on a box without GPUs (`--dry-run-cpu`)."""
    backend = "nccl"
    name = None

    def __init__(self, local_rank):
        import torch
        self.torch, self.local_rank = torch, local_rank
        self.device = torch.device("cuda", local_rank)

    def Stream(self):
        return self.torch.cuda.Stream(device=self.device)   # real (non-null) streams: the kernels and the timing events share them

    def Event(self, enable_timing=False):
        return self.torch.cuda.Event(enable_timing=enable_timing)

    def use(self, stream):
        return self.torch.cuda.stream(stream)

    def synchronize(self):
        self.torch.cuda.synchronize(self.device)

    def make_batch(self, model, n):
        from cassie_amd import Batch
        return Batch(model, n, device=self.local_rank)

    def init_sensordata(self, model, hfield):
        return HostChainEnvs(model, [0], hfield).init_sensordata()

    def replay_envs(self, drive):
        return HostChainEnvs if drive else OracleEnvs

    def host_threads(self):
        from cassie_amd._lib import lib
        return lib().cassie_host_cpu_count()


def device_rollout(model, mode, n, steps, warmup, rank, world, local_rank, substeps_per_launch=HOLD, parity_envs=64, hfield=None, collect=None,
                   all_outputs=False, repeats=1, nstreams=1, rt=None, randomise=None):
    """One device-resident rollout of the workload in `mode`, timed as `repeats` fenced regions of exactly `steps` steps:

      "drive-pd"  CM_DRIVE_PD: every substep runs pd_input's motor PD on the ENCODER measurements of the
                  previous step (13 / 18-bit truncation, integer FIR / IIR velocity filters), the motor model with its
                  speed-torque curve and six-cycle torque delay, then the physics -- the drive-level semantics of
                  cassie_sim_step_pd, bit for bit the host chain of csrc/cassie_hostpath.c, minus the closed Agility blocks'
                  safety layer and estimator.  An episode restart is a fresh cassie_sim_t (init pose, zero filters / delays).
      "drive-pd-safe"  CM_DRIVE_PD_SAFE: the same with cassie_core_sim's safety layer between the PD law and the motor model
                  (joint-limit attenuation / restoring torques, torque-limit clamp, STO; csrc/pk_safety.h, bit for bit the closed
                  Agility block): the WHOLE torque path of cassie_sim_step_pd (reference src/cassiemujoco.c:1147-1157); what stays on
                  the host is the state estimator, which feeds nothing back into the simulation.
      "exact-pd"  the PD law on the exact joint state + the motor's speed-torque limit, no delay, no quantisation.

    nstreams > 1: the rank's batch is stepped as that many contiguous env ranges, each on its own stream at its own pace
    (phys_batch_step_range): the CPU issues policy step p of every range, then p + 1, ..., but nothing on the device joins the
    ranges between policy steps, so one range's workgroups fill the wave slots the other leaves idle at the end and the
    start of its launches.  Restarts, PD targets and (N > 1) the observation all-gather are per range.

    State and inputs are resident in HBM before the timed regions.  Every region is bracketed by barrier +
    torch.cuda.synchronize on both sides; a region's time is the maximum over the ranks.  Sampled envs of EVERY rank are
    snapshotted at a region boundary and compared on rank 0 with their replay on the CPU reference."""
    import torch
    import torch.distributed as dist
    from cassie_amd import phys as P
    rt = rt or GpuRuntime(local_rank)
    pod = model.pod
    drive = mode in ("drive-pd", "drive-pd-safe")
    safe = mode == "drive-pd-safe"
    collect = world > 1 if collect is None else collect       # the observation all-gather, barriers, max-over-ranks reduction
    env_ids = shard_env_ids(rank, world, n)
    snap_r = snapshot_region(steps, warmup, repeats)
    total_steps = PREROLL + warmup + repeats * steps
    replay_steps = PREROLL + warmup + (snap_r + 1) * steps
    npolicy = (total_steps + HOLD - 1) // HOLD + 1
    dev = rt.device
    b = rt.make_batch(model, n)
    if all_outputs:
        b.set_all_outputs_every_substep(True)
    if os.environ.get("CASSIE_NO_FAST_ROWS"):
        b.set_fast_rows(False)          # A/B switch: the full step kernel alone (the row-capped fast kernel)
    if os.environ.get("CASSIE_WAVES_PER_ENV"):
        b.set_waves_per_env(int(os.environ["CASSIE_WAVES_PER_ENV"]))   # A/B switch: one wave per env instead of two
    if os.environ.get("CASSIE_FAST_KERNEL_FORM"):
        b.set_inplace({"plain": 0, "in-place": 1, "auto": 2}[os.environ["CASSIE_FAST_KERNEL_FORM"]])   # A/B switch (the in-place form)
    if os.environ.get("CASSIE_NO_BALANCE"):
        b.set_balance(False)            # A/B switch for the longest-job-first launch order
    if hfield is not None:
        b.set_hfield(hfield)
    # qpos | qvel | sensordata are column blocks of ONE observation tensor the kernel reads and writes in place -- the
    # very buffer the all-gather sends
    nq, nv, nsd, nu = pod.nq, pod.nv, pod.nsensordata, pod.nu
    nobs = nq + nv + nsd
    init_row = torch.zeros(nobs, dtype=torch.float64, device=dev)
    init_row[:nq] = torch.from_numpy(model.qpos_init()).to(dev)
    if drive:   # the drive-level models read the previous step's sensordata: a restarted env carries the init pose's
        init_row[nq + nv:] = torch.from_numpy(rt.init_sensordata(model, hfield)).to(dev)
    obs = ObservationBlock(b, pod, dev, init_row).tensor      # (cassie_amd.distributed: bound with a row stride, no staging copy)
    warm = torch.zeros((n, nv), dtype=torch.float64, device=dev)
    esz = obs.element_size()
    b.bind(P.F_QACC_WARMSTART, warm.data_ptr())
    targets_host = pd_targets(env_ids, npolicy)
    targets = torch.from_numpy(targets_host).to(dev)       # [npolicy][n][10]
    kp = torch.from_numpy(np.tile(PD_KP, (n, 1))).to(dev)
    kd = torch.from_numpy(np.tile(PD_KD, (n, 1))).to(dev)
    b.bind(P.F_PD_KP, kp.data_ptr())
    b.bind(P.F_PD_KD, kd.data_ptr())
    if drive:
        actvel = torch.zeros((n, nu), dtype=torch.float64, device=dev)
        meas = torch.zeros((n, P.MEAS_DIM), dtype=torch.float64, device=dev)
        b.bind(P.F_ACTUATOR_VELOCITY, actvel.data_ptr())
        b.bind(P.F_MEAS, meas.data_ptr())
        b.set_drive_mode(P.DRIVE_PD_SAFE if safe else P.DRIVE_PD)
    else:
        b.set_pd_mode(True)
    rand_info = None
    if randomise is not None:
        # Per-env domain randomisation ON THE DEVICE: every env of the batch gets masses x U(0.8, 1.2) (principal
        # inertias scaled alike), inertial offsets + U(-5, 5) mm, joint damping x U(0.5, 1.5) and sliding friction U(0.4, 1.3) on every
        # collision geom, drawn by torch ON the GPU and handed over as device pointers (phys_batch_randomize), then mj_setConst per env
        # in one launch (phys_batch_set_const).  The shared 95 KB model stays shared; an env reads its own 10 KB block.
        nb, ng = pod.nbody, pod.ngeom
        gen = torch.Generator(device=dev)
        gen.manual_seed(int(randomise) + 7919 * rank)
        u = lambda *shape: torch.rand(*shape, generator=gen, dtype=torch.float64, device=dev) * 2 - 1
        t64 = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), device=dev)
        m0, d0 = t64(pod.body_mass[:nb]), t64(pod.dof_damping[:nv])
        i0, in0 = t64([list(pod.body_ipos[k]) for k in range(nb)]), t64([list(pod.body_inertia[k]) for k in range(nb)])
        f0 = t64([list(pod.geom_friction[g]) for g in range(ng)])
        s_ = 1 + 0.2 * u(n, nb)
        fr = f0.repeat(n, 1, 1)
        fr[:, :, 0] = 0.85 + 0.45 * u(n, ng)
        rows = {P.P_BODY_MASS: (m0 * s_).contiguous(), P.P_BODY_INERTIA: (in0[None] * s_[:, :, None]).reshape(n, -1).contiguous(),
                P.P_BODY_IPOS: (i0[None] + 0.005 * u(n, nb, 3) * (m0 > 0)[None, :, None]).reshape(n, -1).contiguous(),
                P.P_DOF_DAMPING: (d0 * (1 + 0.5 * u(n, nv))).contiguous(), P.P_GEOM_FRICTION: fr.reshape(n, -1).contiguous()}
        rt.synchronize()
        t0 = time.perf_counter()
        for k_, t_ in rows.items():
            b.randomize(k_, None, device_ptr=t_.data_ptr(), n=n)
        b.set_const()
        b.sync()
        rand_info = {"seed": int(randomise), "randomise_and_set_const_ms": 1e3 * (time.perf_counter() - t0), "envs": n,
                     "what": "body_mass x U(0.8, 1.2) with body_inertia scaled alike, body_ipos + U(-5, 5) mm, dof_damping x U(0.5, 1.5), geom_friction[0] ~ U(0.4, 1.3): "
                             "EVERY env, drawn on the GPU (torch), phys_batch_randomize from device pointers + ONE phys_batch_set_const launch (mj_setConst per env on the device)",
                     "bytes_per_env_block": int(ctypes.sizeof(type(b.params(0, 1)[0])))}
    ranges = half_ranges(n, nstreams)
    streams = [rt.Stream() for _ in ranges]

    def restart(group):
        """The envs of a phase group start a new episode: a fresh cassie_sim_t -- init pose, zero velocities and warm start,
        zero filter histories / delay lines, the init pose's sensordata -- in one small launch per range, on its stream
        (phys_batch_reset_envs)."""
        for (first, cnt), st in zip(ranges, streams):
            r0, k = rows_of_group_in_range(group, int(env_ids[0]), first, cnt)
            if k:
                b.reset_envs(r0, NGROUP, k, init_row.data_ptr(), init_row.data_ptr() + (nq + nv) * esz if drive else None, st.cuda_stream)

    # The all-gather runs beside the next launch (cassie_amd.distributed.OverlappedGather): a range's observation block is
    # snapshotted on its launch stream (device to device), RCCL sends the snapshot from a second stream, and the range's next
    # snapshot waits for that gather to have read it.
    og = OverlappedGather(obs, ranges, streams, world, rt) if collect else None
    gather = og.gather if collect else None

    last_launch = {"nsub": 0}

    def step(nsub):
        last_launch["nsub"] = nsub
        for (first, cnt), st in zip(ranges, streams):
            if len(ranges) == 1:
                b.step(nsub, st.cuda_stream)
            else:
                b.step_range(first, cnt, nsub, st.cuda_stream)

    sch = Schedule(step=step,
                   bind_targets=lambda p: b.bind(P.F_PD_PTARGET, targets[p].data_ptr()),
                   restart=restart,
                   gather=gather if collect else None,
                   substeps_per_launch=substeps_per_launch)

    def fence():
        if collect:
            dist.barrier()
        rt.synchronize()

    sample = parity_rows(n, world, parity_envs)
    sample_dev = torch.from_numpy(sample).to(dev)
    rt.synchronize()
    region_s, region_ev, region_launches = [], [], []
    q_sample = info_sample = None
    sch.run(0, PREROLL + warmup)
    b.enable_kernel_timing(True)        # a HIP event pair around the work-doing kernel of every stepping launch from here on
    kern_n, kern_ms_total = 0, 0.0
    for r in range(repeats):
        evs = [(rt.Event(enable_timing=True), rt.Event(enable_timing=True)) for _ in streams]
        region_s.append(timed_region(sch, PREROLL + warmup + r * steps, steps, fence,
                                     mark=lambda i: [e[i].record(st) for e, st in zip(evs, streams)]))
        region_ev.append([e[0].elapsed_time(e[1]) for e in evs])     # per stream: its time over the region
        region_launches.append(sch.launches)
        kn, kms = b.kernel_timing()       # (after the fence: the region's launches are complete)
        kern_n, kern_ms_total = kern_n + kn, kern_ms_total + kms
        if r == snap_r:     # between two fenced regions: the sampled rows of this rank for the CPU replay
            q_sample = obs[sample_dev, :nq].clone()
            info_sample = torch.from_numpy(b.warnings()[1][sample][:, :3].astype(np.int64)).to(dev)
    if collect:   # a region's time is the slowest rank's
        t = torch.tensor(region_s, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        region_s = [float(x) for x in t.cpu()]
    launches = int(sum(region_launches))
    res = {"mode": mode, "n": n, "steps": steps, "warmup": warmup, "repeats": repeats, "launches": launches, "streams": len(ranges),
           # mean stream time of ONE range's launch (includes the rare restart / gather); with several streams the ranges' launches
           # overlap, each sharing the GPU with the others'
           "stream_ms": float(np.mean([np.sum([ev[i] for ev in region_ev]) for i in range(len(streams))])) / launches,
           # the dominant kernel itself: mean duration over all its launches of the timed regions, from a HIP event pair around each
           # on its launch stream (phys_batch_kernel_timing) -- the quantity rocprofv3 --kernel-trace --stats averages
           "kernel_ms": kern_ms_total / max(1, kern_n), "kernel_launches": kern_n,
           "region_s": region_s, "elapsed": float(np.median(region_s))}
    if collect and rank == 0:
        # the gathered block of the last policy boundary must hold this rank's rows (global env order, rank-major)
        res["gather_ok"] = bool(sch.gathers > 0 and og.holds_own_rows(rank))
    w, info = b.warnings()
    # envs the row-capped fast kernel handed over to the full kernel in the last launch: those whose record of
    # completed substeps is short of THAT launch's substep count; NaN (-> null in the line) for a model without a fast kernel
    handed = handed_over_in_last_launch(b.fast_rows_progress(), last_launch["nsub"]) if has_fast_kernel(model.name) else float("nan")
    stats = torch.tensor([float(np.count_nonzero(w))] + [float(info[:, k].sum()) for k in (1, 2, 3)] + [handed], dtype=torch.float64, device=dev)
    if collect:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    stats = stats.cpu().numpy()
    res["envs_with_warnings"] = int(stats[0])
    res["mean_constraint_rows"], res["mean_pgs_iterations"], res["mean_pgs_guarded_sweeps"] = (float(stats[k] / (world * n)) for k in (1, 2, 3))
    res["frac_envs_handed_over_last_launch"] = None if np.isnan(stats[4]) else float(stats[4] / (world * n))
    # what the last launch cost an env, first to last instruction, in shader clocks per substep (phys_batch_download_cost: batches
    # that keep a launch order) -> how full the GPU's workgroup slots were over the timed regions, see main()
    res["fast_kernel_launches_plain_in_place"] = list(b.form_launches()) if hasattr(b, "form_launches") else None   # (the CPU dry run's stand-in has no kernels)
    try:
        res["wide_pass_envs_last_launch"] = sum(b.wide_pass_envs(first) for first, _ in ranges) if hasattr(b, "wide_pass_envs") else None
        res["env_clocks_per_substep"] = float(np.mean(b.launch_cost())) / max(1, last_launch["nsub"])
        res["shader_clock_hz"] = b.measured_shader_clock() if hasattr(b, "measured_shader_clock") else None
    except (RuntimeError, AttributeError):
        res["env_clocks_per_substep"] = None
        res["shader_clock_hz"] = None
        res.setdefault("wide_pass_envs_last_launch", None)
    b_params = b.params() if rand_info is not None and rank == 0 else None
    # ---- the metric's second half: sampled envs of EVERY rank against the CPU reference, same schedule ----
    ids_sample = torch.from_numpy(env_ids[sample].astype(np.int64)).to(dev)
    if collect and world > 1:
        q_sample, info_sample, ids_sample = (gather_rows(x, world) for x in (q_sample, info_sample, ids_sample))
    if rank == 0:
        q_gpu, counts_gpu, ids = q_sample.cpu().numpy(), info_sample.cpu().numpy(), ids_sample.cpu().numpy()
        threads = rt.host_threads()
        tg_replay = pd_targets(ids, (replay_steps + HOLD - 1) // HOLD + 1)      # seeds depend on the global env id only
        pods = None
        if rand_info is not None:
            # the oracle gets a per-env cm_model_t for every replayed env: a HOST model edited through the views the reference's setters
            # write, phys_model_set_const + phys_model_compile -- and the device's parameter block must equal that compile's bit for bit
            import randomise_check as rcheck
            assert world == 1, "the randomised leg replays rank 0's envs"
            hosts = rcheck.HostEnvModels(model.name, flags=int(model.pod.flags))   # (with the shared model's contact options)
            blocks = [b_params[int(e)] for e in sample]
            par = {f: [rcheck.params_as_arrays(blk, pod)[f].reshape(-1) for blk in blocks] for f in rcheck.INPUT_FIELDS}
            pods, same = [], 0
            for i in range(len(blocks)):
                pods.append(hosts.pod(par, i))
                try:
                    rcheck.assert_blocks_equal(blocks[i], pods[-1].params, pod, "env %d" % int(sample[i]))
                    same += 1
                except AssertionError as ex:
                    rand_info.setdefault("first_difference", str(ex))
            rand_info["blocks_equal_to_the_host_compile_bit_for_bit"] = "%d of %d replayed envs" % (same, len(blocks))
            mi = np.array([blk.meaninertia for blk in b_params])
            rand_info["meaninertia_over_the_batch"] = {"min": float(mi.min()), "max": float(mi.max()), "shared_model": float(pod.meaninertia), "distinct": int(len(np.unique(mi)))}
        envs_cls = rt.replay_envs(drive)
        if safe and envs_cls is HostChainEnvs:
            envs_cls = SafeHostChainEnvs     # the REAL cassie_core_sim_step between the PD law and the motor model
        orc = replay_on_oracle(model, ids, lambda p: tg_replay[p], replay_steps, hfield, threads, envs=envs_cls, pods=pods)
        if safe:
            bits = np.array([st.safety_msg for st in b.get_drive_state()])[sample] if hasattr(b, "get_drive_state") else None
            res["safety_messages"] = None if bits is None else {"envs_with_code_635_joint_limit": int(np.count_nonzero(bits & 1)), "envs_with_code_630_torque_limit": int(np.count_nonzero(bits & 2)),
                                                               "of_replayed_envs": int(len(bits))}
        q_ref = orc.qpos()
        err_abs = np.abs(q_gpu - q_ref)
        err_rel = err_abs / np.maximum(1.0, np.abs(q_ref))
        per_rank = {int(r): float(err_abs[ids // n == r].max()) for r in np.unique(ids // n)}
        res["parity"] = {"reference": "oracle/cassie_oracle.c (fp64 CPU restatement of mj_step1 + mj_step2; parity with genuine MuJoCo unpinned)"
                                      + (" + the host chain csrc/cassie_hostpath.c (encoder / motor arithmetic pinned bit-exactly to the reference's code)" if drive else ""),
                         "envs_compared": int(len(ids)), "ranks_compared": int(len(per_rank)), "max_qpos_err_per_rank": per_rank,
                         "steps_replayed": int(replay_steps), "after_timed_region": int(snap_r),
                         "max_qpos_err": float(err_abs.max()), "max_qpos_rel_err": float(err_rel.max()),
                         "frac_envs_with_equal_ncon_nefc_iters": float(np.mean(np.all(counts_gpu == orc.counts(), axis=1))),
                         "tolerance_rel": 1e-6, "ok": bool(err_rel.max() <= 1e-6)}
        if drive:
            res["parity"]["note"] = ("an encoder count that truncates differently on a last-bit physics difference moves a motor torque by "
                                     "kp * 2 pi / 2^bits / gear for one step: agreement is to rounding only as long as no count flips")
    if rand_info is not None:
        res["randomised"] = rand_info
    b.close()
    return res


SLOTS_PER_GPU = 256 * 4        # one env per workgroup, 40 KB of LDS each: four workgroups per CU
SHADER_CLOCK_HZ = 2.4e9        # MI355X peak engine clock: the fallback when the launch did not measure its own (slot_occupancy)


def slot_occupancy(env_clocks_per_substep, n, substeps, elapsed_s, clock_hz=None):
    """Where a timed region's time goes, one level above the kernel: every env-step occupies one of the GPU's 1024 workgroup
    slots for `env_clocks_per_substep`; busy_frac = slot time used / slot time available over the region.  The rest is slots
    waiting for work: the drain at the end of each range's launch (no env of a launch may start the next one before all have
    finished this one) and the gaps between a range's launches."""
    if not env_clocks_per_substep or not elapsed_s:
        return None
    # the shader clock under THIS load, measured by the last launch itself (every env's span in s_memtime clocks over the same span
    # on the constant 100 MHz clock: phys_batch_measured_shader_clock); the nominal peak only where the launch could not measure it
    hz = clock_hz or SHADER_CLOCK_HZ
    busy = n * substeps * env_clocks_per_substep / hz / (SLOTS_PER_GPU * elapsed_s)
    return {"env_clocks_per_substep": env_clocks_per_substep, "slots": SLOTS_PER_GPU, "clock_hz": hz,
            "clock_source": "measured by the last launch (s_memtime over s_memrealtime, all envs)" if clock_hz else "assumed (nominal peak)",
            "busy_frac": busy,
            "rate_with_every_slot_busy": SLOTS_PER_GPU * hz / env_clocks_per_substep,
            "note": "per GPU, from the LAST launch's per-env clocks (first to last instruction of the fast kernel + the pass behind it)"}


def true_reference(model_name, q0, targets, nsteps):
    """Opportunistic run of the genuine reference physics: tests/mujoco_ref.py looks
    for a MuJoCo the reference could have dlopen'd (reference src/cassiemujoco.c:521-555) or a `mujoco` wheel, plus the
    MJCF files (staged under oracle/_ref/model by oracle/build_ref.sh).  Returns a dict, or the string "unavailable"."""
    try:
        import mujoco_ref
        ref = mujoco_ref.find()
        if ref is None:
            print("true reference unavailable -- parity vs restatement only", file=sys.stderr)
            return "unavailable"
        return mujoco_ref.bench_and_compare(ref, model_name, q0, targets, PD_KP, PD_KD, nsteps, HOLD)
    except Exception as exc:  # the harness is opportunistic: never let it take the bench line down
        print("true reference harness failed: %r" % (exc,), file=sys.stderr)
        return "unavailable"


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs-per-gpu", type=int, default=None,
                    help="default: %d with one GPU (BASELINE configs[1]), %d with more (configs[2]: 65536 envs at 8 GPUs)"
                         % (ENVS_PER_GPU_SINGLE, ENVS_PER_GPU_SHARDED))
    ap.add_argument("--total-envs", type=int, default=None,
                    help="shard a FIXED total over the GPUs instead (strong scaling), e.g. 65536 at 1 / 2 / 4 / 8 GPUs")
    ap.add_argument("--repeats", type=int, default=REPEATS,
                    help="fenced timed regions of exactly --steps steps each; `value` is the median region, value_min / value_max the extremes")
    ap.add_argument("--streams", type=int, default=STREAMS,
                    help="a rank's batch is stepped as this many contiguous env ranges, each on its own stream at its own pace "
                         "(1 = the whole batch in one launch per policy step)")
    ap.add_argument("--substeps-per-launch", type=int, default=HOLD,
                    help="physics steps fused into one kernel launch (at most up to the next PD-target re-draw)")
    ap.add_argument("--model", default="cassie", choices=["cassie", "cassie_hfield", "cassie_tray_box"],
                    help="cassie = BASELINE configs[1] (the headline); the other two are configs[3] / configs[4], for the record")
    ap.add_argument("--hfield-contacts", default="default", choices=["default", "prism"],
                    help="cassie_hfield only: `prism` = CM_FLAG_HFPRISM, one contact per penetrated grid triangle (the MuJoCo-shaped contact set; "
                         "up to 32 contacts / 127 rows, envs pass through the 31 / 63 / 127-row instantiations); default = at most two per capsule")
    ap.add_argument("--box-contacts", type=int, default=4, choices=[4, 8],
                    help="cassie_tray_box: contacts a box-box pair keeps -- 4 (default) or 8 = CM_FLAG_BOX8, MuJoCo's count")
    ap.add_argument("--target-spread", type=float, default=TARGET_SPREAD,
                    help="half-width in rad of the uniform PD targets around the standing pose (0.3: the metric's workload; 10: the "
                         "stress targets of reference example/cassietest_jac.py:106, joints driven into their limits)")
    ap.add_argument("--parity-envs", type=int, default=64, help="envs of the timed batch replayed on the CPU reference (spread over all ranks)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-step-pd", action="store_true")
    ap.add_argument("--no-other-mode", action="store_true", help="skip the short run of the other device mode")
    ap.add_argument("--no-randomised", action="store_true", help="skip the short run with every env's physical parameters randomised on the device")
    ap.add_argument("--randomise", type=int, default=None, metavar="SEED", help="randomise every env's masses / inertial offsets / damping / friction on the device for the MAIN timed regions too (SURVEY.md 8f-3)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="validation aid: initialise the process group and run the observation all-gather / barriers even with one rank")
    ap.add_argument("--mode", default="drive-pd-safe", choices=["drive-pd-safe", "drive-pd", "exact-pd"],
                    help="what the device-resident kernel computes per substep (see device_rollout)")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="TEST INFRASTRUCTURE (tests/test_multirank.py): run main()'s launch / sharding / gather / reduction path on CPU "
                         "ranks (gloo) with a stand-in for the GPU batch (tests/bench_standin.py); no physics, nothing is measured")
    args = ap.parse_args(argv)
    globals()["TARGET_SPREAD"] = args.target_spread

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started the way the driver starts N = 1: no launcher around us -- start the ranks ourselves
        return launch_ranks(args.gpus, argv)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("--gpus %d, but the launcher started %d rank(s) (WORLD_SIZE)" % (args.gpus, world))
    if args.dry_run_cpu:
        import bench_standin
        rt = bench_standin.CpuRuntime(local_rank)
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the physics library has no CPU fallback")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit("rank %d wants GPU %d, but this node shows %d" % (rank, local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        rt = GpuRuntime(local_rank)
    collect = world > 1 or args.force_collectives
    if collect:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(free_port())   # (a single rank: nobody else has to find it)
        if args.dry_run_cpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from cassie_amd import Model

    model = Model(args.model)
    if args.hfield_contacts == "prism":
        if args.model != "cassie_hfield":
            raise SystemExit("--hfield-contacts prism applies to --model cassie_hfield")
        from cassie_amd import phys as _P
        model.set_flag(_P.FLAG_HFPRISM, True)
    if args.box_contacts == 8:
        from cassie_amd import phys as _P
        model.set_flag(_P.FLAG_BOX8, True)      # box-box keeps up to eight points like MuJoCo's mjc_BoxBox
    pod = model.pod
    n, scaling, shape = resolve_envs(world, args.envs_per_gpu, args.total_envs)
    repeats = max(1, args.repeats)
    hfield = None
    if args.model == "cassie_hfield":   # terrain of reference example/test_hfield.py:39-41, shared by all envs
        hfield = np.random.default_rng(99).random((200, 200)).astype(np.float32)
        hfield[95:105, 95:105] = 0
    nstreams = max(1, args.streams)
    r = device_rollout(model, args.mode, n, args.steps, args.warmup, rank, world, local_rank, args.substeps_per_launch, args.parity_envs, hfield,
                       collect=collect, repeats=repeats, nstreams=nstreams, rt=rt, randomise=args.randomise if world == 1 else None)

    if rank == 0:
        elapsed, kern_ms, timed_launches = r["elapsed"], r["kernel_ms"], r["launches"]
        steps_per_launch = repeats * args.steps / timed_launches
        # read qpos+qvel+qacc_warmstart+ctrl, write qpos+qvel+qacc+sensordata+actuator_velocity (1976 B for cassie)
        algo_bytes = 8 * ((pod.nq + 2 * pod.nv + pod.nu) + (pod.nq + 2 * pod.nv + pod.nsensordata + pod.nu))
        assert args.model != "cassie" or algo_bytes == ALGO_BYTES_PER_ENV_STEP
        # whole GPU: algorithmic bytes of every env-step of a timed region / the region's time (= value x bytes per env-step);
        # one launch of the dominant kernel: its env-steps x bytes / its own mean duration
        launch_env_steps = n * steps_per_launch / r["streams"]
        achieved_one = algo_bytes * launch_env_steps / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else None   # (None: the CPU stand-in times no kernel)
        achieved = algo_bytes * world * n * args.steps / elapsed / 1e9 / world
        rate = lambda sec: world * n * args.steps / sec
        value = rate(elapsed)
        # (per launch of the dominant kernel, like `achieved_one_launch`: one env range's launch)
        chunks = launch_chunks(n // r["streams"], steps_per_launch, r["streams"] == 1)
        traffic, traffic_src = pmc_traffic(n * steps_per_launch / r["streams"], args.model, envs_per_launch=n / r["streams"], pod=pod, chunks=chunks)
        api = {"drive-pd-safe": "phys_batch_step in CM_DRIVE_PD_SAFE mode (device-resident, include/cassie_phys.h): pd_input's motor PD on the encoder measurements + "
                                "cassie_core_sim's safety layer (joint-limit attenuation / restoring torques, torque-limit clamp, STO: csrc/pk_safety.h, bit for bit the "
                                "closed Agility block) + motor model with torque delay + physics in one kernel -- the whole torque path of cassie_sim_step_pd "
                                "(reference src/cassiemujoco.c:1147-1157); only the state estimator (which feeds nothing back) stays on the host",
               "drive-pd": "phys_batch_step in CM_DRIVE_PD mode (device-resident, include/cassie_phys.h): pd_input's motor PD on the encoder "
                           "measurements + motor model with torque delay + physics in one kernel -- cassie_sim_step_pd's drive-level semantics "
                           "without the Agility safety layer / estimator",
               "exact-pd": "phys_batch_step with phys_batch_set_pd_mode (device-resident): PD law on the exact joint state + motor speed-torque "
                           "limit + physics in one kernel (no encoder quantisation, no torque delay)"}[args.mode]
        out = {
            "metric": "env-steps/sec (whole node) at N envs; max |qpos_err| vs CPU ref", "value": value, "unit": "env-steps/s",
            "value_min": rate(max(r["region_s"])), "value_max": rate(min(r["region_s"])),
            "max_qpos_err": r["parity"]["max_qpos_err"], "max_qpos_rel_err": r["parity"]["max_qpos_rel_err"],
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f64",
            "data": "synthetic" if not args.dry_run_cpu else rt.name,
            "config": {"workload": "READ FIRST -- `value` is the device-resident API (phys_batch_step_range, mode `%s`: see api_of_value), the batch stepped as %d "
                                   "env ranges on %d streams that nothing joins between policy steps; beside it in this line: `value_step_pd` = "
                                   "cassie_sim_step_pd itself, batched (the API BASELINE configs[1] names; Agility blocks on host threads, PCIe every "
                                   "step), `value_one_stream` = the whole batch as ONE launch per policy step, `value_all_outputs_every_substep` = "
                                   "every output of every P-row formed by every substep.  Workload: %d envs/GPU x %d GPU = %d envs (%s), %s.xml, "
                                   "%d-step episodes from the cassie_sim_init pose restarted at staggered phases (untimed pre-roll of %d steps), "
                                   "random joint-PD targets re-drawn every %d steps; `value` is the MEDIAN of %d fenced timed regions of %d steps "
                                   "each (value_min / value_max: the slowest / fastest region)"
                                   % (args.mode, r["streams"], r["streams"], n, world, world * n, shape, args.model, EPISODE, PREROLL, HOLD, repeats, args.steps),
                       "api_of_value": api, "mode": args.mode,
                       "hfield_contacts": args.hfield_contacts if args.model == "cassie_hfield" else None,
                       "box_contacts": args.box_contacts if args.model == "cassie_tray_box" else None,
                       "pd_target_spread_rad": args.target_spread,
                       "envs_per_gpu": n, "envs_total": world * n, "baseline_config": shape, "parallelism": "env-sharded x%d" % world,
                       "obs_allgather_every_steps": HOLD if collect else None,
                       "streams": r["streams"],
                       # a stepping launch is dispatched as this many workgroups per env, each stepping a share of the substeps
                       # (phys_batch_set_chunks; launches of fewer than 10 substeps or 2048 envs stay in one piece)
                       "chunks_per_env_launch": chunks,
                       "wavefronts_per_env": (1 if os.environ.get("CASSIE_WAVES_PER_ENV") == "1" or (args.model == "cassie_tray_box" and os.environ.get("CASSIE_TRAY_TWO_WAVES") == "0") else 2),
                       "streams_note": ("the %d envs of a GPU are stepped as %d contiguous ranges, each on its own stream at its own pace "
                                        "(phys_batch_step_range): per policy step every range gets its restarts, its PD targets and one "
                                        "launch, and nothing on the device joins the ranges between policy steps, so one range's workgroups "
                                        "fill the wave slots the other leaves idle at the end and the start of its launches; "
                                        "`value_one_stream` is the same batch as one launch per policy step" % (n, r["streams"])) if r["streams"] > 1 else None,
                       "timed_regions": repeats, "region_ms": [1e3 * x for x in r["region_s"]],
                       "substeps_per_launch": steps_per_launch, "launches_timed": timed_launches,
                       "outputs_of_a_launch": "its last substep's (sensordata, measurement block, xpos / xquat, solver statistics); IMU words and "
                                              "body quaternions of the substeps in between, which nobody can read, are not formed -- "
                                              "`value_all_outputs_every_substep` is the rate with every output of every P-row formed by every substep",
                       "preroll_steps": PREROLL, "episode_steps": EPISODE},
            "parity": r["parity"],
            **({"safety_messages": r["safety_messages"]} if r.get("safety_messages") else {}),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         # `achieved`, `peak`, `frac`, `traffic` and `kernel_ms` are PER GPU at every N (a rank's env-steps over the
                         # region against ONE GPU's HBM peak); the node moves n_gpus x `achieved` against n_gpus x `peak`
                         "scope": "per GPU", "achieved_node": achieved * world, "peak_node": HBM_PEAK_GBS * world,
                         "concurrent_launches": r["streams"],
                         "achieved_note": ("per GPU: algorithmic bytes of all env-steps of a timed region / the region's time.  The %d env ranges' launches "
                                           "overlap; ONE launch of the dominant kernel (%d env-steps, `kernel_ms` = its mean duration from a HIP "
                                           "event pair around every launch on its stream -- what rocprofv3 averages) moves %.2f GB/s, and a range "
                                           "needs `stream_ms_per_policy_step` per policy step (that kernel + the list-walking pass behind it + "
                                           "order / restart kernels)" % (r["streams"], launch_env_steps, achieved_one or 0.0)) if r["streams"] > 1 else
                                          "algorithmic bytes of one launch / the dominant kernel's mean duration (a HIP event pair around every launch on the launch stream)",
                         "achieved_one_launch": achieved_one, "stream_ms_per_policy_step": r["stream_ms"], "kernel_launches_timed": r["kernel_launches"],
                         "kernel": {"cassie": "ck::cassie_step_kernel<32, ck::TopoCassie32, 0, 31, 2, false, 2> (row-capped fast instantiation, TWO wavefronts per env; "
                                              "<..., 63, 2, true, 2> walks the list of handed-over envs behind it and <..., 127, 2, true, 1> the list of those it passes on)",
                                    "cassie_hfield": "ck::cassie_step_kernel<32, ck::TopoCassie32, 1, 31, 2, false, 2> (row-capped fast instantiation, two wavefronts per env; "
                                                     "<..., 63, 2, true, 2> walks the list of handed-over envs behind it and <..., 127, 2, true, 1> the list of those it passes on)",
                                    "cassie_tray_box": "ck::cassie_step_kernel<40, ck::TopoCassieTray38, 2, 47, 2, false, 2> (row-capped instantiation of 47 rows, TWO wavefronts "
                                                       "per env, Gram matrix on the matrix core; <..., 63, 2, true, 2> walks the list of handed-over envs behind it; "
                                                       "CASSIE_TRAY_TWO_WAVES=0: the one-wave form of round 4, profiles/round5/tray_two_waves_ab.txt)"}[args.model], "kernel_ms": kern_ms, "algorithmic_bytes_per_env_step": algo_bytes, "env_steps_per_launch": launch_env_steps,
                         "note": "latency-bound by design: ~2 KB of state vs ~0.22 MFLOP of serially dependent fp64 per env-step"},
            # the more telling bound: ~0.22 MFLOP of algorithmic fp64 work per env-step against the fp64 vector peak
            "roofline_fp64": {"bound": "fp64-valu", "achieved": value * 0.22e6 / 1e12, "peak": 78.6 * world, "unit": "TFLOP/s",
                              "frac": value * 0.22e6 / 1e12 / (78.6 * world),
                              "note": "algorithmic flops (SURVEY.md 8a estimate), not counting lanes that idle or recompute"},
            "workgroup_slots": slot_occupancy(r["env_clocks_per_substep"], n, r["steps"], r["elapsed"], r.get("shader_clock_hz")),
            "envs_with_warnings": r["envs_with_warnings"],
            **({"randomised": r["randomised"]} if r.get("randomised") else {}),   # (--randomise: the main regions ran with per-env parameters)
            **({"obs_allgather_ok": r.get("gather_ok")} if collect else {}),   # rank 0's rows of the last gathered block = its snapshot
            "frac_envs_handed_over_to_the_full_kernel_in_the_last_launch": r["frac_envs_handed_over_last_launch"],
            # ... and of those, the envs the 63-row pass passed on to the 127-row instantiation (rank 0's batch)
            # stepping launches of rank 0's batch by the form of the fast kernel (phys_batch_set_inplace: picked per env range)
            "fast_kernel_launches_plain_in_place": r.get("fast_kernel_launches_plain_in_place"),
            "frac_envs_in_the_127_row_pass_in_the_last_launch": (r.get("wide_pass_envs_last_launch") / float(n)) if r.get("wide_pass_envs_last_launch") is not None else None,
            "mean_constraint_rows": r["mean_constraint_rows"], "mean_pgs_iterations": r["mean_pgs_iterations"], "mean_pgs_guarded_sweeps": r["mean_pgs_guarded_sweeps"],
        }
        if world == 1 and args.model == "cassie" and args.total_envs is None and not args.dry_run_cpu:
            # the GPU legs first, back to back with the timed region; the CPU legs (tens of seconds with an idle GPU) last
            if not args.no_other_mode:
                other = "exact-pd" if args.mode != "exact-pd" else "drive-pd-safe"
                side = dict(steps=min(args.steps, 400), warmup=min(args.warmup, 50), repeats=min(repeats, 5))
                o = device_rollout(model, other, n, side["steps"], side["warmup"], 0, 1, local_rank, args.substeps_per_launch, 16, hfield, repeats=side["repeats"],
                                   nstreams=nstreams)
                out[other.replace("-", "_")] = {"value": n * o["steps"] / o["elapsed"], "unit": "env-steps/s", "steps": o["steps"], "warmup": o["warmup"],
                                                "timed_regions": o["repeats"], "kernel_ms": o["kernel_ms"], "parity": o["parity"],
                                                "mean_constraint_rows": o["mean_constraint_rows"], "mean_pgs_iterations": o["mean_pgs_iterations"]}
                out["value_" + other.replace("-", "_")] = out[other.replace("-", "_")]["value"]
                if args.mode == "drive-pd-safe":   # what the safety layer costs: the same mode without it (round 5's headline mode)
                    ns = device_rollout(model, "drive-pd", n, side["steps"], side["warmup"], 0, 1, local_rank, args.substeps_per_launch, 4, hfield, repeats=side["repeats"], nstreams=nstreams)
                    out["drive_pd_without_safety_layer"] = {"value": n * ns["steps"] / ns["elapsed"], "unit": "env-steps/s", "steps": ns["steps"], "timed_regions": ns["repeats"],
                                                            "kernel_ms": ns["kernel_ms"], "max_qpos_err": ns["parity"]["max_qpos_err"]}
                    out["value_drive_pd_without_safety_layer"] = out["drive_pd_without_safety_layer"]["value"]
                # `value` with every output evaluated by every substep: a fused launch returns its last substep's outputs, so by
                # default the IMU sensor words and body quaternions of the substeps in between -- values nobody can read -- are
                # not formed; this is what forming them anyway costs
                a = device_rollout(model, args.mode, n, side["steps"], side["warmup"], 0, 1, local_rank, args.substeps_per_launch, 4, hfield,
                                   all_outputs=True, repeats=side["repeats"], nstreams=nstreams)
                out["all_outputs_every_substep"] = {"value": n * a["steps"] / a["elapsed"], "unit": "env-steps/s", "steps": a["steps"],
                                                    "timed_regions": a["repeats"], "kernel_ms": a["kernel_ms"], "max_qpos_err": a["parity"]["max_qpos_err"]}
                out["value_all_outputs_every_substep"] = out["all_outputs_every_substep"]["value"]
                if not args.no_randomised:
                    # every env with its own masses / inertial offsets / damping / friction: what domain randomisation costs
                    # the step kernel (an env's parameters are a 10 KB block of its own instead of L2-hot shared scalars), replayed on the
                    # oracle with per-env models
                    dr = device_rollout(model, args.mode, n, side["steps"], side["warmup"], 0, 1, local_rank, args.substeps_per_launch, 16, hfield,
                                        repeats=side["repeats"], nstreams=nstreams, randomise=4242)
                    out["domain_randomised"] = {"value": n * dr["steps"] / dr["elapsed"], "unit": "env-steps/s", "steps": dr["steps"], "warmup": dr["warmup"],
                                                "timed_regions": dr["repeats"], "kernel_ms": dr["kernel_ms"], "parity": dr["parity"], **dr["randomised"],
                                                "mean_constraint_rows": dr["mean_constraint_rows"], "mean_pgs_iterations": dr["mean_pgs_iterations"]}
                    out["value_domain_randomised"] = out["domain_randomised"]["value"]
                if nstreams > 1:    # the whole batch as one launch per policy step (rounds 1-2, and what a consumer that needs every env's
                    # observation before it acts on any of them gets)
                    one = device_rollout(model, args.mode, n, side["steps"], side["warmup"], 0, 1, local_rank, args.substeps_per_launch, 4, hfield,
                                         repeats=side["repeats"], nstreams=1)
                    out["one_stream"] = {"value": n * one["steps"] / one["elapsed"], "unit": "env-steps/s", "steps": one["steps"], "timed_regions": one["repeats"],
                                         "kernel_ms": one["kernel_ms"], "max_qpos_err": one["parity"]["max_qpos_err"]}
                    out["value_one_stream"] = out["one_stream"]["value"]
            if not args.no_step_pd:
                sp = step_pd_host_api(n)
                sd = step_pd_host_api(n, device_drives=True)
                out["step_pd_host_api"] = sp
                out["step_pd_device_drives"] = sd
                out["value_step_pd"] = max([x["value"] for x in (sp, sd) if x] or [None])
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(model)
            tsample = np.arange(8)
            out["true_reference"] = true_reference(args.model, model.qpos_init(), pd_targets(tsample, EPISODE // HOLD + 1), EPISODE)
        elif world == 1 and args.total_envs is None and not args.dry_run_cpu and not args.no_cpu_baseline:
            if args.model == "cassie_hfield" and args.hfield_contacts == "default" and not args.no_other_mode:
                # config 4 carries BOTH height-field contact definitions: `value` is the default one (one contact per geom pair, the
                # deepest sample), this leg the MuJoCo-shaped set (CM_FLAG_HFPRISM: one contact per penetrated grid triangle, up to 32
                # contacts / 127 rows, three kernel tiers) on the same terrain and schedule, replayed on the oracle with the same flag
                from cassie_amd import phys as _P
                pm = Model(args.model)
                pm.set_flag(_P.FLAG_HFPRISM, True)
                side_steps = min(args.steps, 500)
                pr = device_rollout(pm, args.mode, n, side_steps, min(args.warmup, 100), 0, 1, local_rank, args.substeps_per_launch, 16, hfield,
                                    repeats=min(repeats, 4), nstreams=nstreams)
                out["hfield_contacts_prism"] = {"value": n * pr["steps"] / pr["elapsed"], "unit": "env-steps/s", "steps": pr["steps"], "warmup": pr["warmup"],
                                                "timed_regions": pr["repeats"], "kernel_ms": pr["kernel_ms"], "stream_ms_per_policy_step": pr["stream_ms"],
                                                "parity": pr["parity"], "envs_with_warnings": pr["envs_with_warnings"],
                                                "frac_envs_leaving_the_31_row_tier_in_the_last_launch": pr["frac_envs_handed_over_last_launch"],
                                                "mean_constraint_rows": pr["mean_constraint_rows"], "mean_pgs_iterations": pr["mean_pgs_iterations"],
                                                "what": "CM_FLAG_HFPRISM: MuJoCo's per-prism height-field contact set; off by default (DESIGN.md 4.2, 7.3)"}
                out["value_hfield_contacts_prism"] = out["hfield_contacts_prism"]["value"]
            out["cpu_baseline"] = cpu_baseline(model, hfield=hfield)      # configs 4 / 5: the oracle on this box's cores, same workload
        elif world > 1 and not args.no_cpu_baseline:
            # N > 1 (a SCALE line): rank 0 alone times the CPU oracle on the node's host cores, after the timed regions, while the
            # other ranks wait at the barrier below -- the line must not read as unmeasured for want of the key
            out["cpu_baseline"] = cpu_baseline(model, budget_s=1.0 if args.dry_run_cpu else 12.0, hfield=hfield)
            out["cpu_baseline"]["note"] = "rank 0 only, all host cores of the node, timed behind the GPU regions with the other ranks idle"
        print(json.dumps(out), flush=True)
    if collect and world > 1:
        dist.barrier()      # (the ranks leave together: rank 0 may still be in its CPU leg)
    if collect:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
