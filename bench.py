#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched Cassie physics step on MI355X, and its error against the CPU reference.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1 is launched by
torch.distributed.run, one rank per GPU).  A *step* is one pass of the hot path over
one batch: every env of the batch advances by one 0.5 ms physics step
(= one cassie_sim_step_pd-equivalent; reference src/cassiemujoco.c:1130-1134).

Workload (BASELINE.json configs[1]): 4096 envs per GPU, cassie model, EPISODES of 1000 steps from the
cassie_sim_init pose (reference src/cassiemujoco.c:1023-1029); per-env random PD targets (offset + U(-0.3, 0.3)
rad, gains of reference example/cassietest_jac.py:51-52, :68) re-drawn every 50 steps from a table that is resident
in HBM; the PD law + motor speed-torque limit run on the device inside the step kernel, so no host buffer is touched
in the timed region.  Episodes are STAGGERED: env e restarts from the init pose every 1000 steps at phase
50 * (e mod 20), so at any moment the batch holds every phase of the episode in equal parts and the measured rate is
the episode average whatever --steps / --warmup are (synchronised episodes would make a short run measure whichever
phase it happened to land in -- e.g. the contact-free first 3.7 mm of the drop).  Before the warm-up the schedule is
advanced 1000 untimed steps so that the mix is the stationary one.
With N>1 envs shard across GPUs with no data-path exchange; the only collective is one RCCL all-gather of the
observation block (qpos|qvel|sensordata, 96 doubles/env, ONE tensor the kernel writes in place) per 50 steps.

The JSON line carries, beside the contract's fields:
  max_qpos_err    BASELINE.json's second half of the metric: after the timed region, sampled envs of the timed batch
                  are replayed on the CPU reference (oracle/, the fp64 restatement of mj_step1 + mj_step2) through the
                  same schedule (pre-roll, warm-up, timed steps, restarts) and the final qpos compared
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


def pd_targets(env_ids, npolicy):
    """[npolicy][len(env_ids)][10] targets; env e uses numpy.random.default_rng(1234 + e)."""
    out = np.empty((npolicy, len(env_ids), 10))
    for i, e in enumerate(env_ids):
        out[:, i, :] = PD_OFFSET + np.random.default_rng(1234 + int(e)).uniform(-0.3, 0.3, (npolicy, 10))
    return out


def shard_env_ids(rank, world, envs_per_rank):
    """Contiguous block of global env ids owned by `rank` (weak scaling: the per-rank count is fixed)."""
    return np.arange(rank * envs_per_rank, (rank + 1) * envs_per_rank)


def restart_group(policy_step):
    """The phase group whose envs restart from the init pose at this policy step."""
    return policy_step % NGROUP


def gather_observations(obs, world, out=None):
    """All-gather of the per-rank observation block [n, 96] into [world * n, 96], rank-major = global env order.
    RCCL over xGMI on GPUs ('nccl' backend), gloo in the CPU tests."""
    import torch
    import torch.distributed as dist
    if out is None:
        out = torch.empty((world * obs.shape[0], obs.shape[1]), dtype=obs.dtype, device=obs.device)
    dist.all_gather_into_tensor(out, obs)
    return out


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


def max_over_ranks(seconds, world, device=None):
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def pmc_traffic(env_steps_per_launch):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (tools/gpu_pmc.sh: FETCH_SIZE and WRITE_SIZE in
    separate --pmc runs, corrected as MI355X_MICROARCH.md prescribes), scaled to this run's env-steps per launch.
    The counters cannot be collected from inside the timed process, so this is the last measured figure, or None."""
    import glob
    import re

    def version(path):   # profiles/roundR/vN_pmc_summary.json -> (R, N)
        m = re.search(r"round(\d+).*?v(\d+)_pmc_summary", path)
        return (int(m.group(1)), int(m.group(2))) if m else (0, 0)
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "round*", "*pmc_summary.json")), key=version)
    for path in reversed(files):
        try:
            d = json.load(open(path))["derived"]
            per_env_step = (d["hbm_read_bytes_per_launch"] + d["hbm_write_bytes_per_launch"]) / d["env_steps_per_launch"]
            return per_env_step * env_steps_per_launch, os.path.relpath(path, REPO)
        except (KeyError, ValueError, OSError):
            continue
    return None, None


class OracleEnvs:
    """A set of envs on the CPU reference (oracle/cassie_oracle.c), driven through the benchmark's schedule."""

    def __init__(self, model, env_ids, hfield=None):
        import oracle_py
        self.op, self.model, self.ids = oracle_py, model, np.asarray(env_ids)
        self.L = oracle_py.lib()
        if hfield is not None:
            oracle_py.set_hfield(hfield)
        self.buf = (oracle_py.CoData * len(self.ids))()
        self.q0 = model.qpos_init()
        for i in range(len(self.ids)):
            self.reset(i)
        self.kp = np.tile(PD_KP, (len(self.ids), 1))
        self.kd = np.tile(PD_KD, (len(self.ids), 1))

    def reset(self, i):
        self.L.co_reset(ctypes.byref(self.model.pod), ctypes.byref(self.buf[i]))
        self.op.arr(self.buf[i].qpos)[: self.model.pod.nq] = self.q0

    def restart(self, group):
        for i, e in enumerate(self.ids):
            if int(e) % NGROUP == group:
                self.reset(i)

    def step(self, nsub, targets, threads):
        pt = np.ascontiguousarray(targets)
        self.L.co_step_batch(ctypes.byref(self.model.pod), ctypes.byref(self.buf), len(self.ids), nsub, pt.ctypes.data,
                             self.kp.ctypes.data, self.kd.ctypes.data, threads)

    def qpos(self):
        return np.array([self.op.arr(b.qpos)[: self.model.pod.nq].copy() for b in self.buf])

    def counts(self):
        return np.array([[b.ncon, b.nefc, b.solver_iter] for b in self.buf])


def replay_on_oracle(model, env_ids, targets_of, total_steps, hfield=None, threads=1):
    """The schedule of the timed batch (restarts, PD targets, step count) for the envs `env_ids` on the CPU reference.
    targets_of(p) -> [len(env_ids)][10] targets of policy step p."""
    o = OracleEnvs(model, env_ids, hfield)
    cur = {}
    sch = Schedule(step=lambda nsub: o.step(nsub, cur["t"], threads),
                   bind_targets=lambda p: cur.__setitem__("t", targets_of(p)),
                   restart=o.restart)
    sch.run(0, total_steps)
    return o


def cpu_baseline(model, budget_s=12.0):
    """Times the CPU oracle on a bounded sample of the same workload (staggered 1000-step episodes under the PD
    workload), all usable host cores (OpenMP over envs)."""
    from cassie_amd._lib import lib
    cores = lib().cassie_host_cpu_count()      # affinity- and cgroup-quota-aware
    nenv = 16 * cores
    ids = np.arange(nenv)
    o = OracleEnvs(model, ids)
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


def step_pd_host_api(n, steps=200, warmup=50):
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
    b = L.cassie_batch_create(os.path.join(MODEL_DIR, "cassie.cmodel").encode(), n, 0, 0)
    if not b:
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
            "ms_per_step": 1e3 * dt / steps,
            "what": "cassie_batch_step_pd: pd_input + cassie_core_sim + motor/encoder models + state_output on host threads, "
                    "PCIe ctrl/sensordata copies and the physics kernel every step"}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--substeps-per-launch", type=int, default=HOLD,
                    help="physics steps fused into one kernel launch (at most up to the next PD-target re-draw)")
    ap.add_argument("--model", default="cassie", choices=["cassie", "cassie_hfield", "cassie_tray_box"],
                    help="cassie = BASELINE configs[1] (the headline); the other two are configs[3] / configs[4], for the record")
    ap.add_argument("--parity-envs", type=int, default=64, help="envs of the timed batch replayed on the CPU reference")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-step-pd", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the physics library has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from cassie_amd import Batch, Model
    from cassie_amd import phys as P

    model = Model(args.model)
    pod = model.pod
    n = args.envs_per_gpu
    env_ids = shard_env_ids(rank, world, n)
    total_steps = PREROLL + args.warmup + args.steps
    npolicy = (total_steps + HOLD - 1) // HOLD + 1

    b = Batch(model, n, device=local_rank)
    hfield = None
    if args.model == "cassie_hfield":   # terrain of reference example/test_hfield.py:39-41, shared by all envs
        hfield = np.random.default_rng(99).random((200, 200)).astype(np.float32)
        hfield[95:105, 95:105] = 0
        b.set_hfield(hfield)
    dev = torch.device("cuda", local_rank)
    # state and inputs live in HBM before the timed region starts.  qpos | qvel | sensordata are column blocks of ONE
    # observation tensor the kernel reads and writes in place -- the very buffer the all-gather sends
    nq, nv, nsd = pod.nq, pod.nv, pod.nsensordata
    nobs = nq + nv + nsd
    obs = torch.zeros((n, nobs), dtype=torch.float64, device=dev)
    init_row = torch.zeros(nobs, dtype=torch.float64, device=dev)
    init_row[:nq] = torch.from_numpy(model.qpos_init()).to(dev)
    obs[:] = init_row
    warm = torch.zeros((n, nv), dtype=torch.float64, device=dev)
    esz = obs.element_size()
    b.bind(P.F_QPOS, obs.data_ptr(), row_stride=nobs)
    b.bind(P.F_QVEL, obs.data_ptr() + nq * esz, row_stride=nobs)
    b.bind(P.F_SENSORDATA, obs.data_ptr() + (nq + nv) * esz, row_stride=nobs)
    b.bind(P.F_QACC_WARMSTART, warm.data_ptr())
    targets_host = pd_targets(env_ids, npolicy)
    targets = torch.from_numpy(targets_host).to(dev)       # [npolicy][n][10]
    kp = torch.from_numpy(np.tile(PD_KP, (n, 1))).to(dev)
    kd = torch.from_numpy(np.tile(PD_KD, (n, 1))).to(dev)
    b.bind(P.F_PD_KP, kp.data_ptr())
    b.bind(P.F_PD_KD, kd.data_ptr())
    b.set_pd_mode(True)
    group_rows = [torch.from_numpy(np.nonzero(env_ids % NGROUP == g)[0]).to(dev) for g in range(NGROUP)]
    obs_all = torch.empty((world * n, nobs), dtype=torch.float64, device=dev) if world > 1 else None
    launch_stream = torch.cuda.Stream(device=dev)   # a real (non-null) stream: the kernel and the timing events share it
    stream = launch_stream.cuda_stream

    def restart(group):
        rows = group_rows[group]
        if rows.numel():
            obs[rows, : nq + nv] = init_row[: nq + nv]
            warm[rows] = 0

    sch = Schedule(step=lambda nsub: b.step(nsub, stream),
                   bind_targets=lambda p: b.bind(P.F_PD_PTARGET, targets[p].data_ptr()),
                   restart=restart,
                   gather=(lambda: gather_observations(obs, world, obs_all)) if world > 1 else None,
                   substeps_per_launch=args.substeps_per_launch)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    torch.cuda.synchronize(dev)
    ev = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
    with torch.cuda.stream(launch_stream):
        sch.run(0, PREROLL + args.warmup)
        elapsed = timed_region(sch, PREROLL + args.warmup, args.steps, fence, mark=lambda i: ev[i].record(launch_stream))
    ev0, ev1 = ev
    timed_launches = sch.launches
    launch_ms_stream = ev0.elapsed_time(ev1) / timed_launches   # mean stream time per launch (includes the rare restart / gather)
    elapsed = max_over_ranks(elapsed, world, dev)

    w, info = b.warnings()
    nwarn = int(np.count_nonzero(w))

    if rank == 0:
        # ---- the metric's second half: sampled envs of the timed batch against the CPU reference ----
        nsample = max(1, min(args.parity_envs, n))
        sample = np.unique(np.linspace(0, n - 1, nsample).astype(int))
        q_gpu = obs[:, :nq].cpu().numpy()[sample]
        threads = 1
        try:
            from cassie_amd._lib import lib
            threads = lib().cassie_host_cpu_count()
        except Exception:
            pass
        orc = replay_on_oracle(model, env_ids[sample], lambda p: targets_host[p][sample], total_steps, hfield, threads)
        q_ref = orc.qpos()
        err_abs = np.abs(q_gpu - q_ref)
        err_rel = err_abs / np.maximum(1.0, np.abs(q_ref))
        counts_equal = float(np.mean(np.all(info[sample][:, :3] == orc.counts(), axis=1)))

        # dominant-kernel duration: HIP events on the launch stream around the K timed launches
        kern_ms = launch_ms_stream
        steps_per_launch = args.steps / timed_launches
        # read qpos+qvel+qacc_warmstart+ctrl, write qpos+qvel+qacc+sensordata+actuator_velocity (1976 B for cassie)
        algo_bytes = 8 * ((pod.nq + 2 * pod.nv + pod.nu) + (pod.nq + 2 * pod.nv + pod.nsensordata + pod.nu))
        assert args.model != "cassie" or algo_bytes == ALGO_BYTES_PER_ENV_STEP
        achieved = algo_bytes * n * steps_per_launch / (kern_ms * 1e-3) / 1e9
        value = world * n * args.steps / elapsed
        traffic, traffic_src = pmc_traffic(n * steps_per_launch)
        out = {
            "metric": "env-steps/sec (whole node) at N envs; max |qpos_err| vs CPU ref", "value": value, "unit": "env-steps/s",
            "max_qpos_err": float(err_abs.max()), "max_qpos_rel_err": float(err_rel.max()),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d envs/GPU, %s.xml, %d-step episodes from the cassie_sim_init pose restarted at staggered phases "
                                   "(untimed pre-roll of %d steps), random joint-PD targets re-drawn every %d steps; `value` is the "
                                   "device-resident API (phys_batch_step: PD law + motor limit + physics in one kernel, cassie_sim_step_pd's "
                                   "motor-PD semantics without the Agility host blocks); `value_step_pd` is cassie_sim_step_pd itself, batched"
                                   % (n, args.model, EPISODE, PREROLL, HOLD),
                       "api_of_value": "phys_batch_step (device-resident, include/cassie_phys.h)",
                       "envs_total": world * n, "parallelism": "env-sharded x%d" % world,
                       "obs_allgather_every_steps": HOLD if world > 1 else None,
                       "substeps_per_launch": steps_per_launch, "launches_timed": timed_launches,
                       "preroll_steps": PREROLL, "episode_steps": EPISODE},
            "parity": {"reference": "oracle/cassie_oracle.c (fp64 CPU restatement of mj_step1 + mj_step2; parity with genuine MuJoCo unpinned)",
                       "envs_compared": int(len(sample)), "steps_replayed": int(total_steps),
                       "max_qpos_err": float(err_abs.max()), "max_qpos_rel_err": float(err_rel.max()),
                       "frac_envs_with_equal_ncon_nefc_iters": counts_equal, "tolerance_rel": 1e-6,
                       "ok": bool(err_rel.max() <= 1e-6)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "cassie_step_kernel<%d>" % (32 if pod.nv <= 32 else 40), "kernel_ms": kern_ms, "algorithmic_bytes_per_env_step": algo_bytes, "env_steps_per_launch": n * steps_per_launch,
                         "note": "latency-bound by design: ~2 KB of state vs ~0.22 MFLOP of serially dependent fp64 per env-step"},
            # the more telling bound: ~0.22 MFLOP of algorithmic fp64 work per env-step against the fp64 vector peak
            "roofline_fp64": {"bound": "fp64-valu", "achieved": value * 0.22e6 / 1e12, "peak": 78.6, "unit": "TFLOP/s",
                              "frac": value * 0.22e6 / 1e12 / 78.6,
                              "note": "algorithmic flops (SURVEY.md 8a estimate), not counting lanes that idle or recompute"},
            "envs_with_warnings": nwarn,
            "mean_constraint_rows": float(info[:, 1].mean()), "mean_pgs_iterations": float(info[:, 2].mean()), "mean_pgs_guarded_sweeps": float(info[:, 3].mean()),
        }
        if world == 1 and args.model == "cassie":
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(model)
            out["true_reference"] = true_reference(args.model, model.qpos_init(), targets_host[:, sample[:8]], EPISODE)
            if not args.no_step_pd:
                b.close()
                sp = step_pd_host_api(n)
                out["step_pd_host_api"] = sp
                out["value_step_pd"] = sp["value"] if sp else None
        print(json.dumps(out), flush=True)
    b.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
