#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched Cassie physics step on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1 is launched by
torch.distributed.run, one rank per GPU).  A *step* is one pass of the hot path over
one batch: every env of the batch advances by one 0.5 ms physics step
(= one cassie_sim_step_pd-equivalent; reference src/cassiemujoco.c:1130-1134).

Workload (BASELINE.json configs[1]): 4096 envs per GPU, cassie model, every env starts from
the cassie_sim_init pose (reference src/cassiemujoco.c:1023-1029); per-env random PD targets
(offset + U(-0.3, 0.3) rad, gains of reference example/cassietest_jac.py:51-52, :68) re-drawn
every 50 steps from a table that is resident in HBM; the PD law + motor speed-torque limit run
on the device inside the step kernel, so no host buffer is touched in the timed region.
With N>1 envs shard across GPUs with no data-path exchange; the only collective is one RCCL
all-gather of the observation block (qpos|qvel|sensordata, 96 doubles/env) per 50 steps.

The JSON line also carries:
  roofline      algorithmic HBM bytes of one launch (1976 B per env-step) divided
                by the mean kernel duration measured with HIP events on the launch stream
  cpu_baseline  the fp64 CPU oracle ("port") on the host cores, OpenMP over envs, same workload
                distribution, bounded sample
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


def gather_observations(obs, world, out=None):
    """All-gather of the per-rank observation block [n, 96] into [world * n, 96], rank-major = global env order.
    RCCL over xGMI on GPUs ('nccl' backend), gloo in the CPU tests."""
    import torch
    import torch.distributed as dist
    if out is None:
        out = torch.empty((world * obs.shape[0], obs.shape[1]), dtype=obs.dtype, device=obs.device)
    dist.all_gather_into_tensor(out, obs)
    return out


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


def cpu_baseline(model, budget_s=12.0):
    """Times the CPU oracle on a bounded sample of the same workload, all host cores (OpenMP over envs)."""
    import oracle_py
    from cassie_amd._lib import lib
    cores = lib().cassie_host_cpu_count()      # affinity- and cgroup-quota-aware
    nenv = 16 * cores
    L = oracle_py.lib()
    buf = (oracle_py.CoData * nenv)()
    q0 = model.qpos_init()
    for e in range(nenv):
        L.co_reset(ctypes.byref(model.pod), ctypes.byref(buf[e]))
        oracle_py.arr(buf[e].qpos)[: model.pod.nq] = q0
    kp = np.tile(PD_KP, (nenv, 1))
    kd = np.tile(PD_KD, (nenv, 1))
    done, t_used, pol = 0, 0.0, 0
    tg = pd_targets(range(nenv), 400)
    while t_used < budget_s and pol < 400:
        pt = np.ascontiguousarray(tg[pol])
        t0 = time.perf_counter()
        L.co_step_batch(ctypes.byref(model.pod), ctypes.byref(buf), nenv, HOLD, pt.ctypes.data, kp.ctypes.data,
                        kd.ctypes.data, cores)
        t_used += time.perf_counter() - t0
        done += nenv * HOLD
        pol += 1
    return {"value": done / t_used, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d envs x %d steps, same PD workload, oracle/cassie_oracle.c, OpenMP over envs" % (nenv, pol * HOLD)}


def step_pd_host_api(n, steps=200, warmup=50):
    """The full cassie_sim_step_pd semantics for n envs (include/cassie_batch.h): Agility blocks + encoder / motor
    models on the host thread pool, ctrl / sensordata over PCIe every step, physics on the GPU.  Host-bound by the
    closed Agility code; reported beside `value`, never as `value`."""
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    total_steps = args.warmup + args.steps
    npolicy = (total_steps + HOLD - 1) // HOLD + 1

    b = Batch(model, n, device=local_rank)
    if args.model == "cassie_hfield":   # terrain of reference example/test_hfield.py:39-41, shared by all envs
        hf = np.random.default_rng(99).random((200, 200)).astype(np.float32)
        hf[95:105, 95:105] = 0
        b.set_hfield(hf)
    dev = torch.device("cuda", local_rank)
    # state and inputs live in HBM before the timed region starts (torch owns the observation fields)
    qpos = torch.from_numpy(np.tile(model.qpos_init(), (n, 1))).to(dev)
    qvel = torch.zeros((n, pod.nv), dtype=torch.float64, device=dev)
    sens = torch.zeros((n, pod.nsensordata), dtype=torch.float64, device=dev)
    b.bind(P.F_QPOS, qpos.data_ptr())
    b.bind(P.F_QVEL, qvel.data_ptr())
    b.bind(P.F_SENSORDATA, sens.data_ptr())
    targets = torch.from_numpy(pd_targets(env_ids, npolicy)).to(dev)       # [npolicy][n][10]
    kp = torch.from_numpy(np.tile(PD_KP, (n, 1))).to(dev)
    kd = torch.from_numpy(np.tile(PD_KD, (n, 1))).to(dev)
    b.bind(P.F_PD_KP, kp.data_ptr())
    b.bind(P.F_PD_KD, kd.data_ptr())
    b.set_pd_mode(True)
    obs_all = torch.empty((world * n, pod.nq + pod.nv + pod.nsensordata), dtype=torch.float64, device=dev) if world > 1 else None
    launch_stream = torch.cuda.Stream(device=dev)   # a real (non-null) stream: the kernel and the timing events share it
    stream = launch_stream.cuda_stream

    nlaunch = [0]

    def run(first, count):
        """`count` physics steps starting at step index `first`.  The PD targets are held for HOLD steps, so one
        launch advances every env up to the next re-draw (state stays in LDS between those substeps; set
        --substeps-per-launch 1 to launch every step)."""
        s, end = first, first + count
        while s < end:
            if s % HOLD == 0:
                b.bind(P.F_PD_PTARGET, targets[s // HOLD].data_ptr())
                if world > 1 and s > 0:
                    gather_observations(torch.cat((qpos, qvel, sens), dim=1), world, obs_all)
            nsub = min(args.substeps_per_launch, HOLD - s % HOLD, end - s)
            b.step(nsub, stream)
            nlaunch[0] += 1
            s += nsub

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    torch.cuda.synchronize(dev)
    with torch.cuda.stream(launch_stream):
        run(0, args.warmup)
        fence()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(launch_stream)
        nlaunch[0] = 0
        run(args.warmup, args.steps)
        ev1.record(launch_stream)
        fence()
    elapsed = time.perf_counter() - t0
    timed_launches = nlaunch[0]
    launch_ms_stream = ev0.elapsed_time(ev1) / timed_launches   # mean stream time per launch (includes the rare gather)

    w, info = b.warnings()
    nwarn = int(np.count_nonzero(w))
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
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
            "metric": "env-steps/sec (whole node) at N envs", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d envs/GPU, %s.xml, random joint-PD targets re-drawn every %d steps, "
                                   "PD + motor limit + physics on device (cassie_sim_step_pd motor-PD semantics, "
                                   "Agility host blocks not in the timed region)" % (n, args.model, HOLD),
                       "envs_total": world * n, "parallelism": "env-sharded x%d" % world,
                       "obs_allgather_every_steps": HOLD if world > 1 else None,
                       "substeps_per_launch": steps_per_launch, "launches_timed": timed_launches},
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
        if world == 1 and not args.no_cpu_baseline and args.model == "cassie":
            out["cpu_baseline"] = cpu_baseline(model)
            b.close()
            out["step_pd_host_api"] = step_pd_host_api(n)
        print(json.dumps(out), flush=True)
    b.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
