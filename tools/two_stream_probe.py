#!/usr/bin/env python3
"""What would two half-batches on two streams buy (task 6a)?  The same 4096 envs of the benchmark's PD
workload stepped (a) as one batch on one stream and (b) as two batches of 2048 on two streams, 50-substep launches queued
back to back with the benchmark's staggered episode restarts (fixed PD targets), so that one half's workgroups can fill the wave slots the other half's tail leaves idle.  Prints env-steps/s."""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "cassie-mujoco-sim_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import bench  # noqa: E402
from cassie_amd import Batch, Model  # noqa: E402
from cassie_amd import phys as P  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cassie"
model = Model(name)
hf = None
if name == "cassie_hfield":
    hf = np.random.default_rng(99).random((200, 200)).astype(np.float32)
    hf[95:105, 95:105] = 0
N, LAUNCHES = 4096, 40


q_init = torch.from_numpy(np.concatenate([model.qpos_init(), bench.HostChainEnvs(model, [0], hf).init_sensordata()])).cuda()
NG = bench.NGROUP


def restart(b, n, offset, p, s):
    """the benchmark's staggered episodes: at policy step p the envs whose global id is p mod 20 start over"""
    first = (p - offset) % NG
    count = len(range(first, n, NG))
    b.reset_envs(first, NG, count, q_init.data_ptr(), q_init.data_ptr() + 8 * model.pod.nq, s.cuda_stream)


def make(n, ids):
    b = Batch(model, n)
    if hf is not None:
        b.set_hfield(hf)
    b.set(P.F_QPOS, np.tile(model.qpos_init(), (n, 1)))
    b.set(P.F_PD_KP, np.tile(bench.PD_KP, (n, 1)))
    b.set(P.F_PD_KD, np.tile(bench.PD_KD, (n, 1)))
    b.set(P.F_PD_PTARGET, bench.pd_targets(ids, 1)[0])
    b.forward()
    b.set_drive_mode(P.DRIVE_PD)
    return b


def run(batches, streams, join=False):
    n = N // len(batches)
    for p in range(NG):                         # pre-roll: one episode length, so that the mix of episode phases is the stationary one
        for k, (b, s) in enumerate(zip(batches, streams)):
            restart(b, n, k * n, p, s)
            b.step(50, s.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for p in range(NG, NG + LAUNCHES):
        if join and len(streams) > 1:           # fork-join per policy step: no half starts step p + 1 before both finished step p
            evs = []
            for s in streams:
                e = torch.cuda.Event(); e.record(s); evs.append(e)
            for s in streams:
                for e in evs:
                    s.wait_event(e)
        for k, (b, s) in enumerate(zip(batches, streams)):
            restart(b, n, k * n, p, s)
            b.step(50, s.cuda_stream)
    torch.cuda.synchronize()
    return N * 50 * LAUNCHES / (time.perf_counter() - t0)


for rep in range(2):
    one = [make(N, np.arange(N))]
    r1 = run(one, [torch.cuda.Stream()])
    one[0].close()
    two = [make(N // 2, np.arange(N // 2)), make(N // 2, np.arange(N // 2, N))]
    r2 = run(two, [torch.cuda.Stream(), torch.cuda.Stream()])
    r3 = run(two, [torch.cuda.Stream(), torch.cuda.Stream()], join=True)
    for b in two:
        b.close()
    four = [make(N // 4, np.arange(k * N // 4, (k + 1) * N // 4)) for k in range(4)]
    r4 = run(four, [torch.cuda.Stream() for _ in range(4)])
    for b in four:
        b.close()
    print("%s: one batch of %d on one stream %.3f M env-steps/s; two batches of %d on two streams %.3f M (%+.1f %%); with a join per policy step %.3f M (%+.1f %%); four batches on four streams %.3f M (%+.1f %%)"
          % (name, N, r1 / 1e6, N // 2, r2 / 1e6, 100 * (r2 / r1 - 1), r3 / 1e6, 100 * (r3 / r1 - 1), r4 / 1e6, 100 * (r4 / r1 - 1)))
