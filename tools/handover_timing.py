#!/usr/bin/env python3
"""Rate of a hand-over-heavy workload (needs a GPU): the +-10 rad stress targets of reference example/cassietest_jac.py:106 push
joints into their limits, so a good part of the envs leaves the row-capped fast kernel in the middle of fused launches and is
finished by the pass behind it.  Prints env-steps/s and the share of env-launches handed over."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cassie-mujoco-sim_amd")); sys.path.insert(0, REPO)
import bench
from cassie_amd import Batch, Model
from cassie_amd import phys as P
name = sys.argv[1] if len(sys.argv) > 1 else "cassie"
m = Model(name)
n, npol = 4096, 24
b = Batch(m, n)
if name == "cassie_hfield":
    h = np.random.default_rng(99).random((200, 200)).astype(np.float32); h[95:105, 95:105] = 0
    b.set_hfield(h)
if os.environ.get("WAVES"):
    b.set_waves_per_env(int(os.environ["WAVES"]))
q0 = np.tile(m.qpos_init(), (n, 1)); q0[:, 2] -= 0.2
b.set(P.F_QPOS, q0)
b.set(P.F_PD_KP, np.tile(bench.PD_KP, (n, 1))); b.set(P.F_PD_KD, np.tile(bench.PD_KD, (n, 1)))
b.forward(); b.set_drive_mode(P.DRIVE_PD)
rng = np.random.default_rng(5)
tg = bench.PD_OFFSET + rng.uniform(-10, 10, (npol, n, 10))
handed, t_used = 0, 0.0
for p in range(npol):
    b.set(P.F_PD_PTARGET, tg[p])
    b.sync(); t0 = time.perf_counter()
    b.step(bench.HOLD); b.sync()
    if p >= 4:
        t_used += time.perf_counter() - t0
        handed += int(np.count_nonzero(b.fast_rows_progress() < bench.HOLD))
w, info = b.warnings()
print("%s: %.2f M env-steps/s, %.1f %% of env-launches handed over, rows of the last substep mean %.1f max %d, warnings %d"
      % (name, n * bench.HOLD * (npol - 4) / t_used / 1e6, 100.0 * handed / (n * (npol - 4)), info[:, 1].mean(), info[:, 1].max(), int(np.count_nonzero(w))))
b.close()
