import os, sys, time
import numpy as np
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "cassie-mujoco-sim_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import bench
from cassie_amd import Batch, Model
from cassie_amd import phys as P
m = Model("cassie")
n = 4096
for per_env in (False, True, False, True):
    b = Batch(m, n)
    b.set(P.F_QPOS, np.tile(m.qpos_init(), (n, 1)))
    b.set(P.F_PD_KP, np.tile(bench.PD_KP, (n, 1))); b.set(P.F_PD_KD, np.tile(bench.PD_KD, (n, 1)))
    b.set(P.F_PD_PTARGET, bench.pd_targets(np.arange(n), 1)[0])
    b.set_pd_mode(True)
    if per_env:
        for e in range(n):
            b.set_model(m.pod, e)
    b.step(300); b.sync()
    ms = b.time_steps(50, 8)
    print("per-env model copies" if per_env else "one shared model     ", "%.3f ms per 50-substep launch -> %.2f M env-steps/s" % (ms, n * 50 / ms / 1e3))
    b.close()
