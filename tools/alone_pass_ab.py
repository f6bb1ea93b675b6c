import sys, time, os
sys.path.insert(0, "cassie-mujoco-sim_amd"); sys.path.insert(0, "tests")
import numpy as np
from cassie_amd import Batch, Model
from cassie_amd import phys as P
m = Model("cassie"); n = 4096
b = Batch(m, n); b.set(P.F_QPOS, np.tile(m.qpos_init(), (n, 1)))
ids = [m.name2id(1, "left-foot"), m.name2id(1, "right-foot"), -1, -1, -1, -1]
b.step(50); b.sync()
for rep in range(3):
    t = time.perf_counter()
    for _ in range(20): b.forward()
    b.sync(); tf = (time.perf_counter() - t) / 20
    t = time.perf_counter()
    for _ in range(20): b.derive(ids)
    b.sync(); td = (time.perf_counter() - t) / 20
print("alone512=%s  forward %.3f ms  derive %.3f ms (4096 envs)" % (os.environ.get("CASSIE_ALONE_512"), tf * 1e3, td * 1e3))
