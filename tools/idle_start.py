"""What a launch costs when it starts from an idle GPU: two back-to-back 20-substep launches after 0 / 1 / 10 / 100 ms of
idleness (profiles/round2/micro_idle_start.txt).  The shader clock drops within a millisecond of idleness and takes tens of
milliseconds to come back, which is what the short timed region of `bench.py --steps 20` pays."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cassie-mujoco-sim_amd"))
import torch
from cassie_amd import Batch, Model
from cassie_amd import phys as P
m = Model("cassie"); n = 4096
b = Batch(m, n)
b.set(P.F_QPOS, np.tile(m.qpos_init(), (n, 1)))
rng = np.random.default_rng(0)
b.set(P.F_PD_PTARGET, np.array([0.0045, 0, 0.4973, -1.1997, -1.5968] * 2) + rng.uniform(-0.3, 0.3, (n, 10)))
b.set(P.F_PD_KP, np.tile([70, 70, 100, 100, 50] * 2, (n, 1))); b.set(P.F_PD_KD, np.tile([7, 7, 8, 8, 5] * 2, (n, 1)))
b.set_pd_mode(True)
s = torch.cuda.Stream(); st = s.cuda_stream
for _ in range(10): b.step(50, st)
torch.cuda.synchronize()
def ev(): return torch.cuda.Event(enable_timing=True)
for idle_ms in (0, 0, 1, 10, 100):
    res = []
    for rep in range(5):
        torch.cuda.synchronize(); time.sleep(idle_ms / 1e3)
        e0, e1, e2 = ev(), ev(), ev()
        t0 = time.perf_counter()
        e0.record(s); b.step(20, st); e1.record(s); b.step(20, st); e2.record(s)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        res.append((e0.elapsed_time(e1), e1.elapsed_time(e2), (t1 - t0) * 1e3))
    r = np.array(res).mean(0)
    print("idle %3d ms before: first launch %.3f ms, second %.3f ms, wall for both %.3f ms" % (idle_ms, r[0], r[1], r[2]))
