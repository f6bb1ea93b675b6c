import os, sys, numpy as np
sys.path.insert(0, "/root/repo/cassie-mujoco-sim_amd"); sys.path.insert(0, "/root/repo")
import bench
from cassie_amd import Batch, Model, phys as P
m = Model("cassie"); n = 4096
b = Batch(m, n); b.set_chunks(1)
b.set(P.F_QPOS, np.tile(m.qpos_init(), (n, 1)))
rng = np.random.default_rng(0)
b.set(P.F_PD_PTARGET, np.tile(bench.PD_OFFSET, (n, 1)) + rng.uniform(-0.3, 0.3, (n, 10)))
b.set(P.F_PD_KP, np.tile(bench.PD_KP, (n, 1))); b.set(P.F_PD_KD, np.tile(bench.PD_KD, (n, 1)))
b.forward(); b.set_drive_mode(P.DRIVE_PD_SAFE)
b.step(300); b.sync()
xs, ys = [], []
for nsub in (1, 2, 3, 5, 7, 10, 15, 25, 50):
    ms = b.time_steps(nsub, 30)      # ms per launch of nsub substeps (mean of 30)
    xs.append(nsub); ys.append(ms)
    print("nsub %2d: %.3f ms per launch = %.4f ms per substep" % (nsub, ms, ms / nsub))
a, c = np.polyfit(xs, ys, 1)
print("fit: %.4f ms per substep + %.4f ms per launch (= %.2f substeps' worth)" % (a, c, c / a))
b.close()
