#!/usr/bin/env python3
"""Where a substep of the 127-row instantiation spends its clocks: robots lying on the rough part of the bench terrain with
CM_FLAG_HFPRISM (more than 64 rows: the sweep crosses the waves), the kernel alone, stage stamps of one substep.  Needs a GPU."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cassie-mujoco-sim_amd")); sys.path.insert(0, REPO)
import bench
from cassie_amd import Batch, Model
from cassie_amd import phys as P
m = Model("cassie_hfield")
m.set_flag(P.FLAG_HFPRISM, True)
n = 4096
b = Batch(m, n)
hf = np.random.default_rng(99).random((200, 200)).astype(np.float32)
hf[95:105, 95:105] = 0
b.set_hfield(hf)
b.set(P.F_QPOS, np.tile(m.qpos_init(), (n, 1)))
npol = 20
tg = bench.pd_targets(np.arange(n), npol)       # the bench's workload: the robots that fall lie on the rough terrain
b.set(P.F_PD_KP, np.tile(bench.PD_KP, (n, 1))); b.set(P.F_PD_KD, np.tile(bench.PD_KD, (n, 1)))
b.forward(); b.set_drive_mode(P.DRIVE_PD_SAFE)
for p in range(npol):
    b.set(P.F_PD_PTARGET, tg[p]); b.step(50)
b.sync()
b.set_fast_rows(False)                            # the 127-row instantiation alone
ms = b.time_steps(1, 20)
st = b.profile_step(1)
w, info = b.warnings()
wide = info[:, 1] > 64
print("%d of %d envs have more than 64 rows (rows mean %.1f max %d, sweeps mean %.1f); %.3f ms per one-substep launch of the batch" % (wide.sum(), n, info[wide, 1].mean(), info[:, 1].max(), info[wide, 2].mean(), ms))
st = st[wide]
dur = lambda a, c: (st[:, c] - st[:, a]).astype(float).mean()
tot0 = dur(0, 37)
print("wave 0's substep %.0f clocks" % tot0)
for nm, a, c in [("w0 drive io + kinematics (incl. F)", 0, 1), ("w0 geoms", 1, 17), ("w0 collision (+ drive io)", 17, 33), ("w0 WAIT at X", 33, 5),
                 ("w0 velocity -> cfrc", 5, 24), ("w0 rows+J", 24, 34), ("w0 WAIT at J (+ read-outs)", 34, 8), ("w0 halfsolve", 8, 9),
                 ("w0 A", 9, 10), ("w0 pgs", 10, 11), ("  pgs: warm start", 10, 30), ("  pgs: sweeps", 30, 11), ("w0 hand f over, WAIT for wave 1's qacc + Euler", 11, 37),
                 ("w1 factor M+hB .. qacc", 39, 12), ("w1 euler (then E)", 12, 13), ("w1 accelerometers + outputs (behind E)", 13, 14)]:
    print("  %-48s %9.0f cycles  %5.1f%%" % (nm, dur(a, c), 100 * dur(a, c) / tot0))
sw = (st[:, 11] - st[:, 30]).astype(float)
print("  per sweep: %.0f clocks (mean over envs of sweeps' clocks / sweeps); per (sweep x row): %.1f" % ((sw / info[wide, 2]).mean(), (sw / info[wide, 2] / info[wide, 1]).mean()))
if os.environ.get("CASSIE_LIB", "").endswith("wprof.so"):     # (a library built with -DCK_WIDE_PROFILE: the sweeps' clocks by part)
    names = ["wait", "cross", "rows", "post (change, guard, estimate)", "image", "publish / verdict"]
    it = info[wide, 2].astype(float)
    for w_, base in ((0, 16), (1, 28)):
        print("  wave %d, clocks per sweep:" % w_, ", ".join("%s %.0f" % (nm, (st[:, base + i] / it).mean()) for i, nm in enumerate(names)), "| sum %.0f" % (st[:, base:base + 6].sum(axis=1) / it).mean())
b.close()
