#!/usr/bin/env python3
"""Per-stage shader-clock breakdown of the step kernel (profiling aid; needs a GPU)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cassie-mujoco-sim_amd"))
from cassie_amd import Batch, Model
from cassie_amd import phys as P
names = ["kinematics", "geoms+com+cinert+cdof", "crba", "factor", "collision", "velocity+rne", "qfrc_smooth",
         "rows+J", "halfsolve", "A", "pgs", "qacc", "sensors", "euler"]
m = Model("cassie")
NSUB = int(os.environ.get("NSUB", "1"))   # substeps fused per launch (the bench uses 50)
for n in (int(a) for a in (sys.argv[1:] or ["512", "4096"])):
    b = Batch(m, n)
    if os.environ.get("FULL_KERNEL"):
        b.set_fast_rows(False)      # the stamps of the full instantiation alone (default: the row-capped fast one, where an env fits it)
    b.set(P.F_QPOS, np.tile(m.qpos_init(), (n, 1)))
    rng = np.random.default_rng(0)
    b.set(P.F_PD_PTARGET, np.array([0.0045, 0, 0.4973, -1.1997, -1.5968] * 2) + rng.uniform(-0.3, 0.3, (n, 10)))
    b.set(P.F_PD_KP, np.tile([70, 70, 100, 100, 50] * 2, (n, 1)))
    b.set(P.F_PD_KD, np.tile([7, 7, 8, 8, 5] * 2, (n, 1)))
    b.set_pd_mode(True)
    b.step(300); b.sync()
    ms = b.time_steps(NSUB, 50 if NSUB == 1 else 4) / NSUB
    st = b.profile_step(NSUB)
    w, info = b.warnings()
    d = np.diff(st[:, :15], axis=1).astype(float)
    tot = (st[:, 14] - st[:, 0]).astype(float)
    print("nenv %d, %d substeps per launch: %.3f ms/step; per-env kernel cycles mean %.0f (min %.0f max %.0f); nefc mean %.1f iters mean %.1f (guarded %.2f)"
          % (n, NSUB, ms, tot.mean(), tot.min(), tot.max(), info[:, 1].mean(), info[:, 2].mean(), info[:, 3].mean()))
    span = st[:, 14].max() - st[:, 0].min()
    print("  whole-launch span in clock ticks: %d" % span)
    for i, nm in enumerate(names):
        print("  %-24s %9.0f cycles  %5.1f%%" % (nm, d[:, i].mean(), 100 * d[:, i].mean() / tot.mean()))
    # sub-stage stamps: (label, from stamp, to stamp)
    sub = [("kin: joint locals", 0, 16), ("kin: level loop", 16, 1), ("geoms", 1, 17), ("com reduce", 17, 18), ("cinert+cdof", 18, 2),
           ("crb subtree sums", 2, 19), ("buf", 19, 20), ("M columns", 20, 3), ("coll: block cull", 4, 21), ("coll: lane pass", 21, 22), ("  pair loads", 21, 31), ("  narrow phase", 31, 32), ("  compact+write", 32, 22),
           ("coll: wave pass", 22, 5), ("vel: cvel+cdof_dot", 5, 23), ("vel: cacc+cfrc", 23, 24), ("vel: bias proj", 24, 6),
           ("rows: descriptors", 7, 25), ("rows: geometry", 25, 26), ("rows: impedance", 26, 27), ("rows: J loop", 27, 28),
           ("rows: sensors1", 28, 29), ("rows: dumps+sync", 29, 8), ("pgs: warm start", 10, 30), ("pgs: sweeps", 30, 11)]
    for nm, a, c in sub:
        v = (st[:, c] - st[:, a]).astype(float)
        print("    %-22s %9.0f cycles" % (nm, v.mean()))
    # what one PGS sweep costs: least squares of the sweep-loop clocks of the last substep on (1, sweeps, sweeps x rows)
    sw = (st[:, 11] - st[:, 30]).astype(float)
    it, ne = info[:, 2].astype(float), info[:, 1].astype(float)
    A = np.stack([np.ones_like(it), it, it * np.ceil(ne / 4) * 4], axis=1)
    coef, *_ = np.linalg.lstsq(A, sw, rcond=None)
    print("    pgs sweeps ~ %.0f + %.0f per sweep + %.1f per (sweep x row, rows rounded up to 4); total cycles of the step vs sweeps: %.0f per sweep"
          % (coef[0], coef[1], coef[2], np.polyfit(it, tot, 1)[0]))
    b.close()
