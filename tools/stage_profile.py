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
m = Model(os.environ.get("MODEL", "cassie"))   # MODEL=cassie_tray_box: the 40-dof instantiation (one wave per env by default)
if os.environ.get("PRISM"):
    m.set_flag(P.FLAG_HFPRISM, True)          # PRISM=1 MODEL=cassie_hfield: the MuJoCo-shaped height-field contact set
NSUB = int(os.environ.get("NSUB", "1"))   # substeps fused per launch (the bench uses 50)
for n in (int(a) for a in (sys.argv[1:] or ["512", "4096"])):
    b = Batch(m, n)
    if os.environ.get("FULL_KERNEL"):
        b.set_fast_rows(False)      # the stamps of the full instantiation alone (default: the row-capped fast one, where an env fits it)
    WAVES = int(os.environ.get("WAVES", "2"))
    b.set_waves_per_env(WAVES)
    # launches in one piece by default here (CHUNKS=4 for the product's whole-batch default): the slot analysis below takes an env's
    # start as its end minus its cost, which holds for one workgroup per env only
    b.set_chunks(int(os.environ.get("CHUNKS", "1")))
    if m.name == "cassie_hfield":     # the bench's terrain (reference example/test_hfield.py:39-41), every env at the flat centre patch
        hf = np.random.default_rng(99).random((200, 200)).astype(np.float32)
        hf[95:105, 95:105] = 0
        b.set_hfield(hf)
    b.set(P.F_QPOS, np.tile(m.qpos_init(), (n, 1)))
    rng = np.random.default_rng(0)
    b.set(P.F_PD_PTARGET, np.array([0.0045, 0, 0.4973, -1.1997, -1.5968] * 2) + rng.uniform(-0.3, 0.3, (n, 10)))
    b.set(P.F_PD_KP, np.tile([70, 70, 100, 100, 50] * 2, (n, 1)))
    b.set(P.F_PD_KD, np.tile([7, 7, 8, 8, 5] * 2, (n, 1)))
    b.set_pd_mode(True)
    b.step(300); b.sync()
    ms = b.time_steps(NSUB, 50 if NSUB == 1 else 4) / NSUB
    if os.environ.get("TWO_STREAM_LOAD"):
        # the stamps of a launch that shares the GPU with another batch's launches (the bench's two-range stepping): a second batch is
        # given two launches of its own, asynchronously, right before the profiled launch
        b2 = Batch(m, n)
        b2.set_waves_per_env(WAVES)
        b2.set(P.F_QPOS, b.get(P.F_QPOS)); b2.set(P.F_QVEL, b.get(P.F_QVEL))
        b2.set(P.F_PD_PTARGET, b.get(P.F_PD_PTARGET)); b2.set(P.F_PD_KP, b.get(P.F_PD_KP)); b2.set(P.F_PD_KD, b.get(P.F_PD_KD))
        b2.set_pd_mode(True)
        b2.step(NSUB); b2.sync()
        b2.step(NSUB); b2.step(NSUB); b2.step(NSUB)
    st = b.profile_step(NSUB)
    if os.environ.get("TWO_STREAM_LOAD"):
        b2.sync(); b2.close()
    if n >= 2048:
        c = b.launch_cost()
        print("  whole launch per env: %.0f shader clocks = %.0f per substep (mean; min %.0f max %.0f per substep)" % (c.mean(), c.mean() / NSUB, c.min() / NSUB, c.max() / NSUB))
    w, info = b.warnings()
    # the shader clock against the constant 100 MHz wall clock, both read at each env's end: the clock the chip really ran at
    # under this load (a straight-line fit over every env's end of launch), and the launch's span on the wall clock
    ce, we = st[:, 42].astype(float), st[:, 43].astype(float)
    if os.environ.get("DUMP"):
        np.savez_compressed(os.environ["DUMP"] + "_%d_w%d.npz" % (n, WAVES), st=st, cost=b.launch_cost() if n >= 2048 else np.zeros(0), info=info)
    if np.ptp(we) > 0:
        # (the shader clock has a base of its own in every XCD: one fit per XCD, all with the same slope)
        xcc = (st[:, 40] >> 32).astype(int)
        slopes = []
        for x in np.unique(xcc):
            k = xcc == x
            if np.ptp(we[k]) > 0:
                fit = np.polyfit(we[k] - we[k].min(), ce[k] - ce[k].min(), 1)
                slopes.append(fit[0])
                ce[k] = (ce[k] - ce[k].min() - fit[1]) + (we[k].min() - we.min()) * fit[0]   # onto one time base: that of the wall clock
        slope = float(np.mean(slopes))
        print("  shader clock / wall clock over the envs' ends, per XCD: %s ticks per 10 ns -> %.3f GHz; ends span %.3f ms on the wall clock (launch of %d substeps: %.3f ms)"
              % (np.round(slopes, 2).tolist(), slope * 0.1, np.ptp(we) * 1e-5, NSUB, ms * NSUB))
    if n >= 2048 and np.ptp(we) > 0:
        # workgroup-slot occupancy over the launch: every env's start = its end - its cost (shader clocks; the fast kernel's envs)
        cost = b.launch_cost()
        start, end = ce - cost, ce
        t0, t1 = start.min(), end.max()
        slots = 256 * 4
        print("  workgroup slots: sum of env durations / (%d slots x launch span) = %.3f; span %.0f clocks = %.3f ms at the fitted clock"
              % (slots, cost.sum() / (slots * (t1 - t0)), t1 - t0, (t1 - t0) / (slope * 1e5)))
        grid = np.linspace(t0, t1, 21)
        act = [(int(np.sum((start <= t) & (end > t)))) for t in grid]
        print("  envs in flight at 0, 5, .. 100 %% of the span:", act)
        cuid = ((st[:, 40] >> 32) << 12) | (((st[:, 40] >> 13) & 7) << 5) | (((st[:, 40] >> 12) & 1) << 4) | ((st[:, 40] >> 8) & 15)
        per_cu = np.array([cost[cuid == c].sum() for c in np.unique(cuid)])
        print("  per CU: envs (first 8 CUs) %s, summed env clocks / 4 slots: mean %.0f min %.0f max %.0f (span %.0f)"
              % (np.bincount(np.unique(cuid, return_inverse=True)[1]).tolist()[:8], per_cu.mean() / 4, per_cu.min() / 4, per_cu.max() / 4, t1 - t0))
    if m.name == "cassie_hfield":
        d = lambda a, c: (st[:, c] - st[:, a]).astype(float).mean()
        print("  height-field pre-pass: set-up %.0f, the samples' cells (hfield_spheres_wave) %.0f, capsule rule + records %.0f clocks; pair loop and the rest of the lane pass %.0f" % (d(21, 44), d(44, 45), d(45, 46), d(46, 22)))
    if WAVES == 2 and not os.environ.get("FULL_KERNEL"):
        # two-wave form: wave 0 and wave 1 have timelines of their own, meeting at the barriers F, X and J
        dur = lambda a, c: (st[:, c] - st[:, a]).astype(float).mean()
        tot0 = dur(0, 37)
        print("nenv %d, %d substeps per launch, TWO WAVES PER ENV: %.3f ms/step; wave 0's last substep %.0f cycles; nefc mean %.1f iters mean %.1f"
              % (n, NSUB, ms, tot0, info[:, 1].mean(), info[:, 2].mean()))
        for nm, a, c in [("w0 drive io + kinematics (incl. F)", 0, 1), ("w0 geoms", 1, 17), ("w0 collision (+ drive io)", 17, 33), ("w0 WAIT at X", 33, 5),
                         ("w0 velocity -> cfrc", 5, 24), ("w0 rows+J", 24, 34), ("w0 WAIT at J (+ read-outs)", 34, 8), ("w0 halfsolve", 8, 9),
                         ("w0 A", 9, 10), ("w0 pgs", 10, 11), ("w0 hand f over, WAIT for wave 1's qacc + Euler", 11, 37),
                         ("w1 factor M+hB, stage L row / Y column, WAIT at P, qacc", 39, 12), ("w1 euler (then E)", 12, 13), ("w1 accelerometers + outputs (behind E)", 13, 14),
                         ("w1 WAIT at F", 35, 36), ("w1 com+cinert+cdof", 36, 2), ("w1 crba + M columns", 2, 3), ("w1 WAIT at X", 3, 38),
                         ("w1 drive io + factor", 38, 4), ("w1 wait cfrc + bias proj", 4, 6), ("w1 qfrc_smooth", 6, 7), ("w1 sensors1", 7, 47), ("w1 WAIT at J", 47, 39), ("w1 busy (F..factor done)", 36, 4), ("w0 F -> J (its parallel part)", 1, 8)]:
            print("  %-36s %9.0f cycles  %5.1f%%" % (nm, dur(a, c), 100 * dur(a, c) / tot0))
        hw0, hw1 = st[:, 40], st[:, 41]
        simd = lambda h: (h >> 4) & 3
        cu = lambda h: ((h >> 32) << 12) | (((h >> 13) & 7) << 5) | (((h >> 12) & 1) << 4) | ((h >> 8) & 15)
        pairs = {}
        for a, c in zip(simd(hw0), simd(hw1)):
            pairs[(int(a), int(c))] = pairs.get((int(a), int(c)), 0) + 1
        print("  placement: (SIMD of wave 0, SIMD of wave 1) -> envs:", dict(sorted(pairs.items())))
        print("  same CU for both waves: %d of %d; heavy waves (wave 0) per SIMD id:" % (int(np.sum(cu(hw0) == cu(hw1))), n), np.bincount(simd(hw0), minlength=4).tolist(),
              "; distinct CUs seen: %d; workgroup-slot ids:" % len(set(cu(hw0).tolist())), np.bincount((hw0 >> 16) & 15, minlength=16).tolist())
        b.close()
        continue
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
