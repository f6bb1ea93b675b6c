#!/usr/bin/env python3
"""Two probes on a synchronised batch (every env the same episode phase; GPU box): (1) how much do the model-constant reads
cost -- every env given its OWN copy of the 150 KB model, so nothing is shared in L1 / L2; (2) the two device modes and the
fast kernel on / off, with the hand-over fraction."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "cassie-mujoco-sim_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import bench  # noqa: E402
from cassie_amd import Batch, Model  # noqa: E402
from cassie_amd import phys as P  # noqa: E402

m = Model("cassie")
n = 4096
for mode, per_env, fast in (("exact", False, True), ("exact", True, True), ("drive", False, True), ("drive", False, False), ("exact", False, False)):
    b = Batch(m, n)
    b.set_fast_rows(fast)
    b.set(P.F_QPOS, np.tile(m.qpos_init(), (n, 1)))
    b.set(P.F_PD_KP, np.tile(bench.PD_KP, (n, 1))); b.set(P.F_PD_KD, np.tile(bench.PD_KD, (n, 1)))
    b.set(P.F_PD_PTARGET, bench.pd_targets(np.arange(n), 1)[0])
    if mode == "drive":
        b.forward(); b.set_drive_mode(P.DRIVE_PD)
    else:
        b.set_pd_mode(True)
    if per_env:
        for e in range(n):
            b.set_model(m.pod, e)
    b.step(300); b.sync()
    ms = b.time_steps(50, 8)
    w, info = b.warnings()
    ho = float(np.mean(b.fast_rows_progress() < 50)) if fast else float("nan")
    print("%s-pd, %s, fast kernel %s: %.3f ms per 50-substep launch -> %.2f M env-steps/s; rows mean %.1f max %d, sweeps mean %.1f, handed over %.3f"
          % (mode, "per-env model copies" if per_env else "one shared model", "on" if fast else "off", ms, n * 50 / ms / 1e3, info[:, 1].mean(), info[:, 1].max(), info[:, 2].mean(), ho))
    b.close()
