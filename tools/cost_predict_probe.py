#!/usr/bin/env python3
"""How well does an env's cost in one launch predict its cost in the next (what the launch order is sorted by)?  Needs a GPU."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cassie-mujoco-sim_amd")); sys.path.insert(0, REPO)
import bench
from cassie_amd import Batch, Model
from cassie_amd import phys as P
m = Model("cassie")
n, NSUB = 4096, int(os.environ.get("NSUB", "50"))
b = Batch(m, n)
b.set(P.F_QPOS, np.tile(m.qpos_init(), (n, 1)))
b.set(P.F_PD_KP, np.tile(bench.PD_KP, (n, 1))); b.set(P.F_PD_KD, np.tile(bench.PD_KD, (n, 1)))
b.forward(); b.set_drive_mode(P.DRIVE_PD)
tg = bench.pd_targets(np.arange(n), 40)
costs, work = [], []
for p in range(40):
    b.set(P.F_PD_PTARGET, tg[p])
    b.step(NSUB); b.sync()
    c = b.launch_cost(); w, info = b.warnings()
    costs.append(c.copy()); work.append((info[:, 1] * info[:, 2]).astype(float))
costs = np.array(costs[10:]); work = np.array(work[10:])
cc = [np.corrcoef(costs[i], costs[i + 1])[0, 1] for i in range(len(costs) - 1)]
print("mean cost %.0f clocks per launch, std over envs %.0f (%.1f %%)" % (costs.mean(), costs.std(axis=1).mean(), 100 * costs.std(axis=1).mean() / costs.mean()))
print("correlation of an env's cost with its cost in the previous launch: mean %.3f (min %.3f)" % (np.mean(cc), np.min(cc)))
ema = costs[0].copy(); ce = []
for i in range(1, len(costs)):
    ce.append(np.corrcoef(ema, costs[i])[0, 1]); ema = 0.5 * ema + 0.5 * costs[i]
print("... with the running mean (1/2, 1/2) of its earlier costs: %.3f" % np.mean(ce[3:]))
print("... with the last substep's rows x sweeps of the previous launch: %.3f" % np.mean([np.corrcoef(work[i], costs[i + 1])[0, 1] for i in range(len(costs) - 1)]))
resid = costs[1:] - costs[:-1]
print("std of the launch-to-launch change: %.0f clocks (%.1f %% of the mean)" % (resid.std(), 100 * resid.std() / costs.mean()))
b.close()
