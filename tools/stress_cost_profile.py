"""Where a stress-workload launch spends its time: per-env cost (shader clocks between an env's first and last instruction, summed over
its chunks) against the launch's wall time, by form of the fast kernel.  Run on a GPU box: python tools/stress_cost_profile.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cassie-mujoco-sim_amd"))
import bench
from cassie_amd import Model, Batch, phys as P
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import golden_physics as G

bench.TARGET_SPREAD = float(os.environ.get("SPREAD", "10"))
n, npol = 4096, 40
for name in ("cassie", "cassie_hfield"):
    model = Model(name)
    hf = G.terrain(name)
    tg = bench.pd_targets(np.arange(n), npol)
    q0 = np.tile(model.qpos_init(), (n, 1))
    if name == "cassie_hfield":
        for e in range(n):
            q0[e, 0], q0[e, 1] = G.start_xy(name, e)
    for form, chunks in ((0, 1), (1, 1), (0, 4), (1, 4)):
        b = Batch(model, n)
        b.set_inplace(form); b.set_chunks(chunks)
        if hf is not None:
            b.set_hfield(hf)
        b.set(P.F_QPOS, q0); b.set(P.F_PD_KP, np.tile(bench.PD_KP, (n, 1))); b.set(P.F_PD_KD, np.tile(bench.PD_KD, (n, 1)))
        b.forward(); b.set_drive_mode(P.DRIVE_PD_SAFE)
        wall, rows = [], []
        for p in range(npol):
            b.set(P.F_PD_PTARGET, tg[p]); b.sync()
            t0 = time.perf_counter(); b.step(bench.HOLD); b.sync(); t1 = time.perf_counter()
            if p >= 20:
                hz = b.measured_shader_clock() or 2.4e9
                c = np.sort(b.launch_cost()) / hz * 1e3
                prog = b.fast_rows_progress()
                rows.append((1e3 * (t1 - t0), c.mean(), c[n // 2], c[int(n * 0.99)], c[-8], c[-1], c.sum() / 1024, int(np.count_nonzero(prog < bench.HOLD))))
        r = np.median(np.array(rows), axis=0)
        print("%-14s form %s chunks %d: launch %.2f ms | env cost ms: mean %.2f median %.2f p99 %.2f 8th-longest %.2f longest %.2f | sum/1024 slots %.2f ms | handed over %d"
              % (name, "in-place" if form else "plain   ", chunks, *r[:7], int(r[7])), flush=True)
        b.close()
