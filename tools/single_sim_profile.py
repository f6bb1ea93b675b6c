#!/usr/bin/env python3
"""Where a single cassie_sim_t's step goes (GPU box): the drop-in call, the bare one-env launch + sync through the inner
ABI, and the kernel alone (back-to-back launches timed with HIP events), with and without the 20 KB derived read-out."""
import ctypes
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cassie-mujoco-sim_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))
from cassie_amd import Batch, Model  # noqa: E402
from cassie_amd import phys as P  # noqa: E402
from cassie_amd import iotypes as T  # noqa: E402
from cassie_amd._lib import MODEL_DIR, lib  # noqa: E402

L = lib()
m = Model("cassie")
for ext in (0, 1):
    b = Batch(m, 1)
    b.set(P.F_QPOS, m.qpos_init()[None])
    L.phys_batch_enable_ext.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.phys_batch_enable_ext(b._h, ext)
    b.step(200); b.sync()
    ms = b.time_steps(1, 3000)
    t0 = time.perf_counter()
    for _ in range(3000):
        b.step(1); b.sync()
    dt = (time.perf_counter() - t0) / 3000
    print("ext=%d: kernel alone (back to back, events) %.1f us/step; launch + sync from the host %.1f us/step" % (ext, 1e3 * ms, 1e6 * dt))
    b.close()
VP = ctypes.c_void_p
L.cassie_sim_init.restype = VP
L.cassie_sim_init.argtypes = [ctypes.c_char_p, ctypes.c_bool]
L.cassie_sim_step_pd.argtypes = [VP, VP, VP]
L.cassie_sim_free.argtypes = [VP]
c = L.cassie_sim_init(os.path.join(MODEL_DIR, "cassie.cmodel").encode(), False)
u, y = T.pd_in_t(), T.state_out_t()
for leg in (u.leftLeg, u.rightLeg):
    for i, (kp, kd, pt) in enumerate(zip([70, 70, 100, 100, 50], [7, 7, 8, 8, 5], [0.0045, 0, 0.4973, -1.1997, -1.5968])):
        leg.motorPd.pGain[i], leg.motorPd.dGain[i], leg.motorPd.pTarget[i] = kp, kd, pt
for _ in range(500):
    L.cassie_sim_step_pd(c, ctypes.byref(y), ctypes.byref(u))
t0 = time.perf_counter()
for _ in range(5000):
    L.cassie_sim_step_pd(c, ctypes.byref(y), ctypes.byref(u))
dt = (time.perf_counter() - t0) / 5000
print("cassie_sim_step_pd: %.1f us/step = %.0f steps/s" % (1e6 * dt, 1 / dt))
L.cassie_sim_free(c)
