#!/usr/bin/env python3
"""How cassie_core_sim's law (csrc/pk_safety.h) can be OBSERVED on the closed block itself -- black-box probes of the live binary
(oracle/_ref/libref_hostpath.so = the reference's src/libagilitycassie.a behind oracle/ref_hostpath_harness.c), no source needed:

  bounds       per drive, bisect the position at which a zero user torque starts to come back non-zero
  spring       torque against violation depth fits  kp p + kq p^2  exactly (three depths determine the two gains, a fourth checks)
  damper       d torque / d velocity at depth p is  -kd min(p / 0.15, 1)
  attenuation  a user torque u comes back as u (1 - p / 0.15) minus the spring: zero from 0.15 rad on
  coupling     hip pitch + knee: both drives answer a violation of their SUM
  clamp / STO  |torque| <= the drive's torqueLimit; radio channel 8 != 1 zeroes everything
  messages     radio.channel[1..4]: 635 after a violation, 630 after a clamp, sorted, sticky until setup

Prints the recovered constants next to those csrc/pk_safety.h uses.  Needs /root/reference (build container: bash oracle/build_ref.sh)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cassie-mujoco-sim_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import core_safety_check as C  # noqa: E402

NAMES = ["hip roll", "hip yaw", "hip pitch", "knee", "foot"]


def torque(q, u=None, w=None, ch8=1.0, L=None):
    u = np.zeros(10) if u is None else u
    w = np.zeros(10) if w is None else w
    L = C.LIMITS if L is None else L
    tau, radio, _, _ = C.live(u[None], np.asarray(q)[None], w[None], L[None], np.array([ch8]))
    return tau[0], radio[0]


def bound(k, side):
    """Position of drive k beyond which a restoring torque appears (side -1: lower bound, +1: upper), by bisection."""
    lo, hi = C.NOMINAL[k], C.NOMINAL[k] + side * 3.0
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        q = C.NOMINAL.copy(); q[k] = mid
        if torque(q)[0][k] != 0.0:
            hi = mid
        else:
            lo = mid
        if abs(hi - lo) < 1e-16:
            break
    return hi


print("drive            lower bound        upper bound        kp        kq         kd     (pk_safety.h: bound(), gain_p / gain_q / gain_d)")
for k in range(10):
    lo, hi = bound(k, -1), bound(k, +1)
    # spring: torque at three depths beyond the upper bound (inside the clamp) -> kp, kq by least squares, residual of a fourth
    depths = np.array([0.002, 0.004, 0.008, 0.006])
    t = []
    for p in depths:
        q = C.NOMINAL.copy(); q[k] = hi + p
        if k % 5 == 2:                      # keep the coupled row (hip pitch + knee) out of it
            q[k + 1] = -1.2
        t.append(-torque(q)[0][k])
    A = np.stack([depths[:3], depths[:3] ** 2], axis=1)
    kp, kq = np.linalg.lstsq(A, np.array(t[:3]), rcond=None)[0]
    check = abs(kp * depths[3] + kq * depths[3] ** 2 - t[3])
    # damper at depth 0.05 (s = 1/3) from a velocity difference
    q = C.NOMINAL.copy(); q[k] = hi + 0.05
    w1 = np.zeros(10); w1[k] = 1.0
    big = C.LIMITS * 100
    kd = -(torque(q, w=w1, L=big)[0][k] - torque(q, L=big)[0][k]) / (0.05 / 0.15)
    print("%-5s %-9s  %.15f  %.15f  %8.2f  %9.3f  %6.2f   fit residual %.1e" % ("left" if k < 5 else "right", NAMES[k % 5], lo, hi, kp, kq, kd, check))
# attenuation: user torque 10 on the left knee, knee pushed past its upper bound by p
print("\nattenuation of a user torque of 10 Nm on the left knee (spring removed by subtracting the zero-torque answer), limits lifted:")
big = C.LIMITS * 100
for p in (0.0, 0.03, 0.075, 0.12, 0.15, 0.2):
    q = C.NOMINAL.copy(); q[3] = C.UPPER[3] + p
    u = np.zeros(10); u[3] = 10.0
    print("  depth %.3f: factor %.6f  (1 - depth / 0.15 = %.6f)" % (p, (torque(q, u, L=big)[0][3] - torque(q, L=big)[0][3]) / 10.0, max(0.0, 1 - p / 0.15)))
# coupled constraint
q = C.NOMINAL.copy(); q[2], q[3] = 0.2, -2.356194490192345 - 0.2 - 0.01
print("\nhip pitch + knee 0.01 rad below -3 pi / 4: torques", torque(q)[0][:5], "(both drives answer with kp p + kq p^2 = %.3f)" % (1200 * 0.01 + 8000 * 1e-4))
print("clamp: user torque 1000 everywhere ->", torque(C.NOMINAL, u=np.full(10, 1000.0))[0])
print("STO (radio channel 8 = 0) ->", torque(C.NOMINAL, u=np.full(10, -5.0), ch8=0.0)[0])
tau, radio, _, _ = C.live(np.array([np.zeros(10), np.full(10, 1000.0), np.zeros(10), np.zeros(10)]), np.array([C.NOMINAL, C.NOMINAL, np.zeros(10), C.NOMINAL]),
                          np.zeros((4, 10)), np.tile(C.LIMITS, (4, 1)), np.ones(4), fresh=False)
print("message queue over four steps of ONE block (clean, clamp, limit violation, clean):", radio[:, 1:5].tolist())
