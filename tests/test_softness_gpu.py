"""The documented closed forms of tests/test_softness_pins.py on the DEVICE: the one-row fixtures run through the run-time-topology
instantiation of the step kernel (phys_batch_forward / phys_batch_step through the C ABI), so the kernel's impedance / reference
acceleration / regulariser arithmetic is held to MuJoCo's documentation directly, not only to the oracle."""
import os

import numpy as np
import pytest

from cassie_amd import Batch, Model
from cassie_amd import phys as P
from test_softness_pins import G, HERE, impedance, row

pytestmark = pytest.mark.gpu


def test_contact_connect_and_limit_rows_on_the_device(built):
    # ball on a plane: a sweep of penetrations x velocities, one env each
    m = Model(os.path.join(HERE, "ball_on_plane.xml"))
    cases = [(pen, v) for pen in (1e-5, 2.5e-4, 5e-4, 7.5e-4, 1e-3, 3e-3, 2e-2) for v in (0.0, -0.3, 0.2, -2.0)]
    b = Batch(m, len(cases))
    q = np.zeros((len(cases), 3)); qv = np.zeros((len(cases), 3))
    for i, (pen, v) in enumerate(cases):
        q[i, 2], qv[i, 2] = 0.1 - pen, v
    b.set(P.F_QPOS, q); b.set(P.F_QVEL, qv)
    b.forward()
    qacc = b.get(P.F_QACC)
    for i, (pen, v) in enumerate(cases):
        a1, _ = row(-pen, v, -G, 0.5, 0.5, (0.02, 1.0), True)
        assert abs(qacc[i, 2] - a1) < 1e-9 * max(1.0, abs(a1)), (pen, v, qacc[i, 2], a1)
    # ... and the rest penetration after 2 s of stepping
    b.set(P.F_QPOS, np.tile([0, 0, 0.1], (len(cases), 1))); b.set(P.F_QVEL, np.zeros((len(cases), 3)))
    for _ in range(40):
        b.step(100)
    r = 2e-4
    for _ in range(200):
        d = impedance(r)
        r = (1 - d) * G * (0.95 * 0.02) ** 2 / d ** 2
    z = b.get(P.F_QPOS)[:, 2]
    assert np.max(np.abs((0.1 - z) - r)) < 1e-9
    b.close()
    # mass on a connect (model/cassie.xml:18's equality solref)
    m = Model(os.path.join(HERE, "mass_on_connect.xml"))
    cases = [(rx, v) for rx in (1e-5, 4e-4, 2e-3, -7e-4) for v in (0.0, 0.05, -0.4)]
    b = Batch(m, len(cases))
    q = np.zeros((len(cases), 3)); qv = np.zeros((len(cases), 3))
    for i, (rx, v) in enumerate(cases):
        q[i, 0], qv[i, 0] = rx, v
    b.set(P.F_QPOS, q); b.set(P.F_QVEL, qv)
    b.forward()
    qacc = b.get(P.F_QACC)
    for i, (rx, v) in enumerate(cases):
        a1, _ = row(rx, v, 0.0, 1 / 3.0, 1 / 3.0, (0.005, 1.0), False)
        assert abs(qacc[i, 0] - a1) < 1e-9 * max(1.0, abs(a1)) and abs(qacc[i, 1]) < 1e-12, (rx, v, qacc[i].tolist(), a1)
    b.close()
    # hinge at its range limit
    m = Model(os.path.join(HERE, "hinge_with_limit.xml"))
    I = 0.03 + 1.5 * 0.2 ** 2 + 0.01
    cases = [(q_, side, v) for q_, side in ((-0.5 - 3e-4, 1), (-0.5 - 5e-3, 1), (0.7 + 6e-4, -1), (0.7 + 2e-2, -1)) for v in (0.0, 0.5, -0.5)]
    b = Batch(m, len(cases))
    b.set(P.F_QPOS, np.array([[c[0]] for c in cases])); b.set(P.F_QVEL, np.array([[c[2]] for c in cases]))
    b.forward()
    qacc = b.get(P.F_QACC)
    for i, (q_, side, v) in enumerate(cases):
        dist = (q_ + 0.5) if side == 1 else (0.7 - q_)
        a_row, _ = row(dist, side * v, 0.0, 1 / I, 1 / I, (0.02, 1.0), True)
        assert abs(qacc[i, 0] - side * a_row) < 1e-9 * max(1.0, abs(a_row)), (q_, v, qacc[i, 0], side * a_row)
    w, _ = b.warnings()
    assert not w.any()
    b.close()
