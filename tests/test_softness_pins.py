"""The softness constants of the oracle's constraint rows (oracle/cassie_oracle.c: impedance, reference acceleration, regulariser,
diagonal approximation) against the CLOSED FORMS of MuJoCo's documentation, on one-row models where everything can be written
down by hand -- the part of the restatement that the KKT / Newton / energy pins of tests/test_oracle_pins.py cannot see, because
those hold for ANY (K, B, R).  Independent of the oracle's code: the formulas below are MuJoCo's documented ones
(Computation > Solver parameters): with solimp = (d0, dmax, width, midpoint, power) and solref = (timeconst, dampratio),

    d(r)   = d0 + y(|r| / width) (dmax - d0),  y(x) = x^p / mid^(p-1) for x <= mid, 1 - (1-x)^p / (1-mid)^(p-1) above, 1 for x >= 1
    b      = 2 / (dmax timeconst),   k = d(r) / (dmax^2 timeconst^2 dampratio^2),   aref = -b v - k r
    R      = (1 - d) / d  x  A^,     A^ = sum of the two bodies' translational inverse weights (contact, connect) or the dof's (limit),
                                      inverse weight = 1/3 tr(J M^-1 J^T) at qpos0
    f      = max(0, (aref - a0) / (A + R))  for a unilateral row (no clamp for an equality),   a1 = a0 + A f

so that on a model with A^ = A the constrained acceleration is MuJoCo's interpolation  a1 = (1 - d) a0 + d aref.
The fixtures are this repository's own MJCF files (tests/golden/onedof/): a ball on a plane (contact row, default solref 0.02 1 /
solimp 0.9 0.95 0.001 0.5 2, as every contact of model/cassie.xml), a mass on a `connect` with model/cassie.xml:18's equality solref
0.005 1, a hinge at its range limit."""
import os

import numpy as np
import pytest

from cassie_amd import Model
from oracle_py import Oracle

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "onedof")
SOLIMP = (0.9, 0.95, 0.001, 0.5, 2.0)
G = 9.81


def impedance(r, solimp=SOLIMP):
    d0, dmax, width, mid, p = solimp
    x = abs(r) / width
    if x >= 1:
        return dmax
    y = x ** p / mid ** (p - 1) if x <= mid else 1 - (1 - x) ** p / (1 - mid) ** (p - 1)
    return d0 + y * (dmax - d0)


def row(r, v, a0, A, Ahat, solref, unilateral, solimp=SOLIMP):
    """-> (constrained row acceleration a1, force f) of one constraint row by the documented closed forms."""
    tc, dr = solref
    d, dmax = impedance(r, solimp), solimp[1]
    b, k = 2 / (dmax * tc), d / (dmax ** 2 * tc ** 2 * dr ** 2)
    aref = -b * v - k * r
    R = (1 - d) / d * Ahat
    f = (aref - a0) / (A + R)
    if unilateral:
        f = max(0.0, f)
    return a0 + A * f, f


def test_impedance_sigmoid_known_values():
    assert impedance(0.0) == 0.9 and impedance(-0.002) == 0.95 and impedance(0.001) == 0.95
    assert abs(impedance(0.0005) - 0.925) < 1e-15                  # midpoint: halfway
    assert abs(impedance(0.00025) - (0.9 + 0.125 * 0.05)) < 1e-15   # x = 1/4: y = x^2 / mid = 1/8


@pytest.mark.parametrize("fixture,Ahat_over_A", [("ball_on_plane", 1.0), ("ball_on_rail", 1.0 / 3.0)])
def test_contact_row_against_the_documented_closed_form(fixture, Ahat_over_A):
    m = Model(os.path.join(HERE, fixture + ".xml"))
    pod = m.pod
    mass, radius, iz = 2.0, 0.1, pod.nv - 1
    assert abs(pod.body_invweight0[1][0] - Ahat_over_A / mass) < 1e-15       # the inverse weight IS 1/3 tr(J M^-1 J^T)
    worst = 0.0
    for pen in (1e-5, 2.5e-4, 5e-4, 7.5e-4, 1e-3, 3e-3, 2e-2):
        for v in (0.0, -0.3, 0.2, -2.0):
            o = Oracle(pod)
            o.qpos[:] = 0
            o.qpos[iz] = radius - pen
            o.qvel[:] = 0
            o.qvel[iz] = v
            o.forward()
            a1, f = row(-pen, v, -G, 1 / mass, Ahat_over_A / mass, (0.02, 1.0), True)
            assert o.d.ncon == 1 and o.d.nefc == 1
            err = abs(o.qacc[iz] - a1) / max(1.0, abs(a1))
            assert err < 1e-9, (fixture, pen, v, float(o.qacc[iz]), a1)
            assert abs(o.d.efc_force[0] - f) < 1e-9 * max(1.0, abs(f))
            worst = max(worst, err)
            if Ahat_over_A == 1.0 and f > 0:       # MuJoCo's interpolation between the free and the reference acceleration
                d = impedance(pen)
                tc, dmax = 0.02, 0.95
                aref = -2 / (dmax * tc) * v - d / (dmax * tc) ** 2 * (-pen)
                assert abs(o.qacc[iz] - ((1 - d) * -G + d * aref)) < 1e-9 * max(1.0, abs(aref))
    # a ball lifted off the plane is in free fall; one that separates fast enough gets no force either (f >= 0)
    o = Oracle(pod)
    o.qpos[:] = 0; o.qpos[iz] = radius + 1e-4
    o.forward()
    assert o.d.ncon == 0 and abs(o.qacc[iz] + G) < 1e-12
    o = Oracle(pod)
    o.qpos[:] = 0; o.qpos[iz] = radius - 1e-5; o.qvel[iz] = 5.0
    o.forward()
    assert o.d.ncon == 1 and o.d.efc_force[0] == 0.0 and abs(o.qacc[iz] + G) < 1e-12


def test_ball_comes_to_rest_at_the_documented_penetration():
    """At rest a1 = 0 and v = 0: d k |r| = (1 - d) g, i.e. |r| = (1 - d(r)) g dmax^2 timeconst^2 / d(r)^2 -- a fixed point inside the
    impedance's transition width (1.96e-4 m at d = dmax would lie below the 1 mm width)."""
    m = Model(os.path.join(HERE, "ball_on_plane.xml"))
    r = 2e-4
    for _ in range(200):
        d = impedance(r)
        r = (1 - d) * G * (0.95 * 0.02) ** 2 / d ** 2
    assert 3e-4 < r < 4e-4
    o = Oracle(m.pod)
    o.qpos[:] = 0; o.qpos[2] = 0.1
    o.step(4000)                    # 2 s: dampratio 1, timeconst 20 ms
    assert abs(o.qvel[2]) < 1e-9 and abs((0.1 - o.qpos[2]) - r) < 1e-9, (0.1 - o.qpos[2], r)


def test_connect_row_with_cassies_equality_solref():
    """model/cassie.xml:18 gives the four loop closures solref 0.005 1: a displaced mass is pulled back with a1 = d aref along the
    residual (a0 = 0 without gravity), no clamp; timeconst 5 ms >= 2 dt, so refsafe does not bite."""
    m = Model(os.path.join(HERE, "mass_on_connect.xml"))
    pod = m.pod
    assert abs(pod.body_invweight0[2][0] - 1 / 3.0) < 1e-15 and pod.body_invweight0[1][0] == 0.0
    for rx in (1e-5, 4e-4, 2e-3, -7e-4):
        for v in (0.0, 0.05, -0.4):
            o = Oracle(pod)
            o.qpos[:] = [rx, 0, 0]
            o.qvel[:] = [v, 0, 0]
            o.forward()
            assert o.d.nefc == 3
            a1, f = row(rx, v, 0.0, 1 / 3.0, 1 / 3.0, (0.005, 1.0), False)
            assert abs(o.qacc[0] - a1) < 1e-9 * max(1.0, abs(a1)), (rx, v, float(o.qacc[0]), a1)
            assert abs(o.qacc[1]) < 1e-12 and abs(o.qacc[2]) < 1e-12
    # the decay is the reference's critically damped one slowed by the softness: after 50 ms (10 time constants) the tie is closed
    o = Oracle(pod)
    o.qpos[:] = [2e-3, -1e-3, 5e-4]
    o.step(100)
    assert np.max(np.abs(o.qpos)) < 2e-3 * 1e-2


def test_joint_limit_row():
    """A hinge past its range: the limit row's A^ is the dof's inverse weight 1 / (I + armature) = J M^-1 J^T, so again
    a1 = (1 - d) a0 + d aref on the violated side, and nothing on the other side or inside the range (margin 0)."""
    m = Model(os.path.join(HERE, "hinge_with_limit.xml"))
    pod = m.pod
    I = 0.03 + 1.5 * 0.2 ** 2 + 0.01
    assert abs(pod.dof_invweight0[0] - 1 / I) < 1e-12
    for q, side in ((-0.5 - 3e-4, 1), (-0.5 - 5e-3, 1), (0.7 + 6e-4, -1), (0.7 + 2e-2, -1)):
        for v in (0.0, 0.5, -0.5):
            o = Oracle(pod)
            o.qpos[0], o.qvel[0] = q, v
            o.forward()
            dist = (q + 0.5) if side == 1 else (0.7 - q)          # signed distance to the limit, negative when violated
            a_row, f = row(dist, side * v, 0.0, 1 / I, 1 / I, (0.02, 1.0), True)
            assert o.d.nefc == 1
            assert abs(o.qacc[0] - side * a_row) < 1e-9 * max(1.0, abs(a_row)), (q, v, float(o.qacc[0]), side * a_row)
    o = Oracle(pod)
    o.qpos[0] = 0.3; o.qvel[0] = 1.0
    o.forward()
    assert o.d.nefc == 0 and o.qacc[0] == 0.0
