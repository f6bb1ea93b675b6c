"""Known answers that pin the oracle's physics WITHOUT MuJoCo.

None of these compares the oracle with itself: each checks a stage against independent mathematics -- dense linear
algebra in numpy, the optimality conditions of the constraint problem, an independent QP solve, Newton's second law for
the whole robot, conservation of energy.  They cannot prove parity with MuJoCo's constants (impedance, regulariser:
SURVEY.md App. B marks those as least certain) but they do prove that the restated pipeline solves the problem it
states, which is what the HIP kernel is then held to."""
import ctypes

import numpy as np
import pytest
from scipy.optimize import minimize

from cassie_amd._lib import CmModel
from oracle_py import Oracle, arr

G = 9.81
MASS = 33.312


def _standing(cassie, steps=600, pod=None, kp_scale=1.0):
    """An oracle env held by joint PD at the nominal pose until it stands on the floor."""
    import bench
    pod = pod or cassie.pod
    o = Oracle(pod, cassie.qpos_init())
    for _ in range(steps):
        o.pd_ctrl(bench.PD_OFFSET, kp_scale * bench.PD_KP, bench.PD_KD)
        o.step()
    return o


def _rows(o):
    d = o.d
    n = d.nefc
    J = arr(d.efc_J)[:n, : o.nv].copy()
    return (n, J, arr(d.efc_R)[:n].copy(), arr(d.efc_b)[:n].copy(), arr(d.efc_force)[:n].copy(),
            arr(d.efc_AR)[:n, :n].copy(), np.array(d.efc_type[:n]))


def test_projected_constraint_matrix_is_J_Minv_JT_plus_R(cassie):
    """P9: efc_AR against dense numpy J M^-1 J^T + diag(R) built from the oracle's own M and J (the oracle gets there
    through the tree-sparse L^T D L factor and per-row solves)."""
    o = _standing(cassie, 300)
    n, J, R, b, f, AR, typ = _rows(o)
    assert n >= 20
    M = o.qM
    A = J @ np.linalg.solve(M, J.T) + np.diag(R)
    assert np.max(np.abs(AR - A)) < 1e-9 * np.max(np.abs(A))
    assert np.allclose(AR, AR.T, rtol=0, atol=1e-10 * np.max(np.abs(A)))
    assert np.all(np.linalg.eigvalsh(A) > 0)                                   # regularised: strictly positive definite
    # b = J qacc_smooth - aref with qacc_smooth = M^-1 qfrc_smooth
    qs = np.linalg.solve(M, arr(o.d.qfrc_smooth)[: o.nv])
    assert np.allclose(b, J @ qs - arr(o.d.efc_aref)[:n], rtol=1e-9, atol=1e-9)
    # and the step's qacc is the smooth one plus the constraint forces' effect
    assert np.allclose(o.qacc, qs + np.linalg.solve(M, J.T @ f), rtol=1e-9, atol=1e-8)


def _tight(pod, iterations=4000, tolerance=1e-18):
    p = CmModel.from_buffer_copy(pod)
    p.iterations, p.tolerance = iterations, tolerance
    return p


def test_pgs_solution_satisfies_the_kkt_conditions(cassie):
    """P10: run to convergence, the PGS output must satisfy the optimality conditions of
    min 1/2 f^T A f + b^T f  s.t.  f_i >= 0 on limit / contact rows:  equality rows have zero residual; inequality rows
    have f >= 0, A f + b >= 0 and f (A f + b) = 0."""
    o = _standing(cassie, 400)
    tight = _tight(cassie.pod)
    t = Oracle(tight, o.qpos.copy())
    t.qvel[:] = o.qvel
    t.qacc_warmstart[:] = o.qacc_warmstart
    t.ctrl[:] = o.ctrl
    t.forward()
    n, J, R, b, f, A, typ = _rows(t)
    assert n > 12 and np.any(typ != 0)
    res = A @ f + b
    scale = np.max(np.abs(b))
    eq, ineq = typ == 0, typ != 0
    assert np.max(np.abs(res[eq])) < 1e-9 * scale
    assert np.all(f[ineq] >= 0)
    assert np.all(res[ineq] > -1e-9 * scale)
    assert np.max(np.abs(f[ineq] * res[ineq])) < 1e-9 * scale * max(1.0, np.max(f))
    # the model's own 50-sweep / 1e-8 setting lands close to that optimum
    o.forward()
    f50 = arr(o.d.efc_force)[:n]
    cost = lambda x: 0.5 * x @ A @ x + b @ x
    assert cost(f50) - cost(f) < 1e-3 * abs(cost(f))


def test_pgs_solution_against_an_independent_qp_solver(cassie):
    """The same problem handed to scipy's L-BFGS-B (bounds on the inequality rows): same minimiser."""
    o = _standing(cassie, 400)
    t = Oracle(_tight(cassie.pod), o.qpos.copy())
    t.qvel[:] = o.qvel
    t.ctrl[:] = o.ctrl
    t.forward()
    n, J, R, b, f, A, typ = _rows(t)
    s = 1.0 / np.sqrt(np.diag(A))                                            # Jacobi scaling: the rows span 6 decades
    As, bs = A * np.outer(s, s), b * s
    bounds = [(None, None) if ty == 0 else (0.0, None) for ty in typ]
    r = minimize(lambda x: 0.5 * x @ As @ x + bs @ x, np.zeros(n), jac=lambda x: As @ x + bs, method="L-BFGS-B", bounds=bounds,
                 options=dict(maxiter=20000, maxfun=200000, ftol=1e-16, gtol=1e-12))
    fq = r.x * s
    cost = lambda x: 0.5 * x @ A @ x + b @ x
    assert abs(cost(fq) - cost(f)) < 1e-7 * abs(cost(f))
    assert np.max(np.abs(J.T @ (fq - f))) < 1e-4 * np.max(np.abs(J.T @ f))     # same joint-space constraint force


def _contact_world_forces(o):
    """Per contact: world-frame force on geom2's body, decoded from the pyramid rows like mj_contactForce."""
    d = o.d
    out = []
    for c in range(d.ncon):
        con = d.contact[c]
        a = con.efc_address
        fr = np.array(con.frame).reshape(3, 3)
        if con.dim == 1:
            fc = np.array([d.efc_force[a], 0, 0])
        else:
            e = np.array([d.efc_force[a + i] for i in range(4)])
            mu = con.friction[0]
            fc = np.array([e.sum(), mu * (e[0] - e[1]), mu * (e[2] - e[3])])
        out.append(fr.T @ fc)
    return np.array(out).reshape(-1, 3)


def test_newtons_second_law_for_the_whole_robot(cassie):
    """Sum of the contact forces = m (a_com - g): the contact forces come out of the solver, the centre-of-mass
    acceleration from finite differences of kinematics alone (xipos, body masses) across steps -- two independent
    routes through the pipeline.  Then at rest (joints made stiff by damping so the posture holds): the feet carry the
    weight, 33.312 kg * 9.81 = 326.8 N."""
    pod = cassie.pod
    mass = np.array(pod.body_mass[: pod.nbody])
    assert abs(mass.sum() - MASS) < 1e-9
    h = pod.timestep

    def com(o):
        o.forward()
        return (mass[:, None] * arr(o.d.xipos)[: pod.nbody]).sum(0) / mass.sum()

    o = _standing(cassie, 500)
    import bench
    # three consecutive states around step t: a_com(t) ~ (c(t+1) - 2 c(t) + c(t-1)) / h^2, forces of the step in the middle
    c0 = com(o)
    o.pd_ctrl(bench.PD_OFFSET, bench.PD_KP, bench.PD_KD); o.step()
    c1 = com(o)
    o.pd_ctrl(bench.PD_OFFSET, bench.PD_KP, bench.PD_KD); o.step()
    F = _contact_world_forces(o).sum(0)            # forces computed by the step that took c1 -> c2
    c2 = com(o)
    a = (c2 - 2 * c1 + c0) / h ** 2
    # semi-implicit Euler: the velocity change over the step from c1 is h * a(t1); central differences see a(t1) to O(h)
    assert np.allclose(F, MASS * (a + np.array([0, 0, G])), atol=0.03 * MASS * G)
    assert F[2] > 0.5 * MASS * G

    # rest: damp every joint hard, let it settle, then the floor reaction is the weight
    stiff = CmModel.from_buffer_copy(pod)
    for k in range(6, stiff.nv):
        stiff.dof_damping[k] = 200.0
    s = _standing(cassie, 1500, pod=stiff)
    s.pd_ctrl(bench.PD_OFFSET, bench.PD_KP, bench.PD_KD)
    s.step()
    Fz = _contact_world_forces(s)[:, 2].sum()
    assert abs(s.qvel[2]) < 0.05                   # (sagging / tipping slowly at most: no balance controller)
    assert abs(Fz - MASS * G) < 0.01 * MASS * G, Fz


def test_energy_is_conserved_without_dissipation(cassie):
    """P2/P6/P12: free flight (no contacts), no joint damping, no actuation, loop-closing connects switched off (they
    are soft, i.e. dissipative): kinetic + gravitational + spring energy drifts only by the integrator's O(h)."""
    p = CmModel.from_buffer_copy(cassie.pod)
    for k in range(p.nv):
        p.dof_damping[k] = 0
    for e in range(p.neq):
        p.eq_active[e] = 0
    for j in range(p.njnt):                        # no joint limits in the way either
        p.jnt_limited[j] = 0
    q = cassie.qpos_init()
    q[2] = 8.0
    o = Oracle(p, q)
    rng = np.random.default_rng(2)
    o.qvel[:] = rng.uniform(-1.5, 1.5, p.nv)
    mass = np.array(p.body_mass[: p.nbody])

    def energy():
        o.forward()
        v = o.qvel.copy()
        ke = 0.5 * v @ o.qM @ v
        pe = G * (mass * arr(o.d.xipos)[: p.nbody, 2]).sum()
        se = sum(0.5 * p.jnt_stiffness[j] * (o.qpos[p.jnt_qposadr[j]] - p.qpos_spring[p.jnt_qposadr[j]]) ** 2
                 for j in range(p.njnt) if p.jnt_stiffness[j] > 0)
        return ke + pe + se, ke
    e0, ke0 = energy()
    worst = 0.0
    for _ in range(10):
        o.step(40)
        e, ke = energy()
        assert o.d.ncon == 0 and o.d.nefc == 0
        worst = max(worst, abs(e - e0))
    assert ke0 > 5.0
    assert worst < 0.02 * ke0, (worst, ke0)
    # halving the step halves the drift (first-order integrator, nothing else leaks)
    p2 = CmModel.from_buffer_copy(p)
    p2.timestep = p.timestep / 2
    o = Oracle(p2, q)
    o.qvel[:] = np.random.default_rng(2).uniform(-1.5, 1.5, p.nv)
    e0b, _ = energy()
    o.step(800)
    eb, _ = energy()
    assert abs(eb - e0b) < 0.65 * worst + 1e-9


def test_mass_matrix_against_kinetic_energy_of_the_bodies(cassie):
    """P2: 1/2 v^T M v equals the sum of the bodies' kinetic energies computed from their own velocities and inertias
    (armature aside) -- CRBA against first principles."""
    p = cassie.pod
    rng = np.random.default_rng(3)
    o = Oracle(p, cassie.qpos_init())
    o.qpos[7:] += 0.1 * rng.standard_normal(p.nq - 7)
    o.qvel[:] = rng.uniform(-1, 1, p.nv)
    o.forward()
    v = o.qvel.copy()
    arm = np.array(p.dof_armature[: p.nv])
    ke_M = 0.5 * v @ o.qM @ v - 0.5 * (arm * v * v).sum()
    cvel = arr(o.d.cvel)[: p.nbody]
    com = arr(o.d.subtree_com)[1]
    xipos = arr(o.d.xipos)[: p.nbody]
    ximat = arr(o.d.ximat)[: p.nbody].reshape(-1, 3, 3)
    ke = 0.0
    for b in range(1, p.nbody):
        w = cvel[b, :3]
        vl = cvel[b, 3:] + np.cross(w, xipos[b] - com)
        I = ximat[b] @ np.diag(p.body_inertia[b][:3]) @ ximat[b].T
        ke += 0.5 * p.body_mass[b] * vl @ vl + 0.5 * w @ I @ w
    assert abs(ke - ke_M) < 1e-10 * max(1.0, ke)


def test_yaw_equivariance_of_a_step_on_the_flat_floor(cassie):
    """Physics on a horizontal floor does not care about heading: turn the whole robot (pose and base velocity) about the
    vertical axis by psi and one step later everything is the un-turned result turned by psi -- joint angles identical,
    pelvis position rotated, contact forces rotated.  Checks the frame conventions of kinematics, Jacobians, contact
    frames and the quaternion integration against each other."""
    pod = _tight(cassie.pod)                     # to convergence: Gauss-Seidel iterates depend on the basis the loop-closure
    rng = np.random.default_rng(6)               # rows are written in (world x, y, z), the solution does not
    base = _standing(cassie, 300)
    q0, v0 = base.qpos.copy(), base.qvel.copy()
    v0 += 0.05 * rng.standard_normal(pod.nv)
    ctrl = base.ctrl.copy()
    psi = 0.7
    c, s = np.cos(psi), np.sin(psi)
    Rz = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])

    def quat_mul(a, b):
        return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                         a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])
    qz = np.array([np.cos(psi / 2), 0, 0, np.sin(psi / 2)])
    q1, v1 = q0.copy(), v0.copy()
    q1[0:3] = Rz @ q0[0:3]
    q1[3:7] = quat_mul(qz, q0[3:7])
    v1[0:3] = Rz @ v0[0:3]                       # the pelvis slides are world axes; its angular velocity is body-fixed
    outs = []
    for q, v in ((q0, v0), (q1, v1)):
        o = Oracle(pod, q)
        o.qvel[:] = v
        o.ctrl[:] = ctrl
        o.qacc_warmstart[:] = 0
        o.step()
        outs.append((o.qpos.copy(), o.qvel.copy(), _contact_world_forces(o).sum(0), o.d.ncon, o.sensordata.copy()))
    (qa, va, fa, na, sa), (qb, vb, fb, nb, sb) = outs
    assert na == nb and na >= 2
    assert np.allclose(qb[7:], qa[7:], atol=1e-11) and np.allclose(vb[6:], va[6:], atol=1e-8)      # joints do not notice
    assert np.allclose(qb[0:3], Rz @ qa[0:3], atol=1e-11) and np.allclose(vb[0:3], Rz @ va[0:3], atol=1e-8)
    assert np.allclose(vb[3:6], va[3:6], atol=1e-8)
    assert np.allclose(qb[3:7], quat_mul(qz, qa[3:7]), atol=1e-11) or np.allclose(qb[3:7], -quat_mul(qz, qa[3:7]), atol=1e-11)
    assert np.allclose(fb, Rz @ fa, rtol=1e-7, atol=1e-6) and fa[2] > 100
    assert np.allclose(sb[:16], sa[:16], atol=1e-11) and np.allclose(sb[20:26], sa[20:26], atol=1e-8)   # encoders, gyro, accelerometer: body-fixed


def _perturbed(pod, q, k, eps):
    """q moved by eps along dof k (tangent space: the oracle's own position integrator with a unit velocity)."""
    import ctypes
    from oracle_py import lib
    L = lib()
    L.co_integrate_pos.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double]
    qq = np.ascontiguousarray(q, dtype=np.float64).copy()
    v = np.zeros(pod.nv)
    v[k] = 1.0
    L.co_integrate_pos(ctypes.byref(pod), qq.ctypes.data, v.ctypes.data, eps)
    return qq


def test_jacobians_against_finite_differences_of_the_kinematics(cassie):
    """P5: the point Jacobian of a body (co_jac, what every constraint row is built from) against central differences of
    forward kinematics along every dof; and the equality rows of efc_J against differences of their own residuals."""
    import ctypes
    from oracle_py import lib
    pod = cassie.pod
    rng = np.random.default_rng(12)
    q = cassie.qpos_init()
    q[7:] += 0.1 * rng.standard_normal(pod.nq - 7)
    q[3:7] = [0.9, 0.2, -0.3, 0.1]
    o = Oracle(pod, q)
    o.forward()                                       # (normalises the quaternions it uses internally)
    foot = cassie.name2id(1, "left-foot")
    point = o.xpos[foot].copy()
    jp = ((ctypes.c_double * 40) * 3)()
    jr = ((ctypes.c_double * 40) * 3)()
    lib().co_jac(ctypes.byref(pod), ctypes.byref(o.d), foot, (ctypes.c_double * 3)(*point), jp, jr)
    Jp = np.array(jp)[:, : pod.nv]
    n_eq = o.d.ne
    Jeq = arr(o.d.efc_J)[:n_eq, : pod.nv].copy()
    eps = 1e-6
    Jp_fd, Jeq_fd = np.zeros((3, pod.nv)), np.zeros((n_eq, pod.nv))
    for k in range(pod.nv):
        vals = []
        for sgn in (1, -1):
            t = Oracle(pod, _perturbed(pod, o.qpos, k, sgn * eps))
            t.forward()
            vals.append((t.xpos[foot].copy(), arr(t.d.efc_pos)[:n_eq].copy()))
        Jp_fd[:, k] = (vals[0][0] - vals[1][0]) / (2 * eps)
        Jeq_fd[:, k] = (vals[0][1] - vals[1][1]) / (2 * eps)
    assert np.max(np.abs(Jp - Jp_fd)) < 1e-8
    assert np.max(np.abs(Jeq - Jeq_fd)) < 1e-7 and np.abs(Jeq).max() > 0.1


def test_bias_forces_against_lagranges_equations(cassie):
    """P6: the RNE bias force c(q, v) (Coriolis, centrifugal, gravity) against Lagrange's equations evaluated with finite
    differences of quantities that are pinned elsewhere -- the mass matrix (test above: bodies' kinetic energy) and the
    potential energy from forward kinematics:  c_k = (dM/dt v)_k - 1/2 v^T (dM/dq_k) v + dV/dq_k.  That form holds for true
    coordinates (hinges, slides); the body-fixed angular velocities of the pelvis and of the two ball joints are
    quasi-velocities, so they are kept at zero and only hinge / slide rows are compared."""
    pod = cassie.pod
    rng = np.random.default_rng(13)
    q = cassie.qpos_init()
    q[7:] += 0.1 * rng.standard_normal(pod.nq - 7)
    q[2] = 3.0                                        # in the air
    true_dofs = [k for k in range(pod.nv) if pod.jnt_type[pod.dof_jntid[k]] in (2, 3)]      # slides and hinges
    v = np.zeros(pod.nv)
    v[true_dofs] = rng.uniform(-2, 2, len(true_dofs))
    o = Oracle(pod, q)
    o.qvel[:] = v
    o.forward()
    bias = arr(o.d.qfrc_bias)[: pod.nv].copy()
    mass = np.array(pod.body_mass[: pod.nbody])

    def M_and_V(qq):
        t = Oracle(pod, qq)
        t.forward()
        return t.qM.copy(), 9.81 * float((mass * arr(t.d.xipos)[: pod.nbody, 2]).sum())
    eps = 1e-6
    dM, dV = {}, {}
    for k in true_dofs:
        Mp, Vp = M_and_V(_perturbed(pod, o.qpos, k, eps))
        Mm, Vm = M_and_V(_perturbed(pod, o.qpos, k, -eps))
        dM[k], dV[k] = (Mp - Mm) / (2 * eps), (Vp - Vm) / (2 * eps)
    Mdot = sum(dM[j] * v[j] for j in true_dofs)
    want = np.array([(Mdot @ v)[k] - 0.5 * v @ dM[k] @ v + dV[k] for k in true_dofs])
    got = bias[true_dofs]
    assert np.max(np.abs(got - want)) < 2e-6 * max(1.0, np.max(np.abs(want))), np.max(np.abs(got - want))
    assert np.max(np.abs(want)) > 10                  # gravity and velocity terms are really there


def test_imu_sensors_against_finite_differences_of_the_imu_site(cassie):
    """P11: gyro = angular velocity of the IMU site in its own frame, accelerometer = its linear acceleration minus gravity
    in its own frame -- against first / second differences of the site's pose over consecutive steps of a robot standing
    under PD control (contacts, constraint forces and all)."""
    from scipy.spatial.transform import Rotation as Rot
    import bench
    pod = cassie.pod
    o = _standing(cassie, 250)
    o.qvel[3:6] += [0.3, -0.2, 0.25]                  # make it rock a little
    h = pod.timestep
    imu = 0                                           # the only site with frame sensors on the pelvis
    frames, readings = [], []
    for _ in range(5):
        o.pd_ctrl(bench.PD_OFFSET, bench.PD_KP, bench.PD_KD)
        o.step()                                      # sensors of this step describe the state it STARTED from
        readings.append(o.sensordata.copy())
        t = Oracle(pod, o.qpos.copy())
        t.forward()
        frames.append((arr(t.d.site_xpos)[imu].copy(), arr(t.d.site_xmat)[imu].reshape(3, 3).copy()))
    # frames[i] = pose after step i = the pose that step i + 1's sensors describe
    x0, R0 = frames[1]
    x1, R1 = frames[2]
    xm, _ = frames[0]
    gyro, acc = readings[2][20:23], readings[2][23:26]
    w_fd = Rot.from_matrix(R0.T @ R1).as_rotvec() / h                         # body-frame angular velocity over the step
    a_fd = (x1 - 2 * x0 + xm) / h ** 2                                         # world acceleration, O(h) accurate
    assert np.allclose(gyro, w_fd, atol=2e-2 * max(1.0, np.linalg.norm(w_fd)))
    assert np.allclose(acc, R0.T @ (a_fd + np.array([0, 0, 9.81])), atol=0.35)
    assert 8.5 < np.linalg.norm(acc) < 11.5                                   # standing: mostly gravity
    quat = readings[2][16:20]
    assert np.allclose(np.abs(Rot.from_matrix(R0).as_quat()[[3, 0, 1, 2]] @ quat), 1.0, atol=1e-9)   # framequat = the site's orientation


def test_quaternion_integration_of_ball_and_free_joints(cassie):
    """P12: positions of ball / free joints advance by the exponential of the body-frame angular velocity."""
    from scipy.spatial.transform import Rotation as Rot
    pod = cassie.pod
    rng = np.random.default_rng(14)
    q = cassie.qpos_init()
    q[3:7] = Rot.random(random_state=1).as_quat()[[3, 0, 1, 2]]
    v = rng.uniform(-3, 3, pod.nv)
    q1 = _perturbed_by(pod, q, v, 0.01)
    Rq0, Rq1 = Rot.from_quat(q[[4, 5, 6, 3]]), Rot.from_quat(q1[[4, 5, 6, 3]])
    assert np.allclose((Rq0.inv() * Rq1).as_rotvec(), 0.01 * v[3:6], atol=1e-12)        # pelvis: body-frame omega
    assert np.allclose(q1[0:3], q[0:3] + 0.01 * v[0:3], atol=1e-15)
    ja = pod.jnt_qposadr[[j for j in range(pod.njnt) if pod.jnt_type[j] == 1][0]]        # first ball joint (achilles rod)
    da = pod.jnt_dofadr[[j for j in range(pod.njnt) if pod.jnt_type[j] == 1][0]]
    B0, B1 = Rot.from_quat(q[[ja + 1, ja + 2, ja + 3, ja]]), Rot.from_quat(q1[[ja + 1, ja + 2, ja + 3, ja]])
    assert np.allclose((B0.inv() * B1).as_rotvec(), 0.01 * v[da:da + 3], atol=1e-12)
    assert abs(np.linalg.norm(q1[3:7]) - 1) < 1e-15 and abs(np.linalg.norm(q1[ja:ja + 4]) - 1) < 1e-15


def _perturbed_by(pod, q, v, dt):
    import ctypes
    from oracle_py import lib
    L = lib()
    L.co_integrate_pos.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double]
    qq = np.ascontiguousarray(q, dtype=np.float64).copy()
    vv = np.ascontiguousarray(v, dtype=np.float64)
    L.co_integrate_pos(ctypes.byref(pod), qq.ctypes.data, vv.ctypes.data, dt)
    return qq
