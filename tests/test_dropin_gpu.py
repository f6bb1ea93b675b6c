"""GPU tests of the drop-in C ABI (include/cassiemujoco.h) and of the batched step (include/cassie_batch.h).

The single-simulator path is checked against a CPU re-play of the same pipeline built from the pieces that
are pinned elsewhere: the host chain (bit-exact vs the reference's code, tests/test_hostpath.py) and the
oracle physics (tests/test_model.py, tests/test_gpu_parity.py)."""
import ctypes
import os

import numpy as np
import pytest

from cassie_amd import iotypes as T
from cassie_amd._lib import REPO_DIR, lib
from oracle_py import Oracle
from test_hostpath import HostModel

pytestmark = pytest.mark.gpu
MODEL = os.path.join(REPO_DIR, "models", "cassie.cmodel").encode()
VP = ctypes.c_void_p
DP = ctypes.POINTER(ctypes.c_double)


@pytest.fixture(scope="module")
def L(built):
    L = lib()
    L.cassie_sim_init.restype = VP
    L.cassie_sim_init.argtypes = [ctypes.c_char_p, ctypes.c_bool]
    L.cassie_sim_free.argtypes = [VP]
    for f in ("cassie_sim_qpos", "cassie_sim_qvel", "cassie_sim_time", "cassie_sim_sensordata", "cassie_sim_act_vel"):
        getattr(L, f).restype = DP
        getattr(L, f).argtypes = [VP]
    L.cassie_sim_step.argtypes = [VP, VP, VP]
    L.cassie_sim_step_pd.argtypes = [VP, VP, VP]
    L.cassie_sim_foot_forces.argtypes = [VP, VP]
    L.cassie_sim_heeltoe_forces.argtypes = [VP, VP, VP]
    L.cassie_sim_cm_position.argtypes = [VP, VP]
    L.cassie_sim_full_mass_matrix.argtypes = [VP, VP]
    L.cassie_sim_get_jacobian.argtypes = [VP, VP, ctypes.c_char_p]
    L.cassie_sim_foot_positions.argtypes = [VP, VP]
    L.cassie_state_alloc.restype = VP
    L.cassie_get_state.argtypes = [VP, VP]
    L.cassie_set_state.argtypes = [VP, VP]
    L.cassie_state_free.argtypes = [VP]
    L.cassie_batch_create.restype = VP
    L.cassie_batch_create.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.cassie_batch_free.argtypes = [VP]
    L.cassie_batch_step_pd.argtypes = [VP, VP, VP]
    L.cassie_batch_foot_forces.argtypes = [VP, VP]
    L.cassie_batch_phys.restype = VP
    L.cassie_batch_phys.argtypes = [VP]
    L.cassie_hostenv_alloc.restype = VP
    L.cassie_hostenv_step.argtypes = [VP] * 7
    L.cassie_hostmodel_from_model.argtypes = [VP, VP]
    return L


def pd_input(rng, scale=0.2):
    u = T.pd_in_t()
    off = [0.0045, 0, 0.4973, -1.1997, -1.5968]
    for leg in (u.leftLeg, u.rightLeg):
        for i in range(5):
            leg.motorPd.pTarget[i] = off[i] + scale * rng.uniform(-1, 1)
            leg.motorPd.pGain[i] = [70, 70, 100, 100, 50][i]
            leg.motorPd.dGain[i] = [7, 7, 8, 8, 5][i]
    return u


def test_config1_zero_torque_cassie_sim_step_1000_steps(L, cassie):
    """BASELINE.json configs[0]: one env, zero-torque cassie_sim_step x 1000, replayed on the CPU."""
    c = L.cassie_sim_init(MODEL, False)
    assert c
    hm = HostModel()
    assert L.cassie_hostmodel_from_model(cassie._h, ctypes.byref(hm)) == 0
    env = L.cassie_hostenv_alloc()
    o = Oracle(cassie.pod, cassie.qpos_init())
    o.forward()
    u = T.cassie_user_in_t()
    worst = 0.0
    for t in range(1000):
        y = T.cassie_out_t()
        L.cassie_sim_step(c, ctypes.byref(y), ctypes.byref(u))
        # CPU replay: host chain on the oracle's previous sensordata, then one oracle step
        yr = T.cassie_out_t()
        sd, av = np.ascontiguousarray(o.sensordata), np.ascontiguousarray(o.actuator_velocity)
        ctrl = np.zeros(10)
        L.cassie_hostenv_step(env, ctypes.byref(hm), ctypes.byref(u), sd.ctypes.data, av.ctypes.data, ctrl.ctypes.data, ctypes.byref(yr))
        o.ctrl[:] = ctrl
        o.step()
        q = np.ctypeslib.as_array(L.cassie_sim_qpos(c), (35,))
        worst = max(worst, np.max(np.abs(q - o.qpos)))
        # encoder read-outs agree to one count (a 1e-13 physics difference can flip the truncation)
        assert abs(y.leftLeg.kneeDrive.position - yr.leftLeg.kneeDrive.position) <= 2 * np.pi / (1 << 13) / 16 + 1e-12
        assert abs(y.pelvis.vectorNav.linearAcceleration[2] - yr.pelvis.vectorNav.linearAcceleration[2]) < 1e-6
    assert worst < 1e-8, worst
    assert abs(L.cassie_sim_time(c)[0] - 0.5) < 1e-12
    L.cassie_sim_free(c)


def test_step_pd_stands_and_reports_sane_state(L):
    c = L.cassie_sim_init(MODEL, False)
    u = pd_input(np.random.default_rng(0), scale=0.0)
    y = T.state_out_t()
    for _ in range(600):
        L.cassie_sim_step_pd(c, ctypes.byref(y), ctypes.byref(u))
    q = np.ctypeslib.as_array(L.cassie_sim_qpos(c), (35,))
    assert 0.5 < q[2] < 1.1                          # on its feet under plain joint PD after 0.3 s (no balance controller)
    assert np.allclose(list(y.motor.position), q[[7, 8, 9, 14, 20, 21, 22, 23, 28, 34]], atol=2e-3)
    # on the ground: foot forces carry the weight, heel + toe split adds up to the foot total (reference test_heelforce.c:56-57)
    ff = np.zeros(12); toe = np.zeros(6); heel = np.zeros(6)
    L.cassie_sim_foot_forces(c, ff.ctypes.data)
    L.cassie_sim_heeltoe_forces(c, toe.ctypes.data, heel.ctypes.data)
    total = ff[2] + ff[8]
    assert 150 < total < 600
    assert np.allclose(toe[:3] + heel[:3], ff[0:3], atol=1e-9) and np.allclose(toe[3:] + heel[3:], ff[6:9], atol=1e-9)
    L.cassie_sim_free(c)


def test_derived_getters_against_oracle(L, cassie):
    c = L.cassie_sim_init(MODEL, False)
    o = Oracle(cassie.pod, cassie.qpos_init())
    o.forward()
    cm = np.zeros(3)
    L.cassie_sim_cm_position(c, cm.ctypes.data)
    from oracle_py import arr
    assert np.allclose(cm, arr(o.d.subtree_com)[1], atol=1e-12)
    M = np.zeros(1024)
    L.cassie_sim_full_mass_matrix(c, M.ctypes.data)
    M = M.reshape(32, 32)
    assert np.allclose(M, o.qM, atol=1e-12) and abs(M[0, 0] - 33.312) < 1e-9
    jac = np.zeros(3 * 32)
    L.cassie_sim_get_jacobian(c, jac.ctypes.data, b"left-foot")
    # finite-difference check of the left-foot position Jacobian on a hinge dof (left knee, dof 12)
    eps = 1e-6
    o2 = Oracle(cassie.pod, cassie.qpos_init())
    o2.qpos[14] += eps
    o2.forward()
    lf = cassie.name2id(1, "left-foot")
    fd = (o2.xpos[lf] - o.xpos[lf]) / eps
    assert np.allclose(jac.reshape(3, 32)[:, 12], fd, atol=1e-5)
    fp = np.zeros(6)
    L.cassie_sim_foot_positions(c, fp.ctypes.data)
    assert np.allclose(fp[:2], o.xpos[lf][:2], atol=1e-12)
    L.cassie_sim_free(c)


def test_get_set_state_replays_identically(L):
    c = L.cassie_sim_init(MODEL, False)
    rng = np.random.default_rng(3)
    us = [pd_input(rng) for _ in range(4)]
    y = T.state_out_t()
    for k in range(200):
        L.cassie_sim_step_pd(c, ctypes.byref(y), ctypes.byref(us[k // 50]))
    s = L.cassie_state_alloc()
    L.cassie_get_state(c, s)
    ya, yb = T.state_out_t(), T.state_out_t()
    for k in range(100):
        L.cassie_sim_step_pd(c, ctypes.byref(ya), ctypes.byref(us[2]))
    qa = np.ctypeslib.as_array(L.cassie_sim_qpos(c), (35,)).copy()
    L.cassie_set_state(c, s)
    for k in range(100):
        L.cassie_sim_step_pd(c, ctypes.byref(yb), ctypes.byref(us[2]))
    qb = np.ctypeslib.as_array(L.cassie_sim_qpos(c), (35,)).copy()
    assert np.array_equal(qa, qb) and bytes(ya) == bytes(yb)
    L.cassie_state_free(s)
    L.cassie_sim_free(c)


def test_batch_step_pd_equals_independent_simulators(L):
    n, steps = 6, 300
    rng = np.random.default_rng(7)
    b = L.cassie_batch_create(MODEL, n, 0, 3)
    assert b
    sims = [L.cassie_sim_init(MODEL, False) for _ in range(n)]
    U = (T.pd_in_t * n)()
    Y = (T.state_out_t * n)()
    for k in range(steps):
        if k % 50 == 0:
            for e in range(n):
                U[e] = pd_input(rng, scale=0.3)
        assert L.cassie_batch_step_pd(b, ctypes.byref(U), ctypes.byref(Y)) == 0
        for e in range(n):
            y = T.state_out_t()
            L.cassie_sim_step_pd(sims[e], ctypes.byref(y), ctypes.byref(U[e]))
            assert bytes(y) == bytes(Y[e]), (k, e)
        if k % 100 == 99:
            # batched derived getter: the per-body contact forces left in HBM give every env's foot forces
            ffb = np.zeros((n, 12))
            assert L.cassie_batch_foot_forces(b, ffb.ctypes.data) == 0
            for e in range(n):
                ff = np.zeros(12)
                L.cassie_sim_foot_forces(sims[e], ff.ctypes.data)
                assert np.allclose(ffb[e], ff, rtol=1e-12, atol=1e-9), (k, e)
            assert np.abs(ffb).max() > 1.0          # somebody is standing on something
    for s in sims:
        L.cassie_sim_free(s)
    L.cassie_batch_free(b)


@pytest.mark.parametrize("device_drives", [0, 1])
def test_batch_masked_full_reset_equals_cassie_sim_full_reset(L, device_drives):
    """cassie_batch_full_reset(mask) = cassie_sim_full_reset (reference src/cassiemujoco.c:2008-2033) on the masked envs
    and nothing on the others: every env's state_out_t stays byte-identical to an independent simulator that was reset
    (or not) at the same step -- with the drive-level models on the host threads and on the device."""
    L.cassie_batch_full_reset.argtypes = [VP, VP]
    L.cassie_batch_set_device_drives.argtypes = [VP, ctypes.c_int]
    L.cassie_sim_full_reset.argtypes = [VP]
    n, steps = 5, 150
    rng = np.random.default_rng(11)
    b = L.cassie_batch_create(MODEL, n, 0, 2)
    assert b and L.cassie_batch_set_device_drives(b, device_drives) == 0
    sims = [L.cassie_sim_init(MODEL, False) for _ in range(n)]
    U = (T.pd_in_t * n)()
    Y = (T.state_out_t * n)()
    for k in range(steps):
        if k % 50 == 0:
            for e in range(n):
                U[e] = pd_input(rng, scale=0.3)
        if k in (60, 61, 110):
            mask = np.array([[1, 0, 1, 1, 0], [0, 0, 0, 0, 1], [1, 1, 1, 1, 1]][(60, 61, 110).index(k)], dtype=np.uint8)
            assert L.cassie_batch_full_reset(b, mask.ctypes.data if k != 110 else None) == 0
            for e in range(n):
                if mask[e]:
                    L.cassie_sim_full_reset(sims[e])
        assert L.cassie_batch_step_pd(b, ctypes.byref(U), ctypes.byref(Y)) == 0
        for e in range(n):
            y = T.state_out_t()
            L.cassie_sim_step_pd(sims[e], ctypes.byref(y), ctypes.byref(U[e]))
            assert bytes(y) == bytes(Y[e]), (k, e)
    for s in sims:
        L.cassie_sim_free(s)
    L.cassie_batch_free(b)


# ------------------------------------------------------------------------------------------------------------------
# The reference's UNMODIFIED Python wrapper (example/cassiemujoco.py:31-72) on top of this library.  oracle/build_ref.sh
# stages the two wrapper files and the MJCF models into the git-ignored oracle/_ref/ in the build container; the
# directory travels to the GPU box with the snapshot.
REF_STAGE = os.path.join(REPO_DIR, "oracle", "_ref")

_WRAPPER_SCRIPT = r"""
import json, sys
import numpy as np
from cassiemujoco import *          # the reference's module: dlopens ./libcassiemujoco.so, cassie_mujoco_init at import
sim = CassieSim("../model/cassie.xml")
u = pd_in_t()
off = [0.0045, 0, 0.4973, -1.1997, -1.5968]
for leg in (u.leftLeg, u.rightLeg):
    for i in range(5):
        leg.motorPd.pGain[i], leg.motorPd.dGain[i], leg.motorPd.pTarget[i] = [70, 70, 100, 100, 50][i], [7, 7, 8, 8, 5][i], off[i]
traj = []
for s in range(60):
    y = sim.step_pd(u)
    traj.append(list(sim.qpos()))
print(json.dumps({"qpos": traj, "nq": sim.nq, "nv": sim.nv, "time": sim.time(), "pelvis_z_est": y.pelvis.position[2],
                  "foot_pos": list(sim.foot_pos()), "foot_forces": [float(x) for x in sim.get_foot_forces()]}))
"""


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_STAGE, "example", "cassiemujoco.py")),
                    reason="reference wrapper not staged (oracle/build_ref.sh runs where /root/reference exists)")
def test_unmodified_reference_python_wrapper_end_to_end(L, tmp_path):
    """CassieSim("../model/cassie.xml").step_pd(pd_in_t) x 60 through the reference's own cassiemujoco.py + ctypes
    binding + MJCF file, against the same 60 steps through the C ABI directly."""
    import json
    import shutil
    import subprocess
    import sys
    ex = tmp_path / "example"
    ex.mkdir()
    for f in ("cassiemujoco.py", "cassiemujoco_ctypes.py"):
        shutil.copy(os.path.join(REF_STAGE, "example", f), ex / f)
    os.symlink(os.path.join(REPO_DIR, "cassie-mujoco-sim_amd", "lib", "libcassiemujoco.so"), ex / "libcassiemujoco.so")
    os.symlink(os.path.join(REF_STAGE, "model"), tmp_path / "model")
    out = subprocess.run([sys.executable, "-c", _WRAPPER_SCRIPT], cwd=ex, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert (res["nq"], res["nv"]) == (35, 32)
    assert abs(res["time"] - 60 * 0.0005) < 1e-12

    xml = os.path.join(REF_STAGE, "model", "cassie.xml").encode()
    c = L.cassie_sim_init(xml, False)
    assert c
    u = pd_input(np.random.default_rng(0), scale=0.0)
    y = T.state_out_t()
    traj = []
    for s in range(60):
        L.cassie_sim_step_pd(c, ctypes.byref(y), ctypes.byref(u))
        traj.append(np.ctypeslib.as_array(L.cassie_sim_qpos(c), shape=(35,)).copy())
    L.cassie_sim_free(c)
    assert np.array_equal(np.array(res["qpos"]), np.array(traj))    # same library, same inputs: bit for bit
    assert abs(res["pelvis_z_est"] - y.pelvis.position[2]) < 1e-12
    assert 0.9 < traj[-1][2] < 1.02
    assert len(res["foot_pos"]) == 6 and abs(res["foot_pos"][2]) < 0.2 and all(f >= 0 for f in res["foot_forces"])


_SWEEP_SCRIPT = r"""
import json, sys, warnings
import numpy as np
warnings.simplefilter("ignore")
from cassiemujoco import *
sim = CassieSim("../model/cassie.xml")
assert (sim.nq, sim.nv, sim.nu, sim.nbody, sim.njnt) == (35, 32, 10, 26, 26), (sim.nq, sim.nv, sim.nu, sim.nbody, sim.njnt)
u = pd_in_t()
off = [0.0045, 0, 0.4973, -1.1997, -1.5968]
for leg in (u.leftLeg, u.rightLeg):
    for i in range(5):
        leg.motorPd.pGain[i], leg.motorPd.dGain[i], leg.motorPd.pTarget[i] = [100, 100, 88, 96, 50][i], [10, 10, 8, 9.6, 5][i], off[i]
def run(n):
    for _ in range(n): y = sim.step_pd(u)
    return y
fin = lambda x: bool(np.all(np.isfinite(np.asarray(x, dtype=float))))
def momentum_balance(applied=(0.0, 0.0, 0.0)):
    # Newton for the whole robot over one step: m (v_com' - v_com) / h = foot contact forces + applied force + gravity, to O(h).
    # Ties together center_of_mass_velocity, get_body_mass, the contact-force getters, apply_force and the physics itself.
    m = float(np.sum(sim.get_body_mass())); v0 = np.array(sim.center_of_mass_velocity()); sim.step_pd(u); v1 = np.array(sim.center_of_mass_velocity())
    ff = (ctypes.c_double * 12)(); cassie_sim_foot_forces(sim.c, ff); ff = np.array(ff[:])
    ext = ff[0:3] + ff[6:9] + np.array(applied) + np.array([0, 0, -9.806 * m])
    return float(np.abs(m * (v1 - v0) / 5e-4 - ext).max() / (9.806 * m)), ff
run(150)                                                  # the feet have landed; only the feet touch the floor
W = float(np.sum(sim.get_body_mass())) * 9.806
# --- state vectors and clocks
assert abs(sim.timestep() - 5e-4) < 1e-15 and abs(sim.time() - 150 * 5e-4) < 1e-9
assert len(sim.qpos()) == 35 and len(sim.qpos_full()) == 35 and len(sim.qvel()) == 32 and len(sim.qvel_full()) == 32
assert len(sim.qacc()) == 32 and len(sim.ctrl()) == 10 and fin(sim.qacc()) and fin(sim.ctrl()) and np.abs(sim.ctrl()).max() > 0.1
assert len(sim.jnt_qposadr()) == 26 and len(sim.jnt_dofadr()) == 26 and sim.jnt_qposadr()[3] == 3 and sim.jnt_dofadr()[4] == 6
# xpos / xquat are mj_step's: the pose before the last position update, i.e. within h |v| of qpos
assert np.allclose(sim.xpos("cassie-pelvis"), sim.qpos()[:3], atol=2e-3) and np.allclose(sim.xquat("cassie-pelvis"), sim.qpos()[3:7], atol=2e-3)
assert 0.8 < sim.qpos()[2] < 1.02
# --- contact / force getters
err, ff = momentum_balance(); assert err < 0.02, err
lf, rf = sim.get_foot_forces(); assert abs(lf - np.linalg.norm(ff[0:3])) < 1e-9 and abs(rf - np.linalg.norm(ff[6:9])) < 1e-9 and lf + rf > 0.2 * W
toe, heel = sim.get_heeltoe_forces(); assert np.allclose(toe[:3] + heel[:3], ff[0:3], atol=1e-9) and np.allclose(toe[3:] + heel[3:], ff[6:9], atol=1e-9)
f6 = np.zeros(6); sim.get_body_contact_force(f6, "left-foot"); assert np.allclose(f6[:3], ff[0:3], atol=1e-9) and np.all(f6[3:] == 0)
fp = sim.foot_pos(); assert len(fp) == 6 and abs(fp[2]) < 0.1 and abs(fp[5]) < 0.1
fv = np.zeros(12); sim.foot_vel(fv); assert fin(fv) and np.abs(fv).max() < 2.0
bv = np.zeros(6); sim.body_vel(bv, "cassie-pelvis"); assert fin(bv) and np.allclose(bv[:3], np.array(sim.qvel())[3:6] @ np.eye(3), atol=1.0)
ba = np.zeros(6); sim.get_body_acceleration(ba, "cassie-pelvis"); assert fin(ba)
fq = np.zeros(4); sim.foot_quat(fq); assert abs(np.linalg.norm(fq) - 1) < 1e-9
assert not sim.check_self_collision() and not sim.check_obstacle_collision() and not sim.check_collision(2)
# --- centroidal quantities
com = sim.center_of_mass_position(); assert abs(com[0] - sim.qpos()[0]) < 0.2 and 0.5 < com[2] < 1.0
assert fin(sim.center_of_mass_velocity()) and fin(sim.angular_momentum())
I = np.array(sim.centroid_inertia()).reshape(3, 3); assert np.allclose(I, I.T, atol=1e-9) and np.all(np.linalg.eigvalsh(I) > 0)
M = sim.full_mass_matrix(); assert np.allclose(M, M.T, atol=1e-10) and np.all(np.linalg.eigvalsh(M) > 0)
assert abs(M[0, 0] - W / 9.806) < 1e-9                      # translational block = total mass
v = np.array(sim.qvel()); assert abs(0.5 * v @ M @ v) < 50   # kinetic energy from the getter pair is sane
Jc = sim.constraint_jacobian(); assert Jc.shape == (6, 32) and fin(Jc) and np.abs(Jc).max() > 0.01
assert np.abs(sim.constraint_error()).max() < 5e-3          # the loop closures hold
Mm = sim.minimal_mass_matrix(); assert Mm.shape == (16, 16) and fin(Mm)
# --- Jacobians
for nm in ("left-foot", "right-foot", "cassie-pelvis"):
    jp = sim.get_jacobian(nm); jp2, jr2 = sim.get_jacobian_full(nm)
    assert jp.shape == (96,) and np.array_equal(jp, jp2) and fin(jr2)
jp, jr = sim.get_jacobian_full("cassie-pelvis")
assert np.allclose(jp.reshape(3, 32)[:, :3], np.eye(3), atol=1e-12)
js, jsr = sim.get_jacobian_full_site("left-toe"); assert fin(js) and np.abs(js).max() > 0.1
assert len(sim.get_site_xpos("left-heel")) == 3 and abs(np.linalg.norm(sim.get_site_quat("left-heel")) - 1) < 1e-9
assert abs(sim.get_site_xpos("left-heel")[2]) < 0.1
rel = np.zeros(7); sim.get_object_relative_pose([0, 0, 0, 1, 0, 0, 0], [1, 2, 3, 1, 0, 0, 0], rel); assert np.allclose(rel, [1, 2, 3, 1, 0, 0, 0])
# --- get_state / set_state replay (CassieSim.get_state needs a CassieState class the reference module does not define: use its ctypes functions)
st = cassie_state_alloc(); cassie_get_state(sim.c, st); run(30); q1 = np.array(sim.qpos()); cassie_set_state(sim.c, st); run(30)
assert np.array_equal(q1, np.array(sim.qpos())); cassie_state_free(st)
# --- external force, then a heavier pelvis: Newton's balance keeps holding, with the new terms
sim.apply_force([0, 0, 0.3 * W, 0, 0, 0]); run(10); err, ff2 = momentum_balance((0, 0, 0.3 * W)); assert err < 0.02, err
sim.clear_forces(); run(10); err, ff3 = momentum_balance(); assert err < 0.02, err
m = sim.get_body_mass(); assert m.shape == (26,) and abs(sim.get_body_mass("cassie-pelvis") - m[1]) < 1e-15
sim.set_body_mass(m[1] + 5.0, name="cassie-pelvis"); assert abs(sim.get_body_mass("cassie-pelvis") - m[1] - 5) < 1e-12
run(10); err, ff4 = momentum_balance(); assert err < 0.02, err
Mh = sim.full_mass_matrix(); assert abs(Mh[0, 0] - W / 9.806 - 5) < 1e-9
sim.set_body_mass(m); assert np.array_equal(sim.get_body_mass(), m)
# --- other step entry points
assert fin(sim.step(cassie_user_in_t()).pelvis.vectorNav.orientation[:])
assert fin(sim.step_pd_no2khz(u).pelvis.position[:]) and fin(sim.integrate_pos().pelvis.position[:])
# --- hold / release: the pelvis is pinned by stiff springs
p0 = np.array(sim.qpos()[:3]); sim.hold(); run(300); assert np.abs(np.array(sim.qpos()[:3]) - p0).max() < 0.03 and np.abs(np.array(sim.qvel())[:6]).max() < 0.2; sim.release()
# --- model parameter getters / setters (round trips)
d = sim.get_dof_damping(); assert d.shape == (32,)
assert sim.get_joint_num_dof("left-knee") == 1 and abs(sim.get_dof_damping("left-knee")[0] - d[sim.jnt_dofadr()[sim.mj_name2id("joint", "left-knee")]]) < 1e-15
sim.set_dof_damping(2.5, name="left-knee"); assert sim.get_dof_damping("left-knee")[0] == 2.5
sim.set_dof_damping(d); assert np.array_equal(sim.get_dof_damping(), d)
ip = sim.get_body_ipos(); assert ip.shape == (26, 3) and np.allclose(sim.get_body_ipos("cassie-pelvis"), ip[1])
sim.set_body_ipos(ip[1] + 0.01, name="cassie-pelvis"); assert np.allclose(sim.get_body_ipos("cassie-pelvis"), ip[1] + 0.01)
sim.set_body_ipos(ip.flatten()); assert np.allclose(sim.get_body_ipos(), ip)
bp = sim.get_body_pos("left-foot"); sim.set_body_pos("left-foot", bp + 0.001); assert np.allclose(sim.get_body_pos("left-foot"), bp + 0.001); sim.set_body_pos("left-foot", bp)
fr = sim.get_geom_friction(); assert fr.shape == (sim.ngeom, 3) and np.allclose(sim.get_geom_friction("floor"), fr[0]) and np.allclose(sim.get_geom_name_friction("floor"), fr[0])
sim.set_geom_friction([0.6, 1e-4, 5e-5], name="floor"); assert np.allclose(sim.get_geom_friction("floor"), [0.6, 1e-4, 5e-5])
sim.set_geom_friction(fr.flatten()); assert np.allclose(sim.get_geom_friction(), fr)
for get, set_, w in ((sim.get_geom_rgba, sim.set_geom_rgba, 4), (sim.get_geom_quat, sim.set_geom_quat, 4), (sim.get_geom_pos, sim.set_geom_pos, 3), (sim.get_geom_size, sim.set_geom_size, 3)):
    allv = get(); one = get("box1"); assert allv.shape == (sim.ngeom * w,) and one.shape == (w,) and fin(allv)
    k = sim.mj_name2id("geom", "box1"); assert np.allclose(allv[k * w:(k + 1) * w], one)
    set_(one, name="box1"); set_(allv); assert np.allclose(get(), allv)
sim.set_const(); sim.just_set_const()
assert sim.mj_name2id("body", "cassie-pelvis") == 1 and sim.mj_name2id("body", "no-such-body") == -1
# --- drive-level state: joint filters and the torque delay line (get_drive_filter is unusable in the reference module itself:
#     its ctypes table rebinds cassie_sim_drive_filter with a two-argument signature, cassiemujoco_ctypes.py:886-888)
td = sim.get_torque_delay(); assert td.shape == (10, 6); sim.set_torque_delay(td + 1.0); assert np.allclose(sim.get_torque_delay(), td + 1.0); sim.set_torque_delay(td)
jf = sim.get_joint_filter(); x = [jf[i].x[k] for i in range(6) for k in range(4)]; yv = [jf[i].y[k] for i in range(6) for k in range(3)]
sim.set_joint_filter(x, yv); jf2 = sim.get_joint_filter(); assert [jf2[i].x[k] for i in range(6) for k in range(4)] == x
# --- full_reset, timestep, time
sim.full_reset(); assert abs(sim.qpos()[2] - 1.01) < 1e-12 and np.all(np.array(sim.qvel()) == 0)
sim.set_time(1.5); assert sim.time() == 1.5
sim.set_qpos(list(sim.qpos())); sim.set_qvel(np.zeros(32)); sim.set_ctrl(np.zeros(10)); run(50); assert fin(sim.qpos())
del sim
# --- height-field model: terrain accessors
hs = CassieSim("../model/cassie_hfield.xml")
nr, nc, nd = hs.get_hfield_nrow(), hs.get_hfield_ncol(), hs.get_nhfielddata()
assert nr * nc == nd and nd > 0 and hs.get_hfield_size().shape == (4,)
h = np.clip(0.02 * np.add.outer(np.arange(nr) % 7, np.arange(nc) % 5) / 10.0, 0, 1).astype(np.float32).flatten()
hs.set_hfield_data(h); assert np.allclose(hs.get_hfield_data(), h)
sz = hs.get_hfield_size(); hs.set_hfield_size(sz); assert np.allclose(hs.get_hfield_size(), sz)
for _ in range(200): yy = hs.step_pd(u)
assert fin(hs.qpos()) and 0.7 < hs.qpos()[2] < 1.2
"""


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_STAGE, "example", "cassiemujoco.py")),
                    reason="reference wrapper not staged (oracle/build_ref.sh runs where /root/reference exists)")
def test_unmodified_reference_wrapper_api_sweep(L, tmp_path):
    """Nearly every CassieSim method of the reference's unmodified example/cassiemujoco.py (:31-825) against this
    library, with a physical or round-trip expectation for each: contact force getters carry the robot's weight, the
    mass matrix is symmetric positive definite with the total mass in its translational block, Jacobians, loop-closure
    error, get/set_state replay, hold / apply_force, every model-parameter getter / setter pair, the drive-level
    filter and delay-line accessors, full_reset, and the height-field accessors on cassie_hfield.xml."""
    import json
    import shutil
    import subprocess
    import sys
    ex = tmp_path / "example"
    ex.mkdir()
    for f in ("cassiemujoco.py", "cassiemujoco_ctypes.py"):
        shutil.copy(os.path.join(REF_STAGE, "example", f), ex / f)
    os.symlink(os.path.join(REPO_DIR, "cassie-mujoco-sim_amd", "lib", "libcassiemujoco.so"), ex / "libcassiemujoco.so")
    os.symlink(os.path.join(REF_STAGE, "model"), tmp_path / "model")
    # every top-level statement of the sweep runs even if an earlier one failed, so one run lists all the discrepancies
    driver = r"""
import json, sys, traceback
src = open(sys.argv[1]).read().split("\n")
chunks, cur = [], []
for line in src:
    if line and not line[0].isspace() and cur:
        chunks.append("\n".join(cur)); cur = []
    cur.append(line)
chunks.append("\n".join(cur))
ns, failures = {"__name__": "__main__"}, []
for c in chunks:
    try:
        exec(compile(c, "<sweep>", "exec"), ns)
    except BaseException as e:
        failures.append({"statement": c.strip()[:400], "error": "".join(traceback.format_exception_only(type(e), e)).strip()[:300]})
print(json.dumps({"failures": failures}))
"""
    (ex / "sweep.py").write_text(_SWEEP_SCRIPT)
    out = subprocess.run([sys.executable, "-c", driver, "sweep.py"], cwd=ex, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    failures = json.loads(out.stdout.strip().splitlines()[-1])["failures"]
    assert not failures, "\n".join("%s\n    -> %s" % (f["statement"], f["error"]) for f in failures)


def test_single_simulator_is_faster_than_real_time(L):
    """The reference's implicit requirement: one cassie_sim_step_pd per 0.5 ms of simulated time, i.e. >= 2 kHz
    (reference example/cassiesim.c:284-293 prints SLOWER THAN REAL TIME otherwise)."""
    import time
    c = L.cassie_sim_init(MODEL, False)
    u = pd_input(np.random.default_rng(1))
    y = T.state_out_t()
    for s in range(200):
        L.cassie_sim_step_pd(c, ctypes.byref(y), ctypes.byref(u))
    t0 = time.perf_counter()
    n = 2000
    for s in range(n):
        L.cassie_sim_step_pd(c, ctypes.byref(y), ctypes.byref(u))
    rate = n / (time.perf_counter() - t0)
    L.cassie_sim_free(c)
    print("single cassie_sim_t: %.0f cassie_sim_step_pd per second" % rate)
    os.makedirs(os.path.join(REPO_DIR, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO_DIR, "gpurun_out", "single_sim_rate.txt"), "w") as f:
        f.write("%.1f cassie_sim_step_pd/s (one cassie_sim_t, host API, every step crosses PCIe)\n" % rate)
    assert rate > 2000.0


def test_batched_derived_getters_against_the_single_simulator_getters(L, cassie):
    """cassie_batch_derive: centre of mass / velocity / angular momentum, foot positions / velocities /
    forces, heel-toe forces, foot Jacobians and the mass matrix of every env against the single-simulator getters
    (reference src/cassiemujoco.c:1254-1301, :1604-1712, :1812-1898) evaluated on the same states."""
    from cassie_amd import phys as P
    L.cassie_batch_derive.argtypes = [VP, VP, VP]
    L.cassie_sim_forward.argtypes = [VP]
    for f in ("cassie_sim_cm_velocity", "cassie_sim_angular_momentum", "cassie_sim_foot_velocities"):
        getattr(L, f).argtypes = [VP, VP]
    L.cassie_sim_get_jacobian_full.argtypes = [VP, VP, VP, ctypes.c_char_p]
    n = 6
    bt = L.cassie_batch_create(MODEL, n, 0, 2)
    rng = np.random.default_rng(3)
    us = (T.pd_in_t * n)()
    for e in range(n):
        us[e] = pd_input(rng, scale=0.3)
    ys = (T.state_out_t * n)()
    for _ in range(400):                                   # onto the floor, every env on its own trajectory
        assert L.cassie_batch_step_pd(bt, ctypes.byref(us), ctypes.byref(ys)) == 0
    from cassie_amd._lib import lib as _l
    pb = L.cassie_batch_phys(bt)
    q, v, ct = np.zeros((n, 35)), np.zeros((n, 32)), np.zeros((n, 10))
    _l().phys_batch_download(pb, P.F_QPOS, q.ctypes.data, 0, n)
    _l().phys_batch_download(pb, P.F_QVEL, v.ctypes.data, 0, n)
    _l().phys_batch_download(pb, P.F_CTRL, ct.ctypes.data, 0, n)
    L.cassie_sim_ctrl.restype = DP
    L.cassie_sim_ctrl.argtypes = [VP]
    D, QM = np.zeros((n, P.DRV_DIM)), np.zeros((n, 32 * 32))
    # the fresh simulator below solves its contact forces from a zero warm start: give the batch the same one
    _l().phys_batch_upload(pb, P.F_QACC_WARMSTART, np.zeros((n, 32)).ctypes.data, 0, n)
    assert L.cassie_batch_derive(bt, D.ctypes.data, QM.ctypes.data) == 0
    c = L.cassie_sim_init(MODEL, False)
    saw_force = False
    for e in range(n):
        np.ctypeslib.as_array(L.cassie_sim_qpos(c), (35,))[:] = q[e]
        np.ctypeslib.as_array(L.cassie_sim_qvel(c), (32,))[:] = v[e]
        np.ctypeslib.as_array(L.cassie_sim_ctrl(c), (10,))[:] = ct[e]       # the motor torques act in the forward pass
        L.cassie_sim_forward(c)
        out = lambda k: np.zeros(k)
        cm, cv, am, fp, fv, ff, toe, heel, M = out(3), out(3), out(3), out(6), out(12), out(12), out(6), out(6), out(1024)
        L.cassie_sim_foot_forces(c, ff.ctypes.data)                           # of the forward pass just made
        L.cassie_sim_heeltoe_forces(c, toe.ctypes.data, heel.ctypes.data)
        L.cassie_sim_cm_position(c, cm.ctypes.data)
        L.cassie_sim_cm_velocity(c, cv.ctypes.data)
        L.cassie_sim_angular_momentum(c, am.ctypes.data)
        L.cassie_sim_foot_positions(c, fp.ctypes.data)
        L.cassie_sim_foot_velocities(c, fv.ctypes.data)
        L.cassie_sim_full_mass_matrix(c, M.ctypes.data)
        d = D[e]
        assert np.allclose(d[P.DRV_COM_POS: P.DRV_COM_POS + 3], cm, atol=1e-12)
        assert np.allclose(d[P.DRV_COM_VEL: P.DRV_COM_VEL + 3], cv, atol=1e-12)
        assert np.allclose(d[P.DRV_ANGMOM: P.DRV_ANGMOM + 3], am, atol=1e-11)
        assert np.allclose(d[P.DRV_FOOT_POS: P.DRV_FOOT_POS + 6], fp, atol=1e-12)
        assert np.allclose(d[P.DRV_FOOT_VEL: P.DRV_FOOT_VEL + 12], fv, atol=1e-12)
        assert np.allclose(QM[e], M, atol=1e-11)
        assert np.allclose(d[P.DRV_FOOT_FORCE: P.DRV_FOOT_FORCE + 12], ff, rtol=1e-9, atol=1e-8)
        assert np.allclose(d[P.DRV_TOE_FORCE: P.DRV_TOE_FORCE + 6], toe, rtol=1e-9, atol=1e-8)
        assert np.allclose(d[P.DRV_HEEL_FORCE: P.DRV_HEEL_FORCE + 6], heel, rtol=1e-9, atol=1e-8)
        saw_force |= ff[2] + ff[8] > 100
        for side, name in enumerate((b"left-foot", b"right-foot")):
            jp, jr = np.zeros(96), np.zeros(96)
            L.cassie_sim_get_jacobian_full(c, jp.ctypes.data, jr.ctypes.data, name)
            Jp = d[P.DRV_FOOT_JACP + side * 3 * P.MAXV: P.DRV_FOOT_JACP + (side + 1) * 3 * P.MAXV].reshape(3, P.MAXV)[:, :32]
            Jr = d[P.DRV_FOOT_JACR + side * 3 * P.MAXV: P.DRV_FOOT_JACR + (side + 1) * 3 * P.MAXV].reshape(3, P.MAXV)[:, :32]
            assert np.allclose(Jp, jp.reshape(3, 32), atol=1e-12) and np.allclose(Jr, jr.reshape(3, 32), atol=1e-12)
    assert saw_force
    L.cassie_sim_free(c)
    L.cassie_batch_free(bt)
