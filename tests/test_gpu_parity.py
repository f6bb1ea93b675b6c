"""GPU parity: the HIP kernel (through the C ABI, include/cassie_phys.h) against the CPU oracle.

Tolerances: north_star asks for <= 1e-6 relative qpos error over 1000 steps; the
kernel differs from the oracle only in floating-point operation order, so the
assertions here are several orders tighter (fp64, absolute)."""
import numpy as np
import pytest

from cassie_amd import Batch
from cassie_amd import phys as P
from oracle_py import Oracle

pytestmark = pytest.mark.gpu


def _rollout(cassie, nenv, nsteps, seed, hold=10, ctrl_scale=1.0, check_every=1):
    pod = cassie.pod
    rng = np.random.default_rng(seed)
    q0 = cassie.qpos_init()
    b = Batch(cassie, nenv)
    b.set(P.F_QPOS, np.tile(q0, (nenv, 1)))
    v0 = rng.uniform(-0.3, 0.3, (nenv, pod.nv))
    b.set(P.F_QVEL, v0)
    orcs = [Oracle(pod, q0) for _ in range(nenv)]
    for e, o in enumerate(orcs):
        o.qvel[:] = v0[e]
    hi = np.array([pod.act_ctrlrange[u][1] for u in range(pod.nu)])
    worst = dict(q=0.0, v=0.0, s=0.0)
    for s in range(0, nsteps, hold):
        c = ctrl_scale * hi * rng.uniform(-1, 1, (nenv, pod.nu))
        b.set(P.F_CTRL, c)
        for e, o in enumerate(orcs):
            o.ctrl[:] = c[e]
        for k in range(hold):
            b.step(1)
            for o in orcs:
                o.step()
            if (s + k) % check_every == 0 or s + k == nsteps - 1:
                q, v, sd = b.get(P.F_QPOS), b.get(P.F_QVEL), b.get(P.F_SENSORDATA)
                w, info = b.warnings()
                for e, o in enumerate(orcs):
                    assert (info[e, 0], info[e, 1], info[e, 2]) == (o.d.ncon, o.d.nefc, o.d.solver_iter), (s + k, e)
                    worst["q"] = max(worst["q"], np.max(np.abs(q[e] - o.qpos)))
                    worst["v"] = max(worst["v"], np.max(np.abs(v[e] - o.qvel)))
                    worst["s"] = max(worst["s"], np.max(np.abs(sd[e] - o.sensordata)))
    w, _ = b.warnings()
    assert not w.any()
    b.close()
    return worst


def test_64_envs_200_steps_vs_oracle(cassie):
    w = _rollout(cassie, 64, 200, seed=11)
    assert w["q"] < 1e-11 and w["v"] < 1e-9 and w["s"] < 1e-7, w


def test_1000_step_free_running_rollout(cassie):
    """BASELINE.json bar: max |qpos_err| vs the CPU reference over 1000 steps <= 1e-6 (relative)."""
    w = _rollout(cassie, 8, 1000, seed=12, hold=50, check_every=50)
    assert w["q"] < 1e-8 and w["v"] < 1e-6, w


def test_forward_matches_oracle_and_keeps_state(cassie):
    pod = cassie.pod
    b = Batch(cassie, 4)
    q0 = np.tile(cassie.qpos_init(), (4, 1))
    b.set(P.F_QPOS, q0)
    b.forward()
    o = Oracle(pod, cassie.qpos_init())
    o.forward()
    assert np.array_equal(b.get(P.F_QPOS), q0)
    assert np.allclose(b.get(P.F_QACC)[2], o.qacc, rtol=1e-9, atol=1e-9)
    assert np.allclose(b.get(P.F_SENSORDATA)[3], o.sensordata, atol=1e-10)
    xp = b.get(P.F_XPOS)[1].reshape(pod.nbody, 3)
    assert np.allclose(xp, o.xpos, atol=1e-13)
    b.close()


def test_full_size_batch_properties(cassie):
    """4096 envs (BASELINE config 2 size): identical inputs give bitwise identical rows, different inputs
    stay independent, quaternions stay normalised, nothing diverges, and a sample matches the oracle."""
    pod = cassie.pod
    n = 4096
    rng = np.random.default_rng(3)
    b = Batch(cassie, n)
    q0 = np.tile(cassie.qpos_init(), (n, 1))
    b.set(P.F_QPOS, q0)
    hi = np.array([pod.act_ctrlrange[u][1] for u in range(pod.nu)])
    c = 0.3 * hi * rng.uniform(-1, 1, (n, pod.nu))
    c[1::2] = c[0::2]                      # env 2k+1 is a twin of env 2k
    b.set(P.F_CTRL, c)
    b.step(100)
    q, v = b.get(P.F_QPOS), b.get(P.F_QVEL)
    w, info = b.warnings()
    assert not w.any()
    assert np.array_equal(q[0::2], q[1::2]) and np.array_equal(v[0::2], v[1::2])
    assert np.all(np.isfinite(q)) and np.all(np.isfinite(v))
    for a in (3, 10, 24):                  # pelvis and achilles-rod ball joints
        assert np.allclose(np.linalg.norm(q[:, a:a + 4], axis=1), 1, atol=1e-12)
    assert len(np.unique(q[0::2, 7])) > 2000   # envs really evolve independently
    for e in (0, 777, 4094):
        o = Oracle(pod, cassie.qpos_init())
        o.ctrl[:] = c[e]
        o.step(100)
        assert np.max(np.abs(q[e] - o.qpos)) < 1e-10
    assert abs(b.get(P.F_TIME)[5, 0] - 100 * pod.timestep) < 1e-12
    b.close()


def test_per_env_model_override(cassie):
    """Domain randomisation hook: one env gets a heavier pelvis and must fall differently; others unchanged."""
    from cassie_amd._lib import CmModel
    pod = cassie.pod
    b = Batch(cassie, 3)
    b.set(P.F_QPOS, np.tile(cassie.qpos_init(), (3, 1)))
    heavy = CmModel.from_buffer_copy(pod)
    heavy.body_mass[1] *= 2
    b.set_model(heavy, env=1)
    b.step(50)
    q = b.get(P.F_QPOS)
    assert np.array_equal(q[0], q[2]) and not np.array_equal(q[0], q[1])
    o = Oracle(heavy, cassie.qpos_init())
    o.step(50)
    assert np.max(np.abs(q[1] - o.qpos)) < 1e-11
    # one launch serves every env with the kernel instantiation of the shared model: a per-env model with another dof
    # tree is refused, not run through the wrong sparsity tables
    other = CmModel.from_buffer_copy(pod)
    other.dof_ancmask[7] ^= 1
    with pytest.raises(RuntimeError, match="dof tree"):
        b.set_model(other, env=2)
    b.close()


def test_config5_tray_box_vs_oracle(built):
    """BASELINE config 5 (cassie_tray_box.xml, nv = 38 kernel instantiation, box pair types): 8 envs x 400
    free-running steps against the oracle, then the 4096-env size with property checks."""
    from cassie_amd import Model
    tray = Model("cassie_tray_box")
    pod = tray.pod
    rng = np.random.default_rng(21)
    n = 8
    q0 = tray.qpos_init()
    b = Batch(tray, n)
    b.set(P.F_QPOS, np.tile(q0, (n, 1)))
    orcs = [Oracle(pod, q0) for _ in range(n)]
    hi = np.array([pod.act_ctrlrange[u][1] for u in range(pod.nu)])
    for s in range(0, 400, 50):
        c = 0.4 * hi * rng.uniform(-1, 1, (n, pod.nu))
        b.set(P.F_CTRL, c)
        b.step(50)
        for e, o in enumerate(orcs):
            o.ctrl[:] = c[e]
            o.step(50)
    q = b.get(P.F_QPOS)
    w, info = b.warnings()
    assert not w.any()
    for e, o in enumerate(orcs):
        assert (info[e, 0], info[e, 1]) == (o.d.ncon, o.d.nefc)
        assert np.max(np.abs(q[e] - o.qpos)) < 1e-8
    b.close()
    big = Batch(tray, 4096)
    big.set(P.F_QPOS, np.tile(q0, (4096, 1)))
    big.step(300)
    qb = big.get(P.F_QPOS)
    wb, _ = big.warnings()
    assert not wb.any() and np.all(np.isfinite(qb))
    assert np.all(qb[:, 37] > 0.9)             # every cube still on its tray after 0.15 s of zero-torque sag
    assert np.array_equal(qb[0], qb[4095])
    big.close()


def test_config4_heightfield_vs_oracle(built):
    """BASELINE config 4 (cassie_hfield.xml, terrain of reference example/test_hfield.py:39-41 shared by all envs)."""
    import oracle_py
    from cassie_amd import Model
    hf = Model("cassie_hfield")
    pod = hf.pod
    h = np.random.default_rng(99).random((200, 200)).astype(np.float32)
    h[95:105, 95:105] = 0
    oracle_py.set_hfield(h)
    try:
        n = 4
        q0 = np.tile(hf.qpos_init(), (n, 1))
        q0[:, 0] = [0.0, 0.35, -0.3, 0.6]                         # on the flat patch, straddling its edge, on the rough part
        b = Batch(hf, n)
        b.set_hfield(h)
        b.set(P.F_QPOS, q0)
        orcs = [Oracle(pod, q0[e]) for e in range(n)]
        b.step(400)
        q = b.get(P.F_QPOS)
        w, info = b.warnings()
        assert not w.any()
        for e, o in enumerate(orcs):
            o.step(400)
            assert (info[e, 0], info[e, 1]) == (o.d.ncon, o.d.nefc)
            assert np.max(np.abs(q[e] - o.qpos)) < 1e-8
        assert info[:, 0].max() >= 2                              # contacts with the terrain happened
        b.close()
        big = Batch(hf, 4096)
        big.set_hfield(h)
        big.set(P.F_QPOS, np.tile(hf.qpos_init(), (4096, 1)))
        big.step(200)
        qb = big.get(P.F_QPOS)
        wb, _ = big.warnings()
        assert not wb.any() and np.all(np.isfinite(qb)) and np.array_equal(qb[0], qb[-1])
        big.close()
    finally:
        oracle_py.set_hfield(None)


@pytest.mark.gpu
def test_generic_kernel_agrees_with_the_compile_time_topology_one(cassie, built):
    """The launcher uses a compile-time-topology instantiation for the in-scope models and a generic one (dof tree read
    from the model) for anything else.  Both must give the same trajectories to rounding (they eliminate the mass matrix
    in different orders), for the 32-dof and the 40-dof padded sizes."""
    from cassie_amd import Model
    for model in (cassie, Model("cassie_tray_box")):
        pod = model.pod
        n = 16
        rng = np.random.default_rng(5)
        q0 = np.tile(model.qpos_init(), (n, 1))
        ctrl = rng.uniform(-3, 3, (n, pod.nu))
        out = []
        for generic in (False, True):
            b = Batch(model, n)
            b.set_generic_kernel(generic)
            b.set(P.F_QPOS, q0)
            b.set(P.F_CTRL, ctrl)
            b.step(150)
            out.append((b.get(P.F_QPOS), b.get(P.F_QVEL)))
            b.close()
        assert np.abs(out[0][0] - out[1][0]).max() < 1e-9
        assert np.abs(out[0][1] - out[1][1]).max() < 1e-7


@pytest.mark.gpu
def test_model_that_is_not_kin_simple_runs_the_general_joint_loop(built):
    """The compile-time-topology kernels build a body's local transform from one record (slides + one rotational joint,
    cm_model.h cm_kinrec_t).  A body with two hinges is outside that form: the launcher must pick the run-time-topology kernel,
    whose kinematics stage loops over a body's joints -- against the oracle, 16 envs x 300 steps."""
    from cassie_amd import Model
    from test_kinematics_records import TWO_HINGES, two_hinges_start
    m = Model(TWO_HINGES)
    pod = m.pod
    q0, v0 = two_hinges_start(m)
    n = 16
    rng = np.random.default_rng(11)
    V = np.tile(v0, (n, 1)) + rng.uniform(-0.1, 0.1, (n, pod.nv))
    ctrl = rng.uniform(-2, 2, (n, pod.nu))
    b = Batch(m, n)
    b.set(P.F_QPOS, np.tile(q0, (n, 1)))
    b.set(P.F_QVEL, V)
    b.set(P.F_CTRL, ctrl)
    b.step(300)
    q, v = b.get(P.F_QPOS), b.get(P.F_QVEL)
    w, _ = b.warnings()
    b.close()
    assert not w.any()
    for e in (0, 7, 15):
        o = Oracle(pod, q0)
        o.qvel[:] = V[e]
        o.ctrl[:] = ctrl[e]
        for _ in range(300):
            o.step()
        assert np.abs(q[e] - o.qpos).max() < 1e-9 and np.abs(v[e] - o.qvel).max() < 1e-7


@pytest.mark.gpu
def test_contact_and_row_caps_on_the_gpu(cassie):
    """Poses that overflow the 16-contact and 63-row caps (tests/test_emu_parity.py has the emulator twin): same warning
    bits, counts and trajectory as the oracle."""
    pod = cassie.pod
    poses = ((0.0, [1.0, 0.0, 0.0, 0.0]), (-0.2, [np.cos(np.pi / 4), 0.0, np.sin(np.pi / 4), 0.0]), (-0.5, [np.cos(np.pi / 4), np.sin(np.pi / 4), 0.0, 0.0]))
    q0 = np.tile(cassie.qpos_init(), (len(poses), 1))
    for e, (z, quat) in enumerate(poses):
        q0[e, 2] = z
        q0[e, 3:7] = quat
    b = Batch(cassie, len(poses))
    b.set(P.F_QPOS, q0)
    orcs = [Oracle(pod, q0[e]) for e in range(len(poses))]
    for s in range(12):
        b.step(1)
        q = b.get(P.F_QPOS)
        w, info = b.warnings()
        for e, o in enumerate(orcs):
            o.step()
            assert (info[e, 0], info[e, 1], info[e, 2]) == (o.d.ncon, o.d.nefc, o.d.solver_iter), (s, e)
            want = (1 if o.d.warn_contact_full else 0) | (2 if o.d.warn_constraint_full else 0)
            assert int(w[e]) & 3 == want, (s, e)
            assert np.max(np.abs(q[e] - o.qpos)) < 1e-8, (s, e)
    b.close()


@pytest.mark.gpu
def test_per_env_heightfields(built):
    """Per-env terrain: env 1 gets its own grid, env 0 and 2 keep the shared one; every env must
    follow the oracle run on its own terrain."""
    import oracle_py
    from cassie_amd import Model
    hf = Model("cassie_hfield")
    pod = hf.pod
    shared = np.random.default_rng(99).random((200, 200)).astype(np.float32)
    shared[95:105, 95:105] = 0
    own = (0.5 * np.random.default_rng(7).random((200, 200))).astype(np.float32)
    n = 3
    q0 = np.tile(hf.qpos_init(), (n, 1))
    q0[:, 0] = [0.6, 0.6, -0.3]
    b = Batch(hf, n)
    b.set_hfield(shared)
    b.set_hfield(own, env=1)
    b.set(P.F_QPOS, q0)
    b.step(120)
    q = b.get(P.F_QPOS)
    w, info = b.warnings()
    b.close()
    assert not w.any()
    try:
        for e in range(n):
            oracle_py.set_hfield(own if e == 1 else shared)
            o = Oracle(pod, q0[e])
            o.step(120)
            assert np.max(np.abs(q[e] - o.qpos)) < 1e-8, e
        assert np.max(np.abs(q[0] - q[1])) > 1e-6      # the two terrains really differ under the robot
    finally:
        oracle_py.set_hfield(None)
