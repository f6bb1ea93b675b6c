"""GPU against the ORACLE, directly: the batched derived block, and teacher-forced single steps over
randomised states on all three models (SURVEY.md 8(d) parity protocol (i): identical qpos / qvel / ctrl in, one step, compare)."""
import numpy as np
import pytest

import golden_physics as G
import oracle_py
from cassie_amd import Batch, Model
from cassie_amd import phys as P
from derive_check import check_derived_block, foot_ids
from oracle_py import Oracle

pytestmark = pytest.mark.gpu


def test_batched_derived_block_on_the_gpu_against_the_oracle(cassie):
    """phys_batch_derive on the device -- COM position / velocity, angular momentum, foot positions / velocities, foot and
    heel / toe contact forces, foot Jacobians, the dense mass matrix -- against quantities computed with numpy from the
    oracle's state of the same (qpos, qvel): robots in the air, touching, pressed into the floor and at random joint angles."""
    pod = cassie.pod
    rng = np.random.default_rng(11)
    n = 64
    q = np.tile(cassie.qpos_init(), (n, 1))
    q[:, 2] -= rng.choice([0.0, 0.012, 0.02, 0.03], n)
    q[:, 7:] += 0.05 * rng.standard_normal((n, pod.nq - 7))
    v = rng.uniform(-0.5, 0.5, (n, pod.nv))
    b = Batch(cassie, n)
    try:
        b.set(P.F_QPOS, q); b.set(P.F_QVEL, v)
        ids = foot_ids(cassie)
        b.derive(ids)
        D, QM = b.get(P.F_DERIVED), b.get(P.F_QM).reshape(n, pod.nv, pod.nv)
        assert np.array_equal(b.get(P.F_QPOS), q)                       # a forward pass: the state is untouched
    finally:
        b.close()
    check_derived_block(cassie, q, v, D, QM, ids)
    pressed = q[:, 2] < cassie.qpos_init()[2] - 0.015
    assert (D[pressed, P.DRV_FOOT_FORCE + 2] + D[pressed, P.DRV_FOOT_FORCE + 8] > 20).all()   # pressed into the floor: the feet carry load ...
    assert (D[q[:, 2] == cassie.qpos_init()[2], P.DRV_FOOT_FORCE + 2] == 0).any()   # ... and in the air some carry none


def _random_states(model, n, rng, name):
    """States spread over what a policy can reach: joint positions across their ranges, pelvis heights from free fall to deep
    penetration (up to the contact / row caps), velocities up to +-5 rad/s, random pelvis orientations; ctrl across the torque
    limits; the free box of the tray model on / above / beside the tray."""
    pod = model.pod
    q = np.tile(model.qpos_init(), (n, 1))
    lo = np.array([pod.jnt_range[j][0] for j in range(pod.njnt)]); hi = np.array([pod.jnt_range[j][1] for j in range(pod.njnt)])
    for j in range(pod.njnt):
        if pod.jnt_type[j] == 3 and pod.jnt_limited[j]:                 # hinge with a range: anywhere in (and slightly past) it
            a = pod.jnt_qposadr[j]
            w = hi[j] - lo[j]
            q[:, a] = rng.uniform(lo[j] - 0.02 * w, hi[j] + 0.02 * w, n)
    half = n // 2                                                        # half of the envs stay near the closed-loop-consistent init pose
    q[:half, 7:35] = model.qpos_init()[7:35] + 0.05 * rng.standard_normal((half, 28))
    q[:, 2] = model.qpos_init()[2] + rng.uniform(-0.25, 0.3, n)
    quat = rng.standard_normal((n, 4)) * [0.0, 0.15, 0.15, 0.3] + [1, 0, 0, 0]
    q[:, 3:7] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    lying = np.arange(n) % 4 == 3                                        # a quarter lies on the ground at any orientation: many contacts
    q[lying, 2] = rng.uniform(0.05, 0.35, int(lying.sum()))
    lq = rng.standard_normal((int(lying.sum()), 4))
    q[lying, 3:7] = lq / np.linalg.norm(lq, axis=1, keepdims=True)
    if name == "cassie_hfield":
        for e in range(n):
            q[e, 0], q[e, 1] = G.start_xy(name, e)
    if pod.nq > 35:                                                      # the tray's free cube: position jitter, random orientation
        q[:, 35:38] += rng.uniform(-0.03, 0.03, (n, 3))
        cq = rng.standard_normal((n, 4)) * 0.2 + [1, 0, 0, 0]
        q[:, 38:42] = cq / np.linalg.norm(cq, axis=1, keepdims=True)
    v = rng.uniform(-5.0, 5.0, (n, pod.nv))
    v[:, :3] = rng.uniform(-1.0, 1.0, (n, 3))
    tmax = np.array([pod.act_ctrlrange[u][1] for u in range(pod.nu)])
    ctrl = rng.uniform(-1.1, 1.1, (n, pod.nu)) * tmax
    return q, v, ctrl


@pytest.mark.parametrize("name", ["cassie", "cassie_hfield", "cassie_tray_box"])
def test_teacher_forced_single_steps_over_randomised_states(built, name):
    """One cassie_sim_step-equivalent from identical (qpos, qvel, ctrl, zero warm start) on the device and on the oracle, for
    states that do NOT come from a trajectory started at qpos_init: qpos within 1e-12, qvel within 5e-12 of max(1, |v|, h |a|),
    qacc / sensordata within 1e-9 relative, equal contact / row / sweep counts, and the same envs flagged at a cap."""
    model = Model(name)
    pod = model.pod
    n = 512
    rng = np.random.default_rng({"cassie": 21, "cassie_hfield": 22, "cassie_tray_box": 23}[name])
    q, v, ctrl = _random_states(model, n, rng, name)
    hf = G.terrain(name)
    b = Batch(model, n)
    try:
        if hf is not None:
            b.set_hfield(hf)
            oracle_py.set_hfield(hf)
        b.set(P.F_QPOS, q); b.set(P.F_QVEL, v); b.set(P.F_CTRL, ctrl)
        b.step(1)
        qg, vg, ag, sg = b.get(P.F_QPOS), b.get(P.F_QVEL), b.get(P.F_QACC), b.get(P.F_SENSORDATA)
        w, info = b.warnings()
        rows_seen, worst = [], dict(q=0.0, v=0.0, a=0.0, s=0.0)
        for e in range(n):
            o = Oracle(pod, q[e])
            o.qvel[:] = v[e]; o.ctrl[:] = ctrl[e]
            o.step()
            assert (info[e, 0], info[e, 1], info[e, 2]) == (o.d.ncon, o.d.nefc, o.d.solver_iter), e
            assert bool(w[e] & 1) == bool(o.d.warn_contact_full) and bool(w[e] & 2) == bool(o.d.warn_constraint_full), e
            rows_seen.append(o.d.nefc)
            if o.d.diverged or (w[e] & 8):
                assert bool(o.d.diverged) == bool(w[e] & 8), e
                continue
            scale_a = max(1.0, np.abs(o.qacc).max())
            worst["q"] = max(worst["q"], np.abs(qg[e] - o.qpos).max()); worst["v"] = max(worst["v"], np.abs(vg[e] - o.qvel).max() / max(1.0, np.abs(o.qvel).max(), pod.timestep * np.abs(o.qacc).max()))   # (v' = v + h a: deep penetrations give h |a| of 10 .. 100)
            worst["a"] = max(worst["a"], np.abs(ag[e] - o.qacc).max() / scale_a)
            worst["s"] = max(worst["s"], (np.abs(sg[e] - o.sensordata) / np.maximum(1.0, np.abs(o.sensordata))).max())
        print("%s: rows %d .. %d (mean %.1f), worst errors %s" % (name, min(rows_seen), max(rows_seen), np.mean(rows_seen), worst))
        assert max(rows_seen) > 40 and min(rows_seen) <= 16                  # from free flight to well past the fast kernel's capacity
        assert worst["q"] < 1e-12 and worst["v"] < 5e-12 and worst["a"] < 1e-9 and worst["s"] < 1e-9, worst   # (observed: q 7e-15, v 1.1e-12, a 2.7e-12, s 1.1e-12)
    finally:
        b.close()
        oracle_py.set_hfield(None)
