"""The oracle against the TRUE reference STAGE BY STAGE (genuine MuJoCo, reference src/cassiemujoco.c:521-555 is where the
reference looks for it, :1132-1133 where it steps it) -- skipped wherever no MuJoCo is discoverable, which is the case in the
build container and on the GPU box image (parity of the physics is unpinned there).

tests/test_true_reference.py compares trajectories; this file compares what a trajectory is made of, on the 512 randomised
states per model of tests/test_derive_gpu.py (hinge angles across and past their ranges, pelvis from free flight to 25 cm of
penetration, robots lying on the ground, velocities to +-5 rad/s, torques past their limits): the mass matrix, the contact list
(count, geoms, distance, position, frame), the constraint rows (count, type, id, Jacobian, position, diagApprox, R, aref, b) and
the solver's forces.  The comparisons run in pipeline order and every assertion names the quantity and the state, so the first run
on a machine that has a MuJoCo says WHICH constant of the restatement differs ("the least certain
part": diagApprox, the impedance, the pyramid's R), not merely that a trajectory drifted."""
import numpy as np
import pytest

import mujoco_ref


@pytest.fixture(scope="module")
def ref():
    r = mujoco_ref.find()
    if r is None:
        pytest.skip("true reference unavailable: no mujoco210 directory and no `mujoco` wheel on this machine")
    return r


def _oracle_stages(pod, o):
    import oracle_py
    d = o.d
    nefc, ncon, nv = d.nefc, d.ncon, pod.nv
    g = lambda f: np.array(oracle_py.arr(getattr(d, f))[:nefc], dtype=float)
    con = [d.contact[i] for i in range(ncon)]
    return dict(ncon=ncon, nefc=nefc, qM=np.array(o.qM), efc_J=np.array(oracle_py.arr(d.efc_J))[:nefc, :nv],
                efc_type=np.array(oracle_py.arr(d.efc_type)[:nefc]), efc_id=np.array(oracle_py.arr(d.efc_id)[:nefc]),
                efc_aref=g("efc_aref"), efc_R=g("efc_R"), efc_force=g("efc_force"), efc_pos=g("efc_pos"), efc_diagApprox=g("efc_diagApprox"), efc_b=g("efc_b"),
                contact_dist=np.array([c.dist for c in con]), contact_pos=np.array([list(c.pos) for c in con]).reshape(ncon, 3),
                contact_frame=np.array([list(c.frame) for c in con]).reshape(ncon, 9),
                contact_geom=np.array([[pod.geom_fullid[c.geom1], pod.geom_fullid[c.geom2]] for c in con], dtype=np.int32).reshape(ncon, 2),
                contact_dim=np.array([c.dim for c in con], dtype=np.int32), qacc_smooth=np.array(oracle_py.arr(d.qacc_smooth)[:nv]))


# (quantity, relative tolerance) in pipeline order: the first failure is the earliest stage that differs
STAGES = [("qM", 1e-9), ("qacc_smooth", 1e-8), ("contact_dist", 1e-9), ("contact_pos", 1e-9), ("contact_frame", 1e-9), ("efc_pos", 1e-9), ("efc_J", 1e-9),
          ("efc_diagApprox", 1e-9), ("efc_R", 1e-9), ("efc_aref", 1e-8), ("efc_b", 1e-7), ("efc_force", 1e-5)]


@pytest.mark.parametrize("name", ["cassie", "cassie_hfield", "cassie_tray_box"])
def test_oracle_stage_by_stage_against_true_reference(ref, name):
    from cassie_amd import Model
    import golden_physics as G
    import oracle_py
    from oracle_py import Oracle
    from test_derive_gpu import _random_states
    xml = mujoco_ref.mjcf_path(name)
    if xml is None:
        pytest.skip("true reference found (%s) but the reference's MJCF files are not staged" % ref.kind)
    model = Model(name)
    pod = model.pod
    n = 512
    rng = np.random.default_rng({"cassie": 21, "cassie_hfield": 22, "cassie_tray_box": 23}[name])
    q, v, ctrl = _random_states(model, n, rng, name)
    hf = G.terrain(name)
    s = ref.sim(xml)
    failures = {}
    try:
        if hf is not None:
            s.set_hfield(hf)
            oracle_py.set_hfield(hf)
        for e in range(n):
            s.set_state(q[e], v[e], np.zeros(pod.nv))
            s.set_ctrl(ctrl[e])
            s.forward()
            t = s.stages()
            o = Oracle(pod, q[e])
            o.qvel[:] = v[e]; o.ctrl[:] = ctrl[e]
            o.forward()
            mine = _oracle_stages(pod, o)
            # discrete structure first: a different contact or row count makes every later comparison meaningless for this state
            if (t["ncon"], t["nefc"]) != (mine["ncon"], mine["nefc"]):
                failures.setdefault("counts (ncon, nefc)", []).append((e, (t["ncon"], t["nefc"]), (mine["ncon"], mine["nefc"])))
                continue
            for key in ("contact_geom", "contact_dim", "efc_type", "efc_id"):
                if not np.array_equal(t[key], mine[key]):
                    failures.setdefault(key, []).append((e, t[key].tolist(), mine[key].tolist()))
            for key, tol in STAGES:
                a, b = t[key], mine[key]
                if a is None:        # (a sparse Jacobian in the reference: not dumped)
                    continue
                err = float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(a)))) if a.size else 0.0
                if err > tol:
                    failures.setdefault(key, []).append((e, err))
                    break            # (later stages of this state inherit the difference)
    finally:
        s.close()
        oracle_py.set_hfield(None)
    report = "; ".join("%s: %d of %d states, first %r" % (k, len(vv), n, vv[0]) for k, vv in failures.items())
    assert not failures, "%s %s -- earliest differing stages: %s" % (ref.kind, ref.version, report)
