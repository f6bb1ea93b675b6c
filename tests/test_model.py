"""Known-answer pins for the model compiler and the oracle.

These are the only pins that exist for the physics: the reference ships no tests
or golden vectors and its MuJoCo binary is absent (parity unpinned).
"""
import ctypes
import os

import numpy as np
import pytest

from cassie_amd import Model
from cassie_amd._lib import CmModel
from oracle_py import Oracle, arr

REF_MODEL_DIR = "/root/reference/model"


def test_dims(cassie):
    p = cassie.pod
    assert (p.nq, p.nv, p.nu, p.nbody, p.njnt, p.nsensordata) == (35, 32, 10, 26, 26, 29)
    assert cassie.size(5) == 50          # ngeom, all geoms (reference cassie_sim_params, src/cassiemujoco.c:1566-1574)
    assert p.neq == 4 and p.npair == 153 and p.ngeom == 25


def test_other_models_dims(built):
    t = Model("cassie_tray_box").pod
    assert (t.nq, t.nv, t.nbody) == (42, 38, 28)
    hm = Model("cassie_hfield")
    assert hm.pod.nbody == 27 and (hm.pod.hfield_nrow, hm.pod.hfield_ncol) == (200, 200)
    assert list(hm.pod.hfield_size) == [5, 5, .2, .1]


def test_index_maps(cassie):
    p = cassie.pod
    # motor dofs / qpos addresses (reference src/cassiemujoco.c:1715, example/cassietest_jac.py:56-57)
    assert list(p.act_dofid[:10]) == [6, 7, 8, 12, 18, 19, 20, 21, 25, 31]
    assert list(p.act_qposadr[:10]) == [7, 8, 9, 14, 20, 21, 22, 23, 28, 34]
    # encoder joints (reference src/cassiemujoco.c:856)
    assert [p.sensor_objid[i] for i in (5, 6, 7, 13, 14, 15)] == [9, 10, 14, 20, 21, 25]
    assert [p.sensor_objid[i] for i in (0, 1, 2, 3, 4, 8, 9, 10, 11, 12)] == list(range(10))
    assert list(p.act_gear[:10]) == [25, 25, 16, 16, 50] * 2


def test_total_mass_is_M00(cassie):
    p = cassie.pod
    assert abs(sum(p.body_mass[: p.nbody]) - 33.312) < 1e-9
    o = Oracle(p, cassie.qpos_init())
    o.forward()
    M = o.qM
    for i in range(3):
        assert abs(M[i, i] - 33.312) < 1e-9
    assert np.allclose(M, M.T, atol=1e-13)
    assert np.all(np.linalg.eigvalsh(M) > 0)
    arm = np.array(p.dof_armature[: p.nv])
    assert np.all(np.diag(M) >= arm - 1e-12)


def test_loop_closure_and_census(cassie):
    p = cassie.pod
    o = Oracle(p)                      # qpos0
    o.forward()
    d = o.d
    assert (d.nefc, d.ne, d.nl, d.ncon) == (32, 12, 4, 4)
    assert np.max(np.abs(arr(d.efc_pos)[:12])) < 1e-12          # connects close exactly at qpos0
    assert np.allclose(arr(d.efc_pos)[12:16], -0.5236, atol=1e-4)  # foot / foot-crank limits violated by 30 deg
    o = Oracle(p, cassie.qpos_init())
    o.forward()
    d = o.d
    assert (d.nefc, d.ne, d.nl, d.ncon) == (12, 12, 0, 0)
    res = np.linalg.norm(arr(d.efc_pos)[:12].reshape(4, 3), axis=1)
    assert np.allclose(res, [6.59e-3, 0.856e-3, 6.59e-3, 0.851e-3], atol=2e-5)


def test_fk_at_qpos_init(cassie):
    o = Oracle(cassie.pod, cassie.qpos_init())
    o.forward()
    lf = cassie.name2id(1, "left-foot")
    rf = cassie.name2id(1, "right-foot")
    assert np.allclose(o.xpos[lf], [-0.01999, 0.13477, 0.06073], atol=2e-5)
    assert np.allclose(o.xpos[rf], [-0.01999, -0.13477, 0.06073], atol=2e-5)
    com = arr(o.d.subtree_com)[1]
    assert np.allclose(com, [-0.01754, 0.00012, 0.88147], atol=2e-5)


def test_free_fall_and_accelerometer(cassie):
    """No contact at qpos_init: the base accelerates at ~g and the IMU reads ~0 in free fall."""
    o = Oracle(cassie.pod, cassie.qpos_init())
    o.forward()
    assert abs(o.qacc[2] + 9.81) < 0.2
    assert np.linalg.norm(o.sensordata[23:26]) < 0.5
    assert np.allclose(o.sensordata[16:20], [1, 0, 0, 0])
    assert np.allclose(o.sensordata[26:29], [0, -0.5, 0])


def test_momentum_conservation(cassie):
    """Gravity off, no contacts, no damping, no springs: linear momentum of the tree is conserved."""
    import copy
    p = CmModel.from_buffer_copy(cassie.pod)
    for i in range(3):
        p.gravity[i] = 0
    for k in range(p.nv):
        p.dof_damping[k] = 0
    for j in range(p.njnt):
        p.jnt_stiffness[j] = 0
    q = cassie.qpos_init()
    q[2] = 5.0
    o = Oracle(p, q)
    rng = np.random.default_rng(0)
    o.qvel[:] = rng.uniform(-1, 1, p.nv)
    o.forward()
    mass = np.array(p.body_mass[: p.nbody])

    def momentum():
        o.forward()
        # linear momentum = sum m_b * v(com_b); use finite difference-free formula via cvel at the tree com
        cvel = arr(o.d.cvel)[: p.nbody]
        com = arr(o.d.subtree_com)[1]
        xipos = arr(o.d.xipos)[: p.nbody]
        v = cvel[:, 3:] + np.cross(cvel[:, :3], xipos - com)
        return (mass[:, None] * v).sum(0)

    p0 = momentum()
    o.step(200)
    p1 = momentum()
    assert np.allclose(p0, p1, atol=5e-3 * max(1.0, np.linalg.norm(p0)))


@pytest.mark.skipif(not os.path.isdir(REF_MODEL_DIR), reason="reference MJCF not present on this box")
@pytest.mark.parametrize("name", ["cassie", "cassie_hfield", "cassie_tray_box"])
def test_cmodel_matches_reference_xml(built, name):
    """The committed .cmodel files are exactly what the MJCF loader produces from the reference XML."""
    a = Model(os.path.join(REF_MODEL_DIR, name + ".xml")).pod
    b = Model(name).pod
    assert bytes(a) == bytes(b)
