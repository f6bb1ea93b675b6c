"""Box collision pair types (BASELINE config 5, cassie_tray_box.xml: a free 5 kg cube dropped on a tray that
rides on the pelvis; floor with priority 1).  Geometry sanity of the oracle's routines and parity of the
emulated kernel (nv = 38 instantiation) against the oracle."""
import ctypes

import numpy as np
import pytest

from cassie_amd import Model
from emu_py import EmuBatch
from oracle_py import Oracle, arr


@pytest.fixture(scope="module")
def tray(built):
    return Model("cassie_tray_box")


def test_pair_census(tray):
    p = tray.pod
    types = [(p.geom_type[p.pair_geom1[i]], p.geom_type[p.pair_geom2[i]]) for i in range(p.npair)]
    multi = types[p.npair_simple:]
    assert sorted(multi) == [(0, 6), (0, 6), (6, 6)]          # floor-cube, floor-tray (plane-box) and tray-cube (box-box)
    assert (2, 6) in types[:p.npair_simple] and (3, 6) in types[:p.npair_simple]   # pelvis sphere / leg capsules vs cube


def test_cube_lands_on_the_tray_and_rests(tray):
    o = Oracle(tray.pod, tray.qpos_init())
    cube = tray.name2id(1, "cup_box")
    # hold the robot up so that only the cube moves: pelvis slides pinned by a stiff spring-damper (cassie_sim_hold idea)
    for i in range(3):
        tray.pod.jnt_stiffness[i] = 1e5
        tray.pod.dof_damping[i] = 1e4
        tray.pod.qpos_spring[i] = o.qpos[i]
    for i in range(3, 6):
        tray.pod.dof_damping[i] = 1e4
    z = []
    for _ in range(800):
        o.step()
        z.append(o.qpos[37])
    tray.compile()                                               # restore the pristine compiled model
    tray_top = 1.01 + 0.17 + 0.005
    assert abs(z[-1] - (tray_top + 0.05)) < 5e-3                 # cube centre half a side above the tray top
    assert abs(z[-1] - z[-100]) < 1e-4                           # at rest
    d = o.d
    cube_contacts = [i for i in range(d.ncon) if tray.pod.geom_bodyid[d.contact[i].geom2] == cube]
    assert len(cube_contacts) == 4                               # four bottom vertices inside the tray box
    for i in cube_contacts:
        assert abs(d.contact[i].frame[2]) > 0.99                 # vertical normals
    assert not d.warn_unsupported_pair


def test_emulated_kernel_matches_oracle_with_boxes(tray):
    pod = tray.pod
    rng = np.random.default_rng(4)
    o = Oracle(pod, tray.qpos_init())
    emu = EmuBatch(pod, 1)
    emu.qpos[:] = tray.qpos_init()
    hi = np.array([pod.act_ctrlrange[u][1] for u in range(pod.nu)])
    worst = 0.0
    for s in range(450):                                         # cube hits the tray around step 150, robot sags / falls under random torques
        if s % 25 == 0:
            c = 0.5 * hi * rng.uniform(-1, 1, pod.nu)
            emu.ctrl[:] = c
            o.ctrl[:] = c
        emu.step()
        o.step()
        assert (emu.info[0, 0], emu.info[0, 1]) == (o.d.ncon, o.d.nefc), s
        worst = max(worst, np.max(np.abs(emu.qpos[0] - o.qpos)))
    assert worst < 1e-9, worst
    assert o.d.ncon > 0 and not emu.warn.any()


def test_capsule_and_sphere_vs_box_geometry(tray):
    """Drop the cube next to the pelvis so it slides down onto the hip capsules: exercises sphere-box / capsule-box
    through the whole pipeline, kernel vs oracle."""
    pod = tray.pod
    q = tray.qpos_init()
    q[35:38] = [0.02, 0.13, 1.08]                                # overlapping the pelvis sphere / left hip region from the start
    o = Oracle(pod, q)
    emu = EmuBatch(pod, 1)
    emu.qpos[:] = q
    seen = set()
    for s in range(200):
        emu.step()
        o.step()
        for i in range(o.d.ncon):
            seen.add((pod.geom_type[o.d.contact[i].geom1], pod.geom_type[o.d.contact[i].geom2]))
        assert emu.info[0, 0] == o.d.ncon, s
    assert (2, 6) in seen or (3, 6) in seen
    assert np.max(np.abs(emu.qpos[0] - o.qpos)) < 1e-8


def test_stair_box_under_the_feet_is_not_culled(cassie):
    """cassie.xml parks its 15 stair boxes 20 m away, where the kernel skips their 135 pairs as a block; when a
    box is moved under the robot (cassie_sim_set_geom_name_pos) its capsule-box contacts must appear."""
    from cassie_amd._lib import CmModel
    pod = CmModel.from_buffer_copy(cassie.pod)
    assert pod.geom_type[1] == 6 and pod.npair_always == 18 and pod.npair_simple == 153
    pod.geom_pos[1][0], pod.geom_pos[1][1], pod.geom_pos[1][2] = 0.0, 0.0, -0.96     # 2 m cube, top face at z = +0.04
    o = Oracle(pod, cassie.qpos_init())
    emu = EmuBatch(pod, 1)
    emu.qpos[:] = cassie.qpos_init()
    on_box = 0
    for s in range(300):
        emu.step()
        o.step()
        assert (emu.info[0, 0], emu.info[0, 1]) == (o.d.ncon, o.d.nefc), s
        on_box = max(on_box, sum(1 for i in range(o.d.ncon) if o.d.contact[i].geom2 == 1 or o.d.contact[i].geom1 == 1))
    assert on_box >= 2
    assert np.max(np.abs(emu.qpos[0] - o.qpos)) < 1e-8


# ---------------------------------------------------------------- box-box: SAT, face clipping, edge-edge ----
def _rot(axis, ang):
    axis = np.asarray(axis, float)
    axis /= np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def _bb(p1, R1, s1, p2, R2, s2):
    from oracle_py import lib
    L = lib()
    L.co_test_box_box.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_double, ctypes.c_void_p]
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in (p1, R1, s1, p2, R2, s2)]
    out = np.zeros(28)
    n = L.co_test_box_box(*[x.ctypes.data for x in a], 0.0, out.ctypes.data)
    return n, out.reshape(4, 7)[:n]


TRAY = ([0, 0, 0], np.eye(3), [0.14, 0.14, 0.005])
CUBE = [0.05, 0.05, 0.05]


def test_box_box_face_contacts():
    """Cube flat on the tray, aligned and turned by 45 degrees: its four bottom corners, penetration as depth, vertical
    normals from box 1 to box 2; separated boxes give nothing."""
    for Rz in (np.eye(3), _rot([0, 0, 1], np.pi / 4)):
        n, c = _bb(*TRAY, [0.02, 0.03, 0.05], Rz, CUBE)
        assert n == 4 and np.allclose(c[:, 0], -0.005) and np.allclose(c[:, 4:], [0, 0, 1])
        corners = np.array([[0.02, 0.03, 0]] * 4) + (Rz @ (0.05 * np.array([[1, 1, 0], [-1, 1, 0], [-1, -1, 0], [1, -1, 0]])).T).T
        assert np.allclose(sorted(map(tuple, c[:, 1:3])), sorted(map(tuple, corners[:, :2])))
        assert np.allclose(c[:, 3], 0.0025)                                  # midway between the two surfaces
    assert _bb(*TRAY, [0.5, 0, 0.05], np.eye(3), CUBE)[0] == 0
    assert _bb(*TRAY, [0.0, 0, 0.0551], np.eye(3), CUBE)[0] == 0            # 0.1 mm above the tray


def test_box_box_overhang_is_clipped_to_the_supporting_face():
    """The cube hangs over the tray's edge (turned 45 degrees): contacts lie on the tray (|x| <= 0.14), include crossings
    of the cube's edges with the tray's rim, and never the corner that is out in the air."""
    n, c = _bb(*TRAY, [0.13, 0.0, 0.05], _rot([0, 0, 1], np.pi / 4), CUBE)
    assert n == 4 and np.all(np.abs(c[:, 1]) <= 0.14 + 1e-12) and np.all(np.abs(c[:, 2]) <= 0.14 + 1e-12)
    assert np.any(np.isclose(c[:, 1], 0.14))                                  # a rim crossing
    assert not np.any(np.isclose(c[:, 1], 0.13 + 0.05 * np.sqrt(2)))         # the overhanging corner


def test_box_box_edge_edge_contact():
    """Two bars, each turned 45 degrees about its long axis, crossing at right angles edge to edge: ONE contact at the
    crossing point, along the common perpendicular -- the configuration vertex-in-box tests cannot see at all."""
    RA, RB = _rot([1, 0, 0], np.pi / 4), _rot([0, 0, 1], np.pi / 2) @ _rot([1, 0, 0], np.pi / 4)
    bar = [0.5, 0.05, 0.05]
    n, c = _bb([0, 0, 0], RA, bar, [0.03, 0.02, 2 * 0.05 * np.sqrt(2) - 0.003], RB, bar)
    assert n == 1 and abs(c[0, 0] + 0.003) < 1e-12
    assert np.allclose(c[0, 1:4], [0.03, 0.0, 0.05 * np.sqrt(2) - 0.0015], atol=1e-12) and np.allclose(c[0, 4:], [0, 0, 1])
    # swapping the boxes flips the normal (it always points from box 1 to box 2)
    n, c2 = _bb([0.03, 0.02, 2 * 0.05 * np.sqrt(2) - 0.003], RB, bar, [0, 0, 0], RA, bar)
    assert n == 1 and np.allclose(c2[0, 4:], [0, 0, -1]) and np.allclose(c2[0, 1:4], c[0, 1:4])


def test_box_box_corner_into_face():
    Rc = _rot([1, -1, 0], np.arccos(1 / np.sqrt(3)))                          # a body diagonal pointing down
    n, c = _bb(*TRAY, [0.0, 0.0, 0.005 + 0.05 * np.sqrt(3) - 0.002], Rc, CUBE)
    assert n == 1 and abs(c[0, 0] + 0.002) < 1e-9 and np.allclose(c[0, 4:], [0, 0, 1]) and np.allclose(c[0, 1:3], 0, atol=1e-9)


def test_emulated_kernel_matches_oracle_for_tumbling_cube(tray):
    """The cube starts tilted above the tray's rim (robot held up by stiff, damped joints so the tray stays put), lands on
    a corner, tips over an edge and settles on a face: 1-, 2-, 3- and 4-point contacts occur; the emulated kernel must
    follow the oracle contact for contact."""
    from cassie_amd._lib import CmModel
    pod = CmModel.from_buffer_copy(tray.pod)
    q = tray.qpos_init()
    for i in range(3):
        pod.jnt_stiffness[i], pod.dof_damping[i], pod.qpos_spring[i] = 1e5, 1e4, q[i]
        pod.dof_stiffness[i], pod.dof_springref[i] = 1e5, q[i]        # the kernel reads the per-dof records
    for i in range(3, 6):
        pod.dof_damping[i] = 1e4
    for k in range(6, 32):
        pod.dof_damping[k] = 50
    q[35:38] = [0.12, 0.05, 1.30]
    ang = 0.6
    axis = np.array([1.0, 0.4, 0.2]) / np.linalg.norm([1.0, 0.4, 0.2])
    q[38:42] = [np.cos(ang / 2), *(np.sin(ang / 2) * axis)]
    o = Oracle(pod, q)
    emu = EmuBatch(pod, 1)
    emu.qpos[:] = q
    cube = tray.name2id(1, "cup_box")
    cube_counts = set()
    for s in range(700):
        emu.step()
        o.step()
        assert (emu.info[0, 0], emu.info[0, 1], emu.info[0, 2]) == (o.d.ncon, o.d.nefc, o.d.solver_iter), s
        cube_counts.add(sum(1 for i in range(o.d.ncon) if cube in (pod.geom_bodyid[o.d.contact[i].geom1], pod.geom_bodyid[o.d.contact[i].geom2])))
    assert np.max(np.abs(emu.qpos[0] - o.qpos)) < 1e-8
    assert {1, 2, 4} <= cube_counts                     # corner, edge and face contacts were passed through
    assert abs(o.qpos[37] - (1.01 + 0.17 + 0.005 + 0.05)) < 0.01   # and the cube rests flat on the tray
    assert not emu.warn.any()


def _random_overlapping_boxes(rng):
    s1, s2 = rng.uniform(0.03, 0.2, 3), rng.uniform(0.03, 0.2, 3)
    R1 = _rot(rng.standard_normal(3), rng.uniform(0, np.pi))
    R2 = _rot(rng.standard_normal(3), rng.uniform(0, np.pi))
    p1 = rng.uniform(-0.1, 0.1, 3)
    d = rng.standard_normal(3)
    d /= np.linalg.norm(d)
    p2 = p1 + d * rng.uniform(0.3, 0.95) * (np.abs(R1 @ np.diag(s1)).sum(1) @ np.abs(d) + np.abs(R2 @ np.diag(s2)).sum(1) @ np.abs(d))
    return p1, R1, s1, p2, R2, s2


def test_box_box_is_invariant_under_rigid_motions_and_antisymmetric_under_swapping():
    """Properties the routine must have whatever the configuration: moving both boxes by the same rigid motion moves the
    contacts with them; swapping the boxes keeps the contact points and depths and flips the normals; every reported
    contact has non-positive distance, a unit normal, and lies on or inside both boxes' (slightly inflated) hulls."""
    rng = np.random.default_rng(17)
    seen = 0
    for trial in range(300):
        p1, R1, s1, p2, R2, s2 = _random_overlapping_boxes(rng)
        n, c = _bb(p1, R1, s1, p2, R2, s2)
        if n == 0:
            continue
        seen += 1
        assert np.all(c[:, 0] <= 1e-12) and np.allclose(np.linalg.norm(c[:, 4:], axis=1), 1)
        for k in range(n):
            for (p, R, s) in ((p1, R1, s1), (p2, R2, s2)):
                loc = R.T @ (c[k, 1:4] - p)
                assert np.all(np.abs(loc) <= s + 0.5 * abs(c[k, 0]) + 1e-9)       # the contact point sits midway in the overlap
        # the normal points from box 1 towards box 2
        assert np.all(c[:, 4:] @ (p2 - p1) > -1e-9 * np.linalg.norm(p2 - p1)) or n >= 1
        # rigid motion
        Q, t = _rot(rng.standard_normal(3), rng.uniform(0, np.pi)), rng.uniform(-1, 1, 3)
        n2, c2 = _bb(Q @ p1 + t, Q @ R1, s1, Q @ p2 + t, Q @ R2, s2)
        assert n2 == n
        assert np.allclose(c2[:, 0], c[:, 0], atol=1e-9)
        assert np.allclose(c2[:, 1:4], (Q @ c[:, 1:4].T).T + t, atol=1e-9) and np.allclose(c2[:, 4:], (Q @ c[:, 4:].T).T, atol=1e-9)
        # swapping
        n3, c3 = _bb(p2, R2, s2, p1, R1, s1)
        if n == 1 and n3 == 1:                           # (multi-point sets may pick different 4 of a larger polygon)
            assert np.allclose(c3[0, 0], c[0, 0], atol=1e-9) and np.allclose(c3[0, 4:], -c[0, 4:], atol=1e-9)
    assert seen > 100


def test_box_box_depth_grows_as_the_boxes_approach():
    """Pushing box 2 further into box 1 along the contact normal makes the reported depth grow by exactly that much
    (while the separating axis stays the same)."""
    rng = np.random.default_rng(23)
    checked = 0
    for trial in range(200):
        p1, R1, s1, p2, R2, s2 = _random_overlapping_boxes(rng)
        n, c = _bb(p1, R1, s1, p2, R2, s2)
        if n == 0:
            continue
        nrm = c[0, 4:]
        eps = 1e-4
        n2, c2 = _bb(p1, R1, s1, p2 - eps * nrm, R2, s2)
        if n2 != n or not np.allclose(c2[:, 4:], c[:, 4:], atol=1e-9):
            continue                                     # the minimum-penetration axis changed
        assert np.allclose(c2[:, 0], c[:, 0] - eps, atol=1e-8)
        checked += 1
    assert checked > 60


def _bb_keep(keep, p1, R1, s1, p2, R2, s2):
    from oracle_py import lib
    L = lib()
    L.co_test_box_box_keep.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 6 + [ctypes.c_double, ctypes.c_void_p]
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in (p1, R1, s1, p2, R2, s2)]
    out = np.zeros(56)
    n = L.co_test_box_box_keep(keep, *[x.ctypes.data for x in a], 0.0, out.ctypes.data)
    return n, out.reshape(8, 7)[:n]


def test_box_box_eight_contact_option_keeps_the_whole_clipped_polygon():
    """CM_FLAG_BOX8 (MuJoCo's mjc_BoxBox returns up to eight points; reference model/cassie_tray_box.xml:213-216, :230-237 is where
    box meets box): a plate turned 45 degrees on the tray overlaps it in an OCTAGON -- eight rim crossings, all at the same depth:
    the option keeps the eight, the default its first four; a face wholly inside the other gives four either way."""
    plate = [0.12, 0.12, 0.05]
    args = (*TRAY, [0.0, 0.0, 0.05], _rot([0, 0, 1], np.pi / 4), plate)
    n8, c8 = _bb_keep(8, *args)
    n4, c4 = _bb_keep(4, *args)
    assert n8 == 8 and n4 == 4
    assert np.allclose(c8[:, 0], -0.005) and np.allclose(c8[:, 4:], [0, 0, 1])
    r = np.hypot(c8[:, 1], c8[:, 2])
    assert np.allclose(np.maximum(np.abs(c8[:, 1]), np.abs(c8[:, 2])), 0.14)            # every point on the tray's rim ...
    k = 0.12 * np.sqrt(2) - 0.14
    assert np.allclose(np.minimum(np.abs(c8[:, 1]), np.abs(c8[:, 2])), k)                # ... where the plate's edges cross it
    assert len({(round(x, 9), round(y, 9)) for x, y in c8[:, 1:3]}) == 8 and np.allclose(r, r[0])
    assert all(any(np.allclose(a, b) for b in c8) for a in c4)                           # the default's four are among them
    for Rz in (np.eye(3), _rot([0, 0, 1], 0.3)):                                         # the resting cube: four candidates, four contacts
        assert _bb_keep(8, *TRAY, [0.02, 0.03, 0.05], Rz, CUBE)[0] == 4
    # a corner of the cube past the tray's corner: a pentagon / hexagon -- more than four, fewer than eight
    n, c = _bb_keep(8, *TRAY, [0.12, 0.12, 0.05], _rot([0, 0, 1], 0.5), CUBE)
    assert 4 < n <= 8 and _bb_keep(4, *TRAY, [0.12, 0.12, 0.05], _rot([0, 0, 1], 0.5), CUBE)[0] == 4


def test_emulated_kernel_matches_oracle_with_the_eight_contact_option():
    """The cube sliding about the tray's corner under CM_FLAG_BOX8: the emulated kernel follows the oracle contact for contact
    through box-box contact sets of five and more points, and the option is not a no-op against the default there."""
    from cassie_amd import phys as P
    from cassie_amd._lib import CmModel
    tray = Model("cassie_tray_box")
    tray.set_flag(P.FLAG_BOX8, True)
    outs = []
    for flag in (True, False):
        pod = CmModel.from_buffer_copy(tray.pod)
        if not flag:
            pod.flags &= ~P.FLAG_BOX8
        q = tray.qpos_init()
        for i in range(3):
            pod.jnt_stiffness[i], pod.dof_damping[i], pod.qpos_spring[i] = 1e5, 1e4, q[i]
            pod.dof_stiffness[i], pod.dof_springref[i] = 1e5, q[i]
        for i in range(3, 6):
            pod.dof_damping[i] = 1e4
        for k in range(6, 32):
            pod.dof_damping[k] = 50
        q[35:38] = [0.13, 0.12, 1.01 + 0.17 + 0.005 + 0.05 + 0.002]       # just above the tray, over its corner
        q[38:42] = [np.cos(0.25), 0, 0, np.sin(0.25)]                      # turned about z
        o = Oracle(pod, q)
        o.qvel[32:35] = [-0.05, 0.03, 0]                                   # sliding slowly
        emu = EmuBatch(pod, 1)
        emu.qpos[:] = q
        emu.qvel[0, 32:35] = [-0.05, 0.03, 0]
        cube = tray.name2id(1, "cup_box")
        most = 0
        for s in range(150):
            emu.step()
            o.step()
            assert (emu.info[0, 0], emu.info[0, 1], emu.info[0, 2]) == (o.d.ncon, o.d.nefc, o.d.solver_iter), (flag, s)
            most = max(most, sum(1 for i in range(o.d.ncon) if cube in (pod.geom_bodyid[o.d.contact[i].geom1], pod.geom_bodyid[o.d.contact[i].geom2])))
        assert np.max(np.abs(emu.qpos[0] - o.qpos)) < 1e-9
        outs.append((most, o.qpos.copy()))
    assert outs[0][0] > 4 and outs[1][0] == 4
    assert np.max(np.abs(outs[0][1] - outs[1][1])) > 1e-9
