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
