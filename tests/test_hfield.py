"""Height-field contact path (BASELINE config 4, cassie_hfield.xml): the terrain of reference
example/test_hfield.py:39-41 (random heights, central 10 x 10 samples zeroed), oracle sanity and parity of
the emulated kernel."""
import numpy as np
import pytest

import oracle_py
from cassie_amd import Model
from emu_py import EmuBatch
from oracle_py import Oracle


def terrain(flat=False):
    h = np.random.default_rng(99).random((200, 200)).astype(np.float32)
    h[95:105, 95:105] = 0
    if flat:
        h[:] = 0
    return h


@pytest.fixture(scope="module")
def hf(built):
    return Model("cassie_hfield")


def test_flat_heightfield_behaves_like_the_floor_plane(hf, cassie):
    """All-zero samples: the terrain is the plane z = -0.1 (geom pos), so the robot dropped from the init pose
    must behave exactly like cassie.xml dropped on a floor moved to z = -0.1."""
    oracle_py.set_hfield(terrain(flat=True))
    try:
        o = Oracle(hf.pod, hf.qpos_init())
        from cassie_amd._lib import CmModel
        pl = CmModel.from_buffer_copy(cassie.pod)
        pl.geom_pos[0][2] = -0.1                                 # floor plane is collision geom 0 of cassie.xml
        p = Oracle(pl, cassie.qpos_init())
        for _ in range(400):
            o.step()
            p.step()
        assert o.d.ncon > 0
        assert np.max(np.abs(o.qpos - p.qpos)) < 1e-9
    finally:
        oracle_py.set_hfield(None)


def test_emulated_kernel_matches_oracle_on_rough_terrain(hf):
    h = terrain()
    oracle_py.set_hfield(h)
    try:
        pod = hf.pod
        q = hf.qpos_init()
        q[0] = 0.35                                              # start over the rough part, straddling the flat patch edge
        o = Oracle(pod, q)
        emu = EmuBatch(pod, 1)
        emu.qpos[:] = q
        emu.hfield = h.ravel().copy()
        seen = 0
        for s in range(500):
            emu.step()
            o.step()
            assert (emu.info[0, 0], emu.info[0, 1]) == (o.d.ncon, o.d.nefc), s
            seen = max(seen, o.d.ncon)
        assert seen >= 2
        assert np.max(np.abs(emu.qpos[0] - o.qpos)) < 1e-8
        nz = [o.d.contact[i].frame[2] for i in range(o.d.ncon)]
        assert all(0 < z <= 1 for z in nz)
    finally:
        oracle_py.set_hfield(None)


def test_no_terrain_data_means_no_contact(hf):
    oracle_py.set_hfield(None)
    o = Oracle(hf.pod, hf.qpos_init())
    o.step(300)
    assert o.d.ncon == 0 and o.qpos[2] < 0.9                     # falls freely without samples
