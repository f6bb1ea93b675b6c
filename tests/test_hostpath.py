"""Host half of the step (H2-H8 of SURVEY.md 8a) against golden vectors generated from the REFERENCE's own
code (tools/make_golden_hostpath.py -> oracle/_ref: the reference's drive_encoder / joint_encoder / motor
compiled from src/cassiemujoco.c, plus the Agility blocks of libagilitycassie.a).  Bit-exact: memcmp of
cassie_out_t, its packed wire form, state_out_t and the ctrl doubles."""
import ctypes
import os

import numpy as np
import pytest

from cassie_amd import iotypes as T
from cassie_amd._lib import REPO_DIR, lib

GOLDEN = os.path.join(REPO_DIR, "tests", "golden", "hostpath_v1.npz")


class HostModel(ctypes.Structure):
    _fields_ = [("drive_bits", ctypes.c_int * 10), ("joint_bits", ctypes.c_int * 6), ("gear", ctypes.c_double * 10),
                ("tmax", ctypes.c_double * 10), ("wmax", ctypes.c_double * 10)]


@pytest.fixture(scope="module")
def host(cassie):
    L = lib()
    VP = ctypes.c_void_p
    L.cassie_hostenv_alloc.restype = VP
    L.cassie_hostenv_cassie_out.restype = ctypes.POINTER(T.cassie_out_t)
    L.cassie_hostenv_cassie_out.argtypes = [VP]
    L.cassie_hostenv_free.argtypes = [VP]
    L.cassie_hostmodel_from_model.argtypes = [VP, VP]
    L.cassie_hostenv_step_pd_pre.argtypes = [VP] * 7
    L.cassie_hostenv_step_pd_post.argtypes = [VP] * 3
    L.pack_cassie_out_t.argtypes = [VP, VP]
    hm = HostModel()
    assert L.cassie_hostmodel_from_model(cassie._h, ctypes.byref(hm)) == 0
    return L, hm


def test_hostmodel_constants(host):
    _, hm = host
    assert list(hm.drive_bits) == [13, 13, 13, 13, 18] * 2 and list(hm.joint_bits) == [18, 18, 13] * 2
    assert list(hm.gear) == [25, 25, 16, 16, 50] * 2
    assert np.allclose(list(hm.wmax), np.array([2900, 2900, 1300, 1300, 5500] * 2) * 2 * np.pi / 60)


def test_step_pd_host_chain_matches_reference_bitwise(host):
    L, hm = host
    g = np.load(GOLDEN)
    env = L.cassie_hostenv_alloc()
    nsteps = g["ctrl"].shape[0]
    for t in range(nsteps):
        if t == 150:
            L.cassie_hostenv_cassie_out(env).contents.pelvis.radio.channel[8] = 0   # STO, as in the generator
        if t == 170:
            L.cassie_hostenv_cassie_out(env).contents.pelvis.radio.channel[8] = 1
        u = T.pd_in_t.from_buffer_copy(g["pd_in"][t].tobytes())
        sd, av = np.ascontiguousarray(g["sensordata"][t]), np.ascontiguousarray(g["actvel"][t])
        ctrl = np.zeros(10)
        y, so = T.cassie_out_t(), T.state_out_t()
        L.cassie_hostenv_step_pd_pre(env, ctypes.byref(hm), ctypes.byref(u), sd.ctypes.data, av.ctypes.data, ctrl.ctypes.data,
                                     ctypes.byref(y))
        L.cassie_hostenv_step_pd_post(env, ctypes.byref(y), ctypes.byref(so))
        assert ctrl.tobytes() == g["ctrl"][t].tobytes(), "ctrl differs at step %d" % t
        assert bytes(y) == g["cassie_out"][t].tobytes(), "cassie_out_t differs at step %d" % t
        packed = (ctypes.c_ubyte * 697)()
        L.pack_cassie_out_t(ctypes.byref(y), packed)
        assert bytes(packed) == g["packed"][t].tobytes(), "packed cassie_out_t differs at step %d" % t
        assert bytes(so) == g["state_out"][t].tobytes(), "state_out_t differs at step %d" % t
    L.cassie_hostenv_free(env)


def test_golden_covers_the_interesting_regimes():
    g = np.load(GOLDEN)
    ctrl = g["ctrl"]
    assert np.any(ctrl[150:176] == 0) and np.any(ctrl[100:140] != 0)        # STO window zeroes the torques
    assert np.abs(ctrl).max() <= 12.2 + 1e-12                                # motor-side torque limits respected
    assert (np.abs(ctrl[205:]).max(axis=0) < np.array([4.5, 4.5, 12.2, 12.2, 0.9] * 2)).any()  # speed-torque curve active
