"""Per-env domain randomisation (reference src/cassiemujoco.c:1323-1436 setters, :949-977 mj_setConst) on the
CPU: the device's set_const kernel and the step kernel's reads through the per-env parameter block, executed by the wave
emulator, against the host model compiler and the oracle.  The GPU counterpart is tests/test_randomise_gpu.py."""
import ctypes

import numpy as np
import pytest

import emu_py
import randomise_check as rc
from cassie_amd import Model
from cassie_amd._lib import CmEnvParams, CmModel
from oracle_py import Oracle


def _emu():
    L = emu_py.lib()
    L.emu_set_const.argtypes = [ctypes.POINTER(CmModel), ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    L.emu_set_envparams.argtypes = [ctypes.c_void_p]
    L.emu_sizeof_envparams.restype = ctypes.c_ulong
    return L


def test_parameter_block_layout_is_shared_by_header_emulator_and_library():
    assert _emu().emu_sizeof_envparams() == ctypes.sizeof(CmEnvParams)
    # the model's own block is what the compile derives for the model's own parameters
    for name in ("cassie", "cassie_tray_box"):
        pod = Model(name).pod
        p = rc.params_as_arrays(pod.params, pod)
        assert np.array_equal(p["body_mass"], np.array(pod.body_mass[: pod.nbody]))
        assert p["meaninertia"][0] == pod.meaninertia
        assert np.array_equal(p["pair_invweight"], np.array(pod.pair_invweight[: pod.npair]))
        assert np.array_equal(p["jnt_liminvweight"], np.array(pod.jnt_liminvweight[: pod.njnt]))


@pytest.mark.parametrize("name", ["cassie", "cassie_hfield", "cassie_tray_box"])
def test_device_set_const_reproduces_the_host_compile_bit_for_bit(name):
    """mj_setConst per env as the device kernel computes it (M(qpos0), Cholesky, inverse weights, the per-pair tables) equals
    phys_model_set_const + phys_model_compile of a host model with the same parameters: every field, every bit."""
    nenv = 5
    hosts = rc.HostEnvModels(name)
    pod0 = Model(name).pod
    params = rc.random_params(hosts.m, nenv, seed=11)
    blocks = rc.new_blocks(pod0, nenv, params)
    _emu().emu_set_const(ctypes.byref(pod0), ctypes.addressof(blocks), nenv, 1)
    for e in range(nenv):
        want = hosts.pod(params, e)
        rc.assert_blocks_equal(blocks[e], want.params, pod0, "%s env %d" % (name, e))
    # the randomisation moved what it should: inverse weights follow the masses, the regularisers' inputs differ env by env
    mi = [blocks[e].meaninertia for e in range(nenv)]
    assert len(set(mi)) == nenv and all(abs(x / pod0.meaninertia - 1) < 0.25 for x in mi)
    # unchanged parameters give the model's own block back (the kernel is the compile's arithmetic, not an approximation of it)
    same = rc.new_blocks(pod0, 1, {f: [rc.params_as_arrays(pod0.params, pod0)[f]] for f in rc.INPUT_FIELDS})
    for f in rc.DERIVED_FIELDS:   # start from garbage: everything derived must be rewritten
        if f == "meaninertia":
            same[0].meaninertia = -1.0
        else:
            np.ctypeslib.as_array(getattr(same[0], f)).reshape(-1)[:] = -1.0
    _emu().emu_set_const(ctypes.byref(pod0), ctypes.addressof(same), 1, 1)
    rc.assert_blocks_equal(same[0], pod0.params, pod0, name + " unchanged parameters")


def test_friction_alone_needs_no_set_const():
    """Friction acts through mj_contactParam at every step in the reference: the device refreshes the pairs' mixed values on
    phys_batch_randomize itself (derive_inertial = 0) and leaves the inverse weights alone."""
    hosts = rc.HostEnvModels("cassie")
    pod0 = Model("cassie").pod
    params = rc.random_params(hosts.m, 2, seed=5, mass=0.0, ipos=0.0, damping=0.0)
    blocks = rc.new_blocks(pod0, 2, params)
    _emu().emu_set_const(ctypes.byref(pod0), ctypes.addressof(blocks), 2, 0)
    for e in range(2):
        want = hosts.pod(params, e, set_const=False)
        rc.assert_blocks_equal(blocks[e], want.params, pod0, "env %d" % e)
        assert blocks[e].meaninertia == pod0.meaninertia


@pytest.mark.parametrize("name,two_waves", [("cassie", 1), ("cassie", 0), ("cassie_tray_box", 1)])
def test_step_kernel_reads_the_env_block_like_a_per_env_model(name, two_waves):
    """The same kernel code stepping (a) the shared model + per-env parameter blocks derived on the 'device' and (b) a per-env
    compiled model each -- the round-5 way -- gives the same bits, and both follow the oracle run on the per-env models."""
    nenv, nsub, nlaunch = 3, 10, 6
    hosts = rc.HostEnvModels(name)
    model = Model(name)
    pod0 = model.pod
    params = rc.random_params(hosts.m, nenv, seed=3)
    blocks = rc.new_blocks(pod0, nenv, params)
    L = _emu()
    L.emu_set_const(ctypes.byref(pod0), ctypes.addressof(blocks), nenv, 1)
    L.emu_two_waves(two_waves)
    L.emu_fast_rows(1)
    rng = np.random.default_rng(1)
    hi = np.array([pod0.act_ctrlrange[u][1] for u in range(pod0.nu)])
    ctrl = 0.6 * hi * rng.uniform(-1, 1, (nenv, pod0.nu))
    q0 = model.qpos_init()
    if name == "cassie_tray_box":
        q0 = np.array(pod0.qpos0[: pod0.nq]); q0[7:35] = model.qpos_init()[7:35]
    try:
        a = emu_py.EmuBatch(pod0, nenv)
        a.qpos[:] = q0; a.ctrl[:] = ctrl
        L.emu_set_envparams(ctypes.addressof(blocks))
        for _ in range(nlaunch):
            a.step(nsub)
        L.emu_set_envparams(None)
        for e in range(nenv):
            pe = hosts.pod(params, e)
            b = emu_py.EmuBatch(pe, 1)
            b.qpos[:] = q0; b.ctrl[:] = ctrl[e]
            for _ in range(nlaunch):
                b.step(nsub)
            assert np.array_equal(a.qpos[e], b.qpos[0]) and np.array_equal(a.qvel[e], b.qvel[0]), "env %d: block and per-env model part" % e
            assert np.array_equal(a.info[e], b.info[0])
            o = Oracle(pe, q0)
            o.ctrl[:] = ctrl[e]
            o.step(nsub * nlaunch)
            assert np.max(np.abs(a.qpos[e] - o.qpos)) < 1e-10, "env %d against the oracle" % e
        # ... and the randomisation is not a no-op: the envs part from each other and from the unrandomised model
        c = emu_py.EmuBatch(pod0, 1)
        c.qpos[:] = q0; c.ctrl[:] = ctrl[0]
        for _ in range(nlaunch):
            c.step(nsub)
        assert np.max(np.abs(c.qpos[0] - a.qpos[0])) > 1e-6
    finally:
        L.emu_set_envparams(None)
        L.emu_two_waves(0)
        L.emu_fast_rows(0)
