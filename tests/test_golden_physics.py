"""Frozen physics (tests/golden/physics_v1.npz, tests/golden_physics.py): the regression guard while the oracle cannot be
pinned against MuJoCo itself.  CPU: the oracle must reproduce the committed file BIT FOR BIT (a changed definition needs a
visible new version of the file) and the emulated kernel must stay within 1e-7 of it; GPU: the HIP kernel, through the
C ABI, within 1e-7 relative on every recorded step of all three models in both device modes, with equal
(ncon, nefc, solver iterations)."""
import ctypes

import numpy as np
import pytest

import bench
import golden_physics as G
from cassie_amd import Model
from cassie_amd import phys as P

REL_TOL = 1e-7


@pytest.fixture(scope="module")
def golden():
    """Both files under one roof: physics_v1.npz (the default definitions) and physics_v2.npz (CM_FLAG_HFPRISM, round 5)."""
    g = G.load()
    assert int(g["meta/version"]) == G.VERSION and tuple(g["meta/checkpoints"]) == G.CHECKPOINTS and int(g["meta/nenv"]) == G.NENV
    g2 = G.load(G.PATH2)
    assert int(g2["meta/version"]) == G.VERSION2 and tuple(g2["meta/checkpoints"]) == G.CHECKPOINTS and int(g2["meta/nenv"]) == G.NENV
    g.update({k: v for k, v in g2.items() if not k.startswith("meta/")})
    return g


ALL_MODELS = G.MODELS + G.MODELS2


@pytest.mark.parametrize("name", ALL_MODELS)
@pytest.mark.parametrize("mode", G.MODES)
def test_oracle_reproduces_the_golden_file_bit_for_bit(built, golden, name, mode):
    rec = G.oracle_rollout(G.model_of(name), name, mode)
    for field, v in rec.items():
        want = golden[G.key(name, mode, field)]
        assert v.dtype == want.dtype and v.shape == want.shape, field
        assert np.array_equal(v.view(np.int64) if v.dtype == np.float64 else v, want.view(np.int64) if want.dtype == np.float64 else want), \
            "%s %s %s differs from %s: a deliberate change needs physics_v%d" % (name, mode, field, G.PATH, G.VERSION + 1)


def test_golden_file_is_a_real_workload(golden):
    """Not a trivially green file: contacts everywhere at the end, rough-terrain envs with more rows than the flat ones, box
    contacts in the tray model, and the two modes differ (the torque delay and the encoder quantisation act)."""
    for name in G.MODELS:
        c = golden[G.key(name, "exact-pd", "counts")]
        assert c[-1][:, 0].min() >= 2 and c[-1][:, 1].min() >= 20
        assert not np.array_equal(golden[G.key(name, "exact-pd", "qpos")][-1], golden[G.key(name, "drive-pd", "qpos")][-1])
    assert golden[G.key("cassie_hfield", "exact-pd", "counts")][-1][:, 1].max() > 24
    assert golden[G.key("cassie_tray_box", "exact-pd", "counts")][-1][:, 1].max() >= 36
    assert np.all(np.isfinite(golden[G.key("cassie_tray_box", "drive-pd", "qpos")]))
    # the prism option: more rows than one wavefront has lanes at some checkpoint, and more contacts than the default definition gives
    for mode in G.MODES:
        cp, cd = golden[G.key("cassie_hfield_prism", mode, "counts")], golden[G.key("cassie_hfield", mode, "counts")]
        assert cp[:, :, 1].max() > 64 and cp[-1][:, 0].sum() > cd[-1][:, 0].sum()


def _compare(name, mode, golden, got, upto=None):
    for ci, step in enumerate(G.CHECKPOINTS):
        if upto is not None and step > upto:
            break
        for field in ("qpos", "qvel", "sensordata"):
            want = golden[G.key(name, mode, field)][ci]
            err = np.max(np.abs(got[field][ci] - want) / np.maximum(1.0, np.abs(want)))
            assert err <= REL_TOL, (name, mode, field, step, err)
        assert np.array_equal(got["counts"][ci], golden[G.key(name, mode, "counts")][ci]), (name, mode, step)


def _emu_rollout(model, name, mode, upto):
    from cassie_amd._lib import CmDriveState  # noqa: F401  (layout shared with the device)
    from emu_py import EmuBatch
    pod = model.pod
    emu = EmuBatch(pod, G.NENV)
    emu.qpos[:] = G.initial_qpos(model, name)
    hf = G.terrain(name)
    if hf is not None:
        emu.hfield = np.ascontiguousarray(hf.reshape(-1))
    emu.pd_kp, emu.pd_kd = np.tile(bench.PD_KP, (G.NENV, 1)), np.tile(bench.PD_KD, (G.NENV, 1))
    if mode == "drive-pd":
        emu.forward()                                   # the init pose's sensordata, as cassie_sim_init leaves it
        emu.drive_mode = P.DRIVE_PD
    tg = G.targets()
    rec = {k: [] for k in ("qpos", "qvel", "sensordata", "counts")}
    for first, n in G.segments():
        if first + n > upto:
            break
        emu.pd_ptarget = np.ascontiguousarray(tg[first // bench.HOLD])
        emu.step(n)
        if first + n in G.CHECKPOINTS:
            rec["qpos"].append(emu.qpos.copy()); rec["qvel"].append(emu.qvel.copy()); rec["sensordata"].append(emu.sensordata.copy())
            rec["counts"].append(emu.info[:, :3].astype(np.int64))
    return rec


@pytest.mark.parametrize("name", ALL_MODELS)
@pytest.mark.parametrize("mode", G.MODES)
def test_emulated_kernel_against_the_golden_file(built, golden, name, mode):
    """The kernel source under the wave emulator, first 50 steps of every model / mode (the full 1000 are the GPU test's)."""
    _compare(name, mode, golden, _emu_rollout(G.model_of(name), name, mode, 50), upto=50)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ALL_MODELS)
@pytest.mark.parametrize("mode", G.MODES)
def test_hip_kernel_against_the_golden_file(built, golden, name, mode):
    from cassie_amd import Batch
    model = G.model_of(name)
    b = Batch(model, G.NENV)
    try:
        hf = G.terrain(name)
        if hf is not None:
            b.set_hfield(hf)
        b.set(P.F_QPOS, G.initial_qpos(model, name))
        b.set(P.F_PD_KP, np.tile(bench.PD_KP, (G.NENV, 1)))
        b.set(P.F_PD_KD, np.tile(bench.PD_KD, (G.NENV, 1)))
        if mode == "drive-pd":
            b.forward()
            b.set_drive_mode(P.DRIVE_PD)
        else:
            b.set_pd_mode(True)
        tg = G.targets()
        rec = {k: [] for k in ("qpos", "qvel", "sensordata", "counts")}
        for first, n in G.segments():
            b.set(P.F_PD_PTARGET, tg[first // bench.HOLD])
            b.step(n)
            if first + n in G.CHECKPOINTS:
                w, info = b.warnings()
                assert not (w & ~(P.WARN_CONTACT_FULL | P.WARN_CONSTRAINT_FULL if name in G.MODELS2 else 0)).any()   # (the lying robot of the prism file reaches its 127 rows)
                rec["qpos"].append(b.get(P.F_QPOS)); rec["qvel"].append(b.get(P.F_QVEL)); rec["sensordata"].append(b.get(P.F_SENSORDATA))
                rec["counts"].append(info[:, :3].astype(np.int64))
        _compare(name, mode, golden, rec)
    finally:
        b.close()
