"""Device-side drive-level I/O on the GPU: the encoder / motor models in the step kernel and
in the stand-alone drive pass against the host chain (csrc/cassie_hostpath.c, itself pinned bit for bit to the
reference's own compiled drive_encoder / joint_encoder / motor by tests/test_hostpath.py).

Bar: integer sensor path BIT-EXACT -- identical sensordata / actuator_velocity in => identical ctrl, measurement block
(the cassie_out_t fields), FIR / IIR histories and delay lines, memcmp.  tests/test_drive_io_emu.py is the CPU twin."""
import ctypes

import numpy as np
import pytest

import bench
from cassie_amd import Batch
from cassie_amd import iotypes as T
from cassie_amd import phys as P
from cassie_amd._lib import REPO_DIR, lib
from hostchain_py import HostChain, device_state_bytes, pd_command
from oracle_py import Oracle

pytestmark = pytest.mark.gpu
LIMIT = np.array([112.5, 112.5, 195.2, 195.2, 45.0] * 2)


def _cmd(rng, t, n):
    scale = 3.0 if 60 <= t < 80 else 0.4                      # beyond the torque limits for a while
    u = rng.uniform(-1, 1, (n, 10)) * LIMIT * scale
    sto = np.zeros(n)
    if 40 <= t < 55:
        sto[::2] = 1.0                                        # STO window on every other env
    return u, sto


@pytest.mark.parametrize("standalone", [False, True])
def test_torque_mode_bitwise_against_the_host_chain(cassie, standalone):
    """In-kernel (CM_DRIVE_TORQUE) and as the stand-alone pass + plain physics step: both are one
    cassie_sim_step_ethercat per step for every env."""
    n = 24
    rng = np.random.default_rng(1)
    b = Batch(cassie, n)
    q0 = np.tile(cassie.qpos_init(), (n, 1))
    q0[:, 2] -= rng.uniform(0, 0.02, n)
    b.set(P.F_QPOS, q0)
    b.set(P.F_QVEL, rng.uniform(-0.5, 0.5, (n, cassie.pod.nv)))
    b.forward()
    if not standalone:
        b.set_drive_mode(P.DRIVE_TORQUE)
    chains = [HostChain(cassie) for _ in range(n)]
    for t in range(100):
        u, sto = _cmd(rng, t, n)
        sd, av = b.get(P.F_SENSORDATA), b.get(P.F_ACTUATOR_VELOCITY)          # what the previous step left in HBM
        b.set(P.F_DRIVE_CMD, np.concatenate([u, sto[:, None]], axis=1))
        if standalone:
            b.drive_pass(P.DRIVE_TORQUE)
            ctrl_dev = b.get(P.F_CTRL)
        b.step(1)
        meas = b.get(P.F_MEAS)
        st = b.get_drive_state()
        for e, hc in enumerate(chains):
            ctrl, m, _ = hc.ethercat(u[e], bool(sto[e]), sd[e], av[e])
            assert meas[e].tobytes() == m.tobytes(), (t, e)
            assert device_state_bytes(st[e]) == hc.state_bytes(), (t, e)
            if standalone:
                assert ctrl_dev[e].tobytes() == ctrl.tobytes(), (t, e)
    w, _ = b.warnings()
    assert not w.any()
    b.close()
    for hc in chains:
        hc.close()


def test_fused_substeps_equal_single_steps(cassie):
    n = 16
    rng = np.random.default_rng(2)
    cmd = np.concatenate([rng.uniform(-0.5, 0.5, (n, 10)) * LIMIT, np.zeros((n, 1))], axis=1)
    out = []
    for fused in (True, False):
        b = Batch(cassie, n)
        b.set(P.F_QPOS, np.tile(cassie.qpos_init(), (n, 1)))
        b.forward()
        b.set_drive_mode(P.DRIVE_TORQUE)
        b.set(P.F_DRIVE_CMD, cmd)
        if fused:
            b.step(50)
        else:
            for _ in range(50):
                b.step(1)
        out.append((b.get(P.F_QPOS), b.get(P.F_MEAS), [device_state_bytes(s) for s in b.get_drive_state()], b.get(P.F_SENSORDATA), b.get(P.F_XQUAT)))
        b.close()
    assert out[0][0].tobytes() == out[1][0].tobytes() and out[0][1].tobytes() == out[1][1].tobytes() and out[0][2] == out[1][2]
    # outputs of a fused launch = its last substep's, IMU words and body quaternions included (evaluated by the last two substeps only)
    assert out[0][3].tobytes() == out[1][3].tobytes() and out[0][4].tobytes() == out[1][4].tobytes()


def test_pd_on_measurements_mode_against_host_chain_and_oracle(cassie):
    """CM_DRIVE_PD, the device-resident form of cassie_sim_step_pd's motor-PD path (pd_input's PD law on the encoder
    measurements of the previous step -> motor model with delay -> physics), 50 fused substeps per launch, 1000 steps,
    against numpy PD + host chain + oracle physics run free (bench.HostChainEnvs).  Agreement is to ROUNDING (1e-9
    relative asserted) for every env whose encoder inputs stayed clear of a count boundary -- the replay watches that, see
    tests/test_drive_parity_gpu.py -- and to the encoder quantisation (2e-4) only for an env that did not."""
    n = 12
    ids = np.arange(n)
    tg = bench.pd_targets(ids, 20)
    b = Batch(cassie, n)
    b.set(P.F_QPOS, np.tile(cassie.qpos_init(), (n, 1)))
    b.forward()
    b.set(P.F_PD_KP, np.tile(bench.PD_KP, (n, 1)))
    b.set(P.F_PD_KD, np.tile(bench.PD_KD, (n, 1)))
    b.set_drive_mode(P.DRIVE_PD)
    ref = bench.HostChainEnvs(cassie, ids)
    for p in range(20):
        b.set(P.F_PD_PTARGET, tg[p])
        b.step(50)
        ref.step(50, tg[p])
        q, qr = b.get(P.F_QPOS), ref.qpos()
        w, info = b.warnings()
        assert np.array_equal(info[:, :3], ref.counts()), p
        safe = ref.flip_margin > 1e-6
        err = np.max(np.abs(q - qr) / np.maximum(1.0, np.abs(qr)), axis=1)
        assert np.all(err[safe] <= 1e-9), (p, float(err[safe].max()))
        assert np.all(np.max(np.abs(q - qr), axis=1)[~safe] < 2e-4), p
    assert safe.sum() >= n - 1                      # a count-boundary graze is a ~1e-5 event per env
    assert not w.any()
    assert np.all(b.get(P.F_QPOS)[:, 2] > 0.6)
    # the launch left the torques its last substep applied in the ctrl field (mj_forward reads d->ctrl): a forward pass /
    # phys_batch_derive after a drive-mode step must evaluate the state under THOSE torques
    applied = np.array([o.ctrl.copy() for o in ref.orcs])
    assert np.max(np.abs(b.get(P.F_CTRL) - applied)) < 1e-9 and np.abs(applied).max() > 1.0
    ids6 = [cassie.name2id(1, "left-foot"), cassie.name2id(1, "right-foot"), cassie.name2id(6, "left-heel"), cassie.name2id(6, "right-heel"),
            cassie.name2id(6, "left-toe"), cassie.name2id(6, "right-toe")]
    b.set_drive_mode(P.DRIVE_OFF)
    b.derive(ids6)
    b.sync()
    D = b.get(P.F_DERIVED)
    qacc = b.get(P.F_QACC)
    for e, o in enumerate(ref.orcs):
        o.forward()
        assert np.max(np.abs(qacc[e] - o.qacc)) < 1e-6 * max(1.0, np.abs(o.qacc).max()), e
        ftot = sum(sum(o.d.efc_force[o.d.contact[c].efc_address + i] for i in range(4)) * o.d.contact[c].frame[2] for c in range(o.d.ncon))
        fz = D[e, P.DRV_FOOT_FORCE + 2] + D[e, P.DRV_FOOT_FORCE + 8]
        assert abs(fz - ftot) < 1e-6 * max(1.0, abs(ftot)) and fz > 100, e
    b.close()
    for hc in ref.chains:
        hc.close()


def test_batched_step_pd_with_device_drives_equals_host_drives(built):
    """cassie_batch_step_pd with the encoder / motor models on the device (cassie_batch_set_device_drives) returns the
    same state_out_t bytes as with the models on the host threads, through an STO-free random PD rollout, and the
    drive state moves between host and device without loss when the mode is switched mid-run."""
    L = lib()
    VP = ctypes.c_void_p
    L.cassie_batch_create.restype = VP
    L.cassie_batch_create.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.cassie_batch_free.argtypes = [VP]
    L.cassie_batch_step_pd.argtypes = [VP, VP, VP]
    L.cassie_batch_set_device_drives.argtypes = [VP, ctypes.c_int]
    import os
    model = os.path.join(REPO_DIR, "models", "cassie.cmodel").encode()
    n = 32
    tg = bench.pd_targets(np.arange(n), 8)
    u = np.zeros((n, 119))
    for base in (30, 85):
        u[:, base + 15: base + 20] = bench.PD_KP[:5]
        u[:, base + 20: base + 25] = bench.PD_KD[:5]
    outs = []
    for schedule in ("host", "device", "mixed"):
        bt = L.cassie_batch_create(model, n, 0, 4)
        assert bt
        y = np.zeros((n, 124))
        traj = []
        for s in range(300):
            if schedule == "device" and s == 0:
                assert L.cassie_batch_set_device_drives(bt, 1) == 0
            if schedule == "mixed" and s in (70, 140, 210):
                assert L.cassie_batch_set_device_drives(bt, 1 if s != 140 else 0) == 0
            if s % 50 == 0:
                u[:, 35:40] = tg[s // 50][:, :5]
                u[:, 90:95] = tg[s // 50][:, 5:]
            assert L.cassie_batch_step_pd(bt, u.ctypes.data, y.ctypes.data) == 0
            traj.append(y.copy())
        L.cassie_batch_free(bt)
        outs.append(np.array(traj))
    assert outs[0].tobytes() == outs[1].tobytes()
    assert outs[0].tobytes() == outs[2].tobytes()
    assert np.all(np.isfinite(outs[0]))


def test_batched_step_and_step_ethercat_with_device_drives(built):
    """cassie_batch_step (cassie_user_in_t) and cassie_batch_step_ethercat (cassie_in_t) with the drive-level models on the
    device return the same cassie_out_t bytes as with the models on the host threads."""
    import os
    L = lib()
    VP = ctypes.c_void_p
    L.cassie_batch_create.restype = VP
    L.cassie_batch_create.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.cassie_batch_free.argtypes = [VP]
    L.cassie_batch_step.argtypes = [VP, VP, VP]
    L.cassie_batch_step_ethercat.argtypes = [VP, VP, VP]
    L.cassie_batch_set_device_drives.argtypes = [VP, ctypes.c_int]
    model = os.path.join(REPO_DIR, "models", "cassie.cmodel").encode()
    n = 16
    rng = np.random.default_rng(11)
    for fn, utype, field in ((L.cassie_batch_step, T.cassie_user_in_t, "torque"), (L.cassie_batch_step_ethercat, T.cassie_in_t, None)):
        outs = []
        cmds = rng.uniform(-1, 1, (6, n, 10)) * LIMIT * 0.3
        for device in (0, 1):
            bt = L.cassie_batch_create(model, n, 0, 4)
            assert bt and L.cassie_batch_set_device_drives(bt, device) == 0
            us = (utype * n)()
            ys = (T.cassie_out_t * n)()
            traj = []
            for s in range(150):
                if s % 25 == 0:
                    for e in range(n):
                        c = cmds[s // 25][e]
                        if field:
                            for i in range(10):
                                us[e].torque[i] = c[i]
                        else:
                            legs = [us[e].leftLeg, us[e].rightLeg]
                            for i in range(10):
                                leg = legs[i // 5]
                                [leg.hipRollDrive, leg.hipYawDrive, leg.hipPitchDrive, leg.kneeDrive, leg.footDrive][i % 5].torque = c[i]
                assert fn(bt, ctypes.byref(us), ctypes.byref(ys)) == 0
                traj.append(bytes(ys))
            L.cassie_batch_free(bt)
            outs.append(traj)
        assert outs[0] == outs[1]


def test_outputs_do_not_depend_on_eliding_unread_substep_outputs(cassie):
    """A fused launch evaluates the IMU sensors and body quaternions only on the substeps whose values are read (the last
    two); phys_batch_set_all_outputs_every_substep forms them on every substep.  State and every returned output must be
    bit for bit the same either way, launch after launch."""
    n = 32
    tg = bench.pd_targets(np.arange(n), 4)
    out = []
    for every in (False, True):
        b = Batch(cassie, n)
        b.set(P.F_QPOS, np.tile(cassie.qpos_init(), (n, 1)))
        b.forward()
        b.set(P.F_PD_KP, np.tile(bench.PD_KP, (n, 1)))
        b.set(P.F_PD_KD, np.tile(bench.PD_KD, (n, 1)))
        b.set_drive_mode(P.DRIVE_PD)
        b.set_all_outputs_every_substep(every)
        snaps = []
        for p in range(3):
            b.set(P.F_PD_PTARGET, tg[p])
            b.step(50)
            snaps.append(b"".join(b.get(f).tobytes() for f in (P.F_QPOS, P.F_QVEL, P.F_SENSORDATA, P.F_MEAS, P.F_XQUAT, P.F_XPOS, P.F_ACTUATOR_VELOCITY)))
        out.append(snaps)
        b.close()
    assert out[0] == out[1]
