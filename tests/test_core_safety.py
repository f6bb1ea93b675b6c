"""cassie_core_sim's safety layer (reference include/cassie_core_sim.h:30-35, called at
src/cassiemujoco.c:1141) restated in csrc/pk_safety.h for the device-resident mode CM_DRIVE_PD_SAFE, held BIT FOR BIT to the
closed binary src/libagilitycassie.a(cassie_core_sim.o):

  * tests/golden/core_safety_v1.npz -- inputs and outputs of the real block (tools/make_golden_core_safety.py): hand-made
    corners (every joint at, one ulp inside and beyond each bound, violations of exactly the blend width, torques exactly at the
    limits, signed zeros, zero / negative limits, STO) and 3 000 mixed samples; the ten torques must be identical in every bit
    and the message queue (radio channels 1-4) equal;
  * the live binary where oracle/_ref holds it (this container; the GPU box gets the prebuilt file): 2 x 10^6 fresh samples here,
    10^7 by tools/core_safety_soak.py (profiles/round6/core_safety_soak.txt);
  * the emulated step kernel in CM_DRIVE_PD_SAFE against the HOST chain with the real block in it (pd law -> cassie_core_sim_step
    -> cassie_motor_data / cassie_sensor_data): ctrl, measurement block, filter histories, delay lines, message bits."""
import os

import numpy as np
import pytest

import core_safety_check as C
from cassie_amd import phys as P

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "core_safety_v1.npz")


def _bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint64), np.ascontiguousarray(b).view(np.uint64))


@pytest.mark.parametrize("part", ["adv", "mix"])
def test_restatement_reproduces_the_real_block_on_the_golden_vectors(part):
    g = np.load(GOLD)
    u, q, w, L, ch8 = (g[part + "_" + k] for k in ("u", "q", "w", "L", "ch8"))
    tau, msg = C.restated(u, q, w, L, ch8 != 1.0)
    bad = np.nonzero(np.any(tau.view(np.uint64) != g[part + "_tau"].view(np.uint64), axis=1))[0]
    assert bad.size == 0, (part, int(bad[0]), q[bad[0]].tolist(), tau[bad[0]].tolist(), g[part + "_tau"][bad[0]].tolist())
    radio = g[part + "_radio"]
    assert np.array_equal(C.queue_of(msg), radio[:, 1:5])
    assert not radio[:, 0].any() and not radio[:, 5:].any()
    # the vectors exercise the law: violated constraints, saturated torques, STO (a zero with the torque's sign), untouched pass-through
    assert (msg & 1).any() and (msg & 2).any() and not msg.all()
    if part == "mix":
        clean = (msg == 0) & (ch8 == 1.0)
        assert clean.sum() > 500 and np.array_equal(tau[clean], u[clean])      # (as values: a zero torque may change the sign of its zero)
        assert np.signbit(tau[ch8 != 1.0]).any() and not tau[ch8 != 1.0].any()


def test_message_queue_is_sticky_and_sorted_like_the_real_blocks():
    """One block instance over 400 steps: codes 635 / 630 enter the 4-deep queue once, highest first, and stay; the telemetry
    shorts pass through to radio channels 5-13.  The device keeps the same information as two sticky bits
    (cm_drive_state_t::safety_msg)."""
    g = np.load(GOLD)
    u, q, w, L, ch8 = (g["seq_" + k] for k in ("u", "q", "w", "L", "ch8"))
    tau, msg = C.restated(u, q, w, L, ch8 != 1.0)
    assert _bits_equal(tau, g["seq_tau"])
    sticky = np.bitwise_or.accumulate(msg)
    assert np.array_equal(C.queue_of(sticky), g["seq_radio"][:, 1:5])
    assert np.array_equal(g["seq_radio"][:, 5:14], g["seq_tel"]) and not g["seq_radio"][:, 0].any()


@pytest.mark.skipif(not C.have_live_binary(), reason="oracle/_ref/libref_hostpath.so (the reference's Agility library) is not here")
def test_restatement_against_the_live_binary_two_million_samples():
    worst = 0
    for seed in range(4):
        u, q, w, L, ch8 = C.samples(500000, 1000 + seed)
        tau, radio, flags, cw = C.live(u, q, w, L, ch8)
        mine, msg = C.restated(u, q, w, L, ch8 != 1.0)
        bad = np.nonzero(np.any(tau.view(np.uint64) != mine.view(np.uint64), axis=1))[0]
        assert bad.size == 0, (seed, int(bad[0]), q[bad[0]].tolist(), u[bad[0]].tolist(), w[bad[0]].tolist(), tau[bad[0]].tolist(), mine[bad[0]].tolist())
        assert np.array_equal(C.queue_of(msg), radio[:, 1:5]) and not flags.any() and not cw.any()
        worst = max(worst, int((msg & 1).sum()))
    assert worst > 100000        # (most samples of the two harsher thirds violate something: the law is exercised, not bypassed)


def test_emulated_kernel_in_safe_mode_is_bitwise_the_host_chain_with_the_real_block(cassie):
    """CM_DRIVE_PD_SAFE in the step kernel (wave emulator) against pd law -> the REAL cassie_core_sim_step -> host chain, fed the
    same physics outputs: ctrl, measurement block, filter histories, delay lines bit for bit, message bits equal -- through a fresh
    block's first step (cassie_out_t still zero: every joint 'beyond' a limit), stress targets that drive joints into their limits,
    and an STO window."""
    import bench
    from emu_py import EmuBatch
    from hostchain_py import HostChain, device_state_bytes, pd_command
    pod = cassie.pod
    emu = EmuBatch(pod, 1)
    emu.qpos[:] = cassie.qpos_init()
    emu.forward()
    emu.drive_mode = P.DRIVE_PD_SAFE
    emu.pd_kp, emu.pd_kd = bench.PD_KP[None].copy(), bench.PD_KD[None].copy()
    rng = np.random.default_rng(5)
    hc = HostChain(cassie)
    meas = np.zeros(P.MEAS_DIM)
    seen = 0
    for t in range(240):
        if t % 40 == 0:
            pt = bench.PD_OFFSET + rng.uniform(-1, 1, 10) * (4.0 if t >= 80 else 0.3)      # (from t = 80: targets far beyond the limits)
        sto = 150 <= t < 170
        emu.pd_ptarget = pt[None].copy()
        emu.drive_cmd[0, 10] = 1.0 if sto else 0.0
        hc.L.cassie_hostenv_cassie_out(hc.env).contents.pelvis.radio.channel[8] = 0.0 if sto else 1.0
        sd, av = emu.sensordata[0].copy(), emu.actuator_velocity[0].copy()
        cmd, queue = hc.core_sim(pd_command(meas, pt, bench.PD_KP, bench.PD_KD))
        ctrl, meas, _ = hc.ethercat(cmd, sto, sd, av)
        emu.step()
        assert emu.meas[0].tobytes() == meas.tobytes(), t
        assert device_state_bytes(emu.drive_state[0]) == hc.state_bytes(), t
        assert emu.ctrl[0].tobytes() == ctrl.tobytes(), t
        bits = int(emu.drive_state[0].safety_msg)
        assert list(C.queue_of([bits])[0]) == queue, (t, bits, queue)
        seen |= bits
    assert seen == 3                                 # both codes were raised on the way
    assert np.all(np.isfinite(emu.qpos))
    hc.close()


def test_fused_substeps_equal_single_steps_in_safe_mode(cassie):
    import bench
    from emu_py import EmuBatch
    from hostchain_py import device_state_bytes
    pod = cassie.pod
    a, b = EmuBatch(pod, 1), EmuBatch(pod, 1)
    for e in (a, b):
        e.qpos[:] = cassie.qpos_init()
        e.forward()
        e.drive_mode = P.DRIVE_PD_SAFE
        e.pd_kp, e.pd_kd = bench.PD_KP[None].copy(), bench.PD_KD[None].copy()
        e.pd_ptarget = (bench.PD_OFFSET + np.array([0.4, -0.5, 1.5, -0.2, 0.6, -0.4, 0.5, -1.5, 0.2, -0.6]))[None].copy()
    a.step(15)
    for _ in range(15):
        b.step(1)
    assert a.qpos.tobytes() == b.qpos.tobytes() and a.meas.tobytes() == b.meas.tobytes() and a.ctrl.tobytes() == b.ctrl.tobytes()
    assert device_state_bytes(a.drive_state[0]) == device_state_bytes(b.drive_state[0])
    assert a.drive_state[0].safety_msg == b.drive_state[0].safety_msg != 0
