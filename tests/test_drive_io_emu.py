"""Device-side drive-level I/O on the CPU wave emulator: the encoder / motor models inside
physics_kernel.h against the host chain (csrc/cassie_hostpath.c, pinned to the reference's code) -- BIT FOR BIT:
identical sensordata / actuator_velocity in => identical ctrl, measurement block, filter histories and delay lines."""
import numpy as np

from cassie_amd import phys as P
from emu_py import EmuBatch
from hostchain_py import HostChain, device_state_bytes, pd_command


def _commands(rng, t):
    u = rng.uniform(-1, 1, 10) * np.array([112.5, 112.5, 195.2, 195.2, 45.0] * 2) * (3.0 if 60 <= t < 80 else 0.4)
    return u, 40 <= t < 55           # torques (beyond the limits for a while), STO window


def test_torque_mode_is_bitwise_the_host_chain(cassie):
    pod = cassie.pod
    rng = np.random.default_rng(0)
    emu = EmuBatch(pod, 1)
    emu.qpos[:] = cassie.qpos_init()
    emu.forward()                                  # what cassie_sim_init leaves: sensordata of the init pose
    emu.drive_mode = P.DRIVE_TORQUE
    hc = HostChain(cassie)
    for t in range(110):
        u, sto = _commands(rng, t)
        sd, av = emu.sensordata[0].copy(), emu.actuator_velocity[0].copy()     # outputs of the previous step
        ctrl, meas, _ = hc.ethercat(u, sto, sd, av)
        emu.drive_cmd[0, :10], emu.drive_cmd[0, 10] = u, 1.0 if sto else 0.0
        emu.step()
        assert emu.meas[0].tobytes() == meas.tobytes(), t
        assert device_state_bytes(emu.drive_state[0]) == hc.state_bytes(), t
        if t >= 6:                                 # after the delay line has filled, the applied torques are the motor's
            assert np.any(ctrl != 0) or sto or 46 <= t < 61
    # speed-torque saturation and the limits were exercised
    assert np.abs(np.ctypeslib.as_array(emu.drive_state[0].torque_delay)).max() <= 12.2
    hc.close()


def test_fused_substeps_equal_single_steps_in_torque_mode(cassie):
    """nsub fused substeps with the command held = nsub separate launches (state carried through HBM vs through LDS)."""
    pod = cassie.pod
    a, b = EmuBatch(pod, 1), EmuBatch(pod, 1)
    for e in (a, b):
        e.qpos[:] = cassie.qpos_init()
        e.forward()
        e.drive_mode = P.DRIVE_TORQUE
        e.drive_cmd[0, :10] = [20, -15, 60, -80, 10, -20, 15, -60, 80, -10]
    a.step(12)
    for _ in range(12):
        b.step(1)
    assert a.qpos.tobytes() == b.qpos.tobytes() and a.meas.tobytes() == b.meas.tobytes()
    # the outputs of a fused launch are its last substep's -- IMU words included, which only the last two substeps evaluate
    assert a.sensordata.tobytes() == b.sensordata.tobytes() and a.actuator_velocity.tobytes() == b.actuator_velocity.tobytes()
    assert device_state_bytes(a.drive_state[0]) == device_state_bytes(b.drive_state[0])


def test_pd_mode_on_measurements_against_host_chain_and_oracle(cassie):
    """CM_DRIVE_PD: the command is pd_input's motor PD evaluated on the encoder measurements of the previous step; the
    rest is the ethercat chain.  Checked against numpy PD + host chain + oracle physics, free-running."""
    import bench
    from oracle_py import Oracle
    pod = cassie.pod
    emu = EmuBatch(pod, 1)
    emu.qpos[:] = cassie.qpos_init()
    emu.forward()
    emu.drive_mode = P.DRIVE_PD
    tg = bench.pd_targets([7], 4)[:, 0]
    emu.pd_kp, emu.pd_kd = bench.PD_KP[None].copy(), bench.PD_KD[None].copy()
    o = Oracle(pod, cassie.qpos_init())
    o.forward()
    hc = HostChain(cassie)
    meas = np.zeros(P.MEAS_DIM)
    for t in range(160):
        pt = tg[t // 50]
        emu.pd_ptarget = pt[None].copy()
        ctrl, meas, _ = hc.ethercat(pd_command(meas, pt, bench.PD_KP, bench.PD_KD), False, o.sensordata.copy(), o.actuator_velocity.copy())
        o.ctrl[:] = ctrl
        o.step()
        emu.step()
        assert np.max(np.abs(emu.qpos[0] - o.qpos)) < 1e-10, t
        # encoder counts can differ by one where a 1e-13 physics difference straddles a truncation boundary
        assert np.max(np.abs(emu.meas[0][:10] - meas[:10])) <= 2 * np.pi / (1 << 13) / 16 + 1e-12, t
    assert o.qpos[2] > 0.9                         # PD on quantised, delayed measurements still holds the robot up
    hc.close()


def test_forward_after_a_drive_mode_step_uses_the_applied_torques(cassie):
    """mj_forward reads d->ctrl, the torque last applied (reference src/cassiemujoco.c:971, :1223 after :1120-1134).  In
    the device's drive modes the applied torque is the delay line's output and lives in LDS during a launch: the launch
    must leave it in the ctrl field, or a later forward pass / phys_batch_derive (reward getters: foot forces, qacc)
    evaluates the state under stale motor torques -- zeros in a device-resident rollout."""
    import bench
    from test_derive_emu import _ids
    pod = cassie.pod
    ids = np.array([3])
    emu = EmuBatch(pod, 1)
    emu.qpos[:] = cassie.qpos_init()
    emu.forward()
    emu.drive_mode = P.DRIVE_PD
    emu.pd_kp, emu.pd_kd = bench.PD_KP[None].copy(), bench.PD_KD[None].copy()
    ref = bench.HostChainEnvs(cassie, ids)
    tg = bench.pd_targets(ids, 8)
    for p in range(7):                              # 350 steps: the robot has landed and stands on its PD controller
        emu.pd_ptarget = np.ascontiguousarray(tg[p])
        emu.step(50)
        ref.step(50, tg[p])
    o = ref.orcs[0]
    assert o.d.ncon >= 2 and np.abs(o.ctrl).max() > 1.0                 # standing, motors loaded
    assert np.max(np.abs(emu.ctrl[0] - o.ctrl)) < 1e-9                  # the launch left the applied torques in the ctrl field
    emu.drive_mode = P.DRIVE_OFF
    D, _ = emu.derive(_ids(cassie))
    o.forward()                                                          # the reference: forward on the applied torques
    assert np.max(np.abs(emu.qacc[0] - o.qacc)) < 1e-6 * max(1.0, np.abs(o.qacc).max())
    fz = D[0, P.DRV_FOOT_FORCE + 2] + D[0, P.DRV_FOOT_FORCE + 8]
    ftot = 0.0
    for c in range(o.d.ncon):
        con = o.d.contact[c]
        a = con.efc_address
        ftot += sum(o.d.efc_force[a + i] for i in range(4)) * con.frame[2]
    assert abs(fz - ftot) < 1e-6 * max(1.0, abs(ftot)) and fz > 100     # the feet carry the robot
    # with stale (zero) torques the same state gives a visibly different answer: the check above is not vacuous
    z = o.ctrl.copy()
    o.ctrl[:] = 0
    o.forward()
    assert np.max(np.abs(emu.qacc[0] - o.qacc)) > 1.0
    o.ctrl[:] = z
    for hc in ref.chains:
        hc.close()
