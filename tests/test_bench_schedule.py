"""bench.py's schedule on the CPU: the launch / restart / gather logic (bench.Schedule), the oracle replay that
produces `max_qpos_err`, and -- with the wave emulator standing in for the GPU -- the whole measured path (PD mode,
fused substeps, staggered episode restarts) against that replay."""
import numpy as np

import bench
from emu_py import EmuBatch


def test_schedule_launch_pattern():
    log = []
    sch = bench.Schedule(step=lambda k: log.append(("step", k)), bind_targets=lambda p: log.append(("bind", p)),
                         restart=lambda g: log.append(("restart", g)), gather=lambda: log.append(("gather",)),
                         substeps_per_launch=bench.HOLD)
    sch.run(0, 120)
    assert log == [("bind", 0), ("restart", 0), ("step", 50), ("gather",), ("bind", 1), ("restart", 1), ("step", 50),
                   ("gather",), ("bind", 2), ("restart", 2), ("step", 20)]
    log.clear()
    sch.run(120, 40)                                       # continues inside policy step 2, crosses into 3
    assert log == [("step", 30), ("gather",), ("bind", 3), ("restart", 3), ("step", 10)]
    assert sch.launches == 5 and sch.gathers == 3
    assert bench.restart_group(bench.NGROUP + 3) == 3
    log.clear()
    bench.Schedule(step=lambda k: log.append(k), bind_targets=lambda p: None, restart=lambda g: None, substeps_per_launch=20).run(0, 100)
    assert log == [20, 20, 10, 20, 20, 10]                 # never across a PD-target re-draw


def test_emulated_bench_path_against_the_oracle_replay(cassie, monkeypatch):
    """The measured path end to end with the emulator executing the kernel: 4 envs, short episodes (restart phases
    0..3), targets re-drawn every 10 steps; the oracle replay must reproduce every env's final qpos."""
    monkeypatch.setattr(bench, "HOLD", 10)
    monkeypatch.setattr(bench, "EPISODE", 40)
    monkeypatch.setattr(bench, "NGROUP", 4)
    pod = cassie.pod
    n, total = 4, 95
    ids = np.arange(n)
    tg = bench.pd_targets(ids, total // 10 + 2)
    emu = EmuBatch(pod, n)
    emu.qpos[:] = cassie.qpos_init()
    emu.pd_kp, emu.pd_kd = np.tile(bench.PD_KP, (n, 1)), np.tile(bench.PD_KD, (n, 1))

    def bind(p):
        emu.pd_ptarget = np.ascontiguousarray(tg[p])

    def restart(g):
        rows = np.nonzero(ids % bench.NGROUP == g)[0]
        emu.qpos[rows] = cassie.qpos_init()
        emu.qvel[rows] = 0
        emu.qacc_warmstart[rows] = 0
    sch = bench.Schedule(step=emu.step, bind_targets=bind, restart=restart)
    sch.run(0, 60)
    sch.run(60, total - 60)
    orc = bench.replay_on_oracle(cassie, ids, lambda p: tg[p], total)
    assert np.max(np.abs(emu.qpos - orc.qpos())) < 1e-11
    assert np.array_equal(emu.info[:, :3], orc.counts())
    assert len({tuple(np.round(r, 6)) for r in emu.qpos}) == n      # the envs are at different episode phases


def test_config2_exact_workload_on_the_emulator(cassie):
    """BASELINE config 2 as benchmarked (seeds 1234 + e, PD mode, 50 fused substeps per launch, 1000 steps) for two envs
    on the emulated kernel vs the oracle -- the CPU twin of tests/test_config_parity_gpu.py."""
    pod = cassie.pod
    n = 2
    tg = bench.pd_targets(np.arange(n), 20)
    emu = EmuBatch(pod, n)
    emu.qpos[:] = cassie.qpos_init()
    emu.pd_kp, emu.pd_kd = np.tile(bench.PD_KP, (n, 1)), np.tile(bench.PD_KD, (n, 1))
    orc = bench.OracleEnvs(cassie, np.arange(n))
    for p in range(20):
        emu.pd_ptarget = np.ascontiguousarray(tg[p])
        emu.step(50)
        orc.step(50, tg[p], 2)
        assert np.array_equal(emu.info[:, :3], orc.counts()), p
        qo = orc.qpos()
        assert np.max(np.abs(emu.qpos - qo) / np.maximum(1, np.abs(qo))) < 1e-7, p


def test_emulated_drive_pd_path_against_the_host_chain_replay(cassie, monkeypatch):
    """bench.py's default mode (CM_DRIVE_PD) end to end with the emulator executing the kernel: restarts are a fresh
    cassie_sim_t (init pose and its sensordata, zero filter histories / delay lines); the CPU replay (oracle physics +
    host chain) must reproduce every env's final qpos."""
    import ctypes
    from cassie_amd import phys as P
    from cassie_amd._lib import CmDriveState
    monkeypatch.setattr(bench, "HOLD", 10)
    monkeypatch.setattr(bench, "EPISODE", 40)
    monkeypatch.setattr(bench, "NGROUP", 4)
    pod = cassie.pod
    n, total = 4, 75
    ids = np.arange(n)
    tg = bench.pd_targets(ids, total // 10 + 2)
    ref = bench.HostChainEnvs(cassie, ids)
    init_sens = ref.init_sensordata()
    emu = EmuBatch(pod, n)
    emu.qpos[:] = cassie.qpos_init()
    emu.sensordata[:] = init_sens
    emu.pd_kp, emu.pd_kd = np.tile(bench.PD_KP, (n, 1)), np.tile(bench.PD_KD, (n, 1))
    emu.drive_mode = P.DRIVE_PD

    def bind(p):
        emu.pd_ptarget = np.ascontiguousarray(tg[p])

    def restart(g):
        for r in np.nonzero(ids % bench.NGROUP == g)[0]:
            emu.qpos[r] = cassie.qpos_init()
            emu.qvel[r] = 0
            emu.qacc_warmstart[r] = 0
            emu.sensordata[r] = init_sens
            emu.actuator_velocity[r] = 0
            emu.meas[r] = 0
            ctypes.memset(ctypes.addressof(emu.drive_state[int(r)]), 0, ctypes.sizeof(CmDriveState))
    bench.Schedule(step=emu.step, bind_targets=bind, restart=restart).run(0, total)
    orc = bench.replay_on_oracle(cassie, ids, lambda p: tg[p], total, envs=bench.HostChainEnvs)
    assert np.max(np.abs(emu.qpos - orc.qpos())) < 1e-11
    assert np.array_equal(emu.info[:, :3], orc.counts())
