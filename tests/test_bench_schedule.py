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


def test_hand_over_figure_counts_against_the_last_launch_not_against_hold():
    """the driver's `--steps 20` regions end with a 20-substep launch; the fast kernel records 20 completed
    substeps for an env it did NOT hand over, which must not read as "handed over" because 20 < HOLD."""
    import bench
    progress = np.array([20, 20, 20, 7, 20, 0])
    assert bench.handed_over_in_last_launch(progress, 20) == 2.0          # the envs short of THIS launch's 20 substeps
    assert bench.handed_over_in_last_launch(np.full(4096, 20), 20) == 0.0
    assert bench.handed_over_in_last_launch(np.full(4096, 50), 50) == 0.0
    assert all(bench.has_fast_kernel(m) for m in ("cassie", "cassie_hfield", "cassie_tray_box"))   # (the tray model's: 47 rows, round 4)
    # the schedule of `--steps 20 --warmup 5` after the pre-roll: the last launch of a region has at most 20 substeps
    seen = []
    sch = bench.Schedule(step=seen.append, bind_targets=lambda p: None, restart=lambda g: None)
    sch.run(bench.PREROLL + 5, 20)
    assert sum(seen) == 20 and seen[-1] <= 20


def test_pmc_traffic_scales_only_the_per_substep_part(cassie):
    """roofline.traffic for a launch of another length than the profiled 50 substeps: the per-env launch I/O (state in, state
    and last-substep outputs out) stays, only the per-env-step remainder scales."""
    import bench
    pod = cassie.pod
    fixed = bench.launch_io_bytes_per_env(pod)
    assert 5000 < fixed < 12000
    t50, src = bench.pmc_traffic(4096 * 50, "cassie", envs_per_launch=4096, pod=pod)
    t20, _ = bench.pmc_traffic(4096 * 20, "cassie", envs_per_launch=4096, pod=pod)
    if t50 is None:
        pytest.skip("no committed PMC summary")
    assert t20 > 0.4 * t50 + 0.5 * fixed * 4096 * 0.6          # far above the linear 0.4 x: the fixed part does not shrink
    assert t20 >= fixed * 4096 and t20 <= t50       # (equal when the counters show launch I/O only: model constants stay in cache)
    # launches in chunks: every chunk of an env's launch loads and stores like a launch of its own
    t50_4, _ = bench.pmc_traffic(4096 * 50, "cassie", envs_per_launch=4096, pod=pod, chunks=4)
    assert abs((t50_4 - t50) - 3 * fixed * 4096) < 1e-6 * t50
    # (round 6: whole-batch launches go as 7 chunks where they are long enough, a range's launch of at most 25 substeps as 3)
    assert bench.launch_chunks(2048, 50, False) == 2 and bench.launch_chunks(4096, 50, True) == 7 and bench.launch_chunks(4096, 20, True) == 4
    assert bench.launch_chunks(2048, 20, False) == 3 and bench.launch_chunks(2048, 14, False) == 2 and bench.launch_chunks(2048, 9, False) == 1 and bench.launch_chunks(1024, 50, True) == 1


def test_slot_occupancy_figure():
    """bench.slot_occupancy: 1024 slots, every env-step holding one for its clocks -- a region that keeps every slot busy reads 1.0."""
    import bench
    n, substeps, clocks = 4096, 1000 * bench.HOLD, 80000.0
    full = n * substeps * clocks / bench.SHADER_CLOCK_HZ / bench.SLOTS_PER_GPU      # the region's time with no slot ever idle
    o = bench.slot_occupancy(clocks, n, substeps, full)
    assert abs(o["busy_frac"] - 1.0) < 1e-12
    assert abs(o["rate_with_every_slot_busy"] - n * substeps / full) < 1e-3
    assert abs(bench.slot_occupancy(clocks, n, substeps, 1.25 * full)["busy_frac"] - 0.8) < 1e-12
    assert bench.slot_occupancy(None, n, substeps, full) is None      # a batch that keeps no per-env clocks
