"""GPU parity on the EXACT paths bench.py measures: the on-device PD mode with 50 fused
substeps per launch on BASELINE.json's configs 2, 4 and 5 -- 4096 envs, per-env seeds 1234 + e, 1000 steps -- with a
sample of envs replayed on the CPU oracle (co_step_batch with co_pd_ctrl) and compared at every policy step; plus
the features that were emulator-only so far (applied forces, divergence flag, generic kernel under PD).

Tolerance: north_star asks for <= 1e-6 relative qpos error over 1000 steps; asserted here at 1e-7 relative
(fp64; kernel and oracle differ only in operation order), and (ncon, nefc, solver iterations) must be EQUAL at
every sample point."""
import ctypes

import numpy as np
import pytest

import bench
import oracle_py
from cassie_amd import Batch, Model
from cassie_amd import phys as P
from oracle_py import Oracle

pytestmark = pytest.mark.gpu
REL_TOL = 1e-7


def cassie_model(name):
    return Model(name)


def _pd_rollout(model, n, sample, nsteps=1000, hfield=None, q0_of=None, generic=False, allow_caps=False):
    """n envs under the bench workload on the GPU (PD mode, HOLD fused substeps per launch); envs `sample` also on
    the oracle; returns the worst relative qpos error over all policy steps."""
    pod = model.pod
    npol = nsteps // bench.HOLD
    tg = bench.pd_targets(np.arange(n), npol)
    q0 = np.tile(model.qpos_init(), (n, 1))
    if q0_of is not None:
        for e in range(n):
            q0[e] = q0_of(e, q0[e])
    b = Batch(model, n)
    if generic:
        b.set_generic_kernel(True)
    if hfield is not None:
        b.set_hfield(hfield)
        oracle_py.set_hfield(hfield)
    try:
        b.set(P.F_QPOS, q0)
        b.set(P.F_PD_KP, np.tile(bench.PD_KP, (n, 1)))
        b.set(P.F_PD_KD, np.tile(bench.PD_KD, (n, 1)))
        b.set_pd_mode(True)
        L = oracle_py.lib()
        buf = (oracle_py.CoData * len(sample))()
        for i, e in enumerate(sample):
            L.co_reset(ctypes.byref(pod), ctypes.byref(buf[i]))
            oracle_py.arr(buf[i].qpos)[: pod.nq] = q0[e]
        kp = np.tile(bench.PD_KP, (len(sample), 1))
        kd = np.tile(bench.PD_KD, (len(sample), 1))
        worst, rows_seen = 0.0, 0
        for p in range(npol):
            b.set(P.F_PD_PTARGET, tg[p])
            b.step(bench.HOLD)
            pt = np.ascontiguousarray(tg[p][sample])
            L.co_step_batch(ctypes.byref(pod), ctypes.byref(buf), len(sample), bench.HOLD, pt.ctypes.data, kp.ctypes.data, kd.ctypes.data, 0)
            q = b.get(P.F_QPOS)
            w, info = b.warnings()
            qo = np.array([oracle_py.arr(buf[i].qpos)[: pod.nq] for i in range(len(sample))])
            cnt = np.array([(buf[i].ncon, buf[i].nefc, buf[i].solver_iter) for i in range(len(sample))])
            bad = np.nonzero(np.any(info[sample][:, :3] != cnt, axis=1))[0]
            assert bad.size == 0, (p, sample[bad][:8].tolist(), info[sample][bad][:8].tolist(), cnt[bad][:8].tolist(), w[sample][bad][:8].tolist())
            worst = max(worst, float(np.max(np.abs(q[sample] - qo) / np.maximum(1.0, np.abs(qo)))))
            rows_seen = max(rows_seen, int(cnt[:, 1].max()))
            assert worst <= REL_TOL, (p, worst)
        if allow_caps:          # (rows / contacts past the model's caps are dropped alike by oracle and kernel, and the counts above agree)
            w = w & ~(P.WARN_CONTACT_FULL | P.WARN_CONSTRAINT_FULL)
        assert not w.any(), "warning bits raised: %s" % np.unique(w)
        assert np.all(np.isfinite(q))
        return worst, rows_seen, q
    finally:
        b.close()
        if hfield is not None:
            oracle_py.set_hfield(None)


def test_config2_pd_mode_4096_envs_1000_steps(cassie):
    """BASELINE config 2 as benchmarked: 4096 envs, seeds 1234 + e, PD targets every 50 steps, 1000 steps."""
    n = 4096
    sample = np.arange(n)                               # EVERY env of the batch is replayed on the oracle
    worst, rows, q = _pd_rollout(cassie, n, sample)
    assert rows >= 20                                   # the envs were in contact
    assert len(np.unique(q[:, 7])) > 4000               # and evolved independently


def test_config4_hfield_pd_mode_1000_steps(built):
    """BASELINE config 4 (cassie_hfield.xml, terrain of reference example/test_hfield.py:39-41): the sampled envs start
    spread over the flat patch, its edge and the rough part."""
    hf = Model("cassie_hfield")
    h = np.random.default_rng(99).random((200, 200)).astype(np.float32)
    h[95:105, 95:105] = 0
    n = 4096
    sample = np.arange(0, 16)

    def place(e, q):
        if e < 16:
            q[0], q[1] = [0.0, 0.35, -0.3, 0.6][e % 4], [0.0, 0.2, -0.45, 0.9][e // 4]
        return q
    worst, rows, q = _pd_rollout(hf, n, sample, hfield=h, q0_of=place)
    assert rows >= 16


def test_config5_tray_box_pd_mode_1000_steps(built):
    """BASELINE config 5 (cassie_tray_box.xml, the 40-dof kernel instantiation, box contacts)."""
    tray = Model("cassie_tray_box")
    n = 4096
    sample = np.unique(np.linspace(0, n - 1, 16).astype(int))
    worst, rows, q = _pd_rollout(tray, n, sample)
    assert rows >= 20


def test_generic_kernel_under_pd_mode(cassie):
    """phys_batch_set_generic_kernel with the PD workload (run-time topology instantiation), 256 envs x 500 steps."""
    _pd_rollout(cassie, 256, np.arange(0, 256, 32), nsteps=500, generic=True)


def test_applied_forces_on_the_device(cassie):
    """qfrc_applied and xfrc_applied (cassie_sim_apply_force role, reference src/cassiemujoco.c:1586-1600) under the PD
    workload, fused launches: GPU vs oracle."""
    pod = cassie.pod
    n = 8
    rng = np.random.default_rng(4)
    qf = rng.uniform(-2, 2, (n, pod.nv))
    xf = np.zeros((n, pod.nbody, 6))
    xf[:, 1, :3] = rng.uniform(-40, 40, (n, 3))          # pelvis pushes
    xf[:, 1, 3:] = rng.uniform(-5, 5, (n, 3))
    xf[:, pod.nbody - 1, :3] = rng.uniform(-10, 10, (n, 3))
    tg = bench.pd_targets(np.arange(n), 4)
    b = Batch(cassie, n)
    b.set(P.F_QPOS, np.tile(cassie.qpos_init(), (n, 1)))
    b.set(P.F_QFRC_APPLIED, qf)
    b.set(P.F_XFRC_APPLIED, xf.reshape(n, -1))
    b.set(P.F_PD_KP, np.tile(bench.PD_KP, (n, 1)))
    b.set(P.F_PD_KD, np.tile(bench.PD_KD, (n, 1)))
    b.set_pd_mode(True)
    orcs = [Oracle(pod, cassie.qpos_init()) for _ in range(n)]
    for e, o in enumerate(orcs):
        o.qfrc_applied[:] = qf[e]
        o.xfrc_applied[:] = xf[e]
    for p in range(4):
        b.set(P.F_PD_PTARGET, tg[p])
        b.step(bench.HOLD)
        for e, o in enumerate(orcs):
            for _ in range(bench.HOLD):
                o.pd_ctrl(tg[p][e], bench.PD_KP, bench.PD_KD)
                o.step()
    q, v = b.get(P.F_QPOS), b.get(P.F_QVEL)
    w, info = b.warnings()
    b.close()
    assert not w.any()
    for e, o in enumerate(orcs):
        assert (info[e, 0], info[e, 1], info[e, 2]) == (o.d.ncon, o.d.nefc, o.d.solver_iter)
        assert np.max(np.abs(q[e] - o.qpos)) < 1e-10 and np.max(np.abs(v[e] - o.qvel)) < 1e-8
    free = Oracle(pod, cassie.qpos_init())               # the forces really mattered
    for p in range(4):
        for _ in range(bench.HOLD):
            free.pd_ctrl(tg[p][0], bench.PD_KP, bench.PD_KD)
            free.step()
    assert np.max(np.abs(free.qpos - orcs[0].qpos)) > 1e-3


def test_divergence_flag_on_the_device(cassie):
    """A NaN / out-of-range state raises the sticky WARN_DIVERGED bit and leaves that env's state untouched (the
    documented replacement of MuJoCo's auto-reset); neighbours are unaffected; clear_warnings clears it."""
    n = 6
    q0 = np.tile(cassie.qpos_init(), (n, 1))
    v0 = np.zeros((n, cassie.pod.nv))
    v0[2, 3] = np.nan
    q0[4, 9] = 3e10
    b = Batch(cassie, n)
    b.set(P.F_QPOS, q0)
    b.set(P.F_QVEL, v0)
    b.step(30)
    w, _ = b.warnings()
    q = b.get(P.F_QPOS)
    assert [int(x) & P.WARN_DIVERGED for x in w] == [0, 0, 8, 0, 8, 0]
    assert np.array_equal(q[2], q0[2]) and np.array_equal(q[4], q0[4])
    assert np.array_equal(q[0], q[1]) and np.array_equal(q[0], q[5]) and not np.array_equal(q[0], q0[0])
    b.step(5)
    assert b.warnings()[0][2] & P.WARN_DIVERGED        # sticky
    b.clear_warnings()
    assert not b.warnings()[0].any()
    b.close()


def test_strided_observation_block(cassie):
    """phys_batch_bind_strided: qpos | qvel | sensordata as column blocks of one [n][96] tensor give the same
    trajectory as separate dense fields, and host copies of a strided field are exact."""
    import torch
    pod = cassie.pod
    n = 32
    rng = np.random.default_rng(8)
    c = rng.uniform(-2, 2, (n, pod.nu))
    dense = Batch(cassie, n)
    dense.set(P.F_QPOS, np.tile(cassie.qpos_init(), (n, 1)))
    dense.set(P.F_CTRL, c)
    dense.step(60)
    nobs = pod.nq + pod.nv + pod.nsensordata
    obs = torch.zeros((n, nobs), dtype=torch.float64, device="cuda")
    b = Batch(cassie, n)
    b.bind(P.F_QPOS, obs.data_ptr(), row_stride=nobs)
    b.bind(P.F_QVEL, obs.data_ptr() + 8 * pod.nq, row_stride=nobs)
    b.bind(P.F_SENSORDATA, obs.data_ptr() + 8 * (pod.nq + pod.nv), row_stride=nobs)
    b.set(P.F_QPOS, np.tile(cassie.qpos_init(), (n, 1)))          # 2-D upload into the strided block
    b.set(P.F_CTRL, c)
    b.step(60)
    b.sync()
    o = obs.cpu().numpy()
    assert np.array_equal(o[:, : pod.nq], dense.get(P.F_QPOS))
    assert np.array_equal(o[:, pod.nq: pod.nq + pod.nv], dense.get(P.F_QVEL))
    assert np.array_equal(o[:, pod.nq + pod.nv:], dense.get(P.F_SENSORDATA))
    assert np.array_equal(b.get(P.F_QVEL, 3, 5), o[3:8, pod.nq: pod.nq + pod.nv])   # 2-D download of a row range
    dense.close()
    b.close()


@pytest.mark.parametrize("name,mode", [("cassie", "exact"), ("cassie", "drive"), ("cassie_hfield", "drive"), ("cassie_tray_box", "exact")])
def test_results_do_not_depend_on_what_lds_held_before(name, mode, built):
    """LDS is neither initialised nor cleared between kernels.  Every launch of this rollout is preceded by a kernel that
    fills the LDS of all CUs with NaN bit patterns; the trajectories must be bit for bit those of an undisturbed batch
    (tests/test_emu_parity.py has the emulator twin that found a read of an unwritten row in round 2)."""
    model = cassie_model(name)
    pod = model.pod
    n = 2048
    tg = bench.pd_targets(np.arange(n), 6)
    hfield = None
    if name == "cassie_hfield":
        hfield = np.random.default_rng(99).random((200, 200)).astype(np.float32)
    out = []
    for poison in (False, True):
        b = Batch(model, n)
        if hfield is not None:
            b.set_hfield(hfield)
        q0 = np.tile(model.qpos_init(), (n, 1))
        q0[:, 0] = np.linspace(-1, 1, n) if hfield is not None else 0.0
        b.set(P.F_QPOS, q0)
        b.set(P.F_PD_KP, np.tile(bench.PD_KP, (n, 1)))
        b.set(P.F_PD_KD, np.tile(bench.PD_KD, (n, 1)))
        if mode == "drive":
            b.forward()
            b.set_drive_mode(P.DRIVE_PD)
        else:
            b.set_pd_mode(True)
        for p in range(6):
            b.set(P.F_PD_PTARGET, tg[p])
            if poison:
                b.poison_lds()
            b.step(50)
        w, info = b.warnings()
        out.append((b.get(P.F_QPOS), b.get(P.F_QVEL), b.get(P.F_SENSORDATA), w, info))
        b.close()
    assert not out[1][3].any()
    assert out[1][4][:, 0].max() >= 2
    for a, c in zip(out[0], out[1]):
        assert np.array_equal(a, c)


def test_bench_rollout_with_the_collective_path_on_one_rank(cassie):
    """bench.device_rollout with the multi-GPU machinery switched on (RCCL process group of one rank: barrier fences, the
    observation all-gather that overlaps the next launch through a snapshot and a second stream, the max reduction):
    the gathered block holds this rank's rows, and the rollout still matches its CPU replay."""
    import os
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        r = bench.device_rollout(cassie, "drive-pd", 512, 100, 50, 0, 1, 0, bench.HOLD, 8, None, collect=True, repeats=3, nstreams=2)
    finally:
        if created:
            dist.destroy_process_group()
    assert r["gather_ok"] is True and r["streams"] == 2          # two env ranges on two streams, one all-gather per range and policy step
    assert len(r["region_s"]) == 3 and r["parity"]["after_timed_region"] == 2 and r["parity"]["steps_replayed"] == 1000 + 50 + 300
    assert r["parity"]["ok"] and r["parity"]["frac_envs_with_equal_ncon_nefc_iters"] == 1.0
    assert r["envs_with_warnings"] == 0


def test_dense_heightfield_sampling_flag_on_the_gpu(built):
    """CM_FLAG_HFDENSE (optional: up to eight interior samples per capsule, ten lanes per pair and two passes of the
    pre-pass): robots tipped over on the rough terrain, so that shins and tarsi -- the capsules the flag changes -- reach the
    ground; GPU against the oracle with the same flag, 256 envs x 600 steps, counts equal at every policy step."""
    hf = Model("cassie_hfield")
    hf.set_flag(P.FLAG_HFDENSE, True)
    h = np.random.default_rng(99).random((200, 200)).astype(np.float32)
    h[95:105, 95:105] = 0

    def place(e, q):
        q[0], q[1], q[2] = 0.6 + 0.07 * (e % 16), 0.9 - 0.05 * (e // 16 % 16), 0.75
        q[3:7] = [0.924, 0.0, 0.383, 0.0] if e % 2 else [0.924, 0.383, 0.0, 0.0]      # pitched / rolled 45 degrees
        return q
    worst, rows, q = _pd_rollout(hf, 256, np.arange(0, 256, 16), nsteps=600, hfield=h, q0_of=place)
    assert rows >= 28


def test_multi_contact_heightfield_flag_on_the_gpu(built):
    """CM_FLAG_HFMULTI (optional: up to four contacts per capsule / height-field pair, deepest samples first): robots
    standing on the flat patch (four contacts per foot: 44 rows -- past the row-capped fast kernel, so every launch is handed
    over to the full one) and robots tipped over on the rough part; GPU against the oracle with the same flags."""
    hf = Model("cassie_hfield")
    hf.set_flag(P.FLAG_HFMULTI, True)
    hf.set_flag(P.FLAG_HFDENSE, True)
    h = np.random.default_rng(99).random((200, 200)).astype(np.float32)
    h[95:105, 95:105] = 0

    def place(e, q):
        if e % 2:
            q[0], q[1], q[2] = 0.6 + 0.07 * (e % 16), 0.9 - 0.05 * (e // 16 % 16), 0.75
            q[3:7] = [0.924, 0.0, 0.383, 0.0]
        return q
    worst, rows, q = _pd_rollout(hf, 256, np.arange(0, 256, 8), nsteps=600, hfield=h, q0_of=place)
    assert rows >= 40


def test_prism_contacts_heightfield_flag_on_the_gpu(built):
    """CM_FLAG_HFPRISM (optional: one contact per penetrated grid triangle, up to 32 contacts and 127 rows -- the solve of a substep with
    more than 64 rows spread over both wavefronts of its env): robots standing on the flat patch and robots tipped over on the rough
    part, in exact-PD mode; GPU against the oracle with the same flag, counts equal at every policy step, rows well past one
    wavefront's."""
    hf = Model("cassie_hfield")
    hf.set_flag(P.FLAG_HFPRISM, True)
    h = np.random.default_rng(99).random((200, 200)).astype(np.float32)
    h[95:105, 95:105] = 0

    def place(e, q):
        if e % 2:
            q[0], q[1], q[2] = 0.6 + 0.07 * (e % 16), 0.9 - 0.05 * (e // 16 % 16), 0.75
            q[3:7] = [0.924, 0.0, 0.383, 0.0]
        return q
    worst, rows, q = _pd_rollout(hf, 256, np.arange(0, 256, 8), nsteps=600, hfield=h, q0_of=place, allow_caps=True)
    assert rows > 64


def test_box_box_eight_contact_option_on_the_gpu(built):
    """CM_FLAG_BOX8 (up to eight box-box contacts like MuJoCo's mjc_BoxBox; reference model/cassie_tray_box.xml:213-216, :230-237):
    256 envs of config 5 with the cube started at random places on and around the tray's rim, turned about the vertical, 300 steps:
    the HIP kernel against the oracle, counts equal at every check point, qpos to 1e-9; box-box sets of more than four points occur."""
    from oracle_py import Oracle
    model = Model("cassie_tray_box")
    model.set_flag(P.FLAG_BOX8, True)
    pod = model.pod
    n = 256
    rng = np.random.default_rng(12)
    q0 = np.tile(model.qpos_init(), (n, 1))
    q0[:, 35] = rng.uniform(-0.15, 0.15, n); q0[:, 36] = rng.uniform(-0.15, 0.15, n)
    q0[:, 37] = 1.01 + 0.17 + 0.005 + 0.05 + rng.uniform(0.001, 0.01, n)
    ang = rng.uniform(0, np.pi / 2, n)
    q0[:, 38], q0[:, 41] = np.cos(ang / 2), np.sin(ang / 2)
    b = Batch(model, n)
    try:
        b.set(P.F_QPOS, q0)
        orcs = [Oracle(pod, q0[e]) for e in range(n)]
        cube = model.name2id(1, "cup_box")
        most = 0
        for chunk in range(6):
            b.step(50)
            q = b.get(P.F_QPOS)
            w, info = b.warnings()
            for e, o in enumerate(orcs):
                o.step(50)
                assert (info[e, 0], info[e, 1], info[e, 2]) == (o.d.ncon, o.d.nefc, o.d.solver_iter), (chunk, e, info[e].tolist(), (o.d.ncon, o.d.nefc, o.d.solver_iter))
                most = max(most, sum(1 for i in range(o.d.ncon) if cube in (pod.geom_bodyid[o.d.contact[i].geom1], pod.geom_bodyid[o.d.contact[i].geom2])))
            qo = np.array([o.qpos.copy() for o in orcs])
            assert np.max(np.abs(q - qo) / np.maximum(1.0, np.abs(qo))) < 1e-9, chunk
        assert most > 4, most
        assert not (w & ~(P.WARN_CONTACT_FULL | P.WARN_CONSTRAINT_FULL)).any()
    finally:
        b.close()
