"""Per-env domain randomisation on the device (reference src/cassiemujoco.c:1323-1436 setters, :949-977
mj_setConst), on the GPU through the C ABI: phys_batch_randomize + phys_batch_set_const with EVERY env of a 4096-env batch
randomised, against (a) the host model compiler -- the parameter blocks bit for bit -- and (b) the oracle stepping a per-env
compiled model for every env (50 steps, all 4096) and for 64 sampled envs over 1000 steps in the benchmarked mode."""
import ctypes

import numpy as np
import pytest

import bench
import randomise_check as rc
from cassie_amd import Batch, Model
from cassie_amd import phys as P
from oracle_py import Oracle

pytestmark = pytest.mark.gpu
N = 4096


def _randomised_batch(name, n, seed, through_torch=False):
    model = Model(name)
    params = rc.random_params(model, n, seed=seed)
    b = Batch(model, n)
    keep = []
    for f in rc.INPUT_FIELDS:
        if through_torch and f in ("body_mass", "dof_damping"):
            import torch
            t = torch.from_numpy(np.ascontiguousarray(params[f])).to("cuda:0")   # rows already in HBM: a device pointer, no PCIe
            torch.cuda.synchronize()
            keep.append(t)
            b.randomize(rc.PARAM_IDS[f], None, device_ptr=t.data_ptr(), n=n)
        else:
            b.randomize(rc.PARAM_IDS[f], params[f])
    b.set_const()
    b.sync()
    return model, params, b


@pytest.mark.parametrize("name", ["cassie", "cassie_tray_box"])
def test_device_set_const_equals_the_host_compile_for_every_field(built, name):
    n = 512
    model, params, b = _randomised_batch(name, n, seed=21, through_torch=(name == "cassie"))
    try:
        blocks = b.params()
        hosts = rc.HostEnvModels(name)
        for e in range(0, n, 8):
            rc.assert_blocks_equal(blocks[e], hosts.pod(params, e).params, model.pod, "%s env %d" % (name, e))
        mi = np.array([blk.meaninertia for blk in blocks])
        assert len(np.unique(mi)) == n
        # friction alone: the pairs' mixed values follow at once, the inverse weights stay
        fr = params["geom_friction"].copy()
        fr[:, 0::3] *= 0.5
        b.randomize(P.P_GEOM_FRICTION, fr)
        b.sync()
        after = b.params()
        params2 = dict(params, geom_friction=fr)
        for e in (0, n // 2, n - 1):
            rc.assert_blocks_equal(after[e], hosts.pod(params2, e).params, model.pod, "%s env %d, friction halved" % (name, e))
    finally:
        b.close()


def test_all_4096_envs_randomised_follow_the_oracle_with_per_env_models(built):
    """Torque-level stepping, one launch of 50 fused substeps: every one of the 4096 envs against the oracle run on ITS compiled
    model (counts equal, qpos to rounding) -- and the envs do differ from the unrandomised model."""
    name, nsub = "cassie", 50
    model, params, b = _randomised_batch(name, N, seed=33)
    try:
        q0 = model.qpos_init()
        rng = np.random.default_rng(4)
        hi = np.array([model.pod.act_ctrlrange[u][1] for u in range(model.pod.nu)])
        ctrl = 0.5 * hi * rng.uniform(-1, 1, (N, model.pod.nu))
        b.set(P.F_QPOS, np.tile(q0, (N, 1)))
        b.set(P.F_CTRL, ctrl)
        b.step(nsub)
        q = b.get(P.F_QPOS)
        w, info = b.warnings()
        assert not w.any()
        hosts = rc.HostEnvModels(name)
        worst, moved = 0.0, 0.0
        base = Oracle(model.pod, q0)
        for e in range(N):
            pe = hosts.pod(params, e)
            o = Oracle(pe, q0)
            o.ctrl[:] = ctrl[e]
            o.step(nsub)
            assert (info[e, 0], info[e, 1], info[e, 2]) == (o.d.ncon, o.d.nefc, o.d.solver_iter), (e, info[e].tolist(), (o.d.ncon, o.d.nefc, o.d.solver_iter))
            worst = max(worst, float(np.max(np.abs(q[e] - o.qpos))))
            if e < 16:
                u = Oracle(model.pod, q0)
                u.ctrl[:] = ctrl[e]
                u.step(nsub)
                moved = max(moved, float(np.max(np.abs(u.qpos - o.qpos))))
        print("4096 randomised envs x %d steps: worst |qpos - oracle(per-env model)| %.2e; randomisation moves qpos by up to %.2e" % (nsub, worst, moved))
        assert worst < 1e-11
        assert moved > 1e-5
    finally:
        b.close()


def test_randomised_batch_in_the_benchmarked_mode_1000_steps(built):
    """CM_DRIVE_PD, 4096 randomised envs x 1000 steps in 50-substep launches, 64 sampled envs replayed through the oracle (per-env
    models) + host chain and compared at every policy step like tests/test_drive_parity_gpu.py does for the shared model."""
    name, nsteps = "cassie", 1000
    model, params, b = _randomised_batch(name, N, seed=55)
    sample = np.unique(np.linspace(0, N - 1, 64).astype(int))
    hosts = rc.HostEnvModels(name)
    pods = [hosts.pod(params, int(e)) for e in sample]
    npol = nsteps // bench.HOLD
    tg = bench.pd_targets(sample, npol)
    rng = np.random.default_rng(77)
    tg_all = np.tile(bench.PD_OFFSET, (npol, N, 1)) + rng.uniform(-0.3, 0.3, (npol, N, 10))
    tg_all[:, sample, :] = tg
    ref = bench.HostChainEnvs(model, sample, None, pods=pods)
    try:
        b.set(P.F_QPOS, np.tile(model.qpos_init(), (N, 1)))
        b.forward()
        b.set(P.F_PD_KP, np.tile(bench.PD_KP, (N, 1)))
        b.set(P.F_PD_KD, np.tile(bench.PD_KD, (N, 1)))
        b.set_drive_mode(P.DRIVE_PD)
        worst = 0.0
        for p in range(npol):
            b.set(P.F_PD_PTARGET, tg_all[p])
            b.step(bench.HOLD)
            ref.step(bench.HOLD, tg[p])
            q = b.get(P.F_QPOS)[sample]
            w, info = b.warnings()
            qr, cnt = ref.qpos(), ref.counts()
            assert np.array_equal(info[sample][:, :3], cnt), (p, info[sample][:4].tolist(), cnt[:4].tolist())
            err = np.max(np.abs(q - qr) / np.maximum(1.0, np.abs(qr)), axis=1)
            safe = ref.flip_margin > 1e-6
            assert np.all(err[safe] <= 1e-9), (p, float(err[safe].max()))
            assert np.all(np.max(np.abs(q - qr), axis=1)[~safe] < 2e-4)
            worst = max(worst, float(err[safe].max()) if safe.any() else 0.0)
        assert not w.any()
        print("randomised drive-pd: worst rel err %.2e over 64 envs x %d policy steps" % (worst, npol))
    finally:
        b.close()
        for hc in ref.chains:
            hc.close()


def test_randomised_envs_under_stress_targets_do_not_depend_on_the_form_of_the_fast_kernel(built):
    """Per-env parameter blocks + joints slammed into their limits (envs leave the 31-row tier in the middle of fused launches) + the
    three ways such substeps are finished (phys_batch_set_inplace 0 / 1 / 2, the launch in chunks or in one piece): every env reads
    its own block in every instantiation, so state, outputs and solver statistics must be the same bits throughout."""
    import test_drive_parity_gpu as D
    n, npol = 4096, 12
    tg = D._stress_targets(np.arange(n), npol)
    out = []
    for mode, chunks in ((0, 1), (1, 4), (2, 2)):
        model, params, b = _randomised_batch("cassie", n, seed=91)
        try:
            b.set_inplace(mode); b.set_chunks(chunks)
            b.set(P.F_QPOS, np.tile(model.qpos_init(), (n, 1)))
            b.forward()
            b.set(P.F_PD_KP, np.tile(bench.PD_KP, (n, 1)))
            b.set(P.F_PD_KD, np.tile(bench.PD_KD, (n, 1)))
            b.set_drive_mode(P.DRIVE_PD)
            rows = 0
            for p in range(npol):
                b.set(P.F_PD_PTARGET, tg[p])
                b.step((bench.HOLD, 20, 11)[p % 3])
                rows = max(rows, int(b.warnings()[1][:, 1].max()))
            w, info = b.warnings()
            out.append([b.get(P.F_QPOS), b.get(P.F_QVEL), b.get(P.F_QACC_WARMSTART), b.get(P.F_SENSORDATA), b.get(P.F_MEAS), b.get(P.F_CTRL), w, info[:, :3].copy()])
            forms = b.form_launches()
        finally:
            b.close()
        assert rows > 31, "the workload never left the 31-row tier"
        assert (forms[1] == 0) if mode == 0 else (forms[0] == 0) if mode == 1 else (forms[0] > 0 and forms[1] > 0), (mode, forms)
    for k in (1, 2):
        for a, c in zip(out[0], out[k]):
            assert a.tobytes() == c.tobytes(), k


def test_parameter_blocks_and_per_env_models_do_not_mix(built):
    model = Model("cassie")
    b = Batch(model, 4)
    try:
        b.randomize(P.P_DOF_DAMPING, np.tile(np.array(model.pod.dof_damping[: model.pod.nv]), (4, 1)))
        with pytest.raises(RuntimeError):
            b.set_model(model.pod, 1)
        b.set_model(model.pod, -1)      # replacing the shared model drops the blocks
        b.set_model(model.pod, 1)
        with pytest.raises(RuntimeError):
            b.set_const()
    finally:
        b.close()
