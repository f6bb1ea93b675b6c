"""Opportunistic access to the TRUE reference physics -- test infrastructure only.

The reference dlopens MuJoCo 2.1.0 from ~/.mujoco/mujoco210 (reference src/cassiemujoco.c:521-555) and steps it with
mj_step1 + mj_step2 (reference :1130-1134).  Neither that binary nor a `mujoco` wheel exists in the build container or
on the GPU box image, so everything here degrades to "unavailable"; on a machine that does have one, `find()` returns a
backend that runs the genuine step on the inputs the oracle and the HIP kernel get:

  1. a mujoco210 directory with bin/libmujoco210.so and include/mujoco.h (searched: $MUJOCO_DIR, $MUJOCO_PATH, $MUJOCO_PY_MUJOCO_PATH,
     ~/.mujoco/mujoco210, /opt/mujoco210, /usr/local/mujoco210, and the parent of any LD_LIBRARY_PATH entry that holds
     libmujoco210*.so) -- oracle/mj_harness.c is compiled against it into oracle/_ref/ and driven through ctypes;
  2. an importable `mujoco` Python wheel (a later MuJoCo: its version is reported next to every number).

Both need the reference's MJCF files: $CASSIE_MJCF_DIR, /root/reference/model, or oracle/_ref/model (staged by
oracle/build_ref.sh in the build container, git-ignored, travels to the GPU box with the snapshot).
"""
import ctypes
import glob
import os
import subprocess
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(REPO, "oracle", "_ref")


def mjcf_path(model_name):
    for d in (os.environ.get("CASSIE_MJCF_DIR"), "/root/reference/model", os.path.join(REF_DIR, "model")):
        if d and os.path.exists(os.path.join(d, model_name + ".xml")):
            return os.path.join(d, model_name + ".xml")
    return None


def _find_mujoco210():
    cands = [os.environ.get("MUJOCO_DIR"), os.environ.get("MUJOCO_PATH"), os.environ.get("MUJOCO_PY_MUJOCO_PATH"),
             os.path.expanduser("~/.mujoco/mujoco210"), "/opt/mujoco210", "/usr/local/mujoco210"]
    # a driver-side install that is only on the loader path: .../mujoco210/bin in LD_LIBRARY_PATH
    for entry in os.environ.get("LD_LIBRARY_PATH", "").split(os.pathsep):
        if entry and glob.glob(os.path.join(entry, "libmujoco210*.so")):
            cands.append(os.path.dirname(entry.rstrip("/")))
    seen = set()
    for d in cands:
        if not d or d in seen:
            continue
        seen.add(d)
        libs = glob.glob(os.path.join(d, "bin", "libmujoco210*.so"))
        libs = [p for p in libs if "nogl" in p] + [p for p in libs if "nogl" not in p]   # the headless build first
        if libs and os.path.exists(os.path.join(d, "include", "mujoco.h")):
            return d, libs[0]
    return None


class _CHarness:
    """libmujoco210 through oracle/mj_harness.c."""

    def __init__(self, mjdir, libpath):
        os.makedirs(REF_DIR, exist_ok=True)
        so = os.path.join(REF_DIR, "libmj_harness.so")
        src = os.path.join(REPO, "oracle", "mj_harness.c")
        libname = os.path.basename(libpath)[3:-3]
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-I" + os.path.join(mjdir, "include"), src, "-L" + os.path.join(mjdir, "bin"),
                               "-l" + libname, "-Wl,-rpath," + os.path.join(mjdir, "bin"), "-o", so])
        self.L = ctypes.CDLL(so)
        self.L.mjh_load.restype = ctypes.c_void_p
        self.L.mjh_load.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
        for fn, args in (("mjh_free", 1), ("mjh_sizes", 2), ("mjh_set_hfield", 3), ("mjh_reset", 1), ("mjh_set_state", 4), ("mjh_forward", 1),
                         ("mjh_step", 4), ("mjh_get", 7), ("mjh_get_consts", 4)):
            getattr(self.L, fn).argtypes = [ctypes.c_void_p] * args
            getattr(self.L, fn).restype = None
        self.L.mjh_set_hfield.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        for fn, args in (("mjh_stage_sizes", 2), ("mjh_get_stages", 14), ("mjh_set_ctrl", 2)):
            getattr(self.L, fn).argtypes = [ctypes.c_void_p] * args
            getattr(self.L, fn).restype = None
        self.kind = "libmujoco210"
        self.version = "mj_version() = %d" % self.L.mjh_version_number()

    def sim(self, xml):
        return _CSim(self.L, xml)


class _CSim:
    def __init__(self, L, xml):
        err = ctypes.create_string_buffer(1000)
        self.L, self.h = L, L.mjh_load(xml.encode(), err, len(err))
        if not self.h:
            raise RuntimeError("mj_loadXML failed: " + err.value.decode())
        sz = (ctypes.c_int * 8)()
        L.mjh_sizes(self.h, sz)
        self.nq, self.nv, self.nu, self.nbody, self.nsensordata = sz[0], sz[1], sz[2], sz[3], sz[4]

    def set_hfield(self, data):
        a = np.ascontiguousarray(data, dtype=np.float32)
        self.L.mjh_set_hfield(self.h, a.ctypes.data, a.size)

    def set_state(self, qpos, qvel, warm=None):
        q, v = np.ascontiguousarray(qpos, dtype=np.float64), np.ascontiguousarray(qvel, dtype=np.float64)
        w = None if warm is None else np.ascontiguousarray(warm, dtype=np.float64)
        self.L.mjh_set_state(self.h, q.ctypes.data, v.ctypes.data, None if w is None else w.ctypes.data)

    def forward(self):
        self.L.mjh_forward(self.h)

    def step(self, ctrl):
        c = np.ascontiguousarray(ctrl, dtype=np.float64)
        self.L.mjh_step(self.h, c.ctypes.data, None, None)

    def get(self):
        q, v, a = np.zeros(self.nq), np.zeros(self.nv), np.zeros(self.nv)
        sd, av, cnt = np.zeros(self.nsensordata), np.zeros(self.nu), (ctypes.c_int * 3)()
        self.L.mjh_get(self.h, q.ctypes.data, v.ctypes.data, a.ctypes.data, sd.ctypes.data, av.ctypes.data, cnt)
        return dict(qpos=q, qvel=v, qacc=a, sensordata=sd, actuator_velocity=av, counts=tuple(cnt))

    def set_ctrl(self, ctrl):
        c = np.ascontiguousarray(ctrl, dtype=np.float64)
        self.L.mjh_set_ctrl(self.h, c.ctypes.data)

    def stages(self):
        """Per-stage quantities of the last forward pass (mj_forward, or the one inside the last step): qM, the constraint rows
        (efc_J when stored dense, aref, R, force, pos, diagApprox, b, type, id), the contact list, qacc_smooth."""
        sz = (ctypes.c_int * 4)()
        self.L.mjh_stage_sizes(self.h, sz)
        ncon, nefc, nv, sparse = sz[0], sz[1], sz[2], sz[3]
        qM, J = np.zeros((nv, nv)), np.zeros((nefc, nv))
        vec = {k: np.zeros(nefc) for k in ("aref", "R", "force", "pos", "diagApprox", "b")}
        typ, eid = np.zeros(nefc, dtype=np.int32), np.zeros(nefc, dtype=np.int32)
        con, coni, qs = np.zeros((ncon, 13)), np.zeros((ncon, 3), dtype=np.int32), np.zeros(nv)
        self.L.mjh_get_stages(self.h, qM.ctypes.data, J.ctypes.data, vec["aref"].ctypes.data, vec["R"].ctypes.data, vec["force"].ctypes.data,
                              vec["pos"].ctypes.data, vec["diagApprox"].ctypes.data, vec["b"].ctypes.data, typ.ctypes.data, eid.ctypes.data,
                              con.ctypes.data, coni.ctypes.data, qs.ctypes.data)
        return dict(ncon=ncon, nefc=nefc, qM=qM, efc_J=None if sparse else J, efc_type=typ, efc_id=eid, contact_dist=con[:, 0], contact_pos=con[:, 1:4],
                    contact_frame=con[:, 4:13], contact_geom=coni[:, :2], contact_dim=coni[:, 2], qacc_smooth=qs, **{"efc_" + k: v for k, v in vec.items()})

    def close(self):
        if self.h:
            self.L.mjh_free(self.h)
            self.h = None


class _Wheel:
    """An importable `mujoco` wheel (MuJoCo >= 2.1.2; NOT the 2.1.0 the reference pins -- its version is reported)."""

    def __init__(self, mod):
        self.mj, self.kind, self.version = mod, "mujoco wheel", getattr(mod, "__version__", "?")

    def sim(self, xml):
        return _WheelSim(self.mj, xml)


class _WheelSim:
    def __init__(self, mj, xml):
        self.mj = mj
        self.m = mj.MjModel.from_xml_path(xml)
        self.d = mj.MjData(self.m)
        self.nq, self.nv, self.nu, self.nbody, self.nsensordata = self.m.nq, self.m.nv, self.m.nu, self.m.nbody, self.m.nsensordata

    def set_hfield(self, data):
        self.m.hfield_data[:] = np.asarray(data, dtype=np.float32).ravel()

    def set_state(self, qpos, qvel, warm=None):
        self.d.qpos[:], self.d.qvel[:] = qpos, qvel
        if warm is not None:
            self.d.qacc_warmstart[:] = warm

    def forward(self):
        self.mj.mj_forward(self.m, self.d)

    def step(self, ctrl):
        self.d.ctrl[:] = ctrl
        self.mj.mj_step1(self.m, self.d)
        self.mj.mj_step2(self.m, self.d)

    def get(self):
        d = self.d
        return dict(qpos=d.qpos.copy(), qvel=d.qvel.copy(), qacc=d.qacc.copy(), sensordata=d.sensordata.copy(),
                    actuator_velocity=d.actuator_velocity.copy(), counts=(int(d.ncon), int(d.nefc), int(d.solver_iter[0]) if np.ndim(d.solver_iter) else int(d.solver_iter)))

    def set_ctrl(self, ctrl):
        self.d.ctrl[:] = ctrl

    def stages(self):
        m, d, mj = self.m, self.d, self.mj
        nv, nefc, ncon = m.nv, int(d.nefc), int(d.ncon)
        qM = np.zeros((nv, nv))
        mj.mj_fullM(m, qM, d.qM)
        sparse = bool(mj.mj_isSparse(m))
        J = None if sparse else np.array(d.efc_J, dtype=float).reshape(-1)[: nefc * nv].reshape(nefc, nv)
        g = lambda name: np.array(getattr(d, name), dtype=float).reshape(-1)[:nefc]
        geoms = np.array([[(c.geom[0] if hasattr(c, "geom") else c.geom1), (c.geom[1] if hasattr(c, "geom") else c.geom2)] for c in d.contact[:ncon]], dtype=np.int32).reshape(ncon, 2)
        return dict(ncon=ncon, nefc=nefc, qM=qM, efc_J=J, efc_type=np.array(d.efc_type, dtype=np.int32).reshape(-1)[:nefc], efc_id=np.array(d.efc_id, dtype=np.int32).reshape(-1)[:nefc],
                    efc_aref=g("efc_aref"), efc_R=g("efc_R"), efc_force=g("efc_force"), efc_pos=g("efc_pos"), efc_diagApprox=g("efc_diagApprox"), efc_b=g("efc_b"),
                    contact_dist=np.array([c.dist for c in d.contact[:ncon]]), contact_pos=np.array([c.pos for c in d.contact[:ncon]]).reshape(ncon, 3),
                    contact_frame=np.array([c.frame for c in d.contact[:ncon]]).reshape(ncon, 9), contact_geom=geoms,
                    contact_dim=np.array([c.dim for c in d.contact[:ncon]], dtype=np.int32), qacc_smooth=np.array(d.qacc_smooth, dtype=float))

    def close(self):
        pass


def find():
    """A backend for the genuine reference physics, or None ("true reference unavailable")."""
    hit = _find_mujoco210()
    if hit is not None:
        try:
            return _CHarness(*hit)
        except Exception as exc:
            print("mujoco_ref: found %s but could not build the harness: %r" % (hit[0], exc))
    try:
        import mujoco
        return _Wheel(mujoco)
    except Exception:
        return None


def pd_ctrl(pod, qpos, qvel, ptarget, kp, kd):
    """The joint PD -> motor-side torque law the kernel's PD mode and oracle co_pd_ctrl implement (pd_input motor law +
    reference motor() speed-torque limit, reference src/cassiemujoco.c:638-664), in numpy."""
    out = np.zeros(pod.nu)
    for u in range(pod.nu):
        ratio, tmax = pod.act_gear[u], pod.act_ctrlrange[u][1]
        q, qd = qpos[pod.act_qposadr[u]], qvel[pod.act_dofid[u]]
        tau = kp[u] * (ptarget[u] - q) - kd[u] * qd
        wmax = pod.act_maxrpm[u] * (2.0 * np.pi / 60.0)
        tlim = min(max(2 * tmax * (1 - abs(ratio * qd) / wmax), 0.0), tmax)
        out[u] = np.copysign(min(abs(tau / ratio), tlim), tau)
    return out


def rollout(ref, model, model_name, q0, targets, kp, kd, nsteps, hold, hfield=None):
    """[k] envs x nsteps of the PD workload on the true reference and on the oracle, same inputs.
    targets: [npolicy][k][nu].  Returns per-env final states of both, and the reference's wall time."""
    from oracle_py import Oracle
    xml = mjcf_path(model_name)
    if xml is None:
        raise FileNotFoundError("no MJCF for %s (see mujoco_ref.mjcf_path)" % model_name)
    pod = model.pod
    k = targets.shape[1]
    res, t_ref = [], 0.0
    for e in range(k):
        s = ref.sim(xml)
        if hfield is not None:
            s.set_hfield(hfield)
        s.set_state(q0, np.zeros(pod.nv), np.zeros(pod.nv))
        o = Oracle(pod, q0)
        worst = 0.0
        for t in range(nsteps):
            pt = targets[t // hold][e]
            st = s.get() if t else dict(qpos=np.array(q0, dtype=float), qvel=np.zeros(pod.nv))
            c = pd_ctrl(pod, st["qpos"], st["qvel"], pt, kp, kd)
            t0 = time.perf_counter()
            s.step(c)
            t_ref += time.perf_counter() - t0
            o.pd_ctrl(pt, kp, kd)
            o.step()
            if (t + 1) % hold == 0:
                worst = max(worst, float(np.max(np.abs(s.get()["qpos"] - o.qpos))))
        g = s.get()
        res.append(dict(qpos_ref=g["qpos"], qpos_oracle=o.qpos.copy(), counts_ref=g["counts"], counts_oracle=(o.d.ncon, o.d.nefc, o.d.solver_iter),
                        worst_qpos_err=worst))
        s.close()
    return res, t_ref


def bench_and_compare(ref, model_name, q0, targets, kp, kd, nsteps, hold):
    from cassie_amd import Model
    model = Model(model_name)
    res, t_ref = rollout(ref, model, model_name, q0, targets, kp, kd, nsteps, hold)
    k = len(res)
    return {"kind": ref.kind, "version": ref.version, "cores": 1, "value": k * nsteps / t_ref, "unit": "env-steps/s",
            "sample": "%d envs x %d steps, genuine mj_step1 + mj_step2, same PD workload" % (k, nsteps),
            "max_qpos_err_oracle_vs_true_reference": max(r["worst_qpos_err"] for r in res),
            "counts_equal": bool(all(tuple(r["counts_ref"]) == tuple(r["counts_oracle"]) for r in res))}
