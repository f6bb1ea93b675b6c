#!/usr/bin/env python3
"""What the collision shortcuts of DESIGN.md 4.2 cost in fidelity -- ORACLE SIDE ONLY, no GPU (task 6).

MuJoCo reports one contact per penetrated height-field prism (reference model/cassie_hfield.xml:4 asks nconmax = 300 for that)
and up to eight contacts per box pair; the shipped definitions report at most two contacts per capsule (four with
CM_FLAG_HFMULTI) and four per box pair, under caps of 16 contacts / 63 rows per env-step.  This script builds a STUDY variant
of the oracle (oracle/_study/, -DCO_STUDY with CM_MAXCON = 192 / CM_MAXEFC = 800; never used by a parity test) whose extra modes
report one contact per penetrated grid triangle and up to eight per box pair, runs the benchmark's PD workload on the terrain of
reference example/test_hfield.py:39-41 and on the tray model with every definition, and writes what differs: contact / row
counts (what a MuJoCo-shaped contact set would need from the kernel), pelvis height, foot force, and how far the trajectories
move apart.  Output: profiles/round4/collision_fidelity.json."""
import ctypes
import json
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # (lives under tests/: it builds and runs the oracle, which only test infrastructure may)
sys.path.insert(0, os.path.join(REPO, "cassie-mujoco-sim_amd")); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
import bench                                             # noqa: E402
import golden_physics as G                               # noqa: E402
from cassie_amd import Model, cstruct                    # noqa: E402
from cassie_amd import phys as P                         # noqa: E402
from cassie_amd._lib import MACROS, CmModel              # noqa: E402

MAXCON, MAXEFC = 192, 800
SO = os.path.join(REPO, "oracle", "_study", "libcassie_oracle_study.so")


def build():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-fPIC", "-shared", "-fopenmp", "-DCO_STUDY", "-DCM_MAXCON=%d" % MAXCON, "-DCM_MAXEFC=%d" % MAXEFC,
                           "-I" + os.path.join(REPO, "cassie-mujoco-sim_amd", "csrc"), "-I" + os.path.join(REPO, "oracle"),
                           os.path.join(REPO, "oracle", "cassie_oracle.c"), "-o", SO, "-lm"])
    with open(os.path.join(REPO, "oracle", "cassie_oracle.h")) as f:
        types = cstruct.parse_structs(f.read(), dict(MACROS, CM_MAXCON=MAXCON, CM_MAXEFC=MAXEFC))
    L = ctypes.CDLL(SO)
    L.co_sizeof_data.restype = ctypes.c_ulong
    assert L.co_sizeof_data() == ctypes.sizeof(types["co_data_t"])
    return L, types["co_data_t"]


def rollout(L, CoData, model, name, hfield_mode=0, box_keep=4, flags=(), nsteps=1000, nenv=8, perturb=0.0):
    """The benchmark's exact-state PD workload (bench.pd_targets, gains of reference example/cassietest_jac.py) on `nenv` envs."""
    for f in flags:
        model.set_flag(f, True)
    pod = model.pod
    L.co_study_set_hfield_mode(hfield_mode); L.co_study_set_box_contacts(box_keep)
    hf = G.terrain(name)
    keep = None if hf is None else np.ascontiguousarray(hf, dtype=np.float32)
    L.co_set_hfield(None if keep is None else ctypes.c_void_p(keep.ctypes.data))
    ds = (CoData * nenv)()
    q0 = G.initial_qpos(model, name)
    for e in range(nenv):
        L.co_reset(ctypes.byref(pod), ctypes.byref(ds[e]))
        np.ctypeslib.as_array(ds[e].qpos)[: pod.nq] = q0[e]
        np.ctypeslib.as_array(ds[e].qpos)[7: pod.nq] += perturb * np.random.default_rng(77 + e).standard_normal(pod.nq - 7)
    tg = bench.pd_targets(np.arange(nenv), nsteps // bench.HOLD + 1)
    kp, kd = np.tile(bench.PD_KP, (nenv, 1)), np.tile(bench.PD_KD, (nenv, 1))
    foot_bodies = [model.name2id(1, "left-foot"), model.name2id(1, "right-foot")]
    rec = dict(ncon=[], nefc=[], z=[], fz=[], qpos={}, iters=[])
    t0 = time.time()
    for s in range(nsteps):
        pt = np.ascontiguousarray(tg[s // bench.HOLD])
        L.co_step_batch(ctypes.byref(pod), ctypes.byref(ds), nenv, 1, ctypes.c_void_p(pt.ctypes.data), ctypes.c_void_p(kp.ctypes.data), ctypes.c_void_p(kd.ctypes.data), 8)
        rec["ncon"].append([d.ncon for d in ds]); rec["nefc"].append([d.nefc for d in ds]); rec["iters"].append([d.solver_iter for d in ds])
        rec["z"].append([d.qpos[2] for d in ds])
        fz = []
        for d in ds:                                     # vertical component of the net contact force on the two feet
            f = 0.0
            for c in range(d.ncon):
                con = d.contact[c]
                b1, b2 = pod.geom_bodyid[con.geom1], pod.geom_bodyid[con.geom2]
                if b1 in foot_bodies or b2 in foot_bodies:
                    a, fr = con.efc_address, con.frame
                    if con.dim == 1:
                        fc = (d.efc_force[a], 0.0, 0.0)
                    else:
                        ef = [d.efc_force[a + i] for i in range(4)]
                        fc = (sum(ef), con.friction[0] * (ef[0] - ef[1]), con.friction[0] * (ef[2] - ef[3]))
                    f += (1.0 if b2 in foot_bodies else -1.0) * (fr[2] * fc[0] + fr[5] * fc[1] + fr[8] * fc[2])
            fz.append(f)
        rec["fz"].append(fz)
        if s + 1 in (50, 200, 500, 1000):
            rec["qpos"][s + 1] = np.array([np.ctypeslib.as_array(d.qpos)[: pod.nq].copy() for d in ds])
    rec["seconds"] = time.time() - t0
    rec["warn"] = dict(contact_full=int(sum(d.warn_contact_full for d in ds)), constraint_full=int(sum(d.warn_constraint_full for d in ds)), diverged=int(sum(d.diverged for d in ds)))
    for f in flags:
        model.set_flag(f, False)
    L.co_set_hfield(None)
    for k in ("ncon", "nefc", "z", "fz", "iters"):
        rec[k] = np.array(rec[k], dtype=float)
    return rec


def summary(rec, ref=None):
    nefc, ncon = rec["nefc"], rec["ncon"]
    standing = rec["z"] > 0.6                           # env-steps with the pelvis up
    out = {"contacts_per_step_mean": float(ncon.mean()), "contacts_per_step_max": int(ncon.max()),
           "rows_per_step_mean": float(nefc.mean()), "rows_per_step_p99": float(np.percentile(nefc, 99)), "rows_per_step_max": int(nefc.max()),
           "frac_env_steps_over_63_rows": float((nefc > 63).mean()), "frac_env_steps_over_127_rows": float((nefc > 127).mean()),
           "frac_env_steps_over_16_contacts": float((ncon > 16).mean()),
           "pgs_sweeps_mean": float(rec["iters"].mean()),
           "pelvis_height_mean_at_step": {str(s): float(rec["z"][s - 1].mean()) for s in (50, 200, 500, 1000)},
           "envs_with_pelvis_above_0.6m_at_step_1000": int((rec["z"][-1] > 0.6).sum()),
           "net_vertical_contact_force_on_the_feet_mean_while_standing_N": float(rec["fz"][standing].mean()) if standing.any() else None,
           "caps_hit_or_diverged_envs": rec["warn"], "oracle_seconds": round(rec["seconds"], 1)}
    if ref is not None:
        out["max_abs_qpos_difference_to_the_default_definition_at_step"] = {
            str(s): {"median_over_envs": float(np.median(np.abs(rec["qpos"][s] - ref["qpos"][s]).max(axis=1))), "max_over_envs": float(np.abs(rec["qpos"][s] - ref["qpos"][s]).max())}
            for s in (50, 200, 500, 1000)}
        out["pelvis_height_rms_difference_to_default_first_200_steps_m"] = float(np.sqrt(((rec["z"][:200] - ref["z"][:200]) ** 2).mean()))
    return out


def main():
    L, CoData = build()
    report = {"what": __doc__.split("\n\n")[0], "workload": "bench.py's PD workload (exact-state PD), envs 0..7 of tests/golden_physics.py, 1000 steps, no restarts",
              "study_oracle": "oracle/cassie_oracle.c built with -DCO_STUDY -DCM_MAXCON=%d -DCM_MAXEFC=%d" % (MAXCON, MAXEFC)}
    hf = Model("cassie_hfield")
    runs = {"default (<= 2 contacts per capsule; caps lifted)": dict(),
            "YARDSTICK: default, initial joint angles perturbed by 1e-9 rad (how fast ANY difference grows)": dict(perturb=1e-9),
            "CM_FLAG_HFMULTI (<= 4 per capsule)": dict(flags=(P.FLAG_HFMULTI,)),
            "CM_FLAG_HFDENSE (denser sampling, <= 2 per capsule)": dict(flags=(P.FLAG_HFDENSE,)),
            "one contact per penetrated grid triangle (MuJoCo-shaped)": dict(hfield_mode=1)}
    ref = None
    report["cassie_hfield"] = {}
    for label, kw in runs.items():
        rec = rollout(L, CoData, hf, "cassie_hfield", **kw)
        if ref is None:
            ref = rec
        report["cassie_hfield"][label] = summary(rec, None if rec is ref else ref)
        print(label, json.dumps(report["cassie_hfield"][label])[:400], flush=True)
    # sensitivity yardstick: the default definition against itself with qpos perturbed by 1e-9 (how fast ANY difference grows)
    tray = Model("cassie_tray_box")
    report["cassie_tray_box"] = {}
    ref = None
    for label, kw in {"default (<= 4 contacts per box pair)": dict(), "up to 8 contacts per box pair (MuJoCo-shaped)": dict(box_keep=8)}.items():
        rec = rollout(L, CoData, tray, "cassie_tray_box", **kw)
        if ref is None:
            ref = rec
        report["cassie_tray_box"][label] = summary(rec, None if rec is ref else ref)
        print(label, json.dumps(report["cassie_tray_box"][label])[:400], flush=True)
    os.makedirs(os.path.join(REPO, "profiles", "round4"), exist_ok=True)
    with open(os.path.join(REPO, "profiles", "round4", "collision_fidelity.json"), "w") as f:
        json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
