"""Shared by the emulator and the GPU tests of per-env domain randomisation: random parameter sets in the
reference's own units, and the HOST's answer for them -- a host model edited through the views the reference's setters write
(reference src/cassiemujoco.c:1323-1436), then phys_model_set_const + phys_model_compile -- as a per-env cm_model_t for the
oracle and as the cm_envparams_t the device's set_const kernel has to reproduce bit for bit.  Test infrastructure only."""
import ctypes

import numpy as np

from cassie_amd import Model
from cassie_amd import phys as P
from cassie_amd._lib import CmEnvParams, CmModel

INPUT_FIELDS = ("body_mass", "body_ipos", "body_inertia", "dof_damping", "geom_friction")
DERIVED_FIELDS = ("meaninertia", "body_invweight0", "dof_invweight0", "jnt_liminvweight", "eq_invweight", "pair_invweight", "pair_friction")


def random_params(model, nenv, seed=0, mass=0.2, ipos=0.005, damping=0.5, friction=(0.4, 1.3)):
    """Per-env parameter sets around the model's own: masses x U(1 - mass, 1 + mass) with the principal inertias scaled alike,
    inertial offsets + U(-ipos, ipos) m, joint damping x U(1 - damping, 1 + damping), sliding friction of every collision
    geom ~ U(friction).  Arrays are in the layouts Batch.randomize takes ([nenv][dim], collision geoms in compiled order)."""
    pod = model.pod
    rng = np.random.default_rng(seed)
    nb, nv, ng = pod.nbody, pod.nv, pod.ngeom
    m0 = np.array(pod.body_mass[:nb])
    i0 = np.array([list(pod.body_ipos[b]) for b in range(nb)])
    in0 = np.array([list(pod.body_inertia[b]) for b in range(nb)])
    d0 = np.array(pod.dof_damping[:nv])
    f0 = np.array([list(pod.geom_friction[g]) for g in range(ng)])
    s = rng.uniform(1 - mass, 1 + mass, (nenv, nb))
    out = {
        "body_mass": m0 * s,
        "body_ipos": (i0[None] + rng.uniform(-ipos, ipos, (nenv, nb, 3)) * (m0[None, :, None] > 0)).reshape(nenv, -1),
        "body_inertia": (in0[None] * s[:, :, None]).reshape(nenv, -1),
        "dof_damping": d0 * rng.uniform(1 - damping, 1 + damping, (nenv, nv)),
        "geom_friction": np.tile(f0, (nenv, 1, 1)),
    }
    out["geom_friction"][:, :, 0] = rng.uniform(friction[0], friction[1], (nenv, ng))
    out["geom_friction"] = out["geom_friction"].reshape(nenv, -1)
    return out


PARAM_IDS = {"body_mass": P.P_BODY_MASS, "body_ipos": P.P_BODY_IPOS, "body_inertia": P.P_BODY_INERTIA,
             "dof_damping": P.P_DOF_DAMPING, "geom_friction": P.P_GEOM_FRICTION}


class HostEnvModels:
    """One host model edited env by env: pod(e) = the compiled model of env e (a copy), params(e) = its parameter block."""

    def __init__(self, name, flags=0):
        self.m = Model(name)
        for bit in (P.FLAG_HFDENSE, P.FLAG_HFMULTI, P.FLAG_HFPRISM, P.FLAG_BOX8):     # the contact options of the model the envs share
            if flags & bit:
                self.m.set_flag(bit, True)
        pod = self.m.pod
        self.nb, self.nv, self.ng = pod.nbody, pod.nv, pod.ngeom
        self.ngeom_full = self.m.size(P.SIZE_NGEOM)
        self.fullid = [pod.geom_fullid[g] for g in range(self.ng)]
        self.v_mass = self.m.array(P.M_BODY_MASS, self.nb)
        self.v_ipos = self.m.array(P.M_BODY_IPOS, 3 * self.nb)
        self.v_inertia = self.m.array(P.M_BODY_INERTIA, 3 * self.nb)
        self.v_damping = self.m.array(P.M_DOF_DAMPING, self.nv)
        self.v_friction = self.m.array(P.M_GEOM_FRICTION, 3 * self.ngeom_full)

    def pod(self, params, e, set_const=True):
        self.v_mass[:] = params["body_mass"][e]
        self.v_ipos[:] = params["body_ipos"][e]
        self.v_inertia[:] = params["body_inertia"][e]
        self.v_damping[:] = params["dof_damping"][e]
        fr = params["geom_friction"][e].reshape(self.ng, 3)
        for g, full in enumerate(self.fullid):
            self.v_friction[3 * full: 3 * full + 3] = fr[g]
        if set_const:
            self.m.set_const()      # mj_setConst role + compile
        else:
            self.m.compile()
        out = CmModel()
        ctypes.memmove(ctypes.byref(out), ctypes.byref(self.m.pod), ctypes.sizeof(CmModel))
        return out


def params_as_arrays(block, pod):
    """The arrays of a cm_envparams_t (a CmEnvParams or the `params` member of a CmModel) trimmed to the model's sizes."""
    nb, nv, ng, nj, ne, npair = pod.nbody, pod.nv, pod.ngeom, pod.njnt, pod.neq, pod.npair
    a = lambda f, *shape: np.ctypeslib.as_array(getattr(block, f)).reshape(-1)[: int(np.prod(shape))].reshape(shape).copy()
    return {
        "body_mass": a("body_mass", nb), "body_ipos": a("body_ipos", nb, 3), "body_inertia": a("body_inertia", nb, 3),
        "dof_damping": a("dof_damping", nv), "geom_friction": a("geom_friction", ng, 3),
        "meaninertia": np.array([block.meaninertia]), "body_invweight0": a("body_invweight0", nb, 2),
        "dof_invweight0": a("dof_invweight0", nv), "jnt_liminvweight": a("jnt_liminvweight", nj),
        "eq_invweight": a("eq_invweight", ne), "pair_invweight": a("pair_invweight", npair), "pair_friction": a("pair_friction", npair, 3),
    }


def assert_blocks_equal(got, want, pod, what, fields=INPUT_FIELDS + DERIVED_FIELDS):
    """Bit for bit: the device's set_const kernel performs the host compile's operations in the host compile's order."""
    g, w = params_as_arrays(got, pod), params_as_arrays(want, pod)
    for f in fields:
        if not np.array_equal(g[f].view(np.uint64), w[f].view(np.uint64)):
            bad = np.argwhere(g[f] != w[f])
            i = tuple(bad[0]) if len(bad) else ()
            raise AssertionError("%s: %s differs from the host compile at %s: %r vs %r (%d of %d entries)"
                                 % (what, f, i, g[f][i] if len(bad) else None, w[f][i] if len(bad) else None, len(bad), g[f].size))


def new_blocks(pod, nenv, params):
    """[nenv] CmEnvParams starting from the model's own block with the inputs of `params` written in (what
    phys_batch_randomize's scatter does on the device)."""
    blocks = (CmEnvParams * nenv)()
    for e in range(nenv):
        ctypes.memmove(ctypes.byref(blocks[e]), ctypes.byref(pod.params), ctypes.sizeof(CmEnvParams))
        for f in INPUT_FIELDS:
            dst = np.ctypeslib.as_array(getattr(blocks[e], f)).reshape(-1)
            row = np.asarray(params[f][e]).reshape(-1)
            dst[: row.size] = row
    return blocks
