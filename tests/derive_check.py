"""Checks a derived block (phys_batch_derive: F_DERIVED + F_QM) against values computed independently from the ORACLE's state with
numpy: centre of mass / momentum from first principles, Jacobians by the oracle's co_jac, the mass matrix, and the contact-force
sums.  Shared by the emulator test (tests/test_derive_emu.py) and the GPU test (tests/test_derive_gpu.py) -- test infrastructure."""
import ctypes

import numpy as np

from cassie_amd import phys as P
from oracle_py import Oracle, arr, lib as olib


def foot_ids(cassie):
    return [cassie.name2id(1, "left-foot"), cassie.name2id(1, "right-foot"), cassie.name2id(6, "left-heel"), cassie.name2id(6, "right-heel"),
            cassie.name2id(6, "left-toe"), cassie.name2id(6, "right-toe")]


def check_derived_block(cassie, q, v, D, QM, ids, ctrl=None):
    """q, v [n][nq / nv]: the states the block was derived at; D [n][DRV_DIM], QM [n][nv][nv]; ctrl [n][nu] if motor torques acted."""
    pod = cassie.pod
    n = len(q)
    mass = np.array(pod.body_mass[: pod.nbody])
    for e in range(n):
        o = Oracle(pod, q[e])
        o.qvel[:] = v[e]
        if ctrl is not None:
            o.ctrl[:] = ctrl[e]
        o.forward()
        d = o.d
        xipos = arr(d.xipos)[: pod.nbody]
        cvel = arr(d.cvel)[: pod.nbody]
        croot = arr(d.subtree_com)[1]
        com = (mass[:, None] * xipos).sum(0) / mass.sum()
        vb = cvel[:, 3:] + np.cross(cvel[:, :3], xipos - croot)
        vcom = (mass[:, None] * vb).sum(0) / mass.sum()
        assert np.allclose(D[e, P.DRV_COM_POS: P.DRV_COM_POS + 3], com, atol=1e-12)
        assert np.allclose(D[e, P.DRV_COM_VEL: P.DRV_COM_VEL + 3], vcom, atol=1e-12)
        assert abs(D[e, P.DRV_MASS] - 33.312) < 1e-9
        ximat = arr(d.ximat)[: pod.nbody].reshape(-1, 3, 3)
        L = np.zeros(3)
        for b in range(1, pod.nbody):
            I = ximat[b] @ np.diag(pod.body_inertia[b][:3]) @ ximat[b].T
            L += I @ cvel[b, :3] + np.cross(xipos[b] - com, mass[b] * (vb[b] - vcom))
        assert np.allclose(D[e, P.DRV_ANGMOM: P.DRV_ANGMOM + 3], L, atol=1e-11)
        assert np.allclose(QM[e], o.qM, atol=1e-11)
        off = np.sqrt(0.01762 ** 2 + 0.05219 ** 2)
        for side in range(2):
            foot = ids[side]
            assert np.allclose(D[e, P.DRV_FOOT_POS + 3 * side: P.DRV_FOOT_POS + 3 * side + 3], o.xpos[foot] - [0, 0, off], atol=1e-12)
            assert np.allclose(D[e, P.DRV_FOOT_VEL + 6 * side: P.DRV_FOOT_VEL + 6 * side + 6], cvel[foot], atol=1e-12)
            jp = ((ctypes.c_double * 40) * 3)()
            jr = ((ctypes.c_double * 40) * 3)()
            olib().co_jac(ctypes.byref(pod), ctypes.byref(d), foot, (ctypes.c_double * 3)(*o.xpos[foot]), jp, jr)
            Jp = D[e, P.DRV_FOOT_JACP + side * 3 * P.MAXV: P.DRV_FOOT_JACP + (side + 1) * 3 * P.MAXV].reshape(3, P.MAXV)
            Jr = D[e, P.DRV_FOOT_JACR + side * 3 * P.MAXV: P.DRV_FOOT_JACR + (side + 1) * 3 * P.MAXV].reshape(3, P.MAXV)
            assert np.allclose(Jp, np.array(jp), atol=1e-12) and np.allclose(Jr, np.array(jr), atol=1e-12)
            # the foot's origin velocity is J qvel
            assert np.allclose(Jp[:, : pod.nv] @ v[e], cvel[foot, 3:] + np.cross(cvel[foot, :3], o.xpos[foot] - croot), atol=1e-11)
            # contact forces: foot total = heel + toe (reference test_heelforce.c:56-57) = the oracle's contact forces on the foot
            F = np.zeros(3)
            for c in range(d.ncon):
                con = d.contact[c]
                b1, b2 = pod.geom_bodyid[con.geom1], pod.geom_bodyid[con.geom2]
                if foot not in (b1, b2):
                    continue
                fr = np.array(con.frame).reshape(3, 3)
                a = con.efc_address
                if con.dim == 1:
                    fc = np.array([d.efc_force[a], 0, 0])
                else:
                    ef = np.array([d.efc_force[a + i] for i in range(4)])
                    fc = np.array([ef.sum(), con.friction[0] * (ef[0] - ef[1]), con.friction[0] * (ef[2] - ef[3])])
                F += (-1.0 if b1 == foot else 1.0) * (fr.T @ fc)
            ff = D[e, P.DRV_FOOT_FORCE + 6 * side: P.DRV_FOOT_FORCE + 6 * side + 3]
            toe, heel = D[e, P.DRV_TOE_FORCE + 3 * side: P.DRV_TOE_FORCE + 3 * side + 3], D[e, P.DRV_HEEL_FORCE + 3 * side: P.DRV_HEEL_FORCE + 3 * side + 3]
            assert np.allclose(ff, F, rtol=1e-8, atol=1e-7)
            assert np.allclose(toe + heel, ff, atol=1e-10)
