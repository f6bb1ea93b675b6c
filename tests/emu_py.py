"""ctypes access to the CPU wave emulator running the real physics_kernel.h -- test infrastructure only."""
import ctypes
import os

import numpy as np

from cassie_amd._lib import CmDriveState, CmModel, REPO_DIR

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(os.path.join(REPO_DIR, "tests", "emu", "libcassie_emu.so"))
        _lib.emu_phys_run.argtypes = [ctypes.POINTER(CmModel)] + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 18
        _lib.emu_set_drive_io.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 5
        _lib.emu_derive.argtypes = [ctypes.POINTER(CmModel), ctypes.c_int] + [ctypes.c_void_p] * 14
    return _lib


class EmuBatch:
    """Same state arrays as cassie_amd.Batch, stepped by the emulated kernel."""

    def __init__(self, pod, nenv):
        self.pod, self.nenv = pod, nenv
        z = lambda n: np.zeros((nenv, n))
        self.qpos = np.tile(np.array(pod.qpos0[: pod.nq]), (nenv, 1))
        self.qvel, self.qacc_warmstart, self.qacc = z(pod.nv), z(pod.nv), z(pod.nv)
        self.ctrl, self.actuator_velocity = z(pod.nu), z(pod.nu)
        self.sensordata = z(pod.nsensordata)
        self.time = np.zeros(nenv)
        self.qfrc_applied, self.xfrc_applied = None, None
        self.warn = np.zeros(nenv, dtype=np.int32)
        self.info = np.zeros((nenv, 4), dtype=np.int32)
        self.xpos = z(pod.nbody * 3)
        self.xquat = z(pod.nbody * 4)
        self.pd_ptarget = self.pd_kp = self.pd_kd = None
        self.hfield = None          # float32 [nrow * ncol] shared by all envs
        # drive-level I/O (mode 0 = off): filter histories / delay lines, commands [nenv][nu + 1], measurement block
        self.drive_mode = 0
        self.drive_state = (CmDriveState * nenv)()
        self.drive_cmd = z(pod.nu + 1)
        self.meas = z(56)
        self.pd_dtarget = self.pd_torque = None

    def _run(self, nsub, integrate):
        p = lambda a: None if a is None else a.ctypes.data
        lib().emu_set_drive_io(self.drive_mode, ctypes.addressof(self.drive_state), p(self.drive_cmd), p(self.meas),
                               p(self.pd_dtarget), p(self.pd_torque))
        lib().emu_phys_run(ctypes.byref(self.pod), self.nenv, nsub, integrate, p(self.qpos), p(self.qvel),
                           p(self.qacc_warmstart), p(self.time), p(self.ctrl), p(self.qfrc_applied),
                           p(self.xfrc_applied), p(self.qacc), p(self.sensordata), p(self.actuator_velocity),
                           p(self.warn), p(self.info), p(self.xpos), p(self.xquat), p(self.pd_ptarget), p(self.pd_kp),
                           p(self.pd_kd), p(self.hfield))

    def derive(self, ids):
        """phys_batch_derive on the emulator: -> (derived [nenv][CM_DRV_DIM], qM [nenv][nv][nv])."""
        from cassie_amd import phys as P
        p = lambda a: None if a is None else a.ctypes.data
        derived = np.zeros((self.nenv, P.DRV_DIM))
        qM = np.zeros((self.nenv, self.pod.nv, self.pod.nv))
        idarr = np.asarray(ids, dtype=np.int32)
        lib().emu_derive(ctypes.byref(self.pod), self.nenv, p(self.qpos), p(self.qvel), p(self.qacc_warmstart), p(self.time), p(self.ctrl),
                         p(self.qacc), p(self.sensordata), p(self.actuator_velocity), p(self.warn), p(self.info), p(self.hfield),
                         p(idarr), p(derived), p(qM))
        return derived, qM

    def step(self, nsub=1):
        self._run(nsub, 1)

    def forward(self):
        self._run(1, 0)
