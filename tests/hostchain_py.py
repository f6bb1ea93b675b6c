"""The host chain of one env (csrc/cassie_hostpath.c: cassie_hostenv_ethercat = cassie_motor_data + cassie_sensor_data,
pinned bit for bit to the reference's own compiled code by tests/test_hostpath.py) as the CHECKER of the device-side
drive-level models -- test infrastructure."""
import ctypes

import numpy as np

from cassie_amd import iotypes as T
from cassie_amd import phys as P
from cassie_amd._lib import lib

VP = ctypes.c_void_p


class HostModel(ctypes.Structure):
    _fields_ = [("drive_bits", ctypes.c_int * 10), ("joint_bits", ctypes.c_int * 6), ("gear", ctypes.c_double * 10),
                ("tmax", ctypes.c_double * 10), ("wmax", ctypes.c_double * 10)]


def _setup():
    L = lib()
    L.cassie_hostenv_alloc.restype = VP
    L.cassie_hostenv_free.argtypes = [VP]
    L.cassie_hostenv_cassie_out.restype = ctypes.POINTER(T.cassie_out_t)
    L.cassie_hostenv_cassie_out.argtypes = [VP]
    L.cassie_hostenv_ethercat.argtypes = [VP] * 7
    L.cassie_hostenv_drive_filter.restype = ctypes.POINTER(ctypes.c_int)
    L.cassie_hostenv_drive_filter.argtypes = [VP]
    L.cassie_hostenv_joint_filter.restype = ctypes.POINTER(ctypes.c_double)
    L.cassie_hostenv_joint_filter.argtypes = [VP]
    L.cassie_hostenv_torque_delay.restype = ctypes.POINTER(ctypes.c_double)
    L.cassie_hostenv_torque_delay.argtypes = [VP]
    L.cassie_hostmodel_from_model.argtypes = [VP, VP]
    # the closed Agility block itself (libagilitycassie.a, whole-archived into the product library like reference Makefile:18)
    L.cassie_core_sim_alloc.restype = VP
    L.cassie_core_sim_setup.argtypes = [VP]
    L.cassie_core_sim_free.argtypes = [VP]
    L.cassie_core_sim_step.argtypes = [VP] * 4
    return L


def drives(y):
    return [y.leftLeg.hipRollDrive, y.leftLeg.hipYawDrive, y.leftLeg.hipPitchDrive, y.leftLeg.kneeDrive, y.leftLeg.footDrive,
            y.rightLeg.hipRollDrive, y.rightLeg.hipYawDrive, y.rightLeg.hipPitchDrive, y.rightLeg.kneeDrive, y.rightLeg.footDrive]


def joints(y):
    return [y.leftLeg.shinJoint, y.leftLeg.tarsusJoint, y.leftLeg.footJoint, y.rightLeg.shinJoint, y.rightLeg.tarsusJoint, y.rightLeg.footJoint]


def meas_of(y):
    """cassie_out_t -> the CM_MEAS_* block the device writes."""
    m = np.zeros(P.MEAS_DIM)
    for i, d in enumerate(drives(y)):
        m[P.MEAS_DRIVE_POS + i], m[P.MEAS_DRIVE_VEL + i], m[P.MEAS_DRIVE_TORQUE + i] = d.position, d.velocity, d.torque
    for j, q in enumerate(joints(y)):
        m[P.MEAS_JOINT_POS + j], m[P.MEAS_JOINT_VEL + j] = q.position, q.velocity
    vn = y.pelvis.vectorNav
    m[P.MEAS_ORIENTATION: P.MEAS_ORIENTATION + 4] = list(vn.orientation)
    m[P.MEAS_ANGVEL: P.MEAS_ANGVEL + 3] = list(vn.angularVelocity)
    m[P.MEAS_LINACC: P.MEAS_LINACC + 3] = list(vn.linearAcceleration)
    m[P.MEAS_MAG: P.MEAS_MAG + 3] = list(vn.magneticField)
    return m


class HostChain:
    """One env's cassie_sim_step_ethercat host half."""

    def __init__(self, model):
        self.L = _setup()
        self.hm = HostModel()
        assert self.L.cassie_hostmodel_from_model(model._h, ctypes.byref(self.hm)) == 0
        self.env = self.L.cassie_hostenv_alloc()

    def ethercat(self, torques, sto, sensordata, actvel):
        """-> (ctrl[10], meas block, cassie_out_t) for commanded drive torques on the previous step's physics outputs."""
        u = T.cassie_in_t()
        legs = [u.leftLeg, u.rightLeg]
        for i in range(10):
            leg = legs[i // 5]
            [leg.hipRollDrive, leg.hipYawDrive, leg.hipPitchDrive, leg.kneeDrive, leg.footDrive][i % 5].torque = float(torques[i])
        self.L.cassie_hostenv_cassie_out(self.env).contents.pelvis.radio.channel[8] = 0.0 if sto else 1.0
        sd, av = np.ascontiguousarray(sensordata, dtype=np.float64), np.ascontiguousarray(actvel, dtype=np.float64)
        ctrl = np.zeros(10)
        y = T.cassie_out_t()
        self.L.cassie_hostenv_ethercat(self.env, ctypes.byref(self.hm), ctypes.byref(u), sd.ctypes.data, av.ctypes.data, ctrl.ctypes.data, ctypes.byref(y))
        return ctrl, meas_of(y), y

    def core_sim(self, user_torques):
        """cassie_core_sim_step of the REAL Agility block on this env's cassie_out_t (the previous step's measurements, as in
        cassie_sim_step, reference src/cassiemujoco.c:1141) -> (the ten torques of cassie_in_t, radio.channel[1..4] = the message queue)."""
        if getattr(self, "core", None) is None:
            self.core = self.L.cassie_core_sim_alloc()
            self.L.cassie_core_sim_setup(self.core)
        u, cin = T.cassie_user_in_t(), T.cassie_in_t()
        for i in range(10):
            u.torque[i] = float(user_torques[i])
        self.L.cassie_core_sim_step(self.core, ctypes.byref(u), self.L.cassie_hostenv_cassie_out(self.env), ctypes.byref(cin))
        legs = [cin.leftLeg, cin.rightLeg]
        tau = np.array([[legs[i // 5].hipRollDrive, legs[i // 5].hipYawDrive, legs[i // 5].hipPitchDrive, legs[i // 5].kneeDrive, legs[i // 5].footDrive][i % 5].torque for i in range(10)])
        return tau, [int(cin.pelvis.radio.channel[k]) for k in range(1, 5)]

    def reset(self):
        """A fresh cassie_sim_t's host state: zero filter histories and delay lines (and a fresh safety block: empty message queue)."""
        self.L.cassie_hostenv_free(self.env)
        self.env = self.L.cassie_hostenv_alloc()
        if getattr(self, "core", None) is not None:
            self.L.cassie_core_sim_setup(self.core)

    def state_bytes(self):
        """(drive FIR histories, joint IIR histories, torque delay lines) as raw bytes, in cm_drive_state_t's field order."""
        dx = np.ctypeslib.as_array(self.L.cassie_hostenv_drive_filter(self.env), (10, 9)).copy()
        jf = np.ctypeslib.as_array(self.L.cassie_hostenv_joint_filter(self.env), (6, 7)).copy()   # joint_filter_t = x[4], y[3]
        td = np.ctypeslib.as_array(self.L.cassie_hostenv_torque_delay(self.env), (10, 6)).copy()
        return dx.tobytes(), jf[:, :4].copy().tobytes(), jf[:, 4:].copy().tobytes(), td.tobytes()

    def close(self):
        if self.env:
            self.L.cassie_hostenv_free(self.env)
            self.env = None


def device_state_bytes(ds):
    """A cm_drive_state_t (ctypes) in the same four pieces."""
    return (np.ctypeslib.as_array(ds.drive_x).tobytes(), np.ctypeslib.as_array(ds.joint_x).tobytes(),
            np.ctypeslib.as_array(ds.joint_y).tobytes(), np.ctypeslib.as_array(ds.torque_delay).tobytes())


def pd_command(meas, ptarget, kp, kd, dtarget=None, torque=None):
    """pd_input's motor PD on the measured drive positions / velocities, in the operation order of the kernel."""
    p, v = meas[P.MEAS_DRIVE_POS: P.MEAS_DRIVE_POS + 10], meas[P.MEAS_DRIVE_VEL: P.MEAS_DRIVE_VEL + 10]
    dt = np.zeros(10) if dtarget is None else dtarget
    ff = np.zeros(10) if torque is None else torque
    return (ff + kp * (ptarget - p)) + kd * (dt - v)
