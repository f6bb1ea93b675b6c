"""ctypes access to the CPU oracle (oracle/libcassie_oracle.so) -- test infrastructure only."""
import ctypes
import os

import numpy as np

from cassie_amd import cstruct
from cassie_amd._lib import CmModel, MACROS, REPO_DIR

with open(os.path.join(REPO_DIR, "oracle", "cassie_oracle.h")) as _f:
    _types = cstruct.parse_structs(_f.read(), MACROS)
CoContact = _types["co_contact_t"]
CoData = _types["co_data_t"]

_lib = None


def lib():
    global _lib
    if _lib is None:
        # (a library variant built from an older source tree -- tools/build_variant.sh, CASSIE_LIB -- comes with the oracle
        # of that tree next to it: the two share cm_model_t's layout)
        variant = (os.environ.get("CASSIE_LIB") or "") + ".oracle.so"
        _lib = ctypes.CDLL(variant if os.path.exists(variant) else os.path.join(REPO_DIR, "oracle", "libcassie_oracle.so"))
        _lib.co_sizeof_data.restype = ctypes.c_ulong
        assert _lib.co_sizeof_data() == ctypes.sizeof(CoData), "co_data_t layout mismatch"
        for fn in ("co_reset", "co_forward", "co_step", "co_kinematics", "co_com_pos", "co_crb", "co_factor_m",
                   "co_collision", "co_make_constraint"):
            getattr(_lib, fn).argtypes = [ctypes.POINTER(CmModel), ctypes.POINTER(CoData)]
            getattr(_lib, fn).restype = None
        dp = ctypes.POINTER(ctypes.c_double)
        _lib.co_pd_ctrl.argtypes = [ctypes.POINTER(CmModel), ctypes.POINTER(CoData), dp, dp, dp]
        _lib.co_pd_ctrl.restype = None
        _lib.co_step_batch.argtypes = [ctypes.POINTER(CmModel), ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        _lib.co_step_batch.restype = None
        _lib.co_set_hfield.argtypes = [ctypes.c_void_p]
        _lib.co_set_hfield.restype = None
    return _lib


def arr(cfield, *shape):
    """numpy view (no copy) of a ctypes array member."""
    a = np.ctypeslib.as_array(cfield)
    return a.reshape(shape) if shape else a


_hfield_keepalive = None


def set_hfield(data):
    """Height-field samples (float32, nrow*ncol) used by every Oracle instance; None clears them."""
    global _hfield_keepalive
    _hfield_keepalive = None if data is None else np.ascontiguousarray(data, dtype=np.float32)
    lib().co_set_hfield(None if data is None else _hfield_keepalive.ctypes.data)


class Oracle:
    def __init__(self, pod, qpos=None):
        self.pod = pod
        self.d = CoData()
        lib().co_reset(ctypes.byref(pod), ctypes.byref(self.d))
        if qpos is not None:
            self.qpos[:] = qpos

    nq = property(lambda s: s.pod.nq)
    nv = property(lambda s: s.pod.nv)
    qpos = property(lambda s: arr(s.d.qpos)[: s.pod.nq])
    qvel = property(lambda s: arr(s.d.qvel)[: s.pod.nv])
    qacc = property(lambda s: arr(s.d.qacc)[: s.pod.nv])
    qacc_warmstart = property(lambda s: arr(s.d.qacc_warmstart)[: s.pod.nv])
    ctrl = property(lambda s: arr(s.d.ctrl)[: s.pod.nu])
    sensordata = property(lambda s: arr(s.d.sensordata)[: s.pod.nsensordata])
    actuator_velocity = property(lambda s: arr(s.d.actuator_velocity)[: s.pod.nu])
    qfrc_applied = property(lambda s: arr(s.d.qfrc_applied)[: s.pod.nv])
    xfrc_applied = property(lambda s: arr(s.d.xfrc_applied)[: s.pod.nbody])
    xpos = property(lambda s: arr(s.d.xpos)[: s.pod.nbody])
    qM = property(lambda s: arr(s.d.qM)[: s.pod.nv, : s.pod.nv])

    def pd_ctrl(self, ptarget, kp, kd):
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (ptarget, kp, kd)]
        dp = ctypes.POINTER(ctypes.c_double)
        lib().co_pd_ctrl(ctypes.byref(self.pod), ctypes.byref(self.d), *[x.ctypes.data_as(dp) for x in a])

    def forward(self):
        lib().co_forward(ctypes.byref(self.pod), ctypes.byref(self.d))

    def step(self, n=1):
        for _ in range(n):
            lib().co_step(ctypes.byref(self.pod), ctypes.byref(self.d))
