"""ctypes mirrors of the frozen I/O structs (include/cassie_io_types.h), derived from the header itself."""
import os

from . import cstruct
from ._lib import REPO_DIR

with open(os.path.join(REPO_DIR, "include", "cassie_io_types.h")) as _f:
    _t = cstruct.parse_structs(_f.read(), {})

cassie_out_t = _t["cassie_out_t"]
cassie_in_t = _t["cassie_in_t"]
cassie_user_in_t = _t["cassie_user_in_t"]
pd_in_t = _t["pd_in_t"]
state_out_t = _t["state_out_t"]
elmo_out_t = _t["elmo_out_t"]
cassie_joint_out_t = _t["cassie_joint_out_t"]
drive_filter_t = _t["drive_filter_t"]
joint_filter_t = _t["joint_filter_t"]
ALL = _t
