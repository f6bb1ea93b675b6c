"""Python face of the inner C ABI (include/cassie_phys.h): Model and Batch.

``Batch`` is the batched, HBM-resident counterpart of the reference's one
``cassie_sim_t`` per process (reference example/cassiemujoco.py:31-72): N
environments stepped by one HIP launch.  All heavy lifting is in the shared
library; this module only moves numpy arrays across the boundary.
"""
import ctypes
import os

import numpy as np

from ._lib import CmDriveState, CmEnvParams, CmModel, MODEL_DIR, lib

# field ids (enum in cassie_phys.h)
(F_QPOS, F_QVEL, F_QACC_WARMSTART, F_TIME, F_CTRL, F_QFRC_APPLIED, F_XFRC_APPLIED, F_QACC, F_SENSORDATA,
 F_ACTUATOR_VELOCITY, F_XPOS, F_XQUAT, F_PD_PTARGET, F_PD_KP, F_PD_KD, F_BODY_CFRC, F_DRIVE_CMD, F_MEAS, F_PD_DTARGET,
 F_PD_TORQUE, F_DERIVED, F_QM) = range(22)

# layout of the derived block F_DERIVED (CM_DRV_* in cm_model.h); MAXV = CM_MAXV
MAXV = 40
DRV_COM_POS, DRV_COM_VEL, DRV_ANGMOM, DRV_FOOT_POS, DRV_FOOT_VEL, DRV_FOOT_FORCE, DRV_TOE_FORCE, DRV_HEEL_FORCE, DRV_MASS = 0, 3, 6, 9, 15, 27, 39, 45, 51
DRV_FOOT_JACP, DRV_FOOT_JACR, DRV_DIM = 52, 52 + 6 * MAXV, 52 + 12 * MAXV

# drive modes (CM_DRIVE_* in cm_model.h) and the layout of the measurement block F_MEAS (CM_MEAS_*)
DRIVE_OFF, DRIVE_TORQUE, DRIVE_PD, DRIVE_PD_SAFE = 0, 1, 2, 3     # (PD_SAFE: + cassie_core_sim's safety layer, csrc/pk_safety.h)
SAFETY_MSG_LIMIT, SAFETY_MSG_TORQUE = 1, 2    # bits of cm_drive_state_t::safety_msg: diagnostic codes 635 / 630
MEAS_DRIVE_POS, MEAS_DRIVE_VEL, MEAS_DRIVE_TORQUE, MEAS_JOINT_POS, MEAS_JOINT_VEL = 0, 10, 20, 30, 36
MEAS_ORIENTATION, MEAS_ANGVEL, MEAS_LINACC, MEAS_MAG, MEAS_DIM = 42, 46, 49, 52, 56

FLAG_EULERDAMP, FLAG_WARMSTART, FLAG_REFSAFE, FLAG_HFDENSE, FLAG_HFMULTI, FLAG_HFPRISM, FLAG_BOX8 = 1, 2, 4, 8, 16, 32, 64      # CM_FLAG_* (cm_model.h)
WARN_CONTACT_FULL, WARN_CONSTRAINT_FULL, WARN_UNSUPPORTED_PAIR, WARN_DIVERGED = 1, 2, 4, 8
WARN_CHUNK_PLACEMENT = 16   # a chunk of a stepping launch found its predecessor on another XCD: the env's state may be stale, discard its results

# per-env physical parameters (CM_P_* in cm_model.h): what Batch.randomize takes
P_BODY_MASS, P_BODY_IPOS, P_BODY_INERTIA, P_DOF_DAMPING, P_GEOM_FRICTION = range(5)
# views of the host model's arrays (PHYS_M_* in cassie_phys.h) that Model.array hands out
(M_BODY_MASS, M_BODY_IPOS, M_BODY_POS, M_BODY_QUAT, M_DOF_DAMPING, M_JNT_STIFFNESS, M_QPOS_SPRING, M_GEOM_POS, M_GEOM_QUAT,
 M_GEOM_SIZE, M_GEOM_FRICTION, M_ACTUATOR_GEAR, M_ACTUATOR_CTRLRANGE, M_ACTUATOR_USER, M_SENSOR_USER, M_HFIELD_SIZE, M_TIMESTEP,
 M_QPOS0, M_JNT_RANGE, M_STAT_CENTER, M_STAT_EXTENT, M_GEOM_USER, M_BODY_INERTIA) = range(23)
SIZE_NGEOM = 5               # PHYS_NGEOM: geoms in the host model's full list

# joint configuration the reference writes at init (reference src/cassiemujoco.c:1023-1028)
QPOS_INIT_JOINTS = np.array(
    [0.0045, 0, 0.4973, 0.9785, -0.0164, 0.01787, -0.2049, -1.1997, 0, 1.4267, 0, -1.5244, 1.5244, -1.5968,
     -0.0045, 0, 0.4973, 0.9786, 0.00386, -0.01524, -0.2051, -1.1997, 0, 1.4267, 0, -1.5244, 1.5244, -1.5968])


def model_path(name):
    """Resolves 'cassie' / 'cassie_hfield' / 'cassie_tray_box' (or a path) to a loadable model file."""
    if os.path.exists(name):
        return name
    p = os.path.join(MODEL_DIR, name + ".cmodel")
    if os.path.exists(p):
        return p
    raise FileNotFoundError("no model file for %r (looked in %s)" % (name, MODEL_DIR))


class Model:
    """Host model (names, all geoms, ...) plus its compiled pointer-free form ``.pod`` (cm_model_t)."""

    def __init__(self, path_or_name="cassie"):
        L = lib()
        err = ctypes.create_string_buffer(1024)
        self.name = os.path.splitext(os.path.basename(str(path_or_name)))[0]   # "cassie", "cassie_hfield", ...
        self._h = L.phys_model_load(model_path(path_or_name).encode(), err, len(err))
        if not self._h:
            raise RuntimeError("model load failed: " + err.value.decode())
        self.pod = CmModel()
        self.compile()

    def compile(self):
        err = ctypes.create_string_buffer(1024)
        if lib().phys_model_compile(self._h, ctypes.byref(self.pod), err, len(err)) != 0:
            raise RuntimeError("model compile failed: " + err.value.decode())
        return self.pod

    def set_const(self):
        lib().phys_model_set_const(self._h)
        self.compile()

    def set_flag(self, flag, on=True):
        """Option flag of the model (CM_FLAG_*: FLAG_HFDENSE = denser capsule sampling against height fields); recompiles."""
        if lib().phys_model_set_flag(self._h, int(flag), 1 if on else 0) != 0:
            raise RuntimeError("bad model flag")
        self.compile()

    def save(self, path):
        if lib().phys_model_save(self._h, path.encode()) != 0:
            raise RuntimeError("cannot write " + path)

    def array(self, which, n):
        """Read-write numpy view of `n` doubles of a host-model array (M_*: the mjModel arrays the reference's setters write);
        follow edits of masses / inertial frames with set_const(), of anything with compile()."""
        p = lib().phys_model_array(self._h, int(which))
        if not p:
            raise ValueError("no such model array")
        return np.ctypeslib.as_array(p, shape=(int(n),))

    def name2id(self, objtype, name):
        return lib().phys_model_name2id(self._h, objtype, name.encode())

    def size(self, what):
        return lib().phys_model_size(self._h, what)

    def qpos_init(self):
        """The state cassie_sim_init leaves the robot in (qpos0 with the nominal joint pose)."""
        q = np.array(self.pod.qpos0[: self.pod.nq])
        q[7:35] = QPOS_INIT_JOINTS
        return q

    def __del__(self):
        try:
            if self._h:
                lib().phys_model_free(self._h)
                self._h = None
        except Exception:
            pass


class Batch:
    """N environments resident in HBM on one MI355X."""

    def __init__(self, model, nenv, device=0):
        self.model = model
        self.nenv = int(nenv)
        pod = model.pod if isinstance(model, Model) else model
        self.pod = pod
        self._h = lib().phys_batch_create(ctypes.byref(pod), self.nenv, device)
        if not self._h:
            raise RuntimeError("phys_batch_create failed: " + (lib().phys_last_error() or b"").decode())

    def dim(self, field):
        return lib().phys_batch_field_dim(self._h, field)

    def set(self, field, arr, env0=0):
        a = np.ascontiguousarray(arr, dtype=np.float64).reshape(-1, self.dim(field))
        if lib().phys_batch_upload(self._h, field, a.ctypes.data, env0, a.shape[0]) != 0:
            raise RuntimeError("upload failed: " + (lib().phys_last_error() or b"").decode())

    def get(self, field, env0=0, n=None):
        n = self.nenv - env0 if n is None else n
        out = np.empty((n, self.dim(field)), dtype=np.float64)
        if lib().phys_batch_download(self._h, field, out.ctypes.data, env0, n) != 0:
            raise RuntimeError("download failed: " + (lib().phys_last_error() or b"").decode())
        return out

    def warnings(self):
        w = np.zeros(self.nenv, dtype=np.int32)
        info = np.zeros((self.nenv, 4), dtype=np.int32)
        if lib().phys_batch_download_warn(self._h, w.ctypes.data, info.ctypes.data) != 0:
            raise RuntimeError("download failed")
        return w, info

    def device_ptr(self, field):
        return lib().phys_batch_device_ptr(self._h, field)

    def bind(self, field, device_ptr, row_stride=None):
        """Aliases a field to caller-owned HBM; `row_stride` (doubles, qpos / qvel / sensordata only) lets the field be a
        column block of a wider tensor, e.g. one [nenv][nq + nv + nsensordata] observation block."""
        rc = (lib().phys_batch_bind(self._h, field, device_ptr) if row_stride is None
              else lib().phys_batch_bind_strided(self._h, field, device_ptr, int(row_stride)))
        if rc != 0:
            raise RuntimeError("bind failed: " + (lib().phys_last_error() or b"").decode())

    def clear_warnings(self, env0=0, n=None):
        if lib().phys_batch_clear_warn(self._h, env0, self.nenv - env0 if n is None else n) != 0:
            raise RuntimeError("clear_warn failed")

    def param_dim(self, param):
        return lib().phys_batch_param_dim(self._h, int(param))

    def randomize(self, param, values, env0=0, device_ptr=None, n=None, stream=None):
        """Per-env physical parameters (P_BODY_MASS [nbody], P_BODY_IPOS [nbody*3], P_BODY_INERTIA [nbody*3], P_DOF_DAMPING [nv],
        P_GEOM_FRICTION [pod.ngeom*3], collision geoms in compiled order) for envs env0 ...: `values` is a host array
        [n][dim], or pass `device_ptr` (+ n) to read rows that are already in HBM (a torch tensor's data_ptr()).  Masses /
        inertial offsets / inertias: follow with set_const()."""
        if device_ptr is not None:
            rc = lib().phys_batch_randomize(self._h, int(param), device_ptr, 1, int(env0), int(n), stream)
        else:
            a = np.ascontiguousarray(values, dtype=np.float64).reshape(-1, self.param_dim(param))
            rc = lib().phys_batch_randomize(self._h, int(param), a.ctypes.data, 0, int(env0), a.shape[0], stream)
        if rc != 0:
            raise RuntimeError("randomize failed: " + (lib().phys_last_error() or b"").decode())

    def set_const(self, env0=0, n=None, stream=None):
        """mj_setConst per env on the device: inverse weights and mean inertia from every env's own masses / inertial frames."""
        if lib().phys_batch_set_const(self._h, int(env0), self.nenv - env0 if n is None else int(n), stream) != 0:
            raise RuntimeError("set_const failed: " + (lib().phys_last_error() or b"").decode())

    def params(self, env0=0, n=None):
        """The envs' parameter blocks (ctypes array of CmEnvParams), downloaded."""
        n = self.nenv - env0 if n is None else n
        out = (CmEnvParams * n)()
        if lib().phys_batch_download_params(self._h, ctypes.byref(out), int(env0), int(n)) != 0:
            raise RuntimeError("parameter download failed")
        return out

    def set_model(self, pod, env=-1):
        if lib().phys_batch_set_model(self._h, ctypes.byref(pod), env) != 0:
            raise RuntimeError("set_model failed: " + (lib().phys_last_error() or b"").decode())

    def step(self, nsub=1, stream=None):
        if lib().phys_batch_step(self._h, nsub, stream) != 0:
            raise RuntimeError("step failed: " + (lib().phys_last_error() or b"").decode())

    def step_range(self, env0, n, nsub=1, stream=None):
        """Steps the env range [env0, env0 + n) only; ranges may be in flight on different streams at once."""
        if lib().phys_batch_step_range(self._h, int(env0), int(n), nsub, stream) != 0:
            raise RuntimeError("step_range failed: " + (lib().phys_last_error() or b"").decode())

    def forward(self, stream=None):
        if lib().phys_batch_forward(self._h, stream) != 0:
            raise RuntimeError("forward failed: " + (lib().phys_last_error() or b"").decode())

    def set_hfield(self, data, env=None):
        """Height-field samples for all envs (env=None, one shared grid) or for one env only (per-env terrain)."""
        a = np.ascontiguousarray(data, dtype=np.float32)
        ptr = a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
        rc = lib().phys_batch_set_hfield(self._h, ptr, a.size) if env is None else lib().phys_batch_set_hfield_env(self._h, int(env), ptr, a.size)
        if rc != 0:
            raise RuntimeError("set_hfield failed")

    def set_pd_mode(self, on=True):
        lib().phys_batch_set_pd_mode(self._h, 1 if on else 0)

    def set_drive_mode(self, mode):
        """DRIVE_OFF / DRIVE_TORQUE (cassie_sim_step_ethercat on the device) / DRIVE_PD (pd_input's motor PD on the encoder
        measurements): the encoder + motor models of reference src/cassiemujoco.c:558-664 run in the step kernel."""
        if lib().phys_batch_set_drive_mode(self._h, int(mode)) != 0:
            raise RuntimeError("set_drive_mode failed: " + (lib().phys_last_error() or b"").decode())

    def derive(self, ids, stream=None):
        """Batched derived getters: one forward pass + a reduction kernel fill F_DERIVED and F_QM for every env.
        ids = (left foot body, right foot body, left heel site, right heel site, left toe site, right toe site), -1 = absent."""
        arr = (ctypes.c_int * 6)(*[int(i) for i in ids])
        if lib().phys_batch_derive(self._h, arr, stream) != 0:
            raise RuntimeError("derive failed: " + (lib().phys_last_error() or b"").decode())

    def drive_pass(self, mode=DRIVE_TORQUE, stream=None):
        """The drive-level models alone (no physics): reads the command / PD fields and the last step's sensordata and
        actuator_velocity, writes F_CTRL, F_MEAS and the drive state."""
        if lib().phys_batch_drive_pass(self._h, int(mode), stream) != 0:
            raise RuntimeError("drive_pass failed: " + (lib().phys_last_error() or b"").decode())

    def clear_drive_state(self, first=0, stride=1, count=None, stream=None):
        count = (self.nenv - first + stride - 1) // stride if count is None else count
        if lib().phys_batch_clear_drive_state(self._h, first, stride, count, stream) != 0:
            raise RuntimeError("clear_drive_state failed")

    def reset_envs(self, first, stride, count, qpos_row_ptr, sens_row_ptr=None, stream=None):
        """Episode restart of envs first, first + stride, ... on the device in one launch (phys_batch_reset_envs): qpos from
        the device row `qpos_row_ptr`, velocities / warm start / ctrl / time zero, drive-level state and measurement block
        zero, sensordata from `sens_row_ptr` if given."""
        if lib().phys_batch_reset_envs(self._h, int(first), int(stride), int(count), qpos_row_ptr, sens_row_ptr, stream) != 0:
            raise RuntimeError("reset_envs failed: " + (lib().phys_last_error() or b"").decode())

    def get_drive_state(self, env0=0, n=None):
        n = self.nenv - env0 if n is None else n
        out = (CmDriveState * n)()
        if lib().phys_batch_download_drive_state(self._h, ctypes.byref(out), env0, n) != 0:
            raise RuntimeError("drive state download failed")
        return out

    def set_drive_state(self, states, env0=0):
        if lib().phys_batch_upload_drive_state(self._h, ctypes.byref(states), env0, len(states)) != 0:
            raise RuntimeError("drive state upload failed")

    def sync(self):
        if lib().phys_batch_sync(self._h) != 0:
            raise RuntimeError("sync failed: " + (lib().phys_last_error() or b"").decode())

    def time_steps(self, nsub, reps):
        """Mean milliseconds per launch of `nsub` steps, measured with HIP events on the launch stream."""
        ms = ctypes.c_float(0)
        if lib().phys_batch_time_steps(self._h, nsub, reps, ctypes.byref(ms)) != 0:
            raise RuntimeError("timing failed: " + (lib().phys_last_error() or b"").decode())
        return ms.value

    def set_balance(self, on=True):
        """Longest-job-first launch order from the previous launch's per-env cost (default on for >= 2048 envs)."""
        lib().phys_batch_set_balance(self._h, 1 if on else 0)

    def enable_kernel_timing(self, on=True):
        lib().phys_batch_enable_kernel_timing(self._h, 1 if on else 0)

    def kernel_timing(self):
        """(launches, total ms) of the work-doing kernel of every stepping launch since the last call (HIP event pairs)."""
        n, ms = ctypes.c_int(0), ctypes.c_double(0.0)
        if lib().phys_batch_kernel_timing(self._h, ctypes.byref(n), ctypes.byref(ms)) != 0:
            raise RuntimeError("kernel timing failed")
        return n.value, ms.value

    def set_fast_rows(self, on=True):
        """Row-capped fast kernel ahead of the full one (default on; results are bit for bit the same either way)."""
        lib().phys_batch_set_fast_rows(self._h, 1 if on else 0)

    def set_inplace(self, mode=2):
        """Form of the two-wave fast kernel: 0 = kernel + list-walking pass, 1 = finishes the substeps it cannot hold in place,
        2 = per env range by what its recent launches needed (default).  Same results bit for bit."""
        if lib().phys_batch_set_inplace(self._h, int(mode)) != 0:
            raise ValueError("in-place mode: 0, 1 or 2")

    def form_launches(self):
        """Diagnostics: (plain, in place) -- stepping launches of the two-wave fast kernel so far, by form."""
        import ctypes
        a, c_ = ctypes.c_longlong(0), ctypes.c_longlong(0)
        lib().phys_batch_debug_form_launches(self._h, ctypes.byref(a), ctypes.byref(c_))
        return int(a.value), int(c_.value)

    def inplace_ranges(self):
        """Diagnostics: env ranges whose next stepping launch takes the in-place form of the fast kernel."""
        return lib().phys_batch_debug_inplace_ranges(self._h)

    def set_waves_per_env(self, waves=2):
        """Two-wave form of the fast kernels (default 2; results are bit for bit the same either way)."""
        if lib().phys_batch_set_waves_per_env(self._h, int(waves)) != 0:
            raise ValueError("waves per env: 1 or 2")

    def set_chunks(self, chunks=4):
        """Stepping launches of the fast kernels as `chunks` workgroups per env (1 = off); results are bit for bit the same."""
        if lib().phys_batch_set_chunks(self._h, int(chunks)) != 0:
            raise ValueError("chunks per env-launch: 1 .. 7")

    def launch_cost(self):
        """Shader clocks every env's last stepping launch took, first to last instruction (diagnostics; batches >= 2048 envs)."""
        out = np.zeros(self.nenv, dtype=np.uint32)
        if lib().phys_batch_download_cost(self._h, out.ctypes.data) != 0:
            raise RuntimeError("cost download failed (balancing is off or the batch is small)")
        return out.astype(np.float64) * 64.0

    def measured_shader_clock(self):
        """Hz the last stepping launch ran at: the envs' spans in shader clocks over the same spans on the 100 MHz clock
        (None where the launch-cost arrays do not exist: small batches, balancing off)."""
        import ctypes
        hz = ctypes.c_double(0.0)
        if lib().phys_batch_measured_shader_clock(self._h, ctypes.byref(hz)) != 0:
            return None
        return float(hz.value)

    def wide_pass_envs(self, env0=0):
        """Envs that the last stepping launch over the env range starting at env0 passed on to the 127-row instantiation (they met a
        substep with more than 63 constraint rows or 16 contacts); waits for the batch's streams."""
        r = lib().phys_batch_wide_pass_envs(self._h, int(env0))
        if r < 0:
            raise RuntimeError("phys_batch_wide_pass_envs failed")
        return r

    def handover_pending(self):
        """Validation aid: entries left in the hand-over lists once the batch's streams are idle (0 in every mode)."""
        r = lib().phys_batch_debug_handover_pending(self._h)
        if r < 0:
            raise RuntimeError("hand-over count download failed")
        return r

    def fast_rows_progress(self):
        """Substeps of the last stepping launch the fast kernel completed per env (< the launch's count: handed over there)."""
        out = np.zeros(self.nenv, dtype=np.int32)
        if lib().phys_batch_download_progress(self._h, out.ctypes.data) != 0:
            raise RuntimeError("progress download failed")
        return out

    def set_all_outputs_every_substep(self, on=True):
        """Measurement aid: every substep of a fused launch evaluates every output (IMU sensors, body quaternions), not only
        the substeps whose values can be read."""
        lib().phys_batch_set_all_outputs_every_substep(self._h, 1 if on else 0)

    def poison_lds(self):
        """Validation aid: NaN bit patterns into every CU's LDS before the next launch."""
        if lib().phys_batch_debug_poison_lds(self._h) != 0:
            raise RuntimeError("poison failed")

    def set_generic_kernel(self, on=True):
        """Validation aid: use the run-time-topology instantiation of the step kernel."""
        lib().phys_batch_set_generic_kernel(self._h, 1 if on else 0)

    def profile_step(self, nsub=1):
        """Runs one launch of `nsub` fused substeps and returns the per-env shader-clock stamps [nenv][48] taken at the
        stage boundaries of the last substep."""
        st = np.zeros((self.nenv, 48), dtype=np.int64)
        if lib().phys_batch_profile_substeps(self._h, int(nsub), st.ctypes.data) != 0:
            raise RuntimeError("profile_step failed")
        return st

    def close(self):
        if self._h:
            lib().phys_batch_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
