"""Locates and loads the product shared library and derives the ctypes layout of cm_model_t."""
import ctypes
import os

from . import cstruct

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # .../cassie-mujoco-sim_amd
REPO_DIR = os.path.dirname(PKG_DIR)
# CASSIE_LIB selects another build of the same library (tools/build_variant.sh: compiler-option experiments)
LIB_PATH = os.environ.get("CASSIE_LIB") or os.path.join(PKG_DIR, "lib", "libcassiemujoco.so")
MODEL_DIR = os.path.join(REPO_DIR, "models")

# (a variant built from an older source tree comes with its own header: <variant>.so.cm_model.h next to it)
_variant_hdr = LIB_PATH + ".cm_model.h"
with open(_variant_hdr if os.path.exists(_variant_hdr) else os.path.join(PKG_DIR, "csrc", "cm_model.h")) as _f:
    _hdr = _f.read()
MACROS = cstruct.parse_defines(_hdr)
_structs = cstruct.parse_structs(_hdr, MACROS)
CmModel = _structs["cm_model_t"]
CmDriveState = _structs["cm_drive_state_t"]
CmEnvParams = _structs["cm_envparams_t"]

_lib = None


def lib():
    """The product library.  Raises (loudly) if it has not been built: there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "%s is missing: build it with `make` (or __graft_entry__.build()); "
                "this package has no pure-Python or CPU fallback" % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        _declare(_lib)
        if _lib.phys_sizeof_model() != ctypes.sizeof(CmModel):
            raise RuntimeError("cm_model_t layout mismatch between cm_model.h and the built library")
        if hasattr(_lib, "phys_sizeof_envparams") and _lib.phys_sizeof_envparams() != ctypes.sizeof(CmEnvParams):
            raise RuntimeError("cm_envparams_t layout mismatch between cm_model.h and the built library")
    return _lib


def _declare(L):
    c = ctypes
    vp, ip, dp = c.c_void_p, c.POINTER(c.c_int), c.POINTER(c.c_double)
    L.phys_sizeof_model.restype = c.c_size_t
    L.phys_last_error.restype = c.c_char_p
    L.phys_model_load.restype = vp
    L.phys_model_load.argtypes = [c.c_char_p, c.c_char_p, c.c_int]
    L.phys_model_copy.restype = vp
    L.phys_model_copy.argtypes = [vp]
    L.phys_model_free.argtypes = [vp]
    L.phys_model_save.argtypes = [vp, c.c_char_p]
    L.phys_model_set_const.argtypes = [vp]
    if hasattr(L, "phys_model_set_flag"):
        L.phys_model_set_flag.argtypes = [vp, c.c_uint, c.c_int]
        L.phys_model_flags.argtypes = [vp]
        L.phys_model_flags.restype = c.c_uint
    L.phys_model_compile.argtypes = [vp, c.POINTER(CmModel), c.c_char_p, c.c_int]
    L.phys_model_name2id.argtypes = [vp, c.c_int, c.c_char_p]
    L.phys_model_id2name.restype = c.c_char_p
    L.phys_model_id2name.argtypes = [vp, c.c_int, c.c_int]
    L.phys_model_size.argtypes = [vp, c.c_int]
    L.phys_model_array.restype = dp
    L.phys_model_array.argtypes = [vp, c.c_int]
    L.phys_model_iarray.restype = ip
    L.phys_model_iarray.argtypes = [vp, c.c_int]
    L.phys_model_geom_rgba.restype = c.POINTER(c.c_float)
    L.phys_model_geom_rgba.argtypes = [vp]
    L.phys_model_hfield_data.restype = c.POINTER(c.c_float)
    L.phys_model_hfield_data.argtypes = [vp]
    L.phys_batch_create.restype = vp
    L.phys_batch_create.argtypes = [c.POINTER(CmModel), c.c_int, c.c_int]
    L.phys_batch_free.argtypes = [vp]
    L.phys_batch_nenv.argtypes = [vp]
    L.phys_batch_field_dim.argtypes = [vp, c.c_int]
    L.phys_batch_set_model.argtypes = [vp, c.POINTER(CmModel), c.c_int]
    L.phys_batch_set_hfield.argtypes = [vp, c.POINTER(c.c_float), c.c_int]
    if hasattr(L, "phys_batch_randomize"):   # (absent from older variant builds selected with CASSIE_LIB)
        L.phys_batch_param_dim.argtypes = [vp, c.c_int]
        L.phys_batch_randomize.argtypes = [vp, c.c_int, vp, c.c_int, c.c_int, c.c_int, vp]
        L.phys_batch_set_const.argtypes = [vp, c.c_int, c.c_int, vp]
        L.phys_batch_download_params.argtypes = [vp, vp, c.c_int, c.c_int]
        L.phys_batch_uses_env_params.argtypes = [vp]
        L.phys_sizeof_envparams.restype = c.c_size_t
    L.phys_batch_set_hfield_env.argtypes = [vp, c.c_int, c.POINTER(c.c_float), c.c_int]
    L.phys_batch_upload.argtypes = [vp, c.c_int, vp, c.c_int, c.c_int]
    L.phys_batch_download.argtypes = [vp, c.c_int, vp, c.c_int, c.c_int]
    L.phys_batch_download_warn.argtypes = [vp, vp, vp]
    L.phys_batch_device_ptr.restype = vp
    L.phys_batch_device_ptr.argtypes = [vp, c.c_int]
    L.phys_batch_bind.argtypes = [vp, c.c_int, vp]
    L.phys_batch_bind_strided.argtypes = [vp, c.c_int, vp, c.c_int]
    L.phys_batch_clear_warn.argtypes = [vp, c.c_int, c.c_int]
    L.phys_batch_step.argtypes = [vp, c.c_int, vp]
    L.phys_batch_forward.argtypes = [vp, vp]
    L.phys_batch_sync.argtypes = [vp]
    L.phys_batch_set_pd_mode.argtypes = [vp, c.c_int]
    L.phys_batch_set_drive_mode.argtypes = [vp, c.c_int]
    L.phys_batch_upload_drive_state.argtypes = [vp, vp, c.c_int, c.c_int]
    L.phys_batch_download_drive_state.argtypes = [vp, vp, c.c_int, c.c_int]
    L.phys_batch_uses_applied.argtypes = [vp]
    L.phys_batch_drive_pass.argtypes = [vp, c.c_int, vp]
    L.phys_batch_mark.argtypes = [vp]
    L.phys_batch_debug_poison_lds.argtypes = [vp]
    L.phys_batch_set_balance.argtypes = [vp, c.c_int]
    L.phys_batch_set_fast_rows.argtypes = [vp, c.c_int]
    L.phys_batch_set_waves_per_env.argtypes = [vp, c.c_int]
    L.phys_batch_set_inplace.argtypes = [vp, c.c_int]
    L.phys_batch_debug_inplace_ranges.argtypes = [vp]
    L.phys_batch_debug_form_launches.argtypes = [vp, c.POINTER(c.c_longlong), c.POINTER(c.c_longlong)]
    L.phys_batch_download_cost.argtypes = [vp, vp]
    L.phys_batch_measured_shader_clock.argtypes = [vp, vp]
    L.phys_batch_wide_pass_envs.argtypes = [vp, c.c_int]
    L.phys_batch_set_chunks.argtypes = [vp, c.c_int]
    L.phys_batch_debug_handover_pending.argtypes = [vp]
    L.phys_batch_debug_handover_pending.restype = c.c_int
    if hasattr(L, "phys_batch_kernel_timing"):
        L.phys_batch_enable_kernel_timing.argtypes = [vp, c.c_int]
        L.phys_batch_kernel_timing.argtypes = [vp, c.POINTER(c.c_int), c.POINTER(c.c_double)]
    if hasattr(L, "phys_batch_step_range"):
        L.phys_batch_step_range.argtypes = [vp, c.c_int, c.c_int, c.c_int, vp]
    if hasattr(L, "phys_batch_reset_envs"):
        L.phys_batch_reset_envs.argtypes = [vp, c.c_int, c.c_int, c.c_int, vp, vp, vp]
    if hasattr(L, "phys_batch_download_progress"):   # (absent from older variant builds selected with CASSIE_LIB)
        L.phys_batch_download_progress.argtypes = [vp, vp]
    L.phys_batch_set_all_outputs_every_substep.argtypes = [vp, c.c_int]
    L.phys_batch_derive.argtypes = [vp, c.POINTER(c.c_int), vp]
    L.phys_batch_clear_drive_state.argtypes = [vp, c.c_int, c.c_int, c.c_int, vp]
    L.phys_batch_wait_mark.argtypes = [vp]
    L.phys_batch_set_generic_kernel.argtypes = [vp, ctypes.c_int]
    L.phys_batch_profile_step.argtypes = [vp, vp]
    L.phys_batch_profile_substeps.argtypes = [vp, ctypes.c_int, vp]
    L.phys_batch_time_steps.argtypes = [vp, c.c_int, c.c_int, c.POINTER(c.c_float)]
