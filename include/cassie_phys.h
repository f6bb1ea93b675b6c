/*
 * cassie_phys.h -- the INNER drop-in boundary: the thin C ABI the host C glue
 * (cassiemujoco.c-equivalent) calls where the reference calls into MuJoCo.
 *
 * The reference has no clean ABI at this seam: it is a dlsym'd function-pointer
 * table plus direct mjModel/mjData field access (reference src/cassiemujoco.c:67-122,
 * field census in SURVEY.md 8b).  Each entry point below names the reference
 * call(s) it replaces.  Plain C types only: handles, pointers, sizes.
 *
 * All phys_batch_* entry points run on the MI355X through HIP; creating a batch
 * without a usable GPU fails loudly (NULL + message on stderr) -- there is no CPU
 * fallback in this library.
 */
#ifndef CASSIE_PHYS_H
#define CASSIE_PHYS_H

#include <stddef.h>
#include "cm_model.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ model --- */
typedef struct phys_model phys_model_t; /* host model: names, all geoms, hfield samples ... */

/* mj_loadXML (reference src/cassiemujoco.c:851, :930, :997).  Accepts the MJCF subset of
 * the in-scope models (.xml) or the neutral .cmodel text form; NULL + err text on failure. */
phys_model_t *phys_model_load(const char *path, char *err, int errlen);
/* mj_copyModel (reference :1013, :1016, :1093) */
phys_model_t *phys_model_copy(const phys_model_t *src);
/* mj_deleteModel (reference :883, :1110) */
void phys_model_free(phys_model_t *m);
int phys_model_save(const phys_model_t *m, const char *path);
/* mj_setConst (reference :952, :976): recompute invweight0 / meaninertia after inertial edits */
void phys_model_set_const(phys_model_t *m);
/* derive the pointer-free kernel model; returns 0 on success */
int phys_model_compile(const phys_model_t *m, cm_model_t *out, char *err, int errlen);
/* option flags of the model (CM_FLAG_* in csrc/cm_model.h: implicit joint damping, warm start, refsafe -- the mjOption
 * disable / enable bits the in-scope models use -- and the height-field contact options CM_FLAG_HFDENSE / HFMULTI / HFPRISM);
 * a change takes effect with the next compile / the next step of a cassie_sim_t */
unsigned phys_model_flags(const phys_model_t *m);
int phys_model_set_flag(phys_model_t *m, unsigned flag, int on);
/* a 64-bit fingerprint of every model array / option a caller can change through the views below: compared before a step
 * instead of recompiling (the reference hands out raw mjModel pointers, so writes cannot be observed otherwise) */
unsigned long long phys_model_fingerprint(const phys_model_t *m);
unsigned long long phys_hash_floats(const float *data, size_t n);
/* mj_name2id / mj_id2name (reference :861-866, :1244, ...); objtype uses mjtObj numbering */
int phys_model_name2id(const phys_model_t *m, int objtype, const char *name);
const char *phys_model_id2name(const phys_model_t *m, int objtype, int id);

/* sizes: the mjModel ints the reference reads (nq nv nu nbody njnt ngeom nsensordata ...) */
enum { PHYS_NQ, PHYS_NV, PHYS_NU, PHYS_NBODY, PHYS_NJNT, PHYS_NGEOM, PHYS_NSITE, PHYS_NSENSOR, PHYS_NSENSORDATA,
       PHYS_NEQ, PHYS_NHFIELDDATA, PHYS_HFIELD_NROW, PHYS_HFIELD_NCOL, PHYS_NUSER_SENSOR, PHYS_NUSER_ACTUATOR,
       PHYS_NCAM, PHYS_NUSER_GEOM };
int phys_model_size(const phys_model_t *m, int what);

/* read-write views of the mjModel arrays the reference exposes through accessors
 * (reference :1303-1584, :2050-2112).  Layouts follow MuJoCo. */
enum { PHYS_M_BODY_MASS, PHYS_M_BODY_IPOS, PHYS_M_BODY_POS, PHYS_M_BODY_QUAT, PHYS_M_DOF_DAMPING, PHYS_M_JNT_STIFFNESS,
       PHYS_M_QPOS_SPRING, PHYS_M_GEOM_POS, PHYS_M_GEOM_QUAT, PHYS_M_GEOM_SIZE, PHYS_M_GEOM_FRICTION,
       PHYS_M_ACTUATOR_GEAR, PHYS_M_ACTUATOR_CTRLRANGE, PHYS_M_ACTUATOR_USER, PHYS_M_SENSOR_USER, PHYS_M_HFIELD_SIZE,
       PHYS_M_TIMESTEP, PHYS_M_QPOS0, PHYS_M_JNT_RANGE, PHYS_M_STAT_CENTER, PHYS_M_STAT_EXTENT, PHYS_M_GEOM_USER,
       PHYS_M_BODY_INERTIA /* [nbody][3] principal moments (mjModel.body_inertia) */ };
double *phys_model_array(phys_model_t *m, int which);
float *phys_model_geom_rgba(phys_model_t *m);
float *phys_model_hfield_data(phys_model_t *m);
enum { PHYS_MI_JNT_TYPE, PHYS_MI_JNT_QPOSADR, PHYS_MI_JNT_DOFADR, PHYS_MI_GEOM_BODYID, PHYS_MI_GEOM_GROUP,
       PHYS_MI_SENSOR_OBJID, PHYS_MI_SENSOR_TYPE, PHYS_MI_SENSOR_ADR, PHYS_MI_SENSOR_DIM, PHYS_MI_BODY_PARENTID,
       PHYS_MI_GEOM_TYPE, PHYS_MI_BODY_JNTADR, PHYS_MI_BODY_JNTNUM, PHYS_MI_BODY_DOFADR, PHYS_MI_BODY_DOFNUM };
int *phys_model_iarray(phys_model_t *m, int which);

/* ------------------------------------------------------------ batched data --- */
typedef struct phys_batch phys_batch_t; /* N envs resident in HBM: the mjData role, batched */

/* per-env arrays (env-major, contiguous rows) */
enum { PHYS_F_QPOS, PHYS_F_QVEL, PHYS_F_QACC_WARMSTART, PHYS_F_TIME, PHYS_F_CTRL, PHYS_F_QFRC_APPLIED,
       PHYS_F_XFRC_APPLIED, PHYS_F_QACC, PHYS_F_SENSORDATA, PHYS_F_ACTUATOR_VELOCITY, PHYS_F_XPOS, PHYS_F_XQUAT,
       PHYS_F_PD_PTARGET, PHYS_F_PD_KP, PHYS_F_PD_KD, /* on-device joint PD, see phys_batch_set_pd_mode */
       PHYS_F_BODY_CFRC, /* [nbody][3]: net contact force on every body, world frame (reaction on geom2's body, minus it on
                            geom1's; the foot rows are cassie_sim_foot_forces), of the last substep of a launch */
       PHYS_F_DRIVE_CMD,  /* [nu + 1]: commanded drive torques (the cassie_in_t torques, output side) and the STO flag (non-zero =
                             safe torque off), read every substep in CM_DRIVE_TORQUE mode */
       PHYS_F_MEAS,       /* [CM_MEAS_DIM]: the measurement fields of cassie_out_t written by the device-side encoder / motor
                             models (layout: CM_MEAS_* in cm_model.h) */
       PHYS_F_PD_DTARGET, PHYS_F_PD_TORQUE, /* [nu] each: optional velocity targets / feed-forward torques of CM_DRIVE_PD */
       PHYS_F_DERIVED,    /* [CM_DRV_DIM]: the derived block of phys_batch_derive (layout: CM_DRV_* in cm_model.h) */
       PHYS_F_QM,         /* [nv * nv]: dense joint-space inertia matrix (mj_fullM role), written by phys_batch_derive */
       PHYS_F_COUNT };

/* mj_makeData (reference :441-447) for nenv environments on HIP device `device`;
 * every env starts at qpos0 (mj_resetData role).  NULL + stderr message on failure. */
phys_batch_t *phys_batch_create(const cm_model_t *model, int nenv, int device);
void phys_batch_free(phys_batch_t *b);                                  /* mj_deleteData (reference :452) */
int phys_batch_nenv(const phys_batch_t *b);
int phys_batch_field_dim(const phys_batch_t *b, int field);            /* doubles per env */
/* replace the model of one env (env >= 0) or of all envs (env = -1): per-env domain randomisation.  One launch serves every
 * env with the kernel instantiation picked from the shared model, so a per-env model may vary parameters but must keep the
 * shared model's sizes, dof tree, body-tree depth, joint make-up of the bodies (cm_model_t::kin_simple) and kinds of collision
 * pairs (-1 + phys_last_error() otherwise) */
int phys_batch_set_model(phys_batch_t *b, const cm_model_t *model, int env);
/* Per-env domain randomisation ON THE DEVICE (the batched form of the reference's per-simulator setters
 * cassie_sim_set_body_mass / set_body_ipos / set_dof_damping / set_geom_friction, reference src/cassiemujoco.c:1323-1436, and of
 * cassie_sim_just_set_const = mj_setConst, :974-977).  Every env gets a compact parameter block (cm_envparams_t in csrc/cm_model.h,
 * 10 KB) that the step kernel reads INSTEAD OF the shared model's fields; the rest of the model (95 KB) stays shared.
 *   phys_batch_randomize   writes rows [n][phys_batch_param_dim] of one parameter (CM_P_BODY_MASS [nbody], CM_P_BODY_IPOS
 *                          [nbody][3], CM_P_BODY_INERTIA [nbody][3] principal moments, CM_P_DOF_DAMPING [nv], CM_P_GEOM_FRICTION
 *                          [model->ngeom][3]: the COLLISION geoms in compiled order, cm_model_t::geom_fullid maps them to the
 *                          reference's full geom list) for envs [env0, env0 + n); values may be a DEVICE pointer (on_device != 0:
 *                          e.g. a torch tensor, nothing crosses PCIe) or host memory.  Damping and friction act from the next
 *                          step on, like in the reference; masses / inertial offsets / inertias act on the dynamics at once and
 *                          on the constraint regularisers after
 *   phys_batch_set_const   which recomputes, per env and on the device, what mj_setConst derives: body_invweight0,
 *                          dof_invweight0, meaninertia (M(qpos0) and its Cholesky factor per env) and the per-joint /
 *                          per-equality / per-pair values the constraint stages read -- bit for bit what compiling a host model
 *                          with the same parameters gives (phys_model_set_const + phys_model_compile).
 * Both are asynchronous on `stream` (NULL = the batch's own) and ordered with the stepping launches there.  Replacing the shared
 * model (phys_batch_set_model, env = -1) drops the blocks; per-env MODELS (env >= 0) and per-env parameter blocks do not mix. */
int phys_batch_param_dim(const phys_batch_t *b, int param);
int phys_batch_randomize(phys_batch_t *b, int param, const double *values, int on_device, int env0, int n, void *stream);
int phys_batch_set_const(phys_batch_t *b, int env0, int n, void *stream);
int phys_batch_download_params(phys_batch_t *b, cm_envparams_t *host, int env0, int n); /* (the model's own block where none exist yet) */
int phys_batch_uses_env_params(const phys_batch_t *b);
size_t phys_sizeof_envparams(void);
/* height-field samples (nrow * ncol floats, MuJoCo's normalised 0..1 elevations): one grid shared by all envs, or --
 * per-env terrain randomisation -- a grid of its own for one env (the others keep what they had) */
int phys_batch_set_hfield(phys_batch_t *b, const float *data, int n);
int phys_batch_set_hfield_env(phys_batch_t *b, int env, const float *data, int n);
/* host <-> HBM copies of whole fields or of a row range [env0, env0 + n) */
int phys_batch_upload(phys_batch_t *b, int field, const double *host, int env0, int n);
int phys_batch_download(phys_batch_t *b, int field, double *host, int env0, int n);
/* asynchronous variants on the batch's stream (no host synchronisation; pair with phys_batch_sync);
 * use phys_host_alloc'd (pinned) buffers for true overlap */
int phys_batch_upload_async(phys_batch_t *b, int field, const double *host, int env0, int n);
int phys_batch_download_async(phys_batch_t *b, int field, double *host, int env0, int n);
void *phys_host_alloc(size_t bytes);   /* pinned host memory (hipHostMalloc) */
void phys_host_free(void *p);
int phys_batch_download_warn(phys_batch_t *b, int *host_warn, int *host_info /* [nenv][4] or NULL */);
/* raw device pointer of a field (for torch / RCCL interop); bind replaces it with caller-owned HBM */
void *phys_batch_device_ptr(phys_batch_t *b, int field);
int phys_batch_bind(phys_batch_t *b, int field, void *device_ptr);
/* same with a row stride in doubles (>= the field's dim) for PHYS_F_QPOS / QVEL / SENSORDATA, so that the three can be
 * column blocks of ONE caller-owned [nenv][nq + nv + nsensordata] observation tensor -- the buffer an RCCL all-gather
 * sends as is; uploads / downloads of a strided field are 2-D copies */
int phys_batch_bind_strided(phys_batch_t *b, int field, void *device_ptr, int row_stride);
/* clears the sticky warning bits of envs [env0, env0 + n) (the batched cassie_sim_full_reset does this for the envs it
 * resets; mj_resetData clears mjData.warning the same way) */
int phys_batch_clear_warn(phys_batch_t *b, int env0, int n);
/* non-zero once PHYS_F_QFRC_APPLIED / PHYS_F_XFRC_APPLIED have been uploaded or bound (until then the kernel skips them) */
int phys_batch_uses_applied(const phys_batch_t *b);
/* mj_step1 + mj_step2, nsub times with ctrl held (reference :1130-1134), on `stream`
 * (a hipStream_t passed as void*, NULL = the batch's own stream); asynchronous.  The output fields (sensordata, qacc,
 * xpos / xquat, the measurement block, solver statistics) hold the values of the LAST of the nsub steps -- what nsub
 * single-step launches would leave; the state fields (qpos, qvel, time, warm start, drive-level state) advance nsub steps */
int phys_batch_step(phys_batch_t *b, int nsub, void *stream);
/* mj_forward (reference :971, :1029, :1223, :3293): no integration */
/* The same for the env range [env0, env0 + n) only.  Ranges of one batch may be in flight on different streams at once (the
 * per-env arrays are disjoint): stepping two half-batches on two streams, each at its own pace, lets one half's workgroups
 * fill the wave slots the other half leaves idle at the end and the start of its launches -- +16 % on BASELINE config 2
 *  when nothing joins the halves between policy steps. */
int phys_batch_step_range(phys_batch_t *b, int env0, int n, int nsub, void *stream);
int phys_batch_forward(phys_batch_t *b, void *stream);
/* Episode restarts without leaving the device -- the batched form of what a fresh cassie_sim_t / cassie_sim_full_reset
 * leaves (reference src/cassiemujoco.c:1023-1029, :2008-2034): envs first, first + stride, ... (count of them) get
 * qpos = qpos_row [nq] (DEVICE pointer), zero qvel / qacc_warmstart / ctrl / qacc / actuator_velocity / time and -- once a drive mode is in use -- a zero measurement block and zero drive-level state (encoder filter histories,
 * torque delay lines); sens_row [nsensordata] (device pointer or NULL) becomes their sensordata: the init pose's, which the
 * new episode's first drive-level pass reads.  The sticky warning word stays (phys_batch_clear_warn).  One small launch on
 * `stream`, ordered with the step launches there. */
int phys_batch_reset_envs(phys_batch_t *b, int first, int stride, int count, const double *qpos_row, const double *sens_row, void *stream);
/* the read-out half of mj_forward -- what the reference's getters obtain from mj_kinematics / mj_comPos / mj_comVel /
 * mj_fwdPosition (reference src/cassiemujoco.c:1223-1301, :1604-1770): xpos / xquat / the ext read-out / body_cfrc of the
 * current state, while the fields qacc, sensordata and actuator_velocity keep what the last STEP left (the encoder and
 * motor models of the next step read those) */
int phys_batch_forward_kinematics(phys_batch_t *b, void *stream);
int phys_batch_sync(phys_batch_t *b);
/* on != 0: every substep computes ctrl on the device from PHYS_F_PD_{PTARGET,KP,KD} -- the motor PD law of
 * pd_input_step (reference include/pd_input.h:34) followed by the speed-torque limit of motor()
 * (reference src/cassiemujoco.c:638-664), evaluated on the exact joint state; PHYS_F_CTRL is then ignored */
int phys_batch_set_pd_mode(phys_batch_t *b, int on);
/* Drive-level I/O on the device: with mode != CM_DRIVE_OFF every substep first runs
 * cassie_motor_data + cassie_sensor_data (reference src/cassiemujoco.c:737-803: encoder quantisation, integer FIR / IIR
 * velocity filters, motor speed-torque curve, STO, six-cycle torque delay) for every env, bit for bit the host chain,
 * on the sensordata / actuator_velocity the previous step left in HBM, writes PHYS_F_MEAS and takes ctrl from the
 * delay line.  CM_DRIVE_TORQUE reads the command from PHYS_F_DRIVE_CMD (cassie_sim_step_ethercat semantics);
 * CM_DRIVE_PD computes it as torque + kp (ptarget - position) + kd (dtarget - velocity) on the measured drive
 * position / velocity of the previous step (pd_input_step's motor PD, reference include/pd_input.h:34); CM_DRIVE_PD_SAFE
 * passes that command through cassie_core_sim's safety layer first (reference include/cassie_core_sim.h:34, called at
 * src/cassiemujoco.c:1141: joint-limit attenuation and restoring torques, torque-limit clamp, STO -- restated in
 * csrc/pk_safety.h bit for bit the closed binary), i.e. the whole torque path of cassie_sim_step_pd; the STO switch (radio
 * channel 8) is the last word of PHYS_F_DRIVE_CMD, the block's diagnostic messages collect in cm_drive_state_t::safety_msg.  The filter
 * histories and delay lines live in HBM (one cm_drive_state_t per env, zero-initialised like a fresh cassie_sim_t). */
int phys_batch_set_drive_mode(phys_batch_t *b, int mode);
/* the drive-level pass alone (no physics), whatever the drive mode of the step kernel is: reads PHYS_F_DRIVE_CMD (mode
 * CM_DRIVE_TORQUE) or the PD fields (CM_DRIVE_PD), PHYS_F_SENSORDATA and PHYS_F_ACTUATOR_VELOCITY; writes PHYS_F_CTRL,
 * PHYS_F_MEAS and the drive state.  Follow it with phys_batch_step in CM_DRIVE_OFF mode: together the two launches are
 * one cassie_sim_step_ethercat, and the measurements can travel to the host while the physics runs. */
int phys_batch_drive_pass(phys_batch_t *b, int mode, void *stream);
/* a marker on the batch's stream and a host wait for it (everything queued before the marker has completed) */
int phys_batch_mark(phys_batch_t *b);
int phys_batch_wait_mark(phys_batch_t *b);
/* zeroes the drive state (a fresh cassie_sim_t's filters and delay lines) of envs first, first + stride, ... (count of
 * them), asynchronously on `stream` (NULL = the batch's own): episode restarts of a device-resident rollout */
int phys_batch_clear_drive_state(phys_batch_t *b, int first, int stride, int count, void *stream);
int phys_batch_upload_drive_state(phys_batch_t *b, const cm_drive_state_t *host, int env0, int n);
int phys_batch_download_drive_state(phys_batch_t *b, cm_drive_state_t *host, int env0, int n);

/* times `reps` launches of nsub steps with HIP events on the launch stream; returns mean ms per launch */
int phys_batch_time_steps(phys_batch_t *b, int nsub, int reps, float *mean_ms);

/* extended per-env outputs (cm_ext_t: contacts + contact forces, body velocities, site frames, com): enable once,
 * then every step/forward refreshes them in HBM; download copies envs [env0, env0 + n) to the host */
int phys_batch_enable_ext(phys_batch_t *b, int on);
int phys_batch_download_ext(phys_batch_t *b, cm_ext_t *host, int env0, int n);
int phys_batch_download_ext_async(phys_batch_t *b, cm_ext_t *host, int env0, int n); /* pair with phys_batch_sync */

/* Batched derived getters: one forward pass (mj_forward role: nothing is integrated) that leaves the
 * full read-out of every env in HBM, then a small kernel that reduces it to PHYS_F_DERIVED -- whole-model centre of
 * mass / velocity / angular momentum, foot positions / velocities, foot and heel / toe contact forces, the feet's
 * Jacobians -- and PHYS_F_QM, i.e. what cassie_sim_cm_position / cm_velocity / angular_momentum / foot_positions /
 * foot_velocities / foot_forces / heeltoe_forces / get_jacobian_full / full_mass_matrix return for one simulator
 * (reference src/cassiemujoco.c:1254-1301, :1604-1712, :1812-1898), for every env, as device fields (bindable).
 * ids = {left foot body, right foot body, left heel site, right heel site, left toe site, right toe site}, -1 = absent.
 * The forces are those of the CURRENT state (like after cassie_sim_forward).  Costs about one physics step. */
int phys_batch_derive(phys_batch_t *b, const int ids[6], void *stream);

/* measurement aid: on = every substep of a fused launch evaluates every output (IMU sensors, body quaternions), although
 * only the last substep's values can be read; off (default) = those are formed by the substeps whose values are read */
int phys_batch_set_all_outputs_every_substep(phys_batch_t *b, int on);
/* Launch-order balancing (on by default for batches of 2048 envs and more): every launch records what each env cost
 * and the next launch starts the expensive envs first, so the wave slots finish together instead of the launch waiting
 * for whichever slot drew the slow envs last.  Results do not depend on it (envs are independent). */
int phys_batch_set_balance(phys_batch_t *b, int on);
/* diagnostics (batches of 2048 envs and more, balancing on): what the last stepping launch cost every env, in units of 64
 * shader clocks from the env's first to its last instruction ([nenv] unsigned) -- the figure the launch order is sorted by */
int phys_batch_download_cost(phys_batch_t *b, unsigned *host);
/* the shader clock the last stepping launch ran at, measured by the kernel itself: every env's span in shader clocks (s_memtime)
 * over the same span on the constant 100 MHz clock (s_memrealtime), summed over the envs.  -1 where the launch-cost arrays do not
 * exist (batches under 2048 envs, balancing off). */
int phys_batch_measured_shader_clock(phys_batch_t *b, double *hz);
/* diagnostics: envs that the last stepping launch over the env range starting at env0 passed on to the 127-row instantiation (the
 * count its pass reported: envs that met a substep with more than 63 constraint rows or 16 contacts) */
int phys_batch_wide_pass_envs(phys_batch_t *b, int env0);
/* validation aid: entries (and walker tickets) left in the hand-over lists of the batch's env ranges once its streams are idle
 * -- 0 whatever the mode: the pass behind the fast kernel clears what it walked, and a fast kernel whose pass does not walk the
 * list is not given one; -1 on error */
int phys_batch_debug_handover_pending(phys_batch_t *b);
/* on (default): stepping launches of the Cassie instantiations run the row-capped fast kernel first and the full kernel
 * only finishes envs that needed more than 31 constraint rows in some substep; off: the full kernel alone (same results,
 * bit for bit -- a validation / measurement aid) */
/* measurement aid: with timing enabled every stepping launch records a HIP event pair around the kernel that does its work
 * (the row-capped fast kernel where one exists, else the step kernel) on the launch's stream; phys_batch_kernel_timing waits
 * for the batch's streams and returns the number of launches since the last call and the sum of their kernel durations -- the
 * same quantity rocprofv3 --kernel-trace --stats averages, also when launches of several env ranges overlap */
int phys_batch_enable_kernel_timing(phys_batch_t *b, int on);
int phys_batch_kernel_timing(phys_batch_t *b, int *launches, double *total_ms);
int phys_batch_set_fast_rows(phys_batch_t *b, int on);
/* Which form of the two-wave fast kernel the stepping launches of an env range take (round 6): 0 = the fast kernel and, behind it, the
 * pass that walks the list of envs it handed over (an env's remaining substeps then run as one serial chain while the range's stream
 * waits); 1 = the fast kernel that finishes a substep it cannot hold IN PLACE -- the 63-row code inside the same workgroup -- and goes
 * on with the next one itself; 2 (default) = per range by what its recent launches needed: in place once a launch handed envs over,
 * back to the plain form after eight launches in a row (as the device reports them) in which no env left the fast tier.  Same results, bit for bit.  The in-place form costs
 * a workload that never leaves the fast tier 1.8 % and gains 8 - 24 % on one that does (profiles/round6/inplace_ab.txt). */
int phys_batch_set_inplace(phys_batch_t *b, int mode);
/* diagnostics: env ranges whose next stepping launch takes the in-place form */
int phys_batch_debug_inplace_ranges(const phys_batch_t *b);
/* ... and the stepping launches of the two-wave fast kernel so far, by form */
int phys_batch_debug_form_launches(const phys_batch_t *b, long long *plain, long long *inplace);
/* 2 (default): the row-capped fast kernels run in their two-wave form -- two wavefronts per env, the mass-matrix stage group
 * (centres of mass, composite inertias, M, its two factorisations) on the second wave beside the first wave's collision,
 * velocity and constraint-row stages; 1: one wavefront per env.  Same results, bit for bit (a measurement aid). */
int phys_batch_set_waves_per_env(phys_batch_t *b, int waves);
/* Stepping launches of the row-capped fast kernels in CHUNKS (1 = off .. 7; default: 7 for launches over the whole batch, 2 for
 * launches over an env range, whose neighbours' launches fill the end of its queue anyway -- 3 for a range's launch of at most 25
 * substeps, the shape of a consumer that fences every few substeps): a launch of at least 2048 envs and 10 substeps is
 * dispatched as `chunks` workgroups per env, each stepping a share of the substeps (5 at least) from the state the chunk before it
 * stored -- the jobs the GPU's workgroup slots queue up are that much shorter, and so is the time the slots stand idle at the
 * end of a launch while the last-started jobs finish.  Same results, bit for bit (a chunk ends and the next begins exactly
 * like two launches). */
int phys_batch_set_chunks(phys_batch_t *b, int chunks);
/* diagnostics: how many substeps of the last stepping launch the fast kernel completed for every env ([nenv] ints; less than
 * the launch's substep count = the env was handed over to the full kernel there).  Meaningful after a launch that ran the fast
 * kernel: not with the read-out enabled, fast rows off, or a batch of at most 512 envs stepping at most 4 substeps per launch
 * (those go through the full kernel alone: one launch instead of two) */
int phys_batch_download_progress(phys_batch_t *b, int *host);

/* validation aid: fills every CU's LDS with NaN bit patterns before the next launch (LDS is neither initialised nor
 * cleared between kernels) -- a step kernel that read LDS it had not written would then show it */
int phys_batch_debug_poison_lds(phys_batch_t *b);

/* validation aid: run the generic instantiation of the step kernel (dof-tree topology read from the model at run
 * time) even when the model matches one of the compile-time-topology instantiations */
int phys_batch_set_generic_kernel(phys_batch_t *b, int on);

/* per-stage shader-clock stamps of the next launches: [nenv][48] long long on the host after the call (profiling aid) */
int phys_batch_profile_step(phys_batch_t *b, long long *host_stamps);
/* same for one launch of nsub fused substeps: the stamps are those of the last substep */
int phys_batch_profile_substeps(phys_batch_t *b, int nsub, long long *host_stamps);

size_t phys_sizeof_model(void);
const char *phys_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
