/*
 * cassie_batch.h -- batched form of the reference's step entry points.
 *
 * The reference steps one cassie_sim_t per call and users parallelise by running one
 * process per simulator.  This header is the batched counterpart of
 * cassie_sim_step_pd / cassie_sim_step / cassie_sim_step_ethercat (reference
 * src/cassiemujoco.c:1115-1157): N environments, physics on the MI355X through
 * include/cassie_phys.h, the host-side blocks (pd_input, cassie_core_sim, encoder + motor
 * models, state_output) on a pool of host threads, overlapped with the kernel.
 * Results per env are identical to N independent cassie_sim_t objects.
 */
#ifndef CASSIE_BATCH_H
#define CASSIE_BATCH_H

#include "cassie_io_types.h"
#include "cassie_phys.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- one environment's host-side state: everything of `struct cassie_sim` that is not mjModel/mjData
 *      (reference src/cassiemujoco.c:255-265): Agility block states, cassie_out, filters, torque delay ---- */
typedef struct cassie_hostenv cassie_hostenv_t;

/* constants the encoder / motor models read from the model (reference :558-664) */
typedef struct cassie_hostmodel {
    int drive_bits[10], joint_bits[6]; /* encoder resolutions (sensor user data) */
    double gear[10], tmax[10], wmax[10]; /* gear ratio, motor-side torque limit, no-load speed [rad/s] */
} cassie_hostmodel_t;

int cassie_hostmodel_from_model(const phys_model_t *m, cassie_hostmodel_t *out);
cassie_hostenv_t *cassie_hostenv_alloc(void);          /* cassie_out_init + block alloc + setup (reference :989-991, :1020, :1032-1034) */
void cassie_hostenv_free(cassie_hostenv_t *e);
void cassie_hostenv_copy(cassie_hostenv_t *dst, const cassie_hostenv_t *src);
cassie_out_t *cassie_hostenv_cassie_out(cassie_hostenv_t *e);
drive_filter_t *cassie_hostenv_drive_filter(cassie_hostenv_t *e);   /* [10] */
joint_filter_t *cassie_hostenv_joint_filter(cassie_hostenv_t *e);   /* [6] */
double *cassie_hostenv_torque_delay(cassie_hostenv_t *e);           /* [10][6] */
cassie_core_sim_t *cassie_hostenv_core(cassie_hostenv_t *e);
state_output_t *cassie_hostenv_estimator(cassie_hostenv_t *e);
pd_input_t *cassie_hostenv_pd(cassie_hostenv_t *e);
void cassie_hostenv_reset(cassie_hostenv_t *e);        /* host part of cassie_sim_full_reset (reference :2026-2032) */

/* cassie_motor_data + cassie_sensor_data + measurement copy-out: the host half of cassie_sim_step_ethercat
 * (reference :1115-1127).  sensordata / actuator_velocity are the physics outputs of the previous step; ctrl
 * receives the 10 delayed motor-side torques for the next physics step. */
void cassie_hostenv_ethercat(cassie_hostenv_t *e, const cassie_hostmodel_t *hm, const cassie_in_t *u,
                             const double *sensordata, const double *actuator_velocity, double *ctrl, cassie_out_t *y);
/* + cassie_core_sim_step in front (reference :1137-1145) */
void cassie_hostenv_step(cassie_hostenv_t *e, const cassie_hostmodel_t *hm, const cassie_user_in_t *u,
                         const double *sensordata, const double *actuator_velocity, double *ctrl, cassie_out_t *y);
/* + pd_input_step in front (reference :1147-1154); the estimator runs separately so it can overlap the kernel */
void cassie_hostenv_step_pd_pre(cassie_hostenv_t *e, const cassie_hostmodel_t *hm, const pd_in_t *u,
                                const double *sensordata, const double *actuator_velocity, double *ctrl, cassie_out_t *y);
void cassie_hostenv_step_pd_post(cassie_hostenv_t *e, const cassie_out_t *y, state_out_t *out); /* state_output_step, :1156 */

/* Host half of a step when the drive-level models run on the device (phys_batch_drive_pass): the
 * Agility blocks produce the command -- cmd[0..9] the cassie_in_t drive torques, cmd[10] the STO flag -- and the
 * device's measurement block (CM_MEAS_* layout) is copied into cassie_out afterwards. */
void cassie_hostenv_command_ethercat(cassie_hostenv_t *e, const cassie_in_t *u, double *cmd);
void cassie_hostenv_command(cassie_hostenv_t *e, const cassie_user_in_t *u, double *cmd);     /* cassie_core_sim_step first */
void cassie_hostenv_command_pd(cassie_hostenv_t *e, const pd_in_t *u, double *cmd);           /* pd_input_step first */
void cassie_hostenv_apply_meas(cassie_hostenv_t *e, const double *meas, cassie_out_t *y);
void cassie_hostenv_get_drive_state(const cassie_hostenv_t *e, cm_drive_state_t *out);
void cassie_hostenv_set_drive_state(cassie_hostenv_t *e, const cm_drive_state_t *in);

/* flat byte image of an env's host state (cassie_out, filters, delay lines, the three Agility block states): checkpoints */
size_t cassie_hostenv_image_size(void);
void cassie_hostenv_to_image(const cassie_hostenv_t *e, void *image);
void cassie_hostenv_from_image(cassie_hostenv_t *e, const void *image);
/* whether the closed library's block states have the sizes the flat image assumes (checked against the allocator when the
 * first env is created); the image functions must not be used otherwise */
bool cassie_hostenv_blocks_verified(void);

/* extension (not in the reference): capsule-vs-height-field contact samples the capsule's axis up to a grid cell apart for
 * capsules as long as Cassie's shin (eight interior samples instead of four); costs about 10 % on cassie_hfield.xml */
struct cassie_sim;
void cassie_sim_set_hfield_dense_sampling(struct cassie_sim *c, bool on);
/* extension: up to four contacts per capsule / height-field pair (its deepest sample spheres) instead of two -- towards
 * MuJoCo's one contact per penetrated prism; a Cassie standing on both feet then needs 44 constraint rows instead of 28 */
void cassie_sim_set_hfield_multi_contact(struct cassie_sim *c, bool on);
/* extension: the MuJoCo-shaped contact set -- ONE CONTACT PER PENETRATED GRID TRIANGLE under every sphere / capsule that has a
 * height-field pair (the deepest of the capsule's sample spheres over that triangle), the set reference model/cassie_hfield.xml:4
 * sizes nconmax = 300 for.  Up to 32 contacts / 127 constraint rows per env (a warning bit past that); substeps with more than
 * 31 / 63 rows are finished by the 63- / 127-row passes.  Off by default: about 3.3 x the cost on the bench terrain. */
void cassie_sim_set_hfield_prism_contacts(struct cassie_sim *c, bool on);

/* cores this process may really use: min(affinity mask, cgroup CPU quota) */
int cassie_host_cpu_count(void);

/* ---- N environments ---- */
typedef struct cassie_batch cassie_batch_t;

/* modelfile: MJCF (.xml) or .cmodel; nthreads <= 0 picks cassie_host_cpu_count() (capped at nenv).
 * Every env starts like cassie_sim_init leaves a simulator.  NULL + stderr message on failure. */
cassie_batch_t *cassie_batch_create(const char *modelfile, int nenv, int device, int nthreads);
void cassie_batch_free(cassie_batch_t *b);
int cassie_batch_nenv(const cassie_batch_t *b);
int cassie_batch_nthreads(const cassie_batch_t *b);
phys_batch_t *cassie_batch_phys(cassie_batch_t *b);   /* the HBM-resident state, for direct field access */
phys_model_t *cassie_batch_model(cassie_batch_t *b);
cassie_hostenv_t *cassie_batch_hostenv(cassie_batch_t *b, int env);

/* one cassie_sim_step_pd for every env: u and y are dense arrays of nenv structs */
int cassie_batch_step_pd(cassie_batch_t *b, const pd_in_t *u, state_out_t *y);
/* one cassie_sim_step / cassie_sim_step_ethercat for every env */
int cassie_batch_step(cassie_batch_t *b, const cassie_user_in_t *u, cassie_out_t *y);
int cassie_batch_step_ethercat(cassie_batch_t *b, const cassie_in_t *u, cassie_out_t *y);
/* on != 0: the encoder / motor models of every env run on the device (one small launch ahead of the physics kernel)
 * instead of on the host threads; only the 11 command doubles go up and the 56 measurement doubles come down per env
 * and step, and the state estimators still overlap the physics kernel.  Outputs are bit for bit those of the host
 * mode.  The filter histories / delay lines move to HBM (and back to the host envs when switched off).  Needs
 * timestep = 0.5 ms (one physics step per control step); returns -1 otherwise. */
int cassie_batch_set_device_drives(cassie_batch_t *b, int on);
/* cassie_sim_foot_forces for every env: cfrc is [nenv][12] (left force xyz at 0..2, right force xyz at 6..8, the
 * other entries zero like the single-env getter), from the per-body contact forces the last step left in HBM */
int cassie_batch_foot_forces(cassie_batch_t *b, double *cfrc);
/* The other reward-side getters for every env in one go (phys_batch_derive: one forward pass + a reduction kernel):
 * derived[nenv][CM_DRV_DIM] holds, per env, what cassie_sim_cm_position / cm_velocity / angular_momentum /
 * foot_positions / foot_velocities / foot_forces / heeltoe_forces / get_jacobian_full("left-foot" | "right-foot") return
 * (layout CM_DRV_* in cm_model.h), qM[nenv][nv * nv] what cassie_sim_full_mass_matrix returns.  Either pointer may be NULL
 * (the values then stay in HBM as PHYS_F_DERIVED / PHYS_F_QM of cassie_batch_phys(b), bindable to caller tensors). */
int cassie_batch_derive(cassie_batch_t *b, double *derived, double *qM);
/* cassie_sim_full_reset for the envs whose mask byte is non-zero (mask NULL = all) */
int cassie_batch_full_reset(cassie_batch_t *b, const unsigned char *mask);

#ifdef __cplusplus
}
#endif
#endif
