/*
 * cassiemujoco.h -- the OUTER drop-in boundary: the C ABI of libcassiemujoco.so.
 *
 * Same entry points, argument meaning and error behaviour as the reference's
 * include/cassiemujoco.h (every `extern "C"` function of reference
 * src/cassiemujoco.c, 162 of which the unmodified ctypes wrapper
 * example/cassiemujoco_ctypes.py binds eagerly at import, plus the 25 Agility
 * block / pack symbols and 7 UDP helpers: SURVEY.md App. E).  Each group below
 * cites the reference lines it replaces.  The physics behind it is the batched
 * HIP kernel reached through include/cassie_phys.h; a cassie_sim_t is a batch of
 * one environment on the GPU with host mirrors of the arrays the reference
 * exposes as read-write pointers.
 *
 * Failure behaviour follows the reference: init returns false / NULL with a message
 * on stderr, stepping cannot fail, cassie_sim_free(NULL) is a no-op.  Name lookups
 * that miss return a pointer to a zeroed scratch buffer instead of the reference's
 * out-of-bounds pointer (src/cassiemujoco.c:1244-1245).
 */
#ifndef CASSIEMUJOCO_H
#define CASSIEMUJOCO_H

#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>

#include "cassie_io_types.h"

typedef struct cassie_sim cassie_sim_t;
typedef struct cassie_vis cassie_vis_t;
typedef struct cassie_state cassie_state_t;


#ifdef __cplusplus
extern "C" {
#endif

/* ---- library lifecycle (reference src/cassiemujoco.c:820-947) ---- */
bool cassie_mujoco_init(const char *modelfile);   /* loads the global model; false + stderr on failure */
void cassie_cleanup(void);
bool cassie_reload_xml(const char *modelfile);
void delete_init_model(void);

/* ---- simulator instances (reference :979-1113) ---- */
cassie_sim_t *cassie_sim_init(const char *modelfile, bool reinit);
cassie_sim_t *cassie_sim_duplicate(const cassie_sim_t *sim);
void cassie_sim_copy(cassie_sim_t *dst, const cassie_sim_t *src);
void cassie_sim_copy_just_sim(cassie_sim_t *dst, const cassie_sim_t *src);
void cassie_sim_free(cassie_sim_t *sim);

/* ---- stepping: the hot path (reference :1115-1189) ---- */
void cassie_sim_step_ethercat(cassie_sim_t *sim, cassie_out_t *y, const cassie_in_t *u);
void cassie_sim_step(cassie_sim_t *sim, cassie_out_t *y, const cassie_user_in_t *u);
void cassie_sim_step_pd(cassie_sim_t *sim, state_out_t *y, const pd_in_t *u);
void cassie_sim_step_pd_no2khz(cassie_sim_t *sim, state_out_t *y, const pd_in_t *u);
void cassie_integrate_pos(cassie_sim_t *sim, state_out_t *y);
int cassie_sim_forward(cassie_sim_t *sim);

/* ---- sizes and raw state pointers (reference :1038-1070, :1191-1252, :1566-1584) ---- */
int cassie_sim_nv(const cassie_sim_t *sim);
int cassie_sim_nq(const cassie_sim_t *sim);
int cassie_sim_nu(const cassie_sim_t *sim);
int cassie_sim_nbody(const cassie_sim_t *sim);
int cassie_sim_njnt(const cassie_sim_t *sim);
int cassie_sim_ngeom(const cassie_sim_t *sim);
void cassie_sim_params(cassie_sim_t *sim, int *params);
int *cassie_sim_jnt_qposadr(cassie_sim_t *sim);
int *cassie_sim_jnt_dofadr(cassie_sim_t *sim);
int cassie_sim_mj_name2id(cassie_sim_t *sim, char *mj_type, char *name);
void *cassie_sim_mjmodel(cassie_sim_t *sim);
void *cassie_sim_mjdata(cassie_sim_t *sim);
double *cassie_sim_time(cassie_sim_t *sim);
double *cassie_sim_timestep(cassie_sim_t *sim);
void cassie_sim_set_timestep(cassie_sim_t *sim, double dt);
double *cassie_sim_qpos(cassie_sim_t *sim);
double *cassie_sim_qvel(cassie_sim_t *sim);
double *cassie_sim_qacc(cassie_sim_t *sim);
double *cassie_sim_accel(cassie_sim_t *sim);
double *cassie_sim_qfrc(cassie_sim_t *sim);
double *cassie_sim_ctrl(cassie_sim_t *sim);
void cassie_sim_setctrl(cassie_sim_t *sim, double *ctrl);
double *cassie_sim_act_vel(cassie_sim_t *sim);
double *cassie_sim_sensordata(cassie_sim_t *sim);
double *cassie_sim_xpos(cassie_sim_t *sim, const char *name);
double *cassie_sim_xquat(cassie_sim_t *sim, const char *name);
double *cassie_sim_site_xpos(cassie_sim_t *sim, const char *name);
void cassie_sim_site_xquat(cassie_sim_t *sim, const char *name, double *xquat);
void cassie_sim_read_rangefinder(cassie_sim_t *sim, double ranges[6]);

/* ---- model parameters (reference :1303-1564, :2050-2080) ---- */
double *cassie_sim_dof_damping(cassie_sim_t *sim);
void cassie_sim_set_dof_damping(cassie_sim_t *sim, double *damp);
void cassie_sim_set_dof_name_damping(cassie_sim_t *sim, const char *name, double *damp);
double *cassie_sim_get_dof_name_damping(cassie_sim_t *sim, const char *name);
int cassie_sim_get_joint_num_dof(cassie_sim_t *sim, const char *name);
double *cassie_sim_body_mass(cassie_sim_t *sim);
void cassie_sim_set_body_mass(cassie_sim_t *sim, double *mass);
void cassie_sim_set_body_name_mass(cassie_sim_t *sim, const char *name, double mass);
double cassie_sim_get_body_name_mass(cassie_sim_t *sim, const char *name);
double *cassie_sim_body_ipos(cassie_sim_t *sim);
void cassie_sim_set_body_ipos(cassie_sim_t *sim, double *ipos);
void cassie_sim_set_body_name_ipos(cassie_sim_t *sim, const char *name, double *ipos);
double *cassie_sim_get_body_name_ipos(cassie_sim_t *sim, const char *name);
void cassie_sim_set_body_name_pos(cassie_sim_t *sim, const char *name, double *data);
double *cassie_sim_get_body_name_pos(cassie_sim_t *sim, const char *name);
double *cassie_sim_geom_friction(cassie_sim_t *sim);
void cassie_sim_set_geom_friction(cassie_sim_t *sim, double *fric);
void cassie_sim_set_geom_name_friction(cassie_sim_t *sim, const char *name, double *fric);
double *cassie_sim_get_geom_name_friction(cassie_sim_t *sim, const char *name);
float *cassie_sim_geom_rgba(cassie_sim_t *sim);
float *cassie_sim_geom_name_rgba(cassie_sim_t *sim, const char *name);
void cassie_sim_set_geom_rgba(cassie_sim_t *sim, float *rgba);
void cassie_sim_set_geom_name_rgba(cassie_sim_t *sim, const char *name, float *rgba);
double *cassie_sim_geom_quat(cassie_sim_t *sim);
double *cassie_sim_geom_name_quat(cassie_sim_t *sim, const char *name);
void cassie_sim_set_geom_quat(cassie_sim_t *sim, double *quat);
void cassie_sim_set_geom_name_quat(cassie_sim_t *sim, const char *name, double *quat);
double *cassie_sim_geom_pos(cassie_sim_t *sim);
double *cassie_sim_geom_name_pos(cassie_sim_t *sim, const char *name);
void cassie_sim_set_geom_pos(cassie_sim_t *sim, double *pos);
void cassie_sim_set_geom_name_pos(cassie_sim_t *sim, const char *name, double *pos);
double *cassie_sim_geom_size(cassie_sim_t *sim);
double *cassie_sim_geom_name_size(cassie_sim_t *sim, const char *name);
void cassie_sim_set_geom_size(cassie_sim_t *sim, double *size);
void cassie_sim_set_geom_name_size(cassie_sim_t *sim, const char *name, double *size);
int cassie_sim_get_hfield_nrow(cassie_sim_t *sim);
int cassie_sim_get_hfield_ncol(cassie_sim_t *sim);
int cassie_sim_get_nhfielddata(cassie_sim_t *sim);
double *cassie_sim_get_hfield_size(cassie_sim_t *sim);
void cassie_sim_set_hfield_size(cassie_sim_t *sim, double size[4]);
float *cassie_sim_hfielddata(cassie_sim_t *sim);
void cassie_sim_set_hfielddata(cassie_sim_t *sim, float *data);
void cassie_sim_set_const(cassie_sim_t *sim);        /* reference :949-977 */
void cassie_sim_just_set_const(cassie_sim_t *sim);

/* ---- derived quantities (reference :1254-1301, :1586-1961) ---- */
void cassie_sim_get_jacobian(cassie_sim_t *sim, double *jac, const char *name);
void cassie_sim_get_jacobian_full(cassie_sim_t *sim, double *jac, double *jac_rot, const char *name);
void cassie_sim_get_jacobian_full_site(cassie_sim_t *sim, double *jac, double *jac_rot, const char *name);
bool cassie_sim_check_obstacle_collision(const cassie_sim_t *sim);
bool cassie_sim_check_self_collision(const cassie_sim_t *sim);
bool cassie_sim_geom_collision(const cassie_sim_t *sim, int geom_group);
void cassie_sim_foot_forces(const cassie_sim_t *sim, double cfrc[12]);
void cassie_sim_heeltoe_forces(const cassie_sim_t *sim, double toe_force[6], double heel_force[6]);
void cassie_sim_foot_positions(const cassie_sim_t *sim, double cpos[6]);
void cassie_sim_foot_velocities(const cassie_sim_t *sim, double cvel[12]);
void cassie_sim_foot_orient(const cassie_sim_t *sim, double corient[4]);
void cassie_sim_cm_position(const cassie_sim_t *sim, double cm_pos[3]);
void cassie_sim_cm_velocity(const cassie_sim_t *sim, double cm_vel[3]);
void cassie_sim_centroid_inertia(const cassie_sim_t *sim, double Icm[9]);
void cassie_sim_angular_momentum(const cassie_sim_t *sim, double Lcm[3]);
void cassie_sim_full_mass_matrix(const cassie_sim_t *sim, double M[1024]);
void cassie_sim_minimal_mass_matrix(const cassie_sim_t *sim, double M[256]);
void cassie_sim_loop_constraint_info(const cassie_sim_t *sim, double J_cl[192], double err_cl[6]);
void cassie_sim_body_velocities(const cassie_sim_t *sim, double cvel[6], const char *name);
void cassie_sim_body_acceleration(const cassie_sim_t *sim, double accel[6], const char *name);
/* cfrc[0:3] = summed contact force on the body, world axes; cfrc[3:6] = 0 for the condim-1/3 contacts of the supported
 * models -- the slots the reference's call of mju_transformSpatial on a [force; torque] vector produces (:1781-1810) */
void cassie_sim_body_contact_force(const cassie_sim_t *sim, double cfrc[6], const char *name);
void cassie_sim_relative_pose(double pos1[3], double quat1[4], double pos2[3], double quat2[4],
                              double pos2_in_pos1[3], double quat2_in_quat1[4]);

/* ---- perturbation, radio, reset, filters (reference :1963-2047, :2082-2240) ---- */
void cassie_sim_apply_force(cassie_sim_t *sim, double xfrc[6], const char *name);
void cassie_sim_clear_forces(cassie_sim_t *sim);
void cassie_sim_hold(cassie_sim_t *sim);
void cassie_sim_release(cassie_sim_t *sim);
void cassie_sim_radio(cassie_sim_t *sim, double channels[16]);
void cassie_sim_full_reset(cassie_sim_t *sim);
void reset_state_est(cassie_sim_t *sim, state_out_t *y);
cassie_out_t cassie_sim_get_cassie_out(cassie_sim_t *sim);
void cassie_sim_copy_cassie_out(cassie_sim_t *dst, cassie_out_t *y);
void cassie_sim_copy_mjd(cassie_sim_t *dst, cassie_sim_t *src);
void cassie_sim_copy_state_est(cassie_sim_t *dst, cassie_sim_t *src);
void cassie_sim_run_state_est(cassie_sim_t *sim, cassie_out_t *cassie_out, state_out_t *y);
void state_out_free(state_out_t *out);
joint_filter_t *cassie_sim_joint_filter(cassie_sim_t *sim);
void cassie_sim_get_joint_filter(cassie_sim_t *sim, double *x, double *y);
void cassie_sim_set_joint_filter(cassie_sim_t *sim, double *x, double *y);
drive_filter_t *cassie_sim_drive_filter(cassie_sim_t *sim);
void cassie_sim_get_drive_filter(cassie_sim_t *sim, int *x);
void cassie_sim_set_drive_filter(cassie_sim_t *sim, int *x);
void cassie_sim_torque_delay(cassie_sim_t *sim, double *t);
void cassie_sim_set_torque_delay(cassie_sim_t *sim, double *t);

/* ---- state snapshots (reference :3380-3452) ---- */
/* checkpoint files (an extension): a cassie_state_t -- physics state, encoder filters, delay lines and
 * the Agility block states -- written to / read from disk; 0 on success, -1 on I/O error or foreign / stale file */
int cassie_state_save(const cassie_state_t *state, const char *path);
int cassie_state_load(cassie_state_t *state, const char *path);
cassie_state_t *cassie_state_alloc(void);
cassie_state_t *cassie_state_duplicate(const cassie_state_t *src);
void cassie_state_copy(cassie_state_t *dst, const cassie_state_t *src);
void cassie_state_free(cassie_state_t *state);
double *cassie_state_time(cassie_state_t *state);
double *cassie_state_qpos(cassie_state_t *state);
double *cassie_state_qvel(cassie_state_t *state);
void cassie_get_state(const cassie_sim_t *sim, cassie_state_t *state);
void cassie_set_state(cassie_sim_t *sim, const cassie_state_t *state);

/* ---- visualisation: rendering is out of scope; these behave like the reference built
 *      without GLFW (cassie_vis_init returns NULL, everything else tolerates NULL) ---- */
cassie_vis_t *cassie_vis_init(cassie_sim_t *sim, const char *modelfile, bool offscreen);
void cassie_vis_close(cassie_vis_t *vis);
void cassie_vis_free(cassie_vis_t *vis);
bool cassie_vis_draw(cassie_vis_t *vis, cassie_sim_t *sim);
bool cassie_vis_valid(cassie_vis_t *vis);
bool cassie_vis_paused(cassie_vis_t *vis);
bool cassie_vis_slowmo(cassie_vis_t *vis);
void cassie_vis_window_resize(cassie_vis_t *vis, int width, int height);
void cassie_vis_add_marker(cassie_vis_t *v, double pos[3], double size[3], double rgba[4], double so3[9]);
void cassie_vis_remove_marker(cassie_vis_t *v, int id);
void cassie_vis_clear_markers(cassie_vis_t *v);
void cassie_vis_update_marker_pos(cassie_vis_t *v, int id, double pos[3]);
void cassie_vis_update_marker_size(cassie_vis_t *v, int id, double size[3]);
void cassie_vis_update_marker_rgba(cassie_vis_t *v, int id, double rgba[4]);
void cassie_vis_update_marker_orient(cassie_vis_t *v, int id, double so3[9]);
void cassie_vis_apply_force(cassie_vis_t *vis, double xfrc[6], const char *name);
void cassie_vis_full_reset(cassie_vis_t *vis);
void cassie_vis_remakeSceneCon(cassie_vis_t *v);
void cassie_vis_set_hfielddata(cassie_vis_t *v, float *data);
float *cassie_vis_hfielddata(cassie_vis_t *v);
void cassie_vis_set_cam(cassie_vis_t *v, const char *body_name, double zoom, double azi, double elev);
void cassie_vis_set_cam_pos(cassie_vis_t *v, double *look_point, double distance, double azi, double elev);
void cassie_vis_attach_cam(cassie_vis_t *v, const char *cam_name);
float cassie_vis_extent(cassie_vis_t *v);
float cassie_vis_znear(cassie_vis_t *v);
float cassie_vis_zfar(cassie_vis_t *v);
void cassie_vis_init_recording(cassie_vis_t *v, const char *videofile, int width, int height);
void cassie_vis_record_frame(cassie_vis_t *v);
void cassie_vis_close_recording(cassie_vis_t *v);
void cassie_vis_init_depth(cassie_vis_t *v, int width, int height);
void cassie_vis_init_rgb(cassie_vis_t *v, int width, int height);
float *cassie_vis_draw_depth(cassie_vis_t *v, cassie_sim_t *c, int width, int height);
unsigned char *cassie_vis_get_rgb(cassie_vis_t *v, cassie_sim_t *c, int width, int height);
int cassie_vis_get_depth_size(cassie_vis_t *v);
void cassie_vis_foot_forces(const cassie_vis_t *v, double cfrc[12]);

#ifdef __cplusplus
}
#endif
#endif /* CASSIEMUJOCO_H */
