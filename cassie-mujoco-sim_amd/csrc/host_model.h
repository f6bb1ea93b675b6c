/*
 * host_model.h -- host-side (C++) Cassie model: everything the drop-in C ABI
 * has to expose about a model (names, all 50 geoms incl. visual ones, rgba,
 * user data, heightfield samples ...) with MuJoCo's array layouts, plus the
 * compile step that derives the pointer-free cm_model_t the kernels consume.
 *
 * Replaces, for the in-scope MJCF subset, what the
 * reference obtains from mj_loadXML / mj_copyModel / mj_setConst
 * (reference src/cassiemujoco.c:851, :1013-1016, :949-977).
 */
#ifndef CM_HOST_MODEL_H
#define CM_HOST_MODEL_H

#include <string>
#include <vector>
#include "cm_model.h"

namespace cm {

/* object kinds for name lookups (numeric values follow mjtObj where the
 * reference passes them through cassie_sim_mj_name2id) */
enum ObjType { OBJ_BODY = 1, OBJ_JOINT = 3, OBJ_GEOM = 5, OBJ_SITE = 6, OBJ_CAMERA = 7,
               OBJ_HFIELD = 11, OBJ_EQUALITY = 16, OBJ_ACTUATOR = 18, OBJ_SENSOR = 19 }; /* MuJoCo 2.1.0 mjtObj values */

struct HostModel {
    /* sizes */
    int nq = 0, nv = 0, nu = 0, nbody = 0, njnt = 0, ngeom = 0, nsite = 0, ncam = 0;
    int nsensor = 0, nsensordata = 0, neq = 0, nhfield = 0, nhfielddata = 0;
    int nuser_sensor = 0, nuser_actuator = 0, nuser_geom = 0;

    /* options */
    double timestep = 0.002, tolerance = 1e-8, impratio = 1.0;
    double gravity[3] = {0, 0, -9.81}, magnetic[3] = {0, -0.5, 0};
    int iterations = 100;
    int solver_pgs = 0;
    unsigned flags = CM_FLAG_EULERDAMP | CM_FLAG_WARMSTART | CM_FLAG_REFSAFE;
    double meaninertia = 1.0;
    double stat_center[3] = {0, 0, 0}, stat_extent = 2.0;
    float vis_znear = 0.01f, vis_zfar = 50.f;

    /* names */
    std::vector<std::string> body_name, jnt_name, geom_name, site_name, act_name, sensor_name,
        eq_name, hfield_name, cam_name;

    /* bodies */
    std::vector<int> body_parentid, body_jntadr, body_jntnum, body_dofadr, body_dofnum, body_geomadr,
        body_geomnum;
    std::vector<double> body_pos, body_quat, body_ipos, body_iquat, body_mass, body_inertia,
        body_invweight0, body_subtreemass;

    /* joints / dofs */
    std::vector<int> jnt_type, jnt_qposadr, jnt_dofadr, jnt_bodyid, jnt_limited;
    std::vector<double> jnt_pos, jnt_axis, jnt_range, jnt_stiffness, jnt_margin, jnt_solref, jnt_solimp;
    std::vector<double> qpos0, qpos_spring;
    std::vector<int> dof_bodyid, dof_jntid, dof_parentid;
    std::vector<double> dof_armature, dof_damping, dof_invweight0;

    /* geoms (ALL of them, collision or not) */
    std::vector<int> geom_type, geom_bodyid, geom_contype, geom_conaffinity, geom_condim, geom_priority,
        geom_group, geom_dataid;
    std::vector<double> geom_pos, geom_quat, geom_size, geom_friction, geom_solref, geom_solimp,
        geom_solmix, geom_margin, geom_gap, geom_rbound, geom_user;
    std::vector<float> geom_rgba;

    /* sites / cameras */
    std::vector<int> site_bodyid;
    std::vector<double> site_pos, site_quat;

    /* equality (connect) */
    std::vector<int> eq_body1, eq_body2, eq_active;
    std::vector<double> eq_data, eq_solref, eq_solimp;

    /* actuators */
    std::vector<int> act_jntid, act_ctrllimited;
    std::vector<double> act_gear /* 6 per actuator */, act_ctrlrange, act_user;

    /* sensors */
    std::vector<int> sensor_type, sensor_objid, sensor_adr, sensor_dim;
    std::vector<double> sensor_cutoff, sensor_noise, sensor_user;

    /* heightfield (at most one in the in-scope models) */
    int hfield_nrow = 0, hfield_ncol = 0;
    double hfield_size[4] = {0, 0, 0, 0};
    std::vector<float> hfield_data;

    /* --- operations --- */
    int name2id(int objtype, const char *name) const;
    const char *id2name(int objtype, int id) const;

    /* recompute the constants that depend on inertial parameters at qpos0
     * (body/dof invweight0, subtree masses, meaninertia): the mj_setConst role */
    void set_const();
    /* fill the kernel-facing POD; returns false (and sets err) if a limit is exceeded */
    bool compile(cm_model_t *out, std::string *err) const;

    /* neutral text serialisation (models/ *.cmodel), so that the GPU box -- which
     * has no /root/reference -- can load the in-scope models */
    bool save(const std::string &path) const;
    bool load(const std::string &path, std::string *err);
};

/* MJCF subset loader. Returns false and fills err on failure. */
bool load_mjcf(const std::string &path, HostModel *out, std::string *err);
/* picks the loader by file extension (.xml -> MJCF, else .cmodel) */
bool load_model_file(const std::string &path, HostModel *out, std::string *err);

}  // namespace cm
#endif
