/*
 * cassiemujoco.c -- host C glue of the drop-in libcassiemujoco.so.
 *
 * Host side of the hot path, in C as in the reference: lifecycle / reset / IO
 * packing, the encoder + motor + delay models and the step wiring around the
 * Agility blocks (pd_input -> cassie_core_sim -> ethercat-level sim ->
 * state_output).  Where the reference calls MuJoCo through dlsym'd pointers
 * (src/cassiemujoco.c:67-122) this file calls the thin HIP C ABI of
 * include/cassie_phys.h; a cassie_sim_t owns a one-environment batch on the GPU
 * plus host mirrors of every array the reference hands out as a read-write
 * pointer.  Function-level citations give the reference lines each piece restates.
 */
#define _GNU_SOURCE
#include "cassiemujoco.h"

#include <math.h>
#include <string.h>

#include "cassie_batch.h"
#include "cassie_phys.h"

#define NUM_DRIVES 10
#define NUM_JOINTS 6
#define TORQUE_DELAY_CYCLES 6
#define DRIVE_FILTER_NB 9
#define JOINT_FILTER_NB 4
#define JOINT_FILTER_NA 3

/* mjtObj values (MuJoCo 2.1.0) used for name lookups */
enum { OBJ_BODY = 1, OBJ_XBODY = 2, OBJ_JOINT = 3, OBJ_GEOM = 5, OBJ_SITE = 6, OBJ_CAMERA = 7, OBJ_HFIELD = 11,
       OBJ_EQUALITY = 16, OBJ_ACTUATOR = 18, OBJ_SENSOR = 19 };

/* nominal pose written at init / reset (reference :1023-1028, :2010-2014) */
static const double qpos_nominal[35] = {
    0, 0, 1.01, 1, 0, 0, 0,
    0.0045, 0, 0.4973, 0.9785, -0.0164, 0.01787, -0.2049, -1.1997, 0, 1.4267, 0, -1.5244, 1.5244, -1.5968,
    -0.0045, 0, 0.4973, 0.9786, 0.00386, -0.01524, -0.2051, -1.1997, 0, 1.4267, 0, -1.5244, 1.5244, -1.5968};

/* host mirror of the per-env physics state: the mjData role */
typedef struct {
    double time;
    double qpos[CM_MAXQ], qvel[CM_MAXV], qacc[CM_MAXV], qacc_warmstart[CM_MAXV], ctrl[CM_MAXU];
    double qfrc_applied[CM_MAXV], xfrc_applied[CM_MAXBODY * 6];
    double sensordata[CM_MAXSENSORDATA + 8]; /* +8: the reference reads 6 rangefinder slots past the end (:776-778) */
    double actuator_velocity[CM_MAXU];
    double xpos[CM_MAXBODY * 3], xquat[CM_MAXBODY * 4];
    double site_xpos[CM_MAXSITE * 3];
    int warmstart_dirty;
} sim_data_t;

struct cassie_sim {
    phys_model_t *m;
    cm_model_t pod;           /* compiled form of m as last uploaded */
    phys_batch_t *b;          /* one environment in HBM */
    sim_data_t d;
    cm_ext_t ext;             /* derived-quantity read-out of the last step / forward */
    cassie_hostenv_t *host;   /* Agility block states, cassie_out, encoder filters, torque delay line */
    cassie_hostmodel_t hm;
    int left_foot_body, right_foot_body, left_heel, right_heel, left_toe, right_toe;
    unsigned long long hfield_hash; /* of the samples last uploaded (callers write through cassie_sim_hfielddata) */
    unsigned long long model_print; /* phys_model_fingerprint of m when pod was compiled from it */
    bool applied_in_use;      /* qfrc_applied / xfrc_applied have been non-zero at some point */
    bool ext_stale;           /* the last launch left a newer read-out in HBM than `ext` holds */
};

struct cassie_state {
    sim_data_t d;
    cassie_hostenv_t *host;
};

struct cassie_vis { int unused; };

/* ------------------------------------------------------------ global state --- */
static phys_model_t *initial_model = NULL; /* the reference's process-global model (:50) */
static bool library_initialized = false;
static double zero_scratch[64];            /* returned for failed name lookups */
static float zero_scratch_f[8];

static int nsize(const cassie_sim_t *c, int what) { return phys_model_size(c->m, what); }

/* A cassie_sim_t lives in pinned, device-mapped host memory, and the batch-of-one's state fields are BOUND to its
 * mirrors (sim_attach_physics): the step kernel loads qpos / qvel / ctrl / warm start straight from `d` over PCIe and
 * stores the new state, qacc, sensordata, actuator_velocity, xpos and xquat straight back -- a step is one kernel launch
 * and one stream synchronisation, with no copy engine in between (round 2 queued ~15 small copies per step).  The 20 KB
 * derived read-out (`ext`) stays in HBM and is fetched when a getter asks for it. */
static cassie_sim_t *sim_alloc(void)
{
    cassie_sim_t *c = phys_host_alloc(sizeof(cassie_sim_t));
    if (!c) fprintf(stderr, "cassiemujoco: cannot allocate the simulator (no usable HIP device?) -- this library has no CPU fallback\n");
    return c;
}

static bool load_global_model(const char *path)
{
    char err[1000] = "Could not load XML model";
    phys_model_t *m = phys_model_load(path, err, sizeof err);
    if (!m) {
        fprintf(stderr, "Load model error: %s\n", err);
        return false;
    }
    /* the reference overwrites sensor_objid[0..19] with fixed ids (:856-859); for the in-scope
     * models those are the natural ids, so only the first-20-sensors rule is kept when it applies */
    int *objid = phys_model_iarray(m, PHYS_MI_SENSOR_OBJID);
    static const int fixed[20] = {0, 1, 2, 3, 4, 9, 10, 14, 5, 6, 7, 8, 9, 20, 21, 25, 0, 0, 0, 0};
    int ns = phys_model_size(m, PHYS_NSENSOR);
    for (int i = 0; i < 20 && i < ns; ++i) objid[i] = fixed[i];
    if (phys_model_name2id(m, OBJ_BODY, "left-foot") < 0 || phys_model_name2id(m, OBJ_BODY, "right-foot") < 0) {
        fprintf(stderr, "Could not find body named left-foot / right-foot\n");
        phys_model_free(m);
        return false;
    }
    /* heel / toe sites exist only in cassie.xml; the reference refuses the other shipped models
     * because of that (:861-866).  Here they are optional: heel/toe force splitting is simply
     * unavailable without them (documented deviation). */
    if (initial_model) phys_model_free(initial_model);
    initial_model = m;
    return true;
}

bool cassie_mujoco_init(const char *modelfile)
{
    if (!library_initialized) {
        if (!load_global_model(modelfile)) return false;
        library_initialized = true;
    }
    return library_initialized;
}

void delete_init_model(void)
{
    if (initial_model) phys_model_free(initial_model);
    initial_model = NULL;
}

void cassie_cleanup(void)
{
    delete_init_model();
    library_initialized = false;
}

bool cassie_reload_xml(const char *modelfile) { return load_global_model(modelfile); }

/* ------------------------------------------------------ host <-> HBM sync --- */
/* The caller may have edited model arrays through the accessors or through the raw pointers they return (reference
 * :1303-1584): a fingerprint of everything reachable that way is compared before every launch, and only a change costs a
 * model compile + upload (round 2 recompiled and memcmp'ed the whole model every step). */
static void sim_recompile(cassie_sim_t *c)
{
    const unsigned long long print = phys_model_fingerprint(c->m);
    if (print == c->model_print) return;
    /* (the fingerprint is recorded only once the edit has reached the device: a failed compile or upload is retried -- and
     * reported -- by every later step instead of leaving the stale device model in use silently) */
    cm_model_t pod;
    char err[256];
    if (phys_model_compile(c->m, &pod, err, sizeof err) != 0) {
        fprintf(stderr, "cassiemujoco: model compile failed (the device keeps the last good model; retried at the next step): %s\n", err);
        return;
    }
    if (memcmp(&pod, &c->pod, sizeof pod) != 0) {
        if (phys_batch_set_model(c->b, &pod, -1) != 0) {
            fprintf(stderr, "cassiemujoco: model upload failed (retried at the next step): %s\n", phys_last_error());
            return;
        }
        c->pod = pod;
        const float *hf = phys_model_hfield_data(c->m);
        if (hf && phys_batch_set_hfield(c->b, hf, phys_model_size(c->m, PHYS_NHFIELDDATA)) != 0) {
            fprintf(stderr, "cassiemujoco: terrain upload failed (retried at the next step): %s\n", phys_last_error());
            return;
        }
    }
    c->model_print = print;
}

static void sim_push_hfield(cassie_sim_t *c)
{
    const float *hf = phys_model_hfield_data(c->m);
    if (!hf) return;
    int n = phys_model_size(c->m, PHYS_NHFIELDDATA);
    const unsigned long long h = phys_hash_floats(hf, (size_t)n);
    if (h != c->hfield_hash) { phys_batch_set_hfield(c->b, hf, n); c->hfield_hash = h; }
}

/* what still has to travel before a launch: model / terrain edits, and the perturbation arrays once they are in use
 * (they stay in HBM: the kernel walks them body by body, which is no access pattern for PCIe) */
static void sim_push(cassie_sim_t *c)
{
    sim_recompile(c);
    sim_push_hfield(c);
    if (!c->applied_in_use) {
        for (size_t i = 0; i < sizeof c->d.qfrc_applied / sizeof(double) && !c->applied_in_use; ++i) c->applied_in_use = c->d.qfrc_applied[i] != 0;
        for (size_t i = 0; i < sizeof c->d.xfrc_applied / sizeof(double) && !c->applied_in_use; ++i) c->applied_in_use = c->d.xfrc_applied[i] != 0;
    }
    if (c->applied_in_use) {
        phys_batch_upload_async(c->b, PHYS_F_QFRC_APPLIED, c->d.qfrc_applied, 0, 1);
        phys_batch_upload_async(c->b, PHYS_F_XFRC_APPLIED, c->d.xfrc_applied, 0, 1);
    }
}

static void sim_unpack_ext(cassie_sim_t *c)
{
    int ns = nsize(c, PHYS_NSITE);
    for (int s = 0; s < ns && s < CM_MAXSITE; ++s)
        for (int i = 0; i < 3; ++i) c->d.site_xpos[3 * s + i] = c->ext.site_xpos[s][i];
    c->ext_stale = false;
}

/* the derived read-out of the last launch, fetched from HBM on first use */
static const cm_ext_t *sim_ext(const cassie_sim_t *cc)
{
    cassie_sim_t *c = (cassie_sim_t *)cc;
    if (c->ext_stale) {
        phys_batch_download_ext(c->b, &c->ext, 0, 1);
        sim_unpack_ext(c);
    }
    return &c->ext;
}

/* mj_step1 + mj_step2 (reference :1130-1134) */
static void physics_step(cassie_sim_t *c, int nsteps)
{
    sim_push(c);
    phys_batch_step(c->b, nsteps, NULL);
    phys_batch_sync(c->b);
    c->ext_stale = true;
}

/* mj_forward (reference :971, :1029, :1223) */
static void physics_forward(cassie_sim_t *c)
{
    sim_push(c);
    phys_batch_forward(c->b, NULL);
    phys_batch_sync(c->b);
    c->ext_stale = true;
}

/* position-dependent quantities only (the mj_kinematics / mj_fwdPosition / mj_comVel calls the
 * reference's getters make): refreshes xpos / xquat / ext but leaves qacc, sensordata and actuator_velocity alone,
 * because the encoder models read the sensordata of the last *step* */
static void refresh_derived(const cassie_sim_t *cc)
{
    cassie_sim_t *c = (cassie_sim_t *)cc;
    sim_push(c);
    phys_batch_forward_kinematics(c->b, NULL);
    phys_batch_download_ext_async(c->b, &c->ext, 0, 1);
    phys_batch_sync(c->b);
    sim_unpack_ext(c);
}

/* --------------------------------------------------------------- instances --- */
static void lookup_ids(cassie_sim_t *c)
{
    c->left_foot_body = phys_model_name2id(c->m, OBJ_BODY, "left-foot");
    c->right_foot_body = phys_model_name2id(c->m, OBJ_BODY, "right-foot");
    c->left_heel = phys_model_name2id(c->m, OBJ_SITE, "left-heel");
    c->right_heel = phys_model_name2id(c->m, OBJ_SITE, "right-heel");
    c->left_toe = phys_model_name2id(c->m, OBJ_SITE, "left-toe");
    c->right_toe = phys_model_name2id(c->m, OBJ_SITE, "right-toe");
}

static bool sim_attach_physics(cassie_sim_t *c)
{
    char err[256];
    if (phys_model_compile(c->m, &c->pod, err, sizeof err) != 0) {
        fprintf(stderr, "cassiemujoco: model compile failed: %s\n", err);
        return false;
    }
    c->b = phys_batch_create(&c->pod, 1, 0);
    if (!c->b) {
        fprintf(stderr, "cassiemujoco: cannot create the GPU simulation: %s\n", phys_last_error());
        return false;
    }
    phys_batch_enable_ext(c->b, 1);
    /* zero-copy: the kernel reads and writes the host mirrors in place (c is pinned and device-mapped) */
    struct { int field; double *host; } bound[] = {
        {PHYS_F_QPOS, c->d.qpos}, {PHYS_F_QVEL, c->d.qvel}, {PHYS_F_QACC_WARMSTART, c->d.qacc_warmstart}, {PHYS_F_TIME, &c->d.time},
        {PHYS_F_CTRL, c->d.ctrl}, {PHYS_F_QACC, c->d.qacc}, {PHYS_F_SENSORDATA, c->d.sensordata},
        {PHYS_F_ACTUATOR_VELOCITY, c->d.actuator_velocity}, {PHYS_F_XPOS, c->d.xpos}, {PHYS_F_XQUAT, c->d.xquat}};
    for (size_t i = 0; i < sizeof bound / sizeof bound[0]; ++i)
        if (phys_batch_bind(c->b, bound[i].field, bound[i].host) != 0) {
            fprintf(stderr, "cassiemujoco: cannot bind the simulator's state to the GPU batch: %s\n", phys_last_error());
            return false;
        }
    const float *hf = phys_model_hfield_data(c->m);
    if (hf) phys_batch_set_hfield(c->b, hf, phys_model_size(c->m, PHYS_NHFIELDDATA));
    c->hfield_hash = hf ? phys_hash_floats(hf, (size_t)phys_model_size(c->m, PHYS_NHFIELDDATA)) : 0;
    c->model_print = phys_model_fingerprint(c->m);
    c->ext_stale = true;
    lookup_ids(c);
    return true;
}

cassie_sim_t *cassie_sim_init(const char *modelfile, bool reinit)
{
    if (!library_initialized && !cassie_mujoco_init(modelfile)) return NULL;
    cassie_sim_t *c = sim_alloc();
    if (!c) return NULL;
    if (reinit && !load_global_model(modelfile)) { phys_host_free(c); return NULL; }
    c->m = phys_model_copy(initial_model);
    c->host = cassie_hostenv_alloc(); /* cassie_out_init + Agility block alloc/setup */
    if (!c->m || !c->host || !sim_attach_physics(c)) { cassie_sim_free(c); return NULL; }
    const double *q0 = phys_model_array(c->m, PHYS_M_QPOS0);
    int nq = nsize(c, PHYS_NQ);
    memcpy(c->d.qpos, q0, sizeof(double) * nq);
    memcpy(&c->d.qpos[7], &qpos_nominal[7], 28 * sizeof(double));
    physics_forward(c);
    return c;
}

void cassie_sim_copy_just_sim(cassie_sim_t *dst, const cassie_sim_t *src)
{
    /* model + data (mj_copyModel + mj_copyData, reference :1093-1100) and the three block states */
    phys_model_free(dst->m);
    dst->m = phys_model_copy(src->m);
    dst->model_print = 0; /* another model object: compare its compiled form with what the batch holds before the next launch */
    dst->d = src->d;
    dst->d.warmstart_dirty = 1;
    dst->ext = *sim_ext(src); dst->ext_stale = false;
    lookup_ids(dst);
    cassie_core_sim_copy(cassie_hostenv_core(dst->host), cassie_hostenv_core(src->host));
    state_output_copy(cassie_hostenv_estimator(dst->host), cassie_hostenv_estimator(src->host));
    pd_input_copy(cassie_hostenv_pd(dst->host), cassie_hostenv_pd(src->host));
}

void cassie_sim_copy(cassie_sim_t *dst, const cassie_sim_t *src)
{
    cassie_hostenv_copy(dst->host, src->host); /* POD part (cassie_out, filters, delay line) and block states */
    cassie_sim_copy_just_sim(dst, src);
}

cassie_sim_t *cassie_sim_duplicate(const cassie_sim_t *src)
{
    /* the reference version dereferences an uninitialised model pointer (:1075-1076); this one works */
    cassie_sim_t *c = sim_alloc();
    if (!c) return NULL;
    c->m = phys_model_copy(src->m);
    c->host = cassie_hostenv_alloc();
    if (!c->m || !c->host || !sim_attach_physics(c)) { cassie_sim_free(c); return NULL; }
    cassie_sim_copy(c, src);
    return c;
}

void cassie_sim_free(cassie_sim_t *c)
{
    if (!c) return;
    if (c->b) phys_batch_free(c->b);
    cassie_hostenv_free(c->host);
    if (c->m) phys_model_free(c->m);
    phys_host_free(c);
}

/* ---------------------------------------------------------------- stepping --- */
/* The encoder / motor constants are re-read from the model before every step because callers may edit gear, ctrlrange
 * or the user data through the exposed pointers.  If the edited model is no longer usable (e.g. an actuator's gear no
 * longer matches its drive encoder's) the previous constants are kept and the problem is reported once, instead of
 * stepping with a half-written or zero table (division by zero -> NaN ctrl on the GPU). */
static bool refresh_hostmodel(cassie_sim_t *c)
{
    cassie_hostmodel_t hm;
    if (cassie_hostmodel_from_model(c->m, &hm) == 0) { c->hm = hm; return true; }
    static bool told = false;
    if (!told) {
        fprintf(stderr, "cassiemujoco: the edited model no longer yields valid encoder / motor constants "
                        "(actuator gear vs sensor objid, nuser_sensor / nuser_actuator); keeping the previous ones\n");
        told = true;
    }
    return false;
}

static void step_physics_after_host(cassie_sim_t *c)
{
    const double dt = *phys_model_array(c->m, PHYS_M_TIMESTEP);
    const int mjsteps = (int)round(5e-4 / dt);
    if (mjsteps > 0) physics_step(c, mjsteps);
}

void cassie_sim_step_ethercat(cassie_sim_t *c, cassie_out_t *y, const cassie_in_t *u)
{
    /* motor model (new torque enters the delay line, oldest goes to ctrl), then the measurement of the
     * current state before the control acts, then the physics */
    refresh_hostmodel(c);
    cassie_hostenv_ethercat(c->host, &c->hm, u, c->d.sensordata, c->d.actuator_velocity, c->d.ctrl, y);
    step_physics_after_host(c);
}

void cassie_sim_step(cassie_sim_t *c, cassie_out_t *y, const cassie_user_in_t *u)
{
    refresh_hostmodel(c);
    cassie_hostenv_step(c->host, &c->hm, u, c->d.sensordata, c->d.actuator_velocity, c->d.ctrl, y);
    step_physics_after_host(c);
}

void cassie_sim_step_pd(cassie_sim_t *c, state_out_t *y, const pd_in_t *u)
{
    cassie_out_t cassie_out;
    refresh_hostmodel(c);
    cassie_hostenv_step_pd_pre(c->host, &c->hm, u, c->d.sensordata, c->d.actuator_velocity, c->d.ctrl, &cassie_out);
    step_physics_after_host(c);
    cassie_hostenv_step_pd_post(c->host, &cassie_out, y);
}

void cassie_sim_step_pd_no2khz(cassie_sim_t *c, state_out_t *y, const pd_in_t *u)
{
    cassie_out_t cassie_out;
    refresh_hostmodel(c);
    cassie_hostenv_step_pd_pre(c->host, &c->hm, u, c->d.sensordata, c->d.actuator_velocity, c->d.ctrl, &cassie_out);
    physics_step(c, 1); /* exactly one physics step whatever the timestep (reference :1175) */
    cassie_hostenv_step_pd_post(c->host, &cassie_out, y);
}

static void quat_integrate(double *q, const double *w, double dt)
{
    double ax[3] = {w[0], w[1], w[2]};
    double n = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    if (n < 1e-15) { ax[0] = 1; ax[1] = ax[2] = 0; } else { ax[0] /= n; ax[1] /= n; ax[2] /= n; }
    double ang = dt * n, s = sin(ang / 2);
    double r[4] = {cos(ang / 2), ax[0] * s, ax[1] * s, ax[2] * s};
    double qn = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (qn < 1e-15) { q[0] = 1; q[1] = q[2] = q[3] = 0; } else { for (int i = 0; i < 4; ++i) q[i] /= qn; }
    double t[4] = {q[0] * r[0] - q[1] * r[1] - q[2] * r[2] - q[3] * r[3], q[0] * r[1] + q[1] * r[0] + q[2] * r[3] - q[3] * r[2],
                   q[0] * r[2] - q[1] * r[3] + q[2] * r[0] + q[3] * r[1], q[0] * r[3] + q[1] * r[2] - q[2] * r[1] + q[3] * r[0]};
    memcpy(q, t, sizeof t);
}

/* mj_integratePos on the host mirrors (reference :1183-1189); the estimator is then fed an
 * uninitialised cassie_out_t in the reference -- here a zeroed one */
void cassie_integrate_pos(cassie_sim_t *c, state_out_t *y)
{
    const int *jt = phys_model_iarray(c->m, PHYS_MI_JNT_TYPE);
    const int *qa = phys_model_iarray(c->m, PHYS_MI_JNT_QPOSADR), *da = phys_model_iarray(c->m, PHYS_MI_JNT_DOFADR);
    const double dt = *phys_model_array(c->m, PHYS_M_TIMESTEP);
    int njnt = nsize(c, PHYS_NJNT);
    for (int j = 0; j < njnt; ++j) {
        int q = qa[j], d = da[j];
        if (jt[j] == CM_JNT_FREE) {
            for (int i = 0; i < 3; ++i) c->d.qpos[q + i] += dt * c->d.qvel[d + i];
            quat_integrate(&c->d.qpos[q + 3], &c->d.qvel[d + 3], dt);
        } else if (jt[j] == CM_JNT_BALL) {
            quat_integrate(&c->d.qpos[q], &c->d.qvel[d], dt);
        } else {
            c->d.qpos[q] += dt * c->d.qvel[d];
        }
    }
    cassie_out_t cassie_out;
    memset(&cassie_out, 0, sizeof cassie_out);
    state_output_step(cassie_hostenv_estimator(c->host), &cassie_out, y);
}

int cassie_sim_forward(cassie_sim_t *c) { physics_forward(c); return 0; }

/* ------------------------------------------------------ sizes and pointers --- */
int cassie_sim_nv(const cassie_sim_t *c) { return nsize(c, PHYS_NV); }
int cassie_sim_nq(const cassie_sim_t *c) { return nsize(c, PHYS_NQ); }
int cassie_sim_nu(const cassie_sim_t *c) { return nsize(c, PHYS_NU); }
int cassie_sim_nbody(const cassie_sim_t *c) { return nsize(c, PHYS_NBODY); }
int cassie_sim_njnt(const cassie_sim_t *c) { return nsize(c, PHYS_NJNT); }
int cassie_sim_ngeom(const cassie_sim_t *c) { return nsize(c, PHYS_NGEOM); }
void cassie_sim_params(cassie_sim_t *c, int *p)
{
    p[0] = nsize(c, PHYS_NQ); p[1] = nsize(c, PHYS_NV); p[2] = nsize(c, PHYS_NU);
    p[3] = nsize(c, PHYS_NSENSORDATA); p[4] = nsize(c, PHYS_NBODY); p[5] = nsize(c, PHYS_NGEOM);
}
int *cassie_sim_jnt_qposadr(cassie_sim_t *c) { return phys_model_iarray(c->m, PHYS_MI_JNT_QPOSADR); }
int *cassie_sim_jnt_dofadr(cassie_sim_t *c) { return phys_model_iarray(c->m, PHYS_MI_JNT_DOFADR); }

int cassie_sim_mj_name2id(cassie_sim_t *c, char *mj_type, char *name)
{
    static const struct { const char *s; int t; } map[] = {
        {"body", OBJ_BODY}, {"xbody", OBJ_XBODY}, {"joint", OBJ_JOINT}, {"geom", OBJ_GEOM}, {"site", OBJ_SITE},
        {"camera", OBJ_CAMERA}, {"hfield", OBJ_HFIELD}, {"equality", OBJ_EQUALITY}, {"actuator", OBJ_ACTUATOR},
        {"sensor", OBJ_SENSOR}};
    for (unsigned i = 0; i < sizeof map / sizeof map[0]; ++i)
        if (strcmp(mj_type, map[i].s) == 0) return phys_model_name2id(c->m, map[i].t, name);
    return -1; /* object kinds the supported MJCF subset does not have (lights, meshes, tendons, ...) */
}

void *cassie_sim_mjmodel(cassie_sim_t *c) { return c->m; }
void *cassie_sim_mjdata(cassie_sim_t *c) { return &c->d; }
double *cassie_sim_time(cassie_sim_t *c) { return &c->d.time; }
double *cassie_sim_timestep(cassie_sim_t *c) { return phys_model_array(c->m, PHYS_M_TIMESTEP); }
void cassie_sim_set_timestep(cassie_sim_t *c, double dt) { *phys_model_array(c->m, PHYS_M_TIMESTEP) = dt; }
double *cassie_sim_qpos(cassie_sim_t *c) { return c->d.qpos; }
double *cassie_sim_qvel(cassie_sim_t *c) { return c->d.qvel; }
double *cassie_sim_qacc(cassie_sim_t *c) { return c->d.qacc; }
double *cassie_sim_accel(cassie_sim_t *c) { return c->d.qacc; }
double *cassie_sim_qfrc(cassie_sim_t *c) { return c->d.qfrc_applied; }
double *cassie_sim_ctrl(cassie_sim_t *c) { return c->d.ctrl; }
void cassie_sim_setctrl(cassie_sim_t *c, double *ctrl) { for (int i = 0; i < nsize(c, PHYS_NU); ++i) c->d.ctrl[i] = ctrl[i]; }
double *cassie_sim_act_vel(cassie_sim_t *c) { return c->d.actuator_velocity; }
double *cassie_sim_sensordata(cassie_sim_t *c) { return c->d.sensordata; }

static int body_id(const cassie_sim_t *c, const char *name) { return phys_model_name2id(c->m, OBJ_BODY, name); }
static int geom_id(const cassie_sim_t *c, const char *name) { return phys_model_name2id(c->m, OBJ_GEOM, name); }
static int site_id(const cassie_sim_t *c, const char *name) { return phys_model_name2id(c->m, OBJ_SITE, name); }
static int joint_id(const cassie_sim_t *c, const char *name) { return phys_model_name2id(c->m, OBJ_JOINT, name); }

double *cassie_sim_xpos(cassie_sim_t *c, const char *name)
{
    int b = body_id(c, name);
    return b < 0 ? zero_scratch : &c->d.xpos[3 * b];
}
double *cassie_sim_xquat(cassie_sim_t *c, const char *name)
{
    int b = body_id(c, name);
    return b < 0 ? zero_scratch : &c->d.xquat[4 * b];
}
double *cassie_sim_site_xpos(cassie_sim_t *c, const char *name)
{
    int s = site_id(c, name);
    (void)sim_ext(c); /* the site positions come with the derived read-out */
    return (s < 0 || s >= CM_MAXSITE) ? zero_scratch : &c->d.site_xpos[3 * s];
}

static void mat2quat(double *q, const double *m)
{
    double tr = m[0] + m[4] + m[8];
    if (tr > 0) { double s = sqrt(tr + 1.0) * 2; q[0] = 0.25 * s; q[1] = (m[7] - m[5]) / s; q[2] = (m[2] - m[6]) / s; q[3] = (m[3] - m[1]) / s; }
    else if (m[0] > m[4] && m[0] > m[8]) { double s = sqrt(1.0 + m[0] - m[4] - m[8]) * 2; q[0] = (m[7] - m[5]) / s; q[1] = 0.25 * s; q[2] = (m[1] + m[3]) / s; q[3] = (m[2] + m[6]) / s; }
    else if (m[4] > m[8]) { double s = sqrt(1.0 + m[4] - m[0] - m[8]) * 2; q[0] = (m[2] - m[6]) / s; q[1] = (m[1] + m[3]) / s; q[2] = 0.25 * s; q[3] = (m[5] + m[7]) / s; }
    else { double s = sqrt(1.0 + m[8] - m[0] - m[4]) * 2; q[0] = (m[3] - m[1]) / s; q[1] = (m[2] + m[6]) / s; q[2] = (m[5] + m[7]) / s; q[3] = 0.25 * s; }
}

void cassie_sim_site_xquat(cassie_sim_t *c, const char *name, double *xquat)
{
    int s = site_id(c, name);
    if (s < 0 || s >= CM_MAXSITE) { xquat[0] = 1; xquat[1] = xquat[2] = xquat[3] = 0; return; }
    mat2quat(xquat, sim_ext(c)->site_xmat[s]);
}

void cassie_sim_read_rangefinder(cassie_sim_t *c, double ranges[6]) { memcpy(ranges, &c->d.sensordata[29], 6 * sizeof(double)); }

/* --------------------------------------------------------- model parameters --- */
double *cassie_sim_dof_damping(cassie_sim_t *c) { return phys_model_array(c->m, PHYS_M_DOF_DAMPING); }
void cassie_sim_set_dof_damping(cassie_sim_t *c, double *damp)
{
    double *a = phys_model_array(c->m, PHYS_M_DOF_DAMPING);
    for (int i = 0; i < nsize(c, PHYS_NV); ++i) a[i] = damp[i];
}
int cassie_sim_get_joint_num_dof(cassie_sim_t *c, const char *name)
{
    int j = joint_id(c, name);
    if (j < 0) return 0;
    int t = phys_model_iarray(c->m, PHYS_MI_JNT_TYPE)[j];
    return t == CM_JNT_FREE ? 6 : (t == CM_JNT_BALL ? 3 : 1);
}
void cassie_sim_set_dof_name_damping(cassie_sim_t *c, const char *name, double *damp)
{
    int j = joint_id(c, name);
    if (j < 0) return;
    double *a = phys_model_array(c->m, PHYS_M_DOF_DAMPING);
    int adr = phys_model_iarray(c->m, PHYS_MI_JNT_DOFADR)[j], n = cassie_sim_get_joint_num_dof(c, name);
    for (int i = 0; i < n; ++i) a[adr + i] = damp[i];
}
double *cassie_sim_get_dof_name_damping(cassie_sim_t *c, const char *name)
{
    int j = joint_id(c, name);
    if (j < 0) return zero_scratch;
    return &phys_model_array(c->m, PHYS_M_DOF_DAMPING)[phys_model_iarray(c->m, PHYS_MI_JNT_DOFADR)[j]];
}
double *cassie_sim_body_mass(cassie_sim_t *c) { return phys_model_array(c->m, PHYS_M_BODY_MASS); }
void cassie_sim_set_body_mass(cassie_sim_t *c, double *mass)
{
    double *a = phys_model_array(c->m, PHYS_M_BODY_MASS);
    for (int i = 0; i < nsize(c, PHYS_NBODY); ++i) a[i] = mass[i];
}
void cassie_sim_set_body_name_mass(cassie_sim_t *c, const char *name, double mass)
{
    int b = body_id(c, name);
    if (b >= 0) phys_model_array(c->m, PHYS_M_BODY_MASS)[b] = mass;
}
double cassie_sim_get_body_name_mass(cassie_sim_t *c, const char *name)
{
    int b = body_id(c, name);
    return b < 0 ? 0.0 : phys_model_array(c->m, PHYS_M_BODY_MASS)[b];
}
double *cassie_sim_body_ipos(cassie_sim_t *c) { return phys_model_array(c->m, PHYS_M_BODY_IPOS); }
void cassie_sim_set_body_ipos(cassie_sim_t *c, double *ipos)
{
    /* dense [nbody][3] input; the reference indexes ipos[i + j] (stride bug, :1389), here 3*i + j */
    double *a = phys_model_array(c->m, PHYS_M_BODY_IPOS);
    for (int i = 0; i < 3 * nsize(c, PHYS_NBODY); ++i) a[i] = ipos[i];
}
void cassie_sim_set_body_name_ipos(cassie_sim_t *c, const char *name, double *ipos)
{
    int b = body_id(c, name);
    if (b >= 0) memcpy(&phys_model_array(c->m, PHYS_M_BODY_IPOS)[3 * b], ipos, 3 * sizeof(double));
}
double *cassie_sim_get_body_name_ipos(cassie_sim_t *c, const char *name)
{
    int b = body_id(c, name);
    return b < 0 ? zero_scratch : &phys_model_array(c->m, PHYS_M_BODY_IPOS)[3 * b];
}
void cassie_sim_set_body_name_pos(cassie_sim_t *c, const char *name, double *data)
{
    int b = body_id(c, name);
    if (b >= 0) memcpy(&phys_model_array(c->m, PHYS_M_BODY_POS)[3 * b], data, 3 * sizeof(double));
}
double *cassie_sim_get_body_name_pos(cassie_sim_t *c, const char *name)
{
    int b = body_id(c, name);
    return b < 0 ? zero_scratch : &phys_model_array(c->m, PHYS_M_BODY_POS)[3 * b];
}
double *cassie_sim_geom_friction(cassie_sim_t *c) { return phys_model_array(c->m, PHYS_M_GEOM_FRICTION); }
void cassie_sim_set_geom_friction(cassie_sim_t *c, double *fric)
{
    double *a = phys_model_array(c->m, PHYS_M_GEOM_FRICTION);
    for (int i = 0; i < 3 * nsize(c, PHYS_NGEOM); ++i) a[i] = fric[i];
}
void cassie_sim_set_geom_name_friction(cassie_sim_t *c, const char *name, double *fric)
{
    /* geom_friction is [ngeom][3]; the reference indexes it with the bare geom id (:1430) */
    int g = geom_id(c, name);
    if (g >= 0) memcpy(&phys_model_array(c->m, PHYS_M_GEOM_FRICTION)[3 * g], fric, 3 * sizeof(double));
}
double *cassie_sim_get_geom_name_friction(cassie_sim_t *c, const char *name)
{
    int g = geom_id(c, name);
    return g < 0 ? zero_scratch : &phys_model_array(c->m, PHYS_M_GEOM_FRICTION)[3 * g];
}
float *cassie_sim_geom_rgba(cassie_sim_t *c) { return phys_model_geom_rgba(c->m); }
float *cassie_sim_geom_name_rgba(cassie_sim_t *c, const char *name)
{
    int g = geom_id(c, name);
    return g < 0 ? zero_scratch_f : &phys_model_geom_rgba(c->m)[4 * g];
}
void cassie_sim_set_geom_rgba(cassie_sim_t *c, float *rgba)
{
    float *a = phys_model_geom_rgba(c->m);
    for (int i = 0; i < 4 * nsize(c, PHYS_NGEOM); ++i) a[i] = rgba[i];
}
void cassie_sim_set_geom_name_rgba(cassie_sim_t *c, const char *name, float *rgba)
{
    int g = geom_id(c, name);
    if (g >= 0) memcpy(&phys_model_geom_rgba(c->m)[4 * g], rgba, 4 * sizeof(float));
}
#define GEOM_ARRAY_ACCESSORS(field, WHICH, N)                                                         \
    double *cassie_sim_geom_##field(cassie_sim_t *c) { return phys_model_array(c->m, WHICH); }        \
    double *cassie_sim_geom_name_##field(cassie_sim_t *c, const char *name)                           \
    {                                                                                                 \
        int g = geom_id(c, name);                                                                     \
        return g < 0 ? zero_scratch : &phys_model_array(c->m, WHICH)[N * g];                          \
    }                                                                                                 \
    void cassie_sim_set_geom_##field(cassie_sim_t *c, double *v)                                      \
    {                                                                                                 \
        double *a = phys_model_array(c->m, WHICH);                                                    \
        for (int i = 0; i < N * nsize(c, PHYS_NGEOM); ++i) a[i] = v[i];                               \
    }                                                                                                 \
    void cassie_sim_set_geom_name_##field(cassie_sim_t *c, const char *name, double *v)               \
    {                                                                                                 \
        int g = geom_id(c, name);                                                                     \
        if (g >= 0) memcpy(&phys_model_array(c->m, WHICH)[N * g], v, N * sizeof(double));             \
    }
GEOM_ARRAY_ACCESSORS(quat, PHYS_M_GEOM_QUAT, 4)
GEOM_ARRAY_ACCESSORS(pos, PHYS_M_GEOM_POS, 3)
GEOM_ARRAY_ACCESSORS(size, PHYS_M_GEOM_SIZE, 3)

int cassie_sim_get_hfield_nrow(cassie_sim_t *c) { return nsize(c, PHYS_HFIELD_NROW); }
int cassie_sim_get_hfield_ncol(cassie_sim_t *c) { return nsize(c, PHYS_HFIELD_NCOL); }
int cassie_sim_get_nhfielddata(cassie_sim_t *c) { return nsize(c, PHYS_NHFIELDDATA); }
double *cassie_sim_get_hfield_size(cassie_sim_t *c) { return phys_model_array(c->m, PHYS_M_HFIELD_SIZE); }
void cassie_sim_set_hfield_size(cassie_sim_t *c, double size[4]) { memcpy(phys_model_array(c->m, PHYS_M_HFIELD_SIZE), size, 4 * sizeof(double)); }
float *cassie_sim_hfielddata(cassie_sim_t *c) { return phys_model_hfield_data(c->m); }
void cassie_sim_set_hfield_dense_sampling(cassie_sim_t *c, bool on) { if (c) phys_model_set_flag(c->m, CM_FLAG_HFDENSE, on ? 1 : 0); }
void cassie_sim_set_hfield_multi_contact(cassie_sim_t *c, bool on) { if (c) phys_model_set_flag(c->m, CM_FLAG_HFMULTI, on ? 1 : 0); }
void cassie_sim_set_hfield_prism_contacts(cassie_sim_t *c, bool on) { if (c) phys_model_set_flag(c->m, CM_FLAG_HFPRISM, on ? 1 : 0); }
void cassie_sim_set_hfielddata(cassie_sim_t *c, float *data)
{
    float *a = phys_model_hfield_data(c->m);
    int n = nsize(c, PHYS_NHFIELDDATA);
    if (!a) return;
    for (int i = 0; i < n; ++i) a[i] = data[i];
    phys_batch_set_hfield(c->b, a, n);
}

void cassie_sim_just_set_const(cassie_sim_t *c) { phys_model_set_const(c->m); }
void cassie_sim_set_const(cassie_sim_t *c)
{
    phys_model_set_const(c->m);
    memcpy(c->d.qpos, qpos_nominal, 35 * sizeof(double));
    memset(c->d.qvel, 0, sizeof c->d.qvel);
    memset(c->d.qacc, 0, sizeof c->d.qacc);
    c->d.time = 0.0;
    physics_forward(c);
}

/* -------------------------------------------------------- derived quantities --- */
static void cross3(double *r, const double *a, const double *b)
{
    double t0 = a[1] * b[2] - a[2] * b[1], t1 = a[2] * b[0] - a[0] * b[2], t2 = a[0] * b[1] - a[1] * b[0];
    r[0] = t0; r[1] = t1; r[2] = t2;
}

/* Jacobian read-out of a world point attached to a body from the kernel's motion axes (mj_jac role) */
static void point_jacobian(const cassie_sim_t *c, int body, const double *point, double *jacp, double *jacr)
{
    int nv = nsize(c, PHYS_NV);
    if (jacp) memset(jacp, 0, 3 * nv * sizeof(double));
    if (jacr) memset(jacr, 0, 3 * nv * sizeof(double));
    if (body <= 0) return;
    const double *com = sim_ext(c)->subtree_com[c->pod.body_rootid[body]];
    double off[3] = {point[0] - com[0], point[1] - com[1], point[2] - com[2]};
    for (int k = 0; k < nv; ++k) {
        if (!((c->pod.body_dofmask[body] >> k) & 1ull)) continue;
        double t[3];
        cross3(t, sim_ext(c)->cdof[k], off);
        for (int i = 0; i < 3; ++i) {
            if (jacp) jacp[i * nv + k] = sim_ext(c)->cdof[k][3 + i] + t[i];
            if (jacr) jacr[i * nv + k] = sim_ext(c)->cdof[k][i];
        }
    }
}

void cassie_sim_get_jacobian(cassie_sim_t *c, double *jac, const char *name)
{
    refresh_derived(c);
    int b = body_id(c, name);
    point_jacobian(c, b, b >= 0 ? &c->d.xpos[3 * b] : zero_scratch, jac, NULL);
}
void cassie_sim_get_jacobian_full(cassie_sim_t *c, double *jac, double *jac_rot, const char *name)
{
    refresh_derived(c);
    int b = body_id(c, name);
    point_jacobian(c, b, b >= 0 ? &c->d.xpos[3 * b] : zero_scratch, jac, jac_rot);
}
void cassie_sim_get_jacobian_full_site(cassie_sim_t *c, double *jac, double *jac_rot, const char *name)
{
    refresh_derived(c);
    int s = site_id(c, name);
    if (s < 0 || s >= CM_MAXSITE) { point_jacobian(c, 0, zero_scratch, jac, jac_rot); return; }
    point_jacobian(c, c->pod.site_bodyid[s], sim_ext(c)->site_xpos[s], jac, jac_rot);
}

static int contact_body(const cassie_sim_t *c, int fullgeom) { return phys_model_iarray(c->m, PHYS_MI_GEOM_BODYID)[fullgeom]; }

bool cassie_sim_check_obstacle_collision(const cassie_sim_t *c)
{
    const double *user = phys_model_array(c->m, PHYS_M_GEOM_USER);
    int nug = nsize(c, PHYS_NUSER_GEOM);
    if (!user || nug < 1) return false;
    for (int i = 0; i < sim_ext(c)->ncon; ++i)
        if (user[nug * sim_ext(c)->con_geom1[i]] == 1 || user[nug * sim_ext(c)->con_geom2[i]] == 1) return true;
    return false;
}
bool cassie_sim_check_self_collision(const cassie_sim_t *c)
{
    const double *user = phys_model_array(c->m, PHYS_M_GEOM_USER);
    int nug = nsize(c, PHYS_NUSER_GEOM);
    if (!user || nug < 1) return false;
    for (int i = 0; i < sim_ext(c)->ncon; ++i)
        if (user[nug * sim_ext(c)->con_geom1[i]] == 2 && user[nug * sim_ext(c)->con_geom2[i]] == 2) return true;
    return false;
}
bool cassie_sim_geom_collision(const cassie_sim_t *c, int geom_group)
{
    const int *grp = phys_model_iarray(c->m, PHYS_MI_GEOM_GROUP);
    for (int i = 0; i < sim_ext(c)->ncon; ++i) {
        int g1 = grp[sim_ext(c)->con_geom1[i]], g2 = grp[sim_ext(c)->con_geom2[i]];
        if ((g1 == 1 && g2 == geom_group) || (g2 == 1 && g1 == geom_group)) return true;
    }
    return false;
}

/* contact force in world axes: frame^T * [normal, tangent1, tangent2] (mj_contactForce + mju_rotVecMatT) */
static void contact_force_world(const cassie_sim_t *c, int i, double *fw)
{
    const double *fr = sim_ext(c)->con_frame[i], *f = sim_ext(c)->con_force[i];
    for (int k = 0; k < 3; ++k) fw[k] = fr[k] * f[0] + fr[3 + k] * f[1] + fr[6 + k] * f[2];
}

void cassie_sim_foot_forces(const cassie_sim_t *c, double cfrc[12])
{
    memset(cfrc, 0, 12 * sizeof(double));
    for (int i = 0; i < sim_ext(c)->ncon; ++i) {
        int b1 = contact_body(c, sim_ext(c)->con_geom1[i]), b2 = contact_body(c, sim_ext(c)->con_geom2[i]);
        double fw[3];
        contact_force_world(c, i, fw);
        for (int side = 0; side < 2; ++side) {
            int foot = side == 0 ? c->left_foot_body : c->right_foot_body;
            if (b1 != foot && b2 != foot) continue;
            double sgn = (b1 == foot) ? -1.0 : 1.0;
            for (int j = 0; j < 3; ++j) cfrc[6 * side + j] += sgn * fw[j];
        }
    }
}

void cassie_sim_heeltoe_forces(const cassie_sim_t *c, double toe_force[6], double heel_force[6])
{
    memset(toe_force, 0, 6 * sizeof(double));
    memset(heel_force, 0, 6 * sizeof(double));
    const int heel[2] = {c->left_heel, c->right_heel}, toe[2] = {c->left_toe, c->right_toe};
    for (int i = 0; i < sim_ext(c)->ncon; ++i) {
        int b1 = contact_body(c, sim_ext(c)->con_geom1[i]), b2 = contact_body(c, sim_ext(c)->con_geom2[i]);
        bool left = b1 == c->left_foot_body || b2 == c->left_foot_body;
        bool right = b1 == c->right_foot_body || b2 == c->right_foot_body;
        if (!left && !right) continue;
        int sign = (b1 == c->left_foot_body || b1 == c->right_foot_body) ? -1 : 1;
        int id = right ? 1 : 0;
        if (heel[id] < 0 || toe[id] < 0 || heel[id] >= CM_MAXSITE || toe[id] >= CM_MAXSITE) continue; /* model without heel/toe sites */
        double fw[3];
        contact_force_world(c, i, fw);
        const double *p = sim_ext(c)->con_pos[i], *tp = sim_ext(c)->site_xpos[toe[id]], *hp = sim_ext(c)->site_xpos[heel[id]];
        double td = hypot(tp[0] - p[0], tp[1] - p[1]), hd = hypot(hp[0] - p[0], hp[1] - p[1]);
        double *dst = td < hd ? toe_force : heel_force;
        for (int j = 0; j < 3; ++j) dst[j + 3 * id] += sign * fw[j];
    }
}

void cassie_sim_foot_positions(const cassie_sim_t *c, double cpos[6])
{
    memset(cpos, 0, 6 * sizeof(double));
    memcpy(cpos, &c->d.xpos[3 * c->left_foot_body], 3 * sizeof(double));
    memcpy(&cpos[3], &c->d.xpos[3 * c->right_foot_body], 3 * sizeof(double));
    double off = sqrt(pow(0.01762, 2) + pow(0.05219, 2)); /* foot joint to mid-foot (reference :1612) */
    cpos[2] -= off;
    cpos[5] -= off;
}

void cassie_sim_foot_velocities(const cassie_sim_t *c, double cvel[12])
{
    refresh_derived(c);
    memcpy(cvel, sim_ext(c)->cvel[c->left_foot_body], 6 * sizeof(double));
    memcpy(&cvel[6], sim_ext(c)->cvel[c->right_foot_body], 6 * sizeof(double));
}

void cassie_sim_body_velocities(const cassie_sim_t *c, double cvel[6], const char *name)
{
    refresh_derived(c);
    memset(cvel, 0, 6 * sizeof(double));
    int b = body_id(c, name);
    if (b >= 0) memcpy(cvel, sim_ext(c)->cvel[b], 6 * sizeof(double));
}

void cassie_sim_foot_orient(const cassie_sim_t *c, double corient[4])
{
    int s = site_id(c, "right-foot-middle"); /* not defined by the shipped models; identity then */
    if (s < 0 || s >= CM_MAXSITE) { corient[0] = 1; corient[1] = corient[2] = corient[3] = 0; return; }
    mat2quat(corient, sim_ext(c)->site_xmat[s]);
}

/* whole-model centre of mass and its velocity / angular momentum from the kernel's read-out */
static double total_mass_com(const cassie_sim_t *c, double com[3])
{
    const double *mass = phys_model_array(c->m, PHYS_M_BODY_MASS);
    int nb = nsize(c, PHYS_NBODY);
    double M = 0;
    com[0] = com[1] = com[2] = 0;
    for (int b = 1; b < nb; ++b) {
        M += mass[b];
        for (int i = 0; i < 3; ++i) com[i] += mass[b] * sim_ext(c)->xipos[b][i];
    }
    if (M > 0) for (int i = 0; i < 3; ++i) com[i] /= M;
    return M;
}
static void body_com_velocity(const cassie_sim_t *c, int b, double v[3])
{
    const double *cv = sim_ext(c)->cvel[b], *rc = sim_ext(c)->subtree_com[c->pod.body_rootid[b]];
    double off[3] = {sim_ext(c)->xipos[b][0] - rc[0], sim_ext(c)->xipos[b][1] - rc[1], sim_ext(c)->xipos[b][2] - rc[2]}, t[3];
    cross3(t, cv, off);
    for (int i = 0; i < 3; ++i) v[i] = cv[3 + i] + t[i];
}

void cassie_sim_cm_position(const cassie_sim_t *c, double cm_pos[3])
{
    refresh_derived(c);
    total_mass_com(c, cm_pos);
}

void cassie_sim_cm_velocity(const cassie_sim_t *c, double cm_vel[3])
{
    refresh_derived(c);
    const double *mass = phys_model_array(c->m, PHYS_M_BODY_MASS);
    int nb = nsize(c, PHYS_NBODY);
    double M = 0;
    cm_vel[0] = cm_vel[1] = cm_vel[2] = 0;
    for (int b = 1; b < nb; ++b) {
        double v[3];
        body_com_velocity(c, b, v);
        M += mass[b];
        for (int i = 0; i < 3; ++i) cm_vel[i] += mass[b] * v[i];
    }
    if (M > 0) for (int i = 0; i < 3; ++i) cm_vel[i] /= M;
}

static void quat2mat(double *m, const double *q)
{
    double q00 = q[0] * q[0], q11 = q[1] * q[1], q22 = q[2] * q[2], q33 = q[3] * q[3];
    double q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3], q12 = q[1] * q[2], q13 = q[1] * q[3], q23 = q[2] * q[3];
    m[0] = q00 + q11 - q22 - q33; m[1] = 2 * (q12 - q03); m[2] = 2 * (q13 + q02);
    m[3] = 2 * (q12 + q03); m[4] = q00 - q11 + q22 - q33; m[5] = 2 * (q23 - q01);
    m[6] = 2 * (q13 - q02); m[7] = 2 * (q23 + q01); m[8] = q00 - q11 - q22 + q33;
}

void cassie_sim_angular_momentum(const cassie_sim_t *c, double Lcm[3])
{
    refresh_derived(c);
    const double *mass = phys_model_array(c->m, PHYS_M_BODY_MASS);
    int nb = nsize(c, PHYS_NBODY);
    double com[3], vcom[3] = {0, 0, 0}, M = total_mass_com(c, com);
    for (int b = 1; b < nb; ++b) { double v[3]; body_com_velocity(c, b, v); for (int i = 0; i < 3; ++i) vcom[i] += mass[b] * v[i] / (M > 0 ? M : 1); }
    Lcm[0] = Lcm[1] = Lcm[2] = 0;
    for (int b = 1; b < nb; ++b) {
        /* spin part: R diag(I) R^T w, with R the inertial frame in world axes */
        double q[4], R[9];
        const double *xq = &c->d.xquat[4 * b], *iq = c->pod.body_iquat[b];
        q[0] = xq[0] * iq[0] - xq[1] * iq[1] - xq[2] * iq[2] - xq[3] * iq[3];
        q[1] = xq[0] * iq[1] + xq[1] * iq[0] + xq[2] * iq[3] - xq[3] * iq[2];
        q[2] = xq[0] * iq[2] - xq[1] * iq[3] + xq[2] * iq[0] + xq[3] * iq[1];
        q[3] = xq[0] * iq[3] + xq[1] * iq[2] - xq[2] * iq[1] + xq[3] * iq[0];
        quat2mat(R, q);
        const double *w = sim_ext(c)->cvel[b], *I = c->pod.body_inertia[b];
        double wl[3] = {R[0] * w[0] + R[3] * w[1] + R[6] * w[2], R[1] * w[0] + R[4] * w[1] + R[7] * w[2], R[2] * w[0] + R[5] * w[1] + R[8] * w[2]};
        for (int i = 0; i < 3; ++i) wl[i] *= I[i];
        for (int i = 0; i < 3; ++i) Lcm[i] += R[3 * i] * wl[0] + R[3 * i + 1] * wl[1] + R[3 * i + 2] * wl[2];
        /* orbital part about the whole-model com */
        double v[3], r[3], t[3];
        body_com_velocity(c, b, v);
        for (int i = 0; i < 3; ++i) { r[i] = sim_ext(c)->xipos[b][i] - com[i]; v[i] = mass[b] * (v[i] - vcom[i]); }
        cross3(t, r, v);
        for (int i = 0; i < 3; ++i) Lcm[i] += t[i];
    }
}

void cassie_sim_full_mass_matrix(const cassie_sim_t *c, double M[1024])
{
    refresh_derived(c);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) M[i * 32 + j] = sim_ext(c)->qM[i][j];
}

void cassie_sim_minimal_mass_matrix(const cassie_sim_t *c, double M[256])
{
    static const int IND[16] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 12, 18, 19, 20, 21, 25, 31}; /* base + 10 motors */
    refresh_derived(c);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) M[i * 16 + j] = sim_ext(c)->qM[IND[i]][IND[j]];
}

void cassie_sim_centroid_inertia(const cassie_sim_t *cc, double Icm[9])
{
    /* literal restatement of reference :1640-1685, including its choice of base orientation */
    cassie_sim_t *c = (cassie_sim_t *)cc;
    double stored[4];
    for (int i = 0; i < 4; ++i) { stored[i] = c->d.qpos[i + 3]; c->d.qpos[i + 3] = 0; }
    c->d.qpos[4] = 1;
    refresh_derived(c);
    double m = sim_ext(c)->qM[0][0], rcm[3];
    total_mass_com(c, rcm);
    for (int i = 0; i < 3; ++i) rcm[i] -= c->d.qpos[i];
    double Ip[3][3], Ic[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Ip[i][j] = sim_ext(c)->qM[i + 3][j + 3];
    Ic[0][0] = Ip[0][0] - m * (rcm[1] * rcm[1] + rcm[2] * rcm[2]);
    Ic[1][1] = Ip[1][1] - m * (rcm[2] * rcm[2] + rcm[0] * rcm[0]);
    Ic[2][2] = Ip[2][2] - m * (rcm[0] * rcm[0] + rcm[1] * rcm[1]);
    Ic[0][1] = Ic[1][0] = Ip[1][0] - m * rcm[1] * rcm[0];
    Ic[1][2] = Ic[2][1] = Ip[2][1] - m * rcm[2] * rcm[1];
    Ic[2][0] = Ic[0][2] = Ip[2][0] - m * rcm[2] * rcm[0];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Icm[3 * i + j] = Ic[i][j];
    for (int i = 0; i < 4; ++i) c->d.qpos[i + 3] = stored[i];
    refresh_derived(c);
}

void cassie_sim_loop_constraint_info(const cassie_sim_t *c, double J_cl[192], double err_cl[6])
{
    refresh_derived(c);
    int idx = 0;
    for (int r = 0; r < sim_ext(c)->ne && r < CM_MAXEQROW && idx < 6; ++r) {
        const char *nm = phys_model_id2name(c->m, OBJ_EQUALITY, sim_ext(c)->eq_id[r]);
        if (!nm || (strcmp(nm, "left-achilles-rod-eq") != 0 && strcmp(nm, "right-achilles-rod-eq") != 0)) continue;
        for (int j = 0; j < 32; ++j) J_cl[idx * 32 + j] = sim_ext(c)->eq_J[r][j];
        err_cl[idx] = sim_ext(c)->eq_pos[r];
        ++idx;
    }
}

void cassie_sim_body_acceleration(const cassie_sim_t *c, double accel[6], const char *name)
{
    /* com-frame acceleration incl. -gravity and the last solved qacc (mj_rnePostConstraint's cacc) */
    refresh_derived(c);
    memset(accel, 0, 6 * sizeof(double));
    int b = body_id(c, name);
    if (b < 0) return;
    for (int i = 0; i < 3; ++i) accel[3 + i] = -c->pod.gravity[i];
    for (int k = 0; k < nsize(c, PHYS_NV); ++k) {
        if (!((c->pod.body_dofmask[b] >> k) & 1ull)) continue;
        for (int i = 0; i < 6; ++i) accel[i] += sim_ext(c)->cdof_dot[k][i] * c->d.qvel[k] + sim_ext(c)->cdof[k][i] * c->d.qacc[k];
    }
}

void cassie_sim_body_contact_force(const cassie_sim_t *c, double cfrc[6], const char *name)
{
    /* What the reference returns (:1781-1810): it hands mj_contactForce's [force; torque] (contact frame) to
     * mju_transformSpatial, whose vectors are [rotational; translational], so the slots come out as
     *   cfrc[0:3] = frame^T (force - (xpos_body - contact_pos) x torque),   cfrc[3:6] = frame^T torque.
     * Every contact of the supported models is condim 1 or 3 -- no contact torque -- which leaves the summed contact
     * force on the body in world axes in cfrc[0:3] (what callers of the reference read) and zeros in cfrc[3:6]. */
    memset(cfrc, 0, 6 * sizeof(double));
    int b = body_id(c, name);
    if (b < 0) return;
    for (int i = 0; i < sim_ext(c)->ncon; ++i) {
        int b1 = contact_body(c, sim_ext(c)->con_geom1[i]), b2 = contact_body(c, sim_ext(c)->con_geom2[i]);
        if (b != b1 && b != b2) continue;
        double fw[3];
        contact_force_world(c, i, fw);
        double sgn = (b == b1) ? -1.0 : 1.0;
        for (int k = 0; k < 3; ++k) cfrc[k] += sgn * fw[k];
    }
}

void cassie_sim_relative_pose(double pos1[3], double quat1[4], double pos2[3], double quat2[4], double pos2_in_pos1[3],
                              double quat2_in_quat1[4])
{
    /* pose 2 expressed in frame 1: q = conj(q1) q2, p = R(q1)^T (p2 - p1) */
    double qc[4] = {quat1[0], -quat1[1], -quat1[2], -quat1[3]}, R[9], d[3];
    quat2mat(R, quat1);
    for (int i = 0; i < 3; ++i) d[i] = pos2[i] - pos1[i];
    for (int i = 0; i < 3; ++i) pos2_in_pos1[i] = R[i] * d[0] + R[3 + i] * d[1] + R[6 + i] * d[2];
    quat2_in_quat1[0] = qc[0] * quat2[0] - qc[1] * quat2[1] - qc[2] * quat2[2] - qc[3] * quat2[3];
    quat2_in_quat1[1] = qc[0] * quat2[1] + qc[1] * quat2[0] + qc[2] * quat2[3] - qc[3] * quat2[2];
    quat2_in_quat1[2] = qc[0] * quat2[2] - qc[1] * quat2[3] + qc[2] * quat2[0] + qc[3] * quat2[1];
    quat2_in_quat1[3] = qc[0] * quat2[3] + qc[1] * quat2[2] - qc[2] * quat2[1] + qc[3] * quat2[0];
}

/* ------------------------------------------- perturbation, radio, reset ... --- */
void cassie_sim_apply_force(cassie_sim_t *c, double xfrc[6], const char *name)
{
    int b = body_id(c, name);
    if (b >= 0) memcpy(&c->d.xfrc_applied[6 * b], xfrc, 6 * sizeof(double));
}
void cassie_sim_clear_forces(cassie_sim_t *c) { memset(c->d.xfrc_applied, 0, sizeof c->d.xfrc_applied); }

void cassie_sim_hold(cassie_sim_t *c)
{
    double *stiff = phys_model_array(c->m, PHYS_M_JNT_STIFFNESS), *damp = phys_model_array(c->m, PHYS_M_DOF_DAMPING);
    double *spring = phys_model_array(c->m, PHYS_M_QPOS_SPRING);
    for (int i = 0; i < 3; ++i) { stiff[i] = 1e5; damp[i] = 1e4; spring[i] = c->d.qpos[i]; }
    for (int i = 3; i < 6; ++i) damp[i] = 1e4;
}
void cassie_sim_release(cassie_sim_t *c)
{
    double *stiff = phys_model_array(c->m, PHYS_M_JNT_STIFFNESS), *damp = phys_model_array(c->m, PHYS_M_DOF_DAMPING);
    for (int i = 0; i < 3; ++i) { stiff[i] = 0; damp[i] = 0; }
    for (int i = 3; i < 6; ++i) damp[i] = 0;
}
void cassie_sim_radio(cassie_sim_t *c, double channels[16]) { for (int i = 0; i < 16; ++i) cassie_hostenv_cassie_out(c->host)->pelvis.radio.channel[i] = channels[i]; }

void cassie_sim_full_reset(cassie_sim_t *c)
{
    /* reference :2008-2033: pose, velocities, controls, perturbations, qacc, torque delay, estimator;
     * time, filters, cassie_out, core, pd and the solver warm start are deliberately left alone */
    memcpy(c->d.qpos, qpos_nominal, 35 * sizeof(double));
    memset(c->d.qvel, 0, sizeof c->d.qvel);
    memset(c->d.ctrl, 0, sizeof c->d.ctrl);
    memset(c->d.qfrc_applied, 0, sizeof c->d.qfrc_applied);
    memset(c->d.xfrc_applied, 0, sizeof c->d.xfrc_applied);
    memset(c->d.qacc, 0, sizeof c->d.qacc);
    cassie_hostenv_reset(c->host);
}

void reset_state_est(cassie_sim_t *c, state_out_t *y)
{
    pd_in_t u;
    memset(&u, 0, sizeof u);
    cassie_user_in_t cassie_user_in;
    cassie_out_t cassie_out;
    cassie_in_t cassie_in;
    memset(&cassie_out, 0, sizeof cassie_out); /* uninitialised in the reference (:2039-2047) */
    cassie_out_t measured;
    (void)cassie_user_in; (void)cassie_in;
    refresh_hostmodel(c);
    cassie_hostenv_step_pd_pre(c->host, &c->hm, &u, c->d.sensordata, c->d.actuator_velocity, c->d.ctrl, &measured);
    state_output_step(cassie_hostenv_estimator(c->host), &cassie_out, y);
}

cassie_out_t cassie_sim_get_cassie_out(cassie_sim_t *c) { return *cassie_hostenv_cassie_out(c->host); }
void cassie_sim_copy_cassie_out(cassie_sim_t *dst, cassie_out_t *y) { memcpy(cassie_hostenv_cassie_out(dst->host), y, sizeof(cassie_out_t)); }
void cassie_sim_copy_mjd(cassie_sim_t *dst, cassie_sim_t *src) { dst->d = src->d; dst->d.warmstart_dirty = 1; dst->ext = *sim_ext(src); dst->ext_stale = false; }
void cassie_sim_copy_state_est(cassie_sim_t *dst, cassie_sim_t *src) { state_output_copy(cassie_hostenv_estimator(dst->host), cassie_hostenv_estimator(src->host)); }
void cassie_sim_run_state_est(cassie_sim_t *c, cassie_out_t *cassie_out, state_out_t *y) { state_output_step(cassie_hostenv_estimator(c->host), cassie_out, y); }
void state_out_free(state_out_t *out) { free(out); }

joint_filter_t *cassie_sim_joint_filter(cassie_sim_t *c) { return cassie_hostenv_joint_filter(c->host); }
void cassie_sim_get_joint_filter(cassie_sim_t *c, double *x, double *y)
{
    for (int j = 0; j < NUM_JOINTS; ++j) {
        for (int i = 0; i < JOINT_FILTER_NB; ++i) x[j * JOINT_FILTER_NB + i] = cassie_hostenv_joint_filter(c->host)[j].x[i];
        for (int i = 0; i < JOINT_FILTER_NA; ++i) y[j * JOINT_FILTER_NA + i] = cassie_hostenv_joint_filter(c->host)[j].y[i];
    }
}
void cassie_sim_set_joint_filter(cassie_sim_t *c, double *x, double *y)
{
    for (int j = 0; j < NUM_JOINTS; ++j) {
        for (int i = 0; i < JOINT_FILTER_NB; ++i) cassie_hostenv_joint_filter(c->host)[j].x[i] = x[j * JOINT_FILTER_NB + i];
        for (int i = 0; i < JOINT_FILTER_NA; ++i) cassie_hostenv_joint_filter(c->host)[j].y[i] = y[j * JOINT_FILTER_NA + i];
    }
}
drive_filter_t *cassie_sim_drive_filter(cassie_sim_t *c) { return cassie_hostenv_drive_filter(c->host); }
void cassie_sim_get_drive_filter(cassie_sim_t *c, int *x)
{
    for (int i = 0; i < NUM_DRIVES; ++i) for (int j = 0; j < DRIVE_FILTER_NB; ++j) x[i * DRIVE_FILTER_NB + j] = cassie_hostenv_drive_filter(c->host)[i].x[j];
}
void cassie_sim_set_drive_filter(cassie_sim_t *c, int *x)
{
    for (int i = 0; i < NUM_DRIVES; ++i) for (int j = 0; j < DRIVE_FILTER_NB; ++j) cassie_hostenv_drive_filter(c->host)[i].x[j] = x[i * DRIVE_FILTER_NB + j];
}
void cassie_sim_torque_delay(cassie_sim_t *c, double *t)
{
    for (int i = 0; i < NUM_DRIVES; ++i) for (int j = 0; j < TORQUE_DELAY_CYCLES; ++j) t[i * TORQUE_DELAY_CYCLES + j] = cassie_hostenv_torque_delay(c->host)[i * TORQUE_DELAY_CYCLES + j];
}
void cassie_sim_set_torque_delay(cassie_sim_t *c, double *t)
{
    for (int i = 0; i < NUM_DRIVES; ++i) for (int j = 0; j < TORQUE_DELAY_CYCLES; ++j) cassie_hostenv_torque_delay(c->host)[i * TORQUE_DELAY_CYCLES + j] = t[i * TORQUE_DELAY_CYCLES + j];
}

/* ------------------------------------------------------------ state snapshots --- */
cassie_state_t *cassie_state_alloc(void)
{
    cassie_state_t *s = calloc(1, sizeof(cassie_state_t));
    if (!s) return NULL;
    s->host = cassie_hostenv_alloc();
    return s;
}
void cassie_state_copy(cassie_state_t *dst, const cassie_state_t *src)
{
    dst->d = src->d;
    cassie_hostenv_copy(dst->host, src->host);
}
cassie_state_t *cassie_state_duplicate(const cassie_state_t *src)
{
    cassie_state_t *s = cassie_state_alloc();
    if (s) cassie_state_copy(s, src);
    return s;
}
void cassie_state_free(cassie_state_t *s)
{
    if (!s) return;
    cassie_hostenv_free(s->host);
    free(s);
}
double *cassie_state_time(cassie_state_t *s) { return &s->d.time; }
double *cassie_state_qpos(cassie_state_t *s) { return s->d.qpos; }
double *cassie_state_qvel(cassie_state_t *s) { return s->d.qvel; }

void cassie_get_state(const cassie_sim_t *c, cassie_state_t *s)
{
    s->d = c->d;
    cassie_hostenv_copy(s->host, c->host);
}
void cassie_set_state(cassie_sim_t *c, const cassie_state_t *s)
{
    c->d = s->d;
    c->d.warmstart_dirty = 1; /* mjData carries qacc_warmstart: push it with the next step */
    cassie_hostenv_copy(c->host, s->host);
}

/* On-disk form of a cassie_state_t (the reference keeps states in memory only, :3380-3452):
 *   8 bytes magic "CASSIEST", u32 version, u32 sizeof(sim_data_t), u32 host image size, u32 reserved,
 *   the sim_data_t (time, qpos, qvel, qacc, warm start, ctrl, applied forces, sensordata, ...), the host image
 *   (cassie_out_t, encoder filters, torque delay lines, Agility block states), u64 FNV-1a checksum of everything before it.
 * Native byte order and layout: a checkpoint of this library for this library, like a memcpy of the struct would be.  The
 * Agility block states hold pointers into themselves; they are written as they are, together with the addresses the
 * blocks lived at (so a loader can rebase them) -- the file therefore contains heap addresses of the writing process.
 * A file that is shorter or longer than its header says, or whose checksum does not match, is rejected. */
#define STATE_MAGIC "CASSIEST"
#define STATE_VERSION 2u
static unsigned long long fnv1a(unsigned long long h, const void *data, size_t n)
{
    const unsigned char *p = data;
    for (size_t i = 0; i < n; ++i) h = (h ^ p[i]) * 1099511628211ull;
    return h;
}
int cassie_state_save(const cassie_state_t *s, const char *path)
{
    if (!s || !path) return -1;
    if (!cassie_hostenv_blocks_verified()) { fprintf(stderr, "cassie_state_save: Agility block sizes unverified, refusing\n"); return -1; }
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    const unsigned hdr[4] = {STATE_VERSION, (unsigned)sizeof(sim_data_t), (unsigned)cassie_hostenv_image_size(), 0u};
    void *img = malloc(hdr[2]);
    if (!img) { fclose(f); return -1; }
    cassie_hostenv_to_image(s->host, img);
    unsigned long long sum = fnv1a(1469598103934665603ull, STATE_MAGIC, 8);
    sum = fnv1a(sum, hdr, sizeof hdr); sum = fnv1a(sum, &s->d, sizeof s->d); sum = fnv1a(sum, img, hdr[2]);
    int ok = fwrite(STATE_MAGIC, 8, 1, f) == 1 && fwrite(hdr, sizeof hdr, 1, f) == 1 && fwrite(&s->d, sizeof s->d, 1, f) == 1 &&
             fwrite(img, hdr[2], 1, f) == 1 && fwrite(&sum, sizeof sum, 1, f) == 1;
    free(img);
    ok = fclose(f) == 0 && ok;
    return ok ? 0 : -1;
}
int cassie_state_load(cassie_state_t *s, const char *path)
{
    if (!s || !path) return -1;
    if (!cassie_hostenv_blocks_verified()) { fprintf(stderr, "cassie_state_load: Agility block sizes unverified, refusing\n"); return -1; }
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    char magic[8];
    unsigned hdr[4];
    int rc = -1;
    if (fread(magic, 8, 1, f) == 1 && memcmp(magic, STATE_MAGIC, 8) == 0 && fread(hdr, sizeof hdr, 1, f) == 1 &&
        hdr[0] == STATE_VERSION && hdr[1] == sizeof(sim_data_t) && hdr[2] == cassie_hostenv_image_size()) {
        sim_data_t *d = malloc(sizeof *d);
        void *img = malloc(hdr[2]);
        unsigned long long sum = 0, want = fnv1a(1469598103934665603ull, STATE_MAGIC, 8);
        char extra;
        if (d && img && fread(d, sizeof *d, 1, f) == 1 && fread(img, hdr[2], 1, f) == 1 && fread(&sum, sizeof sum, 1, f) == 1 &&
            fread(&extra, 1, 1, f) == 0 /* nothing may follow */) {
            want = fnv1a(want, hdr, sizeof hdr); want = fnv1a(want, d, sizeof *d); want = fnv1a(want, img, hdr[2]);
            if (want == sum) {
                s->d = *d;
                cassie_hostenv_from_image(s->host, img);
                rc = 0;
            }
        }
        free(img);
        free(d);
    }
    fclose(f);
    return rc;
}

/* --------------------------------------------------------------- visualisation --- */
/* Rendering (GLFW / OpenGL / ffmpeg, reference :2248-3378) is outside the hot path.  These behave like
 * the reference library when GLFW could not be loaded: no window is ever created. */
cassie_vis_t *cassie_vis_init(cassie_sim_t *c, const char *modelfile, bool offscreen) { (void)c; (void)modelfile; (void)offscreen; return NULL; }
void cassie_vis_close(cassie_vis_t *v) { (void)v; }
void cassie_vis_free(cassie_vis_t *v) { free(v); }
bool cassie_vis_draw(cassie_vis_t *v, cassie_sim_t *c) { (void)v; (void)c; return false; }
bool cassie_vis_valid(cassie_vis_t *v) { (void)v; return false; }
bool cassie_vis_paused(cassie_vis_t *v) { (void)v; return false; }
bool cassie_vis_slowmo(cassie_vis_t *v) { (void)v; return false; }
void cassie_vis_window_resize(cassie_vis_t *v, int w, int h) { (void)v; (void)w; (void)h; }
void cassie_vis_add_marker(cassie_vis_t *v, double pos[3], double size[3], double rgba[4], double so3[9]) { (void)v; (void)pos; (void)size; (void)rgba; (void)so3; }
void cassie_vis_remove_marker(cassie_vis_t *v, int id) { (void)v; (void)id; }
void cassie_vis_clear_markers(cassie_vis_t *v) { (void)v; }
void cassie_vis_update_marker_pos(cassie_vis_t *v, int id, double pos[3]) { (void)v; (void)id; (void)pos; }
void cassie_vis_update_marker_size(cassie_vis_t *v, int id, double size[3]) { (void)v; (void)id; (void)size; }
void cassie_vis_update_marker_rgba(cassie_vis_t *v, int id, double rgba[4]) { (void)v; (void)id; (void)rgba; }
void cassie_vis_update_marker_orient(cassie_vis_t *v, int id, double so3[9]) { (void)v; (void)id; (void)so3; }
void cassie_vis_apply_force(cassie_vis_t *v, double xfrc[6], const char *name) { (void)v; (void)xfrc; (void)name; }
void cassie_vis_full_reset(cassie_vis_t *v) { (void)v; }
void cassie_vis_remakeSceneCon(cassie_vis_t *v) { (void)v; }
void cassie_vis_set_hfielddata(cassie_vis_t *v, float *data) { (void)v; (void)data; }
float *cassie_vis_hfielddata(cassie_vis_t *v) { (void)v; return NULL; }
void cassie_vis_set_cam(cassie_vis_t *v, const char *body_name, double zoom, double azi, double elev) { (void)v; (void)body_name; (void)zoom; (void)azi; (void)elev; }
void cassie_vis_set_cam_pos(cassie_vis_t *v, double *look_point, double distance, double azi, double elev) { (void)v; (void)look_point; (void)distance; (void)azi; (void)elev; }
void cassie_vis_attach_cam(cassie_vis_t *v, const char *cam_name) { (void)v; (void)cam_name; }
float cassie_vis_extent(cassie_vis_t *v) { (void)v; return 0.f; }
float cassie_vis_znear(cassie_vis_t *v) { (void)v; return 0.f; }
float cassie_vis_zfar(cassie_vis_t *v) { (void)v; return 0.f; }
void cassie_vis_init_recording(cassie_vis_t *v, const char *videofile, int w, int h) { (void)v; (void)videofile; (void)w; (void)h; }
void cassie_vis_record_frame(cassie_vis_t *v) { (void)v; }
void cassie_vis_close_recording(cassie_vis_t *v) { (void)v; }
void cassie_vis_init_depth(cassie_vis_t *v, int w, int h) { (void)v; (void)w; (void)h; }
void cassie_vis_init_rgb(cassie_vis_t *v, int w, int h) { (void)v; (void)w; (void)h; }
float *cassie_vis_draw_depth(cassie_vis_t *v, cassie_sim_t *c, int w, int h) { (void)v; (void)c; (void)w; (void)h; return NULL; }
unsigned char *cassie_vis_get_rgb(cassie_vis_t *v, cassie_sim_t *c, int w, int h) { (void)v; (void)c; (void)w; (void)h; return NULL; }
int cassie_vis_get_depth_size(cassie_vis_t *v) { (void)v; return 0; }
void cassie_vis_foot_forces(const cassie_vis_t *v, double cfrc[12]) { (void)v; memset(cfrc, 0, 12 * sizeof(double)); }
