/*
 * phys_host.cpp -- host half of the inner C ABI (include/cassie_phys.h): model
 * lifecycle and the read-write model views the drop-in accessors need.
 * Replaces the reference's mj_loadXML / mj_copyModel / mj_deleteModel / mj_setConst /
 * mj_name2id / mj_id2name calls and its direct mjModel field access
 * (reference src/cassiemujoco.c:851, :1013-1016, :1110, :952, :1244).
 */
#include "cassie_phys.h"
#include "host_model.h"

#include <cstring>
#include <string>
#include <vector>

struct phys_model {
    cm::HostModel h;
};

static thread_local std::string g_err;
static void set_err(char *err, int errlen, const std::string &s) {
    g_err = s;
    if (err && errlen > 0) {
        strncpy(err, s.c_str(), errlen - 1);
        err[errlen - 1] = 0;
    }
}
void phys_set_last_error(const char *s) { g_err = s ? s : ""; }

/* 64-bit fingerprint of everything a caller can change through the read-write views below (and the options), four
 * independent multiply-xor lanes so that the words stream at memory speed: a few microseconds for a Cassie model.  The
 * single-simulator glue compares it before every step instead of recompiling the model: the reference hands out raw
 * mjModel pointers (reference src/cassiemujoco.c:1303-1584), so a write can happen at any time without a call to notice. */
namespace {
struct Fingerprint {
    unsigned long long h[4] = {0x9e3779b97f4a7c15ull, 0xc2b2ae3d27d4eb4full, 0x165667b19e3779f9ull, 0x27d4eb2f165667c5ull};
    void words(const void *data, size_t bytes) {
        const unsigned char *p = (const unsigned char *)data;
        size_t i = 0;
        for (; i + 32 <= bytes; i += 32) {
            unsigned long long w[4];
            memcpy(w, p + i, 32);
            for (int k = 0; k < 4; ++k) h[k] = (h[k] ^ w[k]) * 0x100000001b3ull + 0x632be59bd9b4e019ull;
        }
        unsigned long long tail[4] = {0, 0, 0, 0};
        if (i < bytes) { memcpy(tail, p + i, bytes - i); for (int k = 0; k < 4; ++k) h[k] = (h[k] ^ tail[k]) * 0x100000001b3ull + bytes; }
    }
    template <class T> void vec(const std::vector<T> &v) { if (!v.empty()) words(v.data(), v.size() * sizeof(T)); }
    unsigned long long value() const {
        unsigned long long x = h[0];
        for (int k = 1; k < 4; ++k) x = (x ^ (h[k] >> 29) ^ (h[k] << 35)) * 0x9fb21c651e98df25ull;
        return x ^ (x >> 32);
    }
};
}  // namespace

extern "C" {

const char *phys_last_error(void) { return g_err.c_str(); }
size_t phys_sizeof_model(void) { return sizeof(cm_model_t); }

phys_model_t *phys_model_load(const char *path, char *err, int errlen) {
    phys_model *m = new phys_model;
    std::string e;
    if (!path || !cm::load_model_file(path, &m->h, &e)) {
        set_err(err, errlen, e.empty() ? "no model path" : e);
        delete m;
        return nullptr;
    }
    return m;
}

phys_model_t *phys_model_copy(const phys_model_t *src) {
    if (!src) return nullptr;
    return new phys_model(*src);
}

void phys_model_free(phys_model_t *m) { delete m; }

int phys_model_save(const phys_model_t *m, const char *path) { return m && path && m->h.save(path) ? 0 : -1; }

void phys_model_set_const(phys_model_t *m) {
    if (m) m->h.set_const();
}

int phys_model_compile(const phys_model_t *m, cm_model_t *out, char *err, int errlen) {
    std::string e;
    if (!m || !out || !m->h.compile(out, &e)) {
        set_err(err, errlen, e.empty() ? "null model" : e);
        return -1;
    }
    return 0;
}

unsigned long long phys_model_fingerprint(const phys_model_t *m) {
    if (!m) return 0;
    const cm::HostModel &h = m->h;
    Fingerprint f;
    const double opt[] = {h.timestep, h.tolerance, h.impratio, h.gravity[0], h.gravity[1], h.gravity[2], h.magnetic[0], h.magnetic[1],
                          h.magnetic[2], (double)h.iterations, (double)h.solver_pgs, (double)h.flags, h.meaninertia,
                          h.hfield_size[0], h.hfield_size[1], h.hfield_size[2], h.hfield_size[3], (double)h.hfield_nrow, (double)h.hfield_ncol};
    f.words(opt, sizeof opt);
    f.vec(h.body_pos); f.vec(h.body_quat); f.vec(h.body_ipos); f.vec(h.body_iquat); f.vec(h.body_mass); f.vec(h.body_inertia);
    f.vec(h.body_invweight0); f.vec(h.body_subtreemass);
    f.vec(h.jnt_pos); f.vec(h.jnt_axis); f.vec(h.jnt_range); f.vec(h.jnt_stiffness); f.vec(h.jnt_margin); f.vec(h.jnt_solref); f.vec(h.jnt_solimp);
    f.vec(h.jnt_limited); f.vec(h.qpos0); f.vec(h.qpos_spring); f.vec(h.dof_armature); f.vec(h.dof_damping); f.vec(h.dof_invweight0);
    f.vec(h.geom_type); f.vec(h.geom_contype); f.vec(h.geom_conaffinity); f.vec(h.geom_condim); f.vec(h.geom_priority); f.vec(h.geom_group);
    f.vec(h.geom_pos); f.vec(h.geom_quat); f.vec(h.geom_size); f.vec(h.geom_friction); f.vec(h.geom_solref); f.vec(h.geom_solimp);
    f.vec(h.geom_solmix); f.vec(h.geom_margin); f.vec(h.geom_gap); f.vec(h.geom_rbound); f.vec(h.geom_user);
    f.vec(h.site_pos); f.vec(h.site_quat);
    /* the integer arrays phys_model_iarray hands out as raw pointers */
    f.vec(h.jnt_type); f.vec(h.jnt_qposadr); f.vec(h.jnt_dofadr); f.vec(h.geom_bodyid); f.vec(h.body_parentid); f.vec(h.body_jntadr);
    f.vec(h.body_jntnum); f.vec(h.body_dofadr); f.vec(h.body_dofnum);
    f.vec(h.eq_active); f.vec(h.eq_data); f.vec(h.eq_solref); f.vec(h.eq_solimp);
    f.vec(h.act_ctrllimited); f.vec(h.act_gear); f.vec(h.act_ctrlrange); f.vec(h.act_user);
    f.vec(h.sensor_type); f.vec(h.sensor_objid); f.vec(h.sensor_adr); f.vec(h.sensor_dim); f.vec(h.sensor_cutoff); f.vec(h.sensor_noise); f.vec(h.sensor_user);
    return f.value();
}

/* the same hash over a block of 32-bit samples (the height field, which callers also write through a raw pointer) */
unsigned long long phys_hash_floats(const float *data, size_t n) {
    Fingerprint f;
    if (data && n) f.words(data, n * sizeof(float));
    return f.value();
}

unsigned phys_model_flags(const phys_model_t *m) { return m ? m->h.flags : 0u; }
int phys_model_set_flag(phys_model_t *m, unsigned flag, int on) {
    if (!m || (flag & ~(CM_FLAG_EULERDAMP | CM_FLAG_WARMSTART | CM_FLAG_REFSAFE | CM_FLAG_HFDENSE | CM_FLAG_HFMULTI | CM_FLAG_HFPRISM | CM_FLAG_BOX8)) != 0) return -1;
    if (on) m->h.flags |= flag; else m->h.flags &= ~flag;
    return 0;
}

int phys_model_name2id(const phys_model_t *m, int objtype, const char *name) {
    return m ? m->h.name2id(objtype, name) : -1;
}
const char *phys_model_id2name(const phys_model_t *m, int objtype, int id) {
    return m ? m->h.id2name(objtype, id) : nullptr;
}

int phys_model_size(const phys_model_t *m, int what) {
    if (!m) return 0;
    const cm::HostModel &h = m->h;
    switch (what) {
        case PHYS_NQ: return h.nq;
        case PHYS_NV: return h.nv;
        case PHYS_NU: return h.nu;
        case PHYS_NBODY: return h.nbody;
        case PHYS_NJNT: return h.njnt;
        case PHYS_NGEOM: return h.ngeom;
        case PHYS_NSITE: return h.nsite;
        case PHYS_NSENSOR: return h.nsensor;
        case PHYS_NSENSORDATA: return h.nsensordata;
        case PHYS_NEQ: return h.neq;
        case PHYS_NHFIELDDATA: return h.nhfielddata;
        case PHYS_HFIELD_NROW: return h.hfield_nrow;
        case PHYS_HFIELD_NCOL: return h.hfield_ncol;
        case PHYS_NUSER_SENSOR: return h.nuser_sensor;
        case PHYS_NUSER_ACTUATOR: return h.nuser_actuator;
        case PHYS_NUSER_GEOM: return h.nuser_geom;
        case PHYS_NCAM: return h.ncam;
    }
    return 0;
}

double *phys_model_array(phys_model_t *m, int which) {
    if (!m) return nullptr;
    cm::HostModel &h = m->h;
    switch (which) {
        case PHYS_M_BODY_MASS: return h.body_mass.data();
        case PHYS_M_BODY_IPOS: return h.body_ipos.data();
        case PHYS_M_BODY_INERTIA: return h.body_inertia.data();
        case PHYS_M_BODY_POS: return h.body_pos.data();
        case PHYS_M_BODY_QUAT: return h.body_quat.data();
        case PHYS_M_DOF_DAMPING: return h.dof_damping.data();
        case PHYS_M_JNT_STIFFNESS: return h.jnt_stiffness.data();
        case PHYS_M_QPOS_SPRING: return h.qpos_spring.data();
        case PHYS_M_QPOS0: return h.qpos0.data();
        case PHYS_M_JNT_RANGE: return h.jnt_range.data();
        case PHYS_M_GEOM_POS: return h.geom_pos.data();
        case PHYS_M_GEOM_QUAT: return h.geom_quat.data();
        case PHYS_M_GEOM_SIZE: return h.geom_size.data();
        case PHYS_M_GEOM_FRICTION: return h.geom_friction.data();
        case PHYS_M_GEOM_USER: return h.geom_user.data();
        case PHYS_M_ACTUATOR_GEAR: return h.act_gear.data();
        case PHYS_M_ACTUATOR_CTRLRANGE: return h.act_ctrlrange.data();
        case PHYS_M_ACTUATOR_USER: return h.act_user.data();
        case PHYS_M_SENSOR_USER: return h.sensor_user.data();
        case PHYS_M_HFIELD_SIZE: return h.hfield_size;
        case PHYS_M_TIMESTEP: return &h.timestep;
        case PHYS_M_STAT_CENTER: return h.stat_center;
        case PHYS_M_STAT_EXTENT: return &h.stat_extent;
    }
    return nullptr;
}

float *phys_model_geom_rgba(phys_model_t *m) { return m ? m->h.geom_rgba.data() : nullptr; }
float *phys_model_hfield_data(phys_model_t *m) { return m && !m->h.hfield_data.empty() ? m->h.hfield_data.data() : nullptr; }

int *phys_model_iarray(phys_model_t *m, int which) {
    if (!m) return nullptr;
    cm::HostModel &h = m->h;
    switch (which) {
        case PHYS_MI_JNT_TYPE: return h.jnt_type.data();
        case PHYS_MI_JNT_QPOSADR: return h.jnt_qposadr.data();
        case PHYS_MI_JNT_DOFADR: return h.jnt_dofadr.data();
        case PHYS_MI_GEOM_BODYID: return h.geom_bodyid.data();
        case PHYS_MI_GEOM_GROUP: return h.geom_group.data();
        case PHYS_MI_GEOM_TYPE: return h.geom_type.data();
        case PHYS_MI_SENSOR_OBJID: return h.sensor_objid.data();
        case PHYS_MI_SENSOR_TYPE: return h.sensor_type.data();
        case PHYS_MI_SENSOR_ADR: return h.sensor_adr.data();
        case PHYS_MI_SENSOR_DIM: return h.sensor_dim.data();
        case PHYS_MI_BODY_PARENTID: return h.body_parentid.data();
        case PHYS_MI_BODY_JNTADR: return h.body_jntadr.data();
        case PHYS_MI_BODY_JNTNUM: return h.body_jntnum.data();
        case PHYS_MI_BODY_DOFADR: return h.body_dofadr.data();
        case PHYS_MI_BODY_DOFNUM: return h.body_dofnum.data();
    }
    return nullptr;
}

}  // extern "C"
