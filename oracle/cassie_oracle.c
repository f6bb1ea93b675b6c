/*
 * cassie_oracle.c -- TEST INFRASTRUCTURE (see cassie_oracle.h): scalar fp64 CPU
 * restatement of the physics step behind reference src/cassiemujoco.c:1130-1134
 * (mj_step1_fp / mj_step2_fp -> MuJoCo 2.1.0, not in tree).  PARITY UNPINNED.
 *
 * Stage names follow SURVEY.md 8(a) rows P1..P12 / App. B.  The code is written
 * for readability and a fixed, documented operation order; no attempt is made
 * to be fast (bench.py times it only as the "port" CPU baseline).
 */
#include "cassie_oracle.h"

#include <math.h>
#include <string.h>

#define NV_ CM_MAXV

static const float *g_hfield = 0;
void co_set_hfield(const float *data) { g_hfield = data; }
/* contacts one geom pair can report: two (four with CM_FLAG_HFMULTI; box-box four), or -- study builds only -- one per grid
 * triangle under a capsule */
#ifdef CO_STUDY
#define CO_RC_MAX 96
/* ORACLE-ONLY STUDY MODES (tests/collision_fidelity_study.py; built with -DCO_STUDY and raised CM_MAXCON / CM_MAXEFC; never part of
 * the parity oracle): what the collision definitions of DESIGN.md 4.2 cost in fidelity against MuJoCo-shaped contact sets.
 *   hfield mode 1: one contact per PENETRATED GRID TRIANGLE under a sphere / capsule -- MuJoCo builds one prism per grid
 *                  triangle and reports one contact per penetrated prism (why reference model/cassie_hfield.xml:4 asks for
 *                  nconmax = 300); here: per triangle the deepest of the capsule's sample spheres (samples 2.5 mm apart)
 *   box mode 8   : box-box keeps up to eight clipped-face candidates (MuJoCo's mjc_BoxBox returns up to eight) instead of four */
static int g_study_hfield_mode = 0, g_study_box_keep = 4;
void co_study_set_hfield_mode(int mode) { g_study_hfield_mode = mode; }
void co_study_set_box_contacts(int n) { g_study_box_keep = n < 1 ? 1 : (n > 8 ? 8 : n); }
#else
#define CO_RC_MAX 8
#endif
unsigned long co_sizeof_data(void) { return sizeof(co_data_t); }

/* ------------------------------------------------------------ small math --- */
static double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void cross3(double *r, const double *a, const double *b) {
    double t0 = a[1] * b[2] - a[2] * b[1], t1 = a[2] * b[0] - a[0] * b[2], t2 = a[0] * b[1] - a[1] * b[0];
    r[0] = t0; r[1] = t1; r[2] = t2;
}
static double norm3(const double *a) { return sqrt(dot3(a, a)); }
static double normalize3(double *a) {
    double n = norm3(a);
    if (n < CM_MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; }
    else { double s = 1 / n; a[0] *= s; a[1] *= s; a[2] *= s; }
    return n;
}
static void normalize4(double *q) {
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n < CM_MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
    else { double s = 1 / n; q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s; }
}
static void mulquat(double *r, const double *a, const double *b) {
    double t0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    double t1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    double t2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    double t3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    r[0] = t0; r[1] = t1; r[2] = t2; r[3] = t3;
}
static void quat2mat(double *m, const double *q) {
    double q00 = q[0] * q[0], q11 = q[1] * q[1], q22 = q[2] * q[2], q33 = q[3] * q[3];
    double q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
    double q12 = q[1] * q[2], q13 = q[1] * q[3], q23 = q[2] * q[3];
    m[0] = q00 + q11 - q22 - q33; m[1] = 2 * (q12 - q03);       m[2] = 2 * (q13 + q02);
    m[3] = 2 * (q12 + q03);       m[4] = q00 - q11 + q22 - q33; m[5] = 2 * (q23 - q01);
    m[6] = 2 * (q13 - q02);       m[7] = 2 * (q23 + q01);       m[8] = q00 - q11 - q22 + q33;
}
static void mulmatvec3(double *r, const double *m, const double *v) {
    double t0 = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
    double t1 = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
    double t2 = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
    r[0] = t0; r[1] = t1; r[2] = t2;
}
static void mulmatTvec3(double *r, const double *m, const double *v) {
    double t0 = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
    double t1 = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
    double t2 = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
    r[0] = t0; r[1] = t1; r[2] = t2;
}
static void mulmat3(double *r, const double *a, const double *b) {
    double t[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
    memcpy(r, t, sizeof t);
}
static void rotvecquat(double *r, const double *v, const double *q) {
    double m[9];
    quat2mat(m, q);
    mulmatvec3(r, m, v);
}
static void axisangle2quat(double *q, const double *axis, double angle) {
    if (angle == 0) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
    double s = sin(angle * 0.5);
    q[0] = cos(angle * 0.5); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}

/* spatial algebra in MuJoCo's convention: [rotational(3); translational(3)] */
static void cross_motion(double *r, const double *vel, const double *v) {
    double a[3], b[3], c[3];
    cross3(a, vel, v);
    cross3(b, vel, v + 3);
    cross3(c, vel + 3, v);
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2];
    r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
static void cross_force(double *r, const double *vel, const double *f) {
    double a[3], b[3], c[3];
    cross3(a, vel, f);
    cross3(b, vel + 3, f + 3);
    cross3(c, vel, f + 3);
    r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2];
    r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}
/* 10-number spatial inertia [Ixx Iyy Izz Ixy Ixz Iyz  m*dx m*dy m*dz  m] times motion vector */
static void mul_inert_vec(double *r, const double *I, const double *v) {
    r[0] = I[0] * v[0] + I[3] * v[1] + I[4] * v[2] - I[8] * v[4] + I[7] * v[5];
    r[1] = I[3] * v[0] + I[1] * v[1] + I[5] * v[2] + I[8] * v[3] - I[6] * v[5];
    r[2] = I[4] * v[0] + I[5] * v[1] + I[2] * v[2] - I[7] * v[3] + I[6] * v[4];
    r[3] = I[8] * v[1] - I[7] * v[2] + I[9] * v[3];
    r[4] = I[6] * v[2] - I[8] * v[0] + I[9] * v[4];
    r[5] = I[7] * v[0] - I[6] * v[1] + I[9] * v[5];
}

/* ---------------------------------------------------------------- reset --- */
void co_reset(const cm_model_t *m, co_data_t *d) {
    memset(d, 0, sizeof *d);
    for (int i = 0; i < m->nq; ++i) d->qpos[i] = m->qpos0[i];
}

/* ---------------------------------------------------- P1: kinematics ------ */
void co_kinematics(const cm_model_t *m, co_data_t *d) {
    /* world */
    memset(d->xpos[0], 0, sizeof d->xpos[0]);
    d->xquat[0][0] = 1; d->xquat[0][1] = d->xquat[0][2] = d->xquat[0][3] = 0;
    quat2mat(d->xmat[0], d->xquat[0]);
    memset(d->xipos[0], 0, sizeof d->xipos[0]);
    quat2mat(d->ximat[0], d->xquat[0]);

    for (int b = 1; b < m->nbody; ++b) {
        int p = m->body_parentid[b];
        double pos[3], quat[4];
        int j0 = m->body_jntadr[b], jn = m->body_jntnum[b];
        if (jn == 1 && m->jnt_type[j0] == CM_JNT_FREE) {
            int qa = m->jnt_qposadr[j0];
            for (int i = 0; i < 3; ++i) pos[i] = d->qpos[qa + i];
            for (int i = 0; i < 4; ++i) quat[i] = d->qpos[qa + 3 + i];
            normalize4(quat);
            for (int i = 0; i < 3; ++i) { d->xanchor[j0][i] = pos[i]; d->xaxis[j0][i] = (i == 2); }
        } else {
            mulmatvec3(pos, d->xmat[p], m->body_pos[b]);
            for (int i = 0; i < 3; ++i) pos[i] += d->xpos[p][i];
            mulquat(quat, d->xquat[p], m->body_quat[b]);
            for (int jj = 0; jj < jn; ++jj) {
                int j = j0 + jj, qa = m->jnt_qposadr[j];
                double anchor[3], axis[3];
                rotvecquat(anchor, m->jnt_pos[j], quat);
                for (int i = 0; i < 3; ++i) anchor[i] += pos[i];
                rotvecquat(axis, m->jnt_axis[j], quat);
                for (int i = 0; i < 3; ++i) { d->xanchor[j][i] = anchor[i]; d->xaxis[j][i] = axis[i]; }
                if (m->jnt_type[j] == CM_JNT_SLIDE) {
                    double s = d->qpos[qa] - m->qpos0[qa];
                    for (int i = 0; i < 3; ++i) pos[i] += axis[i] * s;
                } else {
                    double ql[4];
                    if (m->jnt_type[j] == CM_JNT_BALL) {
                        for (int i = 0; i < 4; ++i) ql[i] = d->qpos[qa + i];
                        normalize4(ql);
                    } else {
                        axisangle2quat(ql, m->jnt_axis[j], d->qpos[qa] - m->qpos0[qa]);
                    }
                    mulquat(quat, quat, ql);
                    /* keep the anchor fixed: off-centre rotation */
                    double r[3];
                    rotvecquat(r, m->jnt_pos[j], quat);
                    for (int i = 0; i < 3; ++i) pos[i] = anchor[i] - r[i];
                }
            }
        }
        normalize4(quat);
        for (int i = 0; i < 3; ++i) d->xpos[b][i] = pos[i];
        for (int i = 0; i < 4; ++i) d->xquat[b][i] = quat[i];
        quat2mat(d->xmat[b], quat);
        mulmatvec3(d->xipos[b], d->xmat[b], m->body_ipos[b]);
        for (int i = 0; i < 3; ++i) d->xipos[b][i] += pos[i];
        double qi[4];
        mulquat(qi, quat, m->body_iquat[b]);
        quat2mat(d->ximat[b], qi);
    }
    for (int g = 0; g < m->ngeom; ++g) {
        int b = m->geom_bodyid[g];
        mulmatvec3(d->geom_xpos[g], d->xmat[b], m->geom_pos[g]);
        for (int i = 0; i < 3; ++i) d->geom_xpos[g][i] += d->xpos[b][i];
        double q[4];
        mulquat(q, d->xquat[b], m->geom_quat[g]);
        quat2mat(d->geom_xmat[g], q);
    }
    for (int s = 0; s < m->nsite; ++s) {
        int b = m->site_bodyid[s];
        mulmatvec3(d->site_xpos[s], d->xmat[b], m->site_pos[s]);
        for (int i = 0; i < 3; ++i) d->site_xpos[s][i] += d->xpos[b][i];
        double q[4];
        mulquat(q, d->xquat[b], m->site_quat[s]);
        quat2mat(d->site_xmat[s], q);
    }
}

/* ------------------------------------------------- P1b: com-based frame --- */
void co_com_pos(const cm_model_t *m, co_data_t *d) {
    double mass[CM_MAXBODY];
    for (int b = 0; b < m->nbody; ++b) {
        mass[b] = m->body_mass[b];
        for (int i = 0; i < 3; ++i) d->subtree_com[b][i] = m->body_mass[b] * d->xipos[b][i];
    }
    for (int b = m->nbody - 1; b > 0; --b) {
        int p = m->body_parentid[b];
        mass[p] += mass[b];
        for (int i = 0; i < 3; ++i) d->subtree_com[p][i] += d->subtree_com[b][i];
    }
    for (int b = 0; b < m->nbody; ++b) {
        if (mass[b] < CM_MINVAL) for (int i = 0; i < 3; ++i) d->subtree_com[b][i] = d->xipos[b][i];
        else for (int i = 0; i < 3; ++i) d->subtree_com[b][i] /= mass[b];
    }
    /* body inertia about the com of its kinematic tree, world orientation */
    memset(d->cinert[0], 0, sizeof d->cinert[0]);
    for (int b = 1; b < m->nbody; ++b) {
        const double *R = d->ximat[b], *I = m->body_inertia[b];
        const double *c = d->subtree_com[m->body_rootid[b]];
        double dif[3] = {d->xipos[b][0] - c[0], d->xipos[b][1] - c[1], d->xipos[b][2] - c[2]};
        double mm = m->body_mass[b], d2 = dot3(dif, dif);
        double W[9]; /* R diag(I) R^T */
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                W[3 * i + j] = R[3 * i] * I[0] * R[3 * j] + R[3 * i + 1] * I[1] * R[3 * j + 1] + R[3 * i + 2] * I[2] * R[3 * j + 2];
        double *ci = d->cinert[b];
        ci[0] = W[0] + mm * (d2 - dif[0] * dif[0]);
        ci[1] = W[4] + mm * (d2 - dif[1] * dif[1]);
        ci[2] = W[8] + mm * (d2 - dif[2] * dif[2]);
        ci[3] = W[1] - mm * dif[0] * dif[1];
        ci[4] = W[2] - mm * dif[0] * dif[2];
        ci[5] = W[5] - mm * dif[1] * dif[2];
        ci[6] = mm * dif[0]; ci[7] = mm * dif[1]; ci[8] = mm * dif[2];
        ci[9] = mm;
    }
    /* motion axis of every dof, expressed at the tree's com */
    for (int j = 0; j < m->njnt; ++j) {
        int b = m->jnt_bodyid[j], da = m->jnt_dofadr[j];
        const double *c = d->subtree_com[m->body_rootid[b]];
        double off[3] = {c[0] - d->xanchor[j][0], c[1] - d->xanchor[j][1], c[2] - d->xanchor[j][2]};
        int rot0 = da;
        switch (m->jnt_type[j]) {
            case CM_JNT_SLIDE:
                for (int i = 0; i < 3; ++i) { d->cdof[da][i] = 0; d->cdof[da][3 + i] = d->xaxis[j][i]; }
                break;
            case CM_JNT_HINGE:
                for (int i = 0; i < 3; ++i) d->cdof[da][i] = d->xaxis[j][i];
                cross3(d->cdof[da] + 3, d->xaxis[j], off);
                break;
            case CM_JNT_FREE:
                for (int k = 0; k < 3; ++k)
                    for (int i = 0; i < 3; ++i) { d->cdof[da + k][i] = 0; d->cdof[da + k][3 + i] = (i == k); }
                rot0 = da + 3;
                /* fall through: rotational dofs like a ball joint */
            case CM_JNT_BALL:
                for (int k = 0; k < 3; ++k) {
                    double ax[3] = {d->xmat[b][k], d->xmat[b][3 + k], d->xmat[b][6 + k]};
                    for (int i = 0; i < 3; ++i) d->cdof[rot0 + k][i] = ax[i];
                    cross3(d->cdof[rot0 + k] + 3, ax, off);
                }
                break;
        }
    }
}

/* ------------------------------------------------------------ P2: CRBA ---- */
void co_crb(const cm_model_t *m, co_data_t *d) {
    int nv = m->nv;
    for (int b = 0; b < m->nbody; ++b) memcpy(d->crb[b], d->cinert[b], sizeof d->crb[b]);
    for (int b = m->nbody - 1; b > 0; --b) {
        int p = m->body_parentid[b];
        if (p > 0) for (int i = 0; i < 10; ++i) d->crb[p][i] += d->crb[b][i];
    }
    for (int i = 0; i < nv; ++i) for (int j = 0; j < nv; ++j) d->qM[i][j] = 0;
    for (int i = 0; i < nv; ++i) {
        double buf[6];
        mul_inert_vec(buf, d->crb[m->dof_bodyid[i]], d->cdof[i]);
        for (int j = i; j >= 0; j = m->dof_parentid[j]) {
            double v = 0;
            for (int k = 0; k < 6; ++k) v += d->cdof[j][k] * buf[k];
            d->qM[i][j] = v;
            d->qM[j][i] = v;
        }
        d->qM[i][i] += m->dof_armature[i];
    }
}

/* ------------------------------------------------ P3: M = L^T D L --------- */
static void factor_ld(const cm_model_t *m, double A[NV_][NV_]) {
    /* in place on the lower triangle, exploiting the tree sparsity through dof_parentid */
    for (int k = m->nv - 1; k >= 0; --k) {
        double Dk = A[k][k];
        for (int i = m->dof_parentid[k]; i >= 0; i = m->dof_parentid[i]) {
            double t = A[k][i] / Dk;
            for (int j = i; j >= 0; j = m->dof_parentid[j]) A[i][j] -= t * A[k][j];
            A[k][i] = t;
        }
    }
}
void co_factor_m(const cm_model_t *m, co_data_t *d) {
    for (int i = 0; i < m->nv; ++i) for (int j = 0; j < m->nv; ++j) d->qLD[i][j] = j <= i ? d->qM[i][j] : 0.0;
    factor_ld(m, d->qLD);
}
void co_solve_m(const cm_model_t *m, const double LD[NV_][NV_], double *x) {
    for (int k = m->nv - 1; k >= 0; --k) /* x <- L^-T x */
        for (int i = m->dof_parentid[k]; i >= 0; i = m->dof_parentid[i]) x[i] -= LD[k][i] * x[k];
    for (int k = 0; k < m->nv; ++k) x[k] /= LD[k][k];
    for (int k = 0; k < m->nv; ++k) /* x <- L^-1 x */
        for (int i = m->dof_parentid[k]; i >= 0; i = m->dof_parentid[i]) x[k] -= LD[k][i] * x[i];
}

/* Jacobian of a world point rigidly attached to a body */
void co_jac(const cm_model_t *m, const co_data_t *d, int body, const double point[3],
            double jacp[3][NV_], double jacr[3][NV_]) {
    for (int i = 0; i < 3; ++i) for (int k = 0; k < m->nv; ++k) { jacp[i][k] = 0; jacr[i][k] = 0; }
    if (body <= 0) return;
    const double *c = d->subtree_com[m->body_rootid[body]];
    double off[3] = {point[0] - c[0], point[1] - c[1], point[2] - c[2]};
    for (int k = 0; k < m->nv; ++k) {
        if (!((m->body_dofmask[body] >> k) & 1ull)) continue;
        double t[3];
        cross3(t, d->cdof[k], off);
        for (int i = 0; i < 3; ++i) { jacp[i][k] = d->cdof[k][3 + i] + t[i]; jacr[i][k] = d->cdof[k][i]; }
    }
}

/* -------------------------------------------------- P4: collision --------- */
typedef struct { double dist, pos[3], normal[3], tangent[3]; } raw_contact_t;

static int plane_sphere(raw_contact_t *c, const double *ppos, const double *pmat, const double *spos, double r, double margin) {
    double n[3] = {pmat[2], pmat[5], pmat[8]};
    double dif[3] = {spos[0] - ppos[0], spos[1] - ppos[1], spos[2] - ppos[2]};
    double dist = dot3(dif, n) - r;
    if (dist > margin) return 0;
    c->dist = dist;
    for (int i = 0; i < 3; ++i) {
        c->normal[i] = n[i];
        c->pos[i] = spos[i] - n[i] * (r + 0.5 * dist);
        c->tangent[i] = 0;
    }
    return 1;
}
static int plane_capsule(raw_contact_t *c, const double *ppos, const double *pmat, const double *cpos, const double *cmat,
                         const double *size, double margin) {
    double axis[3] = {cmat[2], cmat[5], cmat[8]};
    int n = 0;
    for (int s = 0; s < 2; ++s) {
        double sgn = s == 0 ? 1.0 : -1.0;
        double e[3] = {cpos[0] + sgn * size[1] * axis[0], cpos[1] + sgn * size[1] * axis[1], cpos[2] + sgn * size[1] * axis[2]};
        if (plane_sphere(c + n, ppos, pmat, e, size[0], margin)) {
            for (int i = 0; i < 3; ++i) c[n].tangent[i] = axis[i]; /* contact frame aligned with the capsule axis */
            ++n;
        }
    }
    return n;
}
static int sphere_sphere(raw_contact_t *c, const double *p1, double r1, const double *p2, double r2, double margin) {
    double dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
    double cd = norm3(dif);
    double dist = cd - r1 - r2;
    if (dist > margin) return 0;
    double n[3];
    if (cd < CM_MINVAL) { n[0] = 1; n[1] = 0; n[2] = 0; }
    else { n[0] = dif[0] / cd; n[1] = dif[1] / cd; n[2] = dif[2] / cd; }
    c->dist = dist;
    for (int i = 0; i < 3; ++i) { c->normal[i] = n[i]; c->pos[i] = p1[i] + n[i] * (r1 + 0.5 * dist); c->tangent[i] = 0; }
    return 1;
}
static double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
/* closest points between segments p1 +- a1*l1 and p2 +- a2*l2 (unit axes); returns parameters */
static void segment_closest(const double *p1, const double *a1, double l1, const double *p2, const double *a2, double l2,
                            double *x1, double *x2) {
    double dif[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    double mb = -dot3(a1, a2), u = -dot3(a1, dif), v = dot3(a2, dif);
    double det = 1.0 - mb * mb;
    double t1, t2;
    if (fabs(det) >= 1e-12) {
        t1 = (u - mb * v) / det;
        t2 = (v - mb * u) / det;
        if (t1 > l1) { t1 = l1; t2 = v - mb * t1; }
        else if (t1 < -l1) { t1 = -l1; t2 = v - mb * t1; }
        if (t2 > l2) { t2 = l2; t1 = clampd(u - mb * t2, -l1, l1); }
        else if (t2 < -l2) { t2 = -l2; t1 = clampd(u - mb * t2, -l1, l1); }
    } else {
        /* parallel axes: centre the contact in the overlap interval */
        double s = -mb; /* +1 or -1: relative direction */
        double c2 = v;  /* projection of p1 on axis 2, in segment-2 coordinates */
        double lo = fmax(-l2, c2 - l1), hi = fmin(l2, c2 + l1);
        if (lo <= hi) t2 = 0.5 * (lo + hi);
        else t2 = clampd(c2, -l2, l2);
        t1 = clampd((t2 - c2) * (s >= 0 ? 1.0 : -1.0), -l1, l1);
    }
    *x1 = t1; *x2 = t2;
}
static int capsule_capsule(raw_contact_t *c, const double *p1, const double *m1, const double *s1, const double *p2,
                           const double *m2, const double *s2, double margin) {
    double a1[3] = {m1[2], m1[5], m1[8]}, a2[3] = {m2[2], m2[5], m2[8]};
    double x1, x2;
    segment_closest(p1, a1, s1[1], p2, a2, s2[1], &x1, &x2);
    double q1[3] = {p1[0] + a1[0] * x1, p1[1] + a1[1] * x1, p1[2] + a1[2] * x1};
    double q2[3] = {p2[0] + a2[0] * x2, p2[1] + a2[1] * x2, p2[2] + a2[2] * x2};
    return sphere_sphere(c, q1, s1[0], q2, s2[0], margin);
}
static int sphere_capsule(raw_contact_t *c, const double *p1, double r1, const double *p2, const double *m2, const double *s2,
                          double margin) {
    double a2[3] = {m2[2], m2[5], m2[8]};
    double dif[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    double x = clampd(dot3(a2, dif), -s2[1], s2[1]);
    double q2[3] = {p2[0] + a2[0] * x, p2[1] + a2[1] * x, p2[2] + a2[2] * x};
    return sphere_sphere(c, p1, r1, q2, s2[0], margin);
}


/* ---- boxes (this repository's own definition of these pair types; MuJoCo's routines differ in detail) ---- */
/* signed distance from point q to a box (centre pb, frame mb, half sizes sb); outward unit normal at the
 * closest surface point, in world axes */
static double point_box(const double *q, const double *pb, const double *mb, const double *sb, double *nworld) {
    double d[3] = {q[0] - pb[0], q[1] - pb[1], q[2] - pb[2]}, loc[3], cl[3];
    mulmatTvec3(loc, mb, d);
    int inside = 1;
    for (int k = 0; k < 3; ++k) { cl[k] = clampd(loc[k], -sb[k], sb[k]); if (cl[k] != loc[k]) inside = 0; }
    double nl[3] = {0, 0, 0}, dist;
    if (!inside) {
        double dif[3] = {loc[0] - cl[0], loc[1] - cl[1], loc[2] - cl[2]};
        dist = norm3(dif);
        for (int k = 0; k < 3; ++k) nl[k] = dif[k] / dist;
    } else {
        int kb = 0;
        double best = sb[0] - fabs(loc[0]);
        for (int k = 1; k < 3; ++k) { double dk = sb[k] - fabs(loc[k]); if (dk < best) { best = dk; kb = k; } }
        nl[kb] = loc[kb] >= 0 ? 1.0 : -1.0;
        dist = -best;
    }
    mulmatvec3(nworld, mb, nl);
    return dist;
}
static int sphere_box(raw_contact_t *c, const double *ps, double r, const double *pb, const double *mb, const double *sb, double margin) {
    double nw[3];
    double dist = point_box(ps, pb, mb, sb, nw) - r;
    if (dist > margin) return 0;
    c->dist = dist;
    for (int i = 0; i < 3; ++i) { c->normal[i] = -nw[i]; c->pos[i] = ps[i] - nw[i] * (r + 0.5 * dist); c->tangent[i] = 0; }
    return 1;
}
/* capsule vs box: the point of the capsule axis closest to the box (golden-section search on the convex
 * distance function, fixed 32 iterations) and, if it also touches, the far end cap: at most 2 contacts */
static int capsule_box(raw_contact_t *c, const double *pc, const double *mc, const double *sc, const double *pb, const double *mb,
                       const double *sb, double margin) {
    const double ax[3] = {mc[2], mc[5], mc[8]}, h = sc[1], gr = 0.6180339887498949;
    double lo = -h, hi = h, nw[3];
    double t1 = hi - gr * (hi - lo), t2 = lo + gr * (hi - lo);
    double q1[3] = {pc[0] + ax[0] * t1, pc[1] + ax[1] * t1, pc[2] + ax[2] * t1}, q2[3] = {pc[0] + ax[0] * t2, pc[1] + ax[1] * t2, pc[2] + ax[2] * t2};
    double f1 = point_box(q1, pb, mb, sb, nw), f2 = point_box(q2, pb, mb, sb, nw);
    for (int it = 0; it < 32; ++it) {
        if (f1 <= f2) { hi = t2; t2 = t1; f2 = f1; t1 = hi - gr * (hi - lo); for (int i = 0; i < 3; ++i) q1[i] = pc[i] + ax[i] * t1; f1 = point_box(q1, pb, mb, sb, nw); }
        else { lo = t1; t1 = t2; f1 = f2; t2 = lo + gr * (hi - lo); for (int i = 0; i < 3; ++i) q2[i] = pc[i] + ax[i] * t2; f2 = point_box(q2, pb, mb, sb, nw); }
    }
    double ts = 0.5 * (lo + hi);
    /* snap to an end cap when the minimum sits there */
    if (ts > h - 1e-9 * (1 + h)) ts = h;
    if (ts < -h + 1e-9 * (1 + h)) ts = -h;
    double tt[2] = {ts, ts >= 0 ? -h : h};
    int n = 0;
    for (int k = 0; k < 2; ++k) {
        if (k == 1 && fabs(tt[1] - tt[0]) < 1e-6 + 1e-3 * h) break;
        double q[3] = {pc[0] + ax[0] * tt[k], pc[1] + ax[1] * tt[k], pc[2] + ax[2] * tt[k]};
        if (sphere_box(c + n, q, sc[0], pb, mb, sb, margin)) { for (int i = 0; i < 3; ++i) c[n].tangent[i] = ax[i]; ++n; }
    }
    return n;
}
/* plane vs box: corners below the box centre, in corner order, at most 4 (same rule as MuJoCo's plane-box) */
static int plane_box(raw_contact_t *c, const double *pp, const double *mp, const double *pb, const double *mb, const double *sb, double margin) {
    double nrm[3] = {mp[2], mp[5], mp[8]}, dif[3] = {pb[0] - pp[0], pb[1] - pp[1], pb[2] - pp[2]};
    double dist = dot3(dif, nrm);
    int n = 0;
    for (int i = 0; i < 8 && n < 4; ++i) {
        double v[3] = {(i & 1) ? sb[0] : -sb[0], (i & 2) ? sb[1] : -sb[1], (i & 4) ? sb[2] : -sb[2]}, corner[3];
        mulmatvec3(corner, mb, v);
        double ld = dot3(nrm, corner);
        if (dist + ld > margin || ld > 0) continue;
        c[n].dist = dist + ld;
        for (int k = 0; k < 3; ++k) { c[n].normal[k] = nrm[k]; c[n].tangent[k] = 0; c[n].pos[k] = corner[k] + pb[k] - nrm[k] * 0.5 * c[n].dist; }
        ++n;
    }
    return n;
}
/* box vs box: separating-axis test over the 15 candidate axes (3 + 3 face normals, 9 edge x edge), then
 *   face axis : the incident face of the other box is clipped against the reference face's rectangle; the vertices of
 *               the clipped polygon -- incident vertices inside the rectangle (candidates 0-3), rectangle corners inside
 *               the projected incident face (4-7), crossings of incident edges with the rectangle's edge lines (8-23) --
 *               that lie below the reference face are contacts, at most the 4 deepest, reported in candidate order;
 *   edge axis : one contact midway between the closest points of the two edges.
 * Face axes win ties (A's before B's), an edge axis must beat them by 5 %.  MuJoCo's own routine (a port of the same
 * idea, engine_collision_box.c) differs in its tie-breaking and in how it culls to 4 points; the contact sets agree for
 * resting and edge-crossing configurations but are not bit-compatible. */
typedef struct { double u, v, w; int valid; } bb_cand_t;
#define BB_TIE 1e-10
static int box_box_keep(raw_contact_t *c, int keep, const double *p1, const double *m1, const double *s1, const double *p2, const double *m2, const double *s2,
                   double margin) {
    double A[3][3], B[3][3], d[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, ta[3], tb[3], C[3][3], Q[3][3];
    for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) { A[i][k] = m1[3 * k + i]; B[i][k] = m2[3 * k + i]; }
    for (int i = 0; i < 3; ++i) { ta[i] = dot3(d, A[i]); tb[i] = dot3(d, B[i]); }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { C[i][j] = dot3(A[i], B[j]); Q[i][j] = fabs(C[i][j]); }
    double score[15];
    int best = -1;
    for (int k = 0; k < 15; ++k) {
        double sep, sc;
        if (k < 3) {
            sep = fabs(ta[k]) - (s1[k] + (s2[0] * Q[k][0] + s2[1] * Q[k][1] + s2[2] * Q[k][2]));
            sc = sep;
        } else if (k < 6) {
            const int j = k - 3;
            sep = fabs(tb[j]) - (s2[j] + (s1[0] * Q[0][j] + s1[1] * Q[1][j] + s1[2] * Q[2][j]));
            sc = sep - BB_TIE;
        } else {
            const int i = (k - 6) / 3, j = (k - 6) % 3, i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            const double len2 = 1.0 - C[i][j] * C[i][j];
            if (len2 < 1e-6) { score[k] = -1e300; continue; }        /* (nearly) parallel edges: no axis */
            const double proj = ta[i2] * C[i1][j] - ta[i1] * C[i2][j];
            const double ra = s1[i1] * Q[i2][j] + s1[i2] * Q[i1][j], rb = s2[j1] * Q[i][j2] + s2[j2] * Q[i][j1];
            sep = (fabs(proj) - (ra + rb)) / sqrt(len2);
            sc = (sep < 0 ? 1.05 * sep : sep) - 2 * BB_TIE;
        }
        if (sep > margin) return 0;                                  /* a separating axis */
        score[k] = sc;
    }
    for (int k = 0; k < 15; ++k) if (best < 0 || score[k] > score[best]) best = k;

    if (best >= 6) {
        const int i = (best - 6) / 3, j = (best - 6) % 3;
        double n[3], pa[3] = {p1[0], p1[1], p1[2]}, pb[3] = {p2[0], p2[1], p2[2]};
        cross3(n, A[i], B[j]);
        normalize3(n);
        if (dot3(n, d) < 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
        for (int k = 0; k < 3; ++k) {
            if (k != i) { const double sg = dot3(n, A[k]) > 0 ? 1.0 : -1.0; for (int x = 0; x < 3; ++x) pa[x] += sg * s1[k] * A[k][x]; }
            if (k != j) { const double sg = dot3(n, B[k]) > 0 ? -1.0 : 1.0; for (int x = 0; x < 3; ++x) pb[x] += sg * s2[k] * B[k][x]; }
        }
        const double ab[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
        const double uaub = C[i][j], q1 = dot3(A[i], ab), q2 = -dot3(B[j], ab), den = 1.0 - uaub * uaub;
        const double al = clampd((q1 + uaub * q2) / den, -s1[i], s1[i]), be = clampd((uaub * q1 + q2) / den, -s2[j], s2[j]);
        double ca[3], cb[3];
        for (int x = 0; x < 3; ++x) { ca[x] = pa[x] + al * A[i][x]; cb[x] = pb[x] + be * B[j][x]; }
        const double cd[3] = {cb[0] - ca[0], cb[1] - ca[1], cb[2] - ca[2]};
        c[0].dist = dot3(cd, n);
        if (c[0].dist > margin) return 0;
        for (int x = 0; x < 3; ++x) { c[0].normal[x] = n[x]; c[0].tangent[x] = 0; c[0].pos[x] = 0.5 * (ca[x] + cb[x]); }
        return 1;
    }

    /* face contact: reference box R owns the face, incident box I is clipped against it */
    const int refA = best < 3, a = best % 3, a1 = (a + 1) % 3, a2 = (a + 2) % 3;
    const double (*R)[3] = refA ? A : B, (*I)[3] = refA ? B : A;
    const double *pr = refA ? p1 : p2, *pi = refA ? p2 : p1, *sr = refA ? s1 : s2, *si = refA ? s2 : s1;
    const double dri[3] = {pi[0] - pr[0], pi[1] - pr[1], pi[2] - pr[2]};
    const double sgn = dot3(dri, R[a]) >= 0 ? 1.0 : -1.0;
    double n[3] = {sgn * R[a][0], sgn * R[a][1], sgn * R[a][2]};    /* outward normal of the reference face */
    int kf = 0;
    double kbest = fabs(dot3(n, I[0]));
    for (int k = 1; k < 3; ++k) { const double v = fabs(dot3(n, I[k])); if (v > kbest + 1e-9) { kbest = v; kf = k; } }
    const int k1 = (kf + 1) % 3, k2 = (kf + 2) % 3;
    const double isg = dot3(n, I[kf]) > 0 ? -1.0 : 1.0;
    double cr[3], ci[3];
    for (int x = 0; x < 3; ++x) { cr[x] = pr[x] + sgn * sr[a] * R[a][x]; ci[x] = pi[x] + isg * si[kf] * I[kf][x]; }
    const double h1 = sr[a1], h2 = sr[a2];
    /* incident face vertices in reference-face coordinates (u, v along the face, w = height above it) */
    static const double su[4] = {1, -1, -1, 1}, sv[4] = {1, 1, -1, -1};
    double pu[4], pv[4], pw[4];
    for (int q = 0; q < 4; ++q) {
        double x[3];
        for (int t = 0; t < 3; ++t) x[t] = ci[t] + su[q] * si[k1] * I[k1][t] + sv[q] * si[k2] * I[k2][t] - cr[t];
        pu[q] = dot3(x, R[a1]); pv[q] = dot3(x, R[a2]); pw[q] = dot3(x, n);
    }
    bb_cand_t cand[24];
    const double tol = 1e-12;
    for (int q = 0; q < 4; ++q) {                                    /* 0-3: incident vertices inside the rectangle */
        cand[q].u = pu[q]; cand[q].v = pv[q]; cand[q].w = pw[q];
        cand[q].valid = fabs(pu[q]) <= h1 + tol && fabs(pv[q]) <= h2 + tol;
    }
    /* the incident face's plane over (u, v): w = w0 + gu (u - u0) + gv (v - v0) from three of its vertices */
    const double e1u = pu[1] - pu[0], e1v = pv[1] - pv[0], e1w = pw[1] - pw[0], e2u = pu[3] - pu[0], e2v = pv[3] - pv[0], e2w = pw[3] - pw[0];
    const double det = e1u * e2v - e1v * e2u;
    const double gu = (e1w * e2v - e1v * e2w) / det, gv = (e1u * e2w - e1w * e2u) / det;
    for (int q = 0; q < 4; ++q) {                                    /* 4-7: rectangle corners inside the projected incident face */
        const double u = su[q] * h1, v = sv[q] * h2;
        int pos = 0, neg = 0;
        for (int e = 0; e < 4; ++e) {
            const int f = (e + 1) & 3;
            const double cr2 = (pu[f] - pu[e]) * (v - pv[e]) - (pv[f] - pv[e]) * (u - pu[e]);
            if (cr2 > tol) ++pos; else if (cr2 < -tol) ++neg;
        }
        cand[4 + q].u = u; cand[4 + q].v = v; cand[4 + q].w = pw[0] + gu * (u - pu[0]) + gv * (v - pv[0]);
        cand[4 + q].valid = !(pos && neg);
    }
    for (int e = 0; e < 4; ++e)                                       /* 8-23: incident edges x rectangle edge lines */
        for (int l = 0; l < 4; ++l) {
            const int f = (e + 1) & 3, q = 8 + 4 * e + l;
            const int along_u = l < 2;                               /* lines u = +h1, u = -h1, v = +h2, v = -h2 */
            const double lim = (l & 1) ? -(along_u ? h1 : h2) : (along_u ? h1 : h2);
            const double x0 = along_u ? pu[e] : pv[e], x1 = along_u ? pu[f] : pv[f];
            const double y0 = along_u ? pv[e] : pu[e], y1 = along_u ? pv[f] : pu[f], hy = along_u ? h2 : h1;
            cand[q].valid = 0; cand[q].u = cand[q].v = cand[q].w = 0;
            const double dx = x1 - x0;
            if (fabs(dx) < 1e-14) continue;
            const double sp = (lim - x0) / dx;
            if (!(sp > 0 && sp < 1)) continue;
            const double y = y0 + sp * (y1 - y0);
            if (fabs(y) > hy) continue;                               /* strictly: corners are candidates 4-7 */
            cand[q].u = along_u ? lim : y; cand[q].v = along_u ? y : lim; cand[q].w = pw[e] + sp * (pw[f] - pw[e]);
            cand[q].valid = 1;
        }
    int nvalid = 0;
    for (int q = 0; q < 24; ++q) { if (cand[q].valid && cand[q].w > margin) cand[q].valid = 0; nvalid += cand[q].valid; }
#ifdef CO_STUDY
    const int bb_keep = g_study_box_keep > keep ? g_study_box_keep : keep;
#else
    const int bb_keep = keep;      /* 4, or 8 with CM_FLAG_BOX8 */
#endif
    if (nvalid > bb_keep)                                             /* keep the 4 deepest; within 1e-9 the earlier candidate wins */
        for (int q = 0; q < 24; ++q) {
            if (!cand[q].valid) continue;
            int rank = 0;
            for (int r = 0; r < 24; ++r) {
                if (r == q || !cand[r].valid) continue;
                if (cand[r].w < cand[q].w - 1e-9 || (fabs(cand[r].w - cand[q].w) <= 1e-9 && r < q)) ++rank;
            }
            if (rank >= bb_keep) cand[q].valid = 2;                   /* dropped (marked, so ranks of the others stay put) */
        }
    int nc = 0;
    for (int q = 0; q < 24 && nc < bb_keep; ++q) {
        if (cand[q].valid != 1) continue;
        c[nc].dist = cand[q].w;
        for (int x = 0; x < 3; ++x) {
            const double px = cr[x] + cand[q].u * R[a1][x] + cand[q].v * R[a2][x] + cand[q].w * n[x];
            c[nc].pos[x] = px - 0.5 * cand[q].w * n[x];
            c[nc].normal[x] = refA ? n[x] : -n[x];
            c[nc].tangent[x] = 0;
        }
        ++nc;
    }
    return nc;
}

/* test hook: box-box contacts for two boxes given by centre, rotation matrix (row-major, axes in columns) and half sizes;
 * out[4][7] = dist, pos, normal */
int co_test_box_box_keep(int keep, const double *p1, const double *m1, const double *s1, const double *p2, const double *m2, const double *s2, double margin, double *out) {
    raw_contact_t c[8];
    int n = box_box_keep(c, keep < 1 ? 1 : (keep > 8 ? 8 : keep), p1, m1, s1, p2, m2, s2, margin);
    for (int k = 0; k < n; ++k) { out[7 * k] = c[k].dist; for (int i = 0; i < 3; ++i) { out[7 * k + 1 + i] = c[k].pos[i]; out[7 * k + 4 + i] = c[k].normal[i]; } }
    return n;
}
int co_test_box_box(const double *p1, const double *m1, const double *s1, const double *p2, const double *m2, const double *s2, double margin, double *out) {
    raw_contact_t c[8];
    int n = box_box_keep(c, 4, p1, m1, s1, p2, m2, s2, margin);
    for (int k = 0; k < n; ++k) { out[7 * k] = c[k].dist; for (int i = 0; i < 3; ++i) { out[7 * k + 1 + i] = c[k].pos[i]; out[7 * k + 4 + i] = c[k].normal[i]; } }
    return n;
}


/* ---- height field.  MuJoCo decomposes the cells under a geom's bounding box into triangular prisms and runs its
 * generic convex routine on every prism; this repository's definition keeps the part of that which matters for a sample
 * sphere -- EVERY grid triangle under the sphere's footprint is a candidate, the closest feature (face, edge or vertex of
 * the terrain surface, so the rims of taller neighbouring cells count) gives the contact, the deepest candidate is kept:
 *   above a triangle's plane:  distance to the closest point of the triangle, normal from that point to the centre;
 *   below it and inside its vertical prism (the centre is in the ground): signed distance to the plane, the plane's normal.
 * Grid layout as in MuJoCo: nrow x ncol samples, x <-> columns over [-sx, sx], y <-> rows over [-sy, sy],
 * elevation = sample * size[2].  Each cell is split along the diagonal joining (col+1,row) and (col,row+1). ---- */
static void closest_on_triangle(const double *p, const double *a, const double *b, const double *c, double *q) {
    /* Ericson, Real-Time Collision Detection 5.1.5 */
    double ab[3], ac[3], ap[3], bp[3], cp[3];
    for (int i = 0; i < 3; ++i) { ab[i] = b[i] - a[i]; ac[i] = c[i] - a[i]; ap[i] = p[i] - a[i]; bp[i] = p[i] - b[i]; cp[i] = p[i] - c[i]; }
    const double d1 = dot3(ab, ap), d2 = dot3(ac, ap), d3 = dot3(ab, bp), d4 = dot3(ac, bp), d5 = dot3(ab, cp), d6 = dot3(ac, cp);
    double v = 0, w = 0;   /* q = a + v ab + w ac */
    const double vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
    if (d1 <= 0 && d2 <= 0) { v = 0; w = 0; }
    else if (d3 >= 0 && d4 <= d3) { v = 1; w = 0; }
    else if (vc <= 0 && d1 >= 0 && d3 <= 0) { v = d1 / (d1 - d3); w = 0; }
    else if (d6 >= 0 && d5 <= d6) { v = 0; w = 1; }
    else if (vb <= 0 && d2 >= 0 && d6 <= 0) { v = 0; w = d2 / (d2 - d6); }
    else if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) { w = (d4 - d3) / ((d4 - d3) + (d5 - d6)); v = 1 - w; }
    else { const double den = 1.0 / (va + vb + vc); v = vb * den; w = vc * den; }
    for (int i = 0; i < 3; ++i) q[i] = a[i] + v * ab[i] + w * ac[i];
}
/* one grid triangle against the sample centre p (height-field frame): updates the best (smallest) distance */
static void hfield_triangle(const double *p, const double *a, const double *b, const double *c, double *best, double *bestn) {
    double ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, ac[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]}, n[3];
    cross3(n, ab, ac);
    if (n[2] < 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
    const double inv = 1.0 / sqrt(dot3(n, n));
    n[0] *= inv; n[1] *= inv; n[2] *= inv;
    const double ap[3] = {p[0] - a[0], p[1] - a[1], p[2] - a[2]}, s = dot3(n, ap);
    if (s >= 0) {
        double q[3];
        closest_on_triangle(p, a, b, c, q);
        const double d[3] = {p[0] - q[0], p[1] - q[1], p[2] - q[2]}, len = sqrt(dot3(d, d));
        if (len < *best) {
            *best = len;
            if (len > 1e-12) { bestn[0] = d[0] / len; bestn[1] = d[1] / len; bestn[2] = d[2] / len; }
            else { bestn[0] = n[0]; bestn[1] = n[1]; bestn[2] = n[2]; }
        }
    } else {
        /* centre under this triangle's plane: counts only inside the triangle's vertical prism */
        const double e0 = (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0]);
        const double e1 = (c[0] - b[0]) * (p[1] - b[1]) - (c[1] - b[1]) * (p[0] - b[0]);
        const double e2 = (a[0] - c[0]) * (p[1] - c[1]) - (a[1] - c[1]) * (p[0] - c[0]);
        const int inside = (e0 >= 0 && e1 >= 0 && e2 >= 0) || (e0 <= 0 && e1 <= 0 && e2 <= 0);
        if (inside && s < *best) { *best = s; bestn[0] = n[0]; bestn[1] = n[1]; bestn[2] = n[2]; }
    }
}
static int hfield_sphere(raw_contact_t *c, const cm_model_t *m, const float *data, const double *ph, const double *mh, const double *ps,
                         double r, double margin) {
    if (!data || m->hfield_nrow < 2 || m->hfield_ncol < 2) return 0;
    const double sx = m->hfield_size[0], sy = m->hfield_size[1], sz = m->hfield_size[2];
    double d[3] = {ps[0] - ph[0], ps[1] - ph[1], ps[2] - ph[2]}, p[3];
    mulmatTvec3(p, mh, d);
    const double reach = r + (margin > 0 ? margin : 0);
    if (fabs(p[0]) > sx + reach || fabs(p[1]) > sy + reach || p[2] - r > sz + margin) return 0;
    const int nc = m->hfield_ncol, nr = m->hfield_nrow;
    const double dx = 2 * sx / (nc - 1), dy = 2 * sy / (nr - 1);
    int j0 = (int)floor((p[0] - reach + sx) / dx), j1 = (int)floor((p[0] + reach + sx) / dx);
    int i0 = (int)floor((p[1] - reach + sy) / dy), i1 = (int)floor((p[1] + reach + sy) / dy);
    if (j0 < 0) j0 = 0;
    if (i0 < 0) i0 = 0;
    if (j1 > nc - 2) j1 = nc - 2;
    if (i1 > nr - 2) i1 = nr - 2;
    double best = 1e300, bn[3] = {0, 0, 1};
    for (int i = i0; i <= i1; ++i)
        for (int j = j0; j <= j1; ++j) {
            const double x0 = -sx + j * dx, y0 = -sy + i * dy;
            const double v00[3] = {x0, y0, sz * data[i * nc + j]}, v10[3] = {x0 + dx, y0, sz * data[i * nc + j + 1]};
            const double v01[3] = {x0, y0 + dy, sz * data[(i + 1) * nc + j]}, v11[3] = {x0 + dx, y0 + dy, sz * data[(i + 1) * nc + j + 1]};
            hfield_triangle(p, v00, v10, v01, &best, bn);
            hfield_triangle(p, v11, v01, v10, &best, bn);
        }
    if (best > 1e299) return 0;
    const double dist = best - r;
    if (dist > margin) return 0;
    double nw[3];
    mulmatvec3(nw, mh, bn);
    c->dist = dist;
    for (int k = 0; k < 3; ++k) { c->normal[k] = nw[k]; c->pos[k] = ps[k] - nw[k] * (r + 0.5 * dist); c->tangent[k] = 0; }
    return 1;
}
#ifdef CO_STUDY
/* study mode 1: one contact per penetrated grid triangle under a capsule (a sphere is a capsule of half length 0): the capsule's
 * axis is sampled every 2.5 mm, every triangle under the footprint keeps its deepest sample sphere, and every triangle whose
 * deepest sample is within the margin gives one contact (position on the sample sphere's surface, normal = that sample's
 * closest-feature direction, as in the default definition).  At most CO_RC_MAX contacts, deepest first. */
static int hfield_capsule_per_triangle(raw_contact_t *c, const cm_model_t *m, const float *data, const double *ph, const double *mh, const double *pc,
                                       const double *axw, double r, double h, double margin) {
    if (!data || m->hfield_nrow < 2 || m->hfield_ncol < 2) return 0;
    const double sx = m->hfield_size[0], sy = m->hfield_size[1], sz = m->hfield_size[2];
    double d[3] = {pc[0] - ph[0], pc[1] - ph[1], pc[2] - ph[2]}, p0[3], ax[3];
    mulmatTvec3(p0, mh, d);
    mulmatTvec3(ax, mh, axw);
    const int nc = m->hfield_ncol, nr = m->hfield_nrow;
    const double dx = 2 * sx / (nc - 1), dy = 2 * sy / (nr - 1), reach = r + (margin > 0 ? margin : 0);
    const double xa = p0[0] - h * fabs(ax[0]) - reach, xb = p0[0] + h * fabs(ax[0]) + reach, ya = p0[1] - h * fabs(ax[1]) - reach, yb = p0[1] + h * fabs(ax[1]) + reach;
    int j0 = (int)floor((xa + sx) / dx), j1 = (int)floor((xb + sx) / dx), i0 = (int)floor((ya + sy) / dy), i1 = (int)floor((yb + sy) / dy);
    if (j0 < 0) j0 = 0;
    if (i0 < 0) i0 = 0;
    if (j1 > nc - 2) j1 = nc - 2;
    if (i1 > nr - 2) i1 = nr - 2;
    const int ns = h > 0 ? 1 + 2 * (int)ceil(h / 0.0025) : 1;
    raw_contact_t all[512];
    int nall = 0;
    for (int i = i0; i <= i1; ++i)
        for (int j = j0; j <= j1; ++j)
            for (int tri = 0; tri < 2; ++tri) {
                const double x0 = -sx + j * dx, y0 = -sy + i * dy;
                const double v00[3] = {x0, y0, sz * data[i * nc + j]}, v10[3] = {x0 + dx, y0, sz * data[i * nc + j + 1]};
                const double v01[3] = {x0, y0 + dy, sz * data[(i + 1) * nc + j]}, v11[3] = {x0 + dx, y0 + dy, sz * data[(i + 1) * nc + j + 1]};
                double best = 1e300, bn[3] = {0, 0, 1}, bp[3] = {0, 0, 0};
                for (int k = 0; k < ns; ++k) {
                    const double t = ns > 1 ? -h + 2 * h * k / (ns - 1) : 0.0;
                    const double p[3] = {p0[0] + t * ax[0], p0[1] + t * ax[1], p0[2] + t * ax[2]};
                    double cur = 1e300, cn[3] = {0, 0, 1};
                    if (tri == 0) hfield_triangle(p, v00, v10, v01, &cur, cn); else hfield_triangle(p, v11, v01, v10, &cur, cn);
                    if (cur < best) { best = cur; for (int q = 0; q < 3; ++q) { bn[q] = cn[q]; bp[q] = p[q]; } }
                }
                if (best > 1e299 || best - r > margin) continue;
                raw_contact_t rc;
                double nw[3], pw[3];
                mulmatvec3(nw, mh, bn);
                mulmatvec3(pw, mh, bp);
                rc.dist = best - r;
                for (int q = 0; q < 3; ++q) { rc.normal[q] = nw[q]; rc.pos[q] = pw[q] + ph[q] - nw[q] * (r + 0.5 * rc.dist); rc.tangent[q] = h > 0 ? axw[q] : 0.0; }
                if (nall < 512) all[nall++] = rc;
            }
    /* deepest first (ties: grid order), at most CO_RC_MAX */
    int n = 0;
    char used[512];
    memset(used, 0, sizeof used);
    while (n < CO_RC_MAX) {
        int best_i = -1;
        for (int k = 0; k < nall; ++k) if (!used[k] && (best_i < 0 || all[k].dist < all[best_i].dist)) best_i = k;
        if (best_i < 0) break;
        used[best_i] = 1;
        c[n++] = all[best_i];
    }
    return n;
}
#endif
/* CM_FLAG_HFPRISM: ONE CONTACT PER PENETRATED GRID TRIANGLE under a sphere / capsule (a sphere is a capsule of half length 0).
 * MuJoCo decomposes the terrain under a geom into one prism per grid triangle and reports one contact per penetrated prism
 * (why reference model/cassie_hfield.xml:4 asks for nconmax = 300); this definition keeps that shape with the repository's
 * closest-feature rule per triangle (hfield_triangle above):
 *   - the capsule's axis carries ns = 1 + ceil(2 h / r) sample spheres (at most CM_HP_MAXS), t_k = h (1 - 2 k / (ns - 1)): no
 *     further apart than the radius, so the union of the spheres sags at most 0.134 r below the capsule's surface;
 *   - every grid triangle under the capsule's bounding rectangle (grown by the reach r + margin), cells in row-major scan order,
 *     the triangle (v00, v10, v01) of a cell before (v11, v01, v10), keeps the DEEPEST of the samples -- smallest closest-feature
 *     distance, ties to the lower k -- and gives a contact when that distance minus r is within the margin: position on the
 *     sample sphere's surface, normal = that sample's closest-feature direction, tangent hint = the capsule's axis;
 *   - contacts in that order; the caller's buffer bounds them (co_collision: the model's contact cap, with the warning bit).
 * Returns the number of contacts found (which may exceed `room`: only the first `room` are written). */
static int hfield_prism_contacts(raw_contact_t *c, int room, const cm_model_t *m, const float *data, const double *ph, const double *mh,
                                 const double *pc, const double *axw, double r, double h, double margin) {
    if (!data || m->hfield_nrow < 2 || m->hfield_ncol < 2) return 0;
    const double sx = m->hfield_size[0], sy = m->hfield_size[1], sz = m->hfield_size[2];
    double d[3] = {pc[0] - ph[0], pc[1] - ph[1], pc[2] - ph[2]}, p0[3], ax[3];
    mulmatTvec3(p0, mh, d);
    mulmatTvec3(ax, mh, axw);
    const int nc = m->hfield_ncol, nr = m->hfield_nrow;
    const double dx = 2 * sx / (nc - 1), dy = 2 * sy / (nr - 1), reach = r + (margin > 0 ? margin : 0);
    if (p0[2] - h * fabs(ax[2]) - r > sz + margin) return 0;     /* the whole capsule above the highest possible terrain */
    const double xa = p0[0] - h * fabs(ax[0]) - reach, xb = p0[0] + h * fabs(ax[0]) + reach;
    const double ya = p0[1] - h * fabs(ax[1]) - reach, yb = p0[1] + h * fabs(ax[1]) + reach;
    int j0 = (int)floor((xa + sx) / dx), j1 = (int)floor((xb + sx) / dx), i0 = (int)floor((ya + sy) / dy), i1 = (int)floor((yb + sy) / dy);
    if (j0 < 0) j0 = 0;
    if (i0 < 0) i0 = 0;
    if (j1 > nc - 2) j1 = nc - 2;
    if (i1 > nr - 2) i1 = nr - 2;
    int ns = h > 0 ? 1 + (int)ceil(2 * h / r) : 1;
    if (ns > CM_HP_MAXS) ns = CM_HP_MAXS;
    int n = 0;
    for (int i = i0; i <= i1; ++i)
        for (int j = j0; j <= j1; ++j) {
            const double x0 = -sx + j * dx, y0 = -sy + i * dy;
            const double v00[3] = {x0, y0, sz * data[i * nc + j]}, v10[3] = {x0 + dx, y0, sz * data[i * nc + j + 1]};
            const double v01[3] = {x0, y0 + dy, sz * data[(i + 1) * nc + j]}, v11[3] = {x0 + dx, y0 + dy, sz * data[(i + 1) * nc + j + 1]};
            for (int tri = 0; tri < 2; ++tri) {
                double best = 1e300, bn[3] = {0, 0, 1}, bt = 0;
                for (int k = 0; k < ns; ++k) {
                    const double t = ns > 1 ? h * (1.0 - 2.0 * k / (ns - 1)) : 0.0;
                    const double p[3] = {p0[0] + t * ax[0], p0[1] + t * ax[1], p0[2] + t * ax[2]};
                    double cur = 1e300, cn[3] = {0, 0, 1};
                    if (tri == 0) hfield_triangle(p, v00, v10, v01, &cur, cn); else hfield_triangle(p, v11, v01, v10, &cur, cn);
                    if (cur < best) { best = cur; bt = t; for (int q = 0; q < 3; ++q) bn[q] = cn[q]; }
                }
                if (best > 1e299) continue;
                const double dist = best - r;
                if (dist > margin) continue;
                if (n < room) {
                    raw_contact_t *rc = &c[n];
                    double nw[3];
                    mulmatvec3(nw, mh, bn);
                    rc->dist = dist;
                    for (int q = 0; q < 3; ++q) {
                        const double psw = pc[q] + bt * axw[q];          /* the sample's centre, world frame */
                        rc->normal[q] = nw[q]; rc->pos[q] = psw - nw[q] * (r + 0.5 * dist); rc->tangent[q] = h > 0 ? axw[q] : 0.0;
                    }
                }
                ++n;
            }
        }
    return n;
}
/* test hook: sphere (halflen 0) / capsule against the height field geom by the CM_FLAG_HFPRISM rule; out[max][7] = dist, pos, normal */
int co_test_hfield_prism(const cm_model_t *m, const double *pc, const double *mc, double radius, double halflen, double margin, int max, double *out) {
    if (m->hfield_geom < 0) return 0;
    raw_contact_t c[CM_MAXCON];
    double mh[9];
    const double axw[3] = {mc[2], mc[5], mc[8]};
    quat2mat(mh, m->geom_quat[m->hfield_geom]);
    if (max > CM_MAXCON) max = CM_MAXCON;
    const int n = hfield_prism_contacts(c, max, m, g_hfield, m->geom_pos[m->hfield_geom], mh, pc, axw, radius, halflen, margin);
    for (int k = 0; k < n && k < max; ++k) { out[7 * k] = c[k].dist; for (int i = 0; i < 3; ++i) { out[7 * k + 1 + i] = c[k].pos[i]; out[7 * k + 4 + i] = c[k].normal[i]; } }
    return n;
}
/* test hook: one sphere (world centre ps, radius r) against the height field geom of the model; out = dist, pos, normal */
int co_test_hfield_sphere(const cm_model_t *m, const double *ps, double r, double margin, double *out) {
    if (m->hfield_geom < 0) return 0;
    raw_contact_t c;
    double mh[9];
    quat2mat(mh, m->geom_quat[m->hfield_geom]);
    int n = hfield_sphere(&c, m, g_hfield, m->geom_pos[m->hfield_geom], mh, ps, r, margin);
    if (n) { out[0] = c.dist; for (int i = 0; i < 3; ++i) { out[1 + i] = c.pos[i]; out[4 + i] = c.normal[i]; } }
    return n;
}
/* capsule: the two end spheres as against a plane, plus sample spheres along the axis no further apart than one grid cell
 * (at most 4 interior ones; 8 with CM_FLAG_HFDENSE).  An interior sample counts only when it is deeper than both ends -- a bump under the middle
 * of the capsule -- and then replaces the shallower end; on flat ground this is exactly plane_capsule.  Contacts are
 * reported in the order of their position along the axis, +h first. */
static int hfield_capsule(raw_contact_t *c, const cm_model_t *m, const float *data, const double *ph, const double *mh, const double *pc,
                          const double *mc, const double *sc, double margin) {
    const double ax[3] = {mc[2], mc[5], mc[8]}, h = sc[1];
    const double cell = 2 * m->hfield_size[0] / (m->hfield_ncol > 1 ? m->hfield_ncol - 1 : 1);
    int ni = (int)ceil(2 * h / cell) - 1;  /* interior samples */
    if (ni < 0) ni = 0;
    const int nimax = ((m->flags & CM_FLAG_HFDENSE) ? CM_HF_SLOTS_DENSE : CM_HF_SLOTS) - 2;
    if (ni > nimax) ni = nimax;
    if (m->flags & CM_FLAG_HFMULTI) {
        /* up to CM_HF_MAXC contacts: the deepest sample spheres (ends = samples 0, 1; interior ones 2 ..), deepest first,
         * ties to the lower sample index */
        raw_contact_t smp[CM_HF_SLOTS_DENSE];
        int have[CM_HF_SLOTS_DENSE], taken[CM_HF_SLOTS_DENSE];
        const int ns = 2 + ni;
        for (int k = 0; k < ns; ++k) {
            const double t = k == 0 ? h : (k == 1 ? -h : h * (1.0 - 2.0 * (k - 1) / (ni + 1)));
            double e[3] = {pc[0] + t * ax[0], pc[1] + t * ax[1], pc[2] + t * ax[2]};
            have[k] = hfield_sphere(&smp[k], m, data, ph, mh, e, sc[0], margin);
            taken[k] = 0;
        }
        int n = 0;
        while (n < CM_HF_MAXC) {
            int best = -1;
            for (int k = 0; k < ns; ++k) if (have[k] && !taken[k] && (best < 0 || smp[k].dist < smp[best].dist)) best = k;
            if (best < 0) break;
            taken[best] = 1;
            c[n] = smp[best];
            for (int k = 0; k < 3; ++k) c[n].tangent[k] = ax[k];
            ++n;
        }
        return n;
    }
    raw_contact_t end[2], mid;
    int have_end[2] = {0, 0}, have_mid = 0;
    double tmid = 0;
    for (int s = 0; s < 2; ++s) {
        const double t = s == 0 ? h : -h;
        double e[3] = {pc[0] + t * ax[0], pc[1] + t * ax[1], pc[2] + t * ax[2]};
        have_end[s] = hfield_sphere(&end[s], m, data, ph, mh, e, sc[0], margin);
    }
    for (int k = 1; k <= ni; ++k) {
        const double t = h * (1.0 - 2.0 * k / (ni + 1));
        double e[3] = {pc[0] + t * ax[0], pc[1] + t * ax[1], pc[2] + t * ax[2]};
        raw_contact_t cur;
        if (hfield_sphere(&cur, m, data, ph, mh, e, sc[0], margin) && (!have_mid || cur.dist < mid.dist)) { mid = cur; have_mid = 1; tmid = t; }
    }
    (void)tmid;
    if (have_mid && (!have_end[0] || mid.dist < end[0].dist) && (!have_end[1] || mid.dist < end[1].dist)) {
        /* the interior sample is the deepest point of the capsule: it takes the place of the shallower (or absent) end */
        const int drop = !have_end[0] ? 0 : (!have_end[1] ? 1 : (end[0].dist <= end[1].dist ? 1 : 0));
        end[drop] = mid;
        have_end[drop] = 1;
    }
    int n = 0;
    for (int s = 0; s < 2; ++s)
        if (have_end[s]) { c[n] = end[s]; for (int k = 0; k < 3; ++k) c[n].tangent[k] = ax[k]; ++n; }
    return n;
}

/* test hook: one capsule (world centre pc, rotation matrix mc with the axis in its third column, radius, half length) against
 * the height field geom; out[CM_HF_MAXC][7] = dist, pos, normal per contact (two without CM_FLAG_HFMULTI) */
int co_test_hfield_capsule(const cm_model_t *m, const double *pc, const double *mc, double radius, double halflen, double margin, double *out) {
    if (m->hfield_geom < 0) return 0;
    raw_contact_t c[CM_HF_MAXC];
    double mh[9], sc[3] = {radius, halflen, 0};
    quat2mat(mh, m->geom_quat[m->hfield_geom]);
    int n = hfield_capsule(c, m, g_hfield, m->geom_pos[m->hfield_geom], mh, pc, mc, sc, margin);
    for (int k = 0; k < n; ++k) { out[7 * k] = c[k].dist; for (int i = 0; i < 3; ++i) { out[7 * k + 1 + i] = c[k].pos[i]; out[7 * k + 4 + i] = c[k].normal[i]; } }
    return n;
}

/* completes a contact frame from its normal and an optional tangent hint */
static void make_frame(double *frame) {
    normalize3(frame);
    if (norm3(frame + 3) < 0.5) {
        frame[3] = frame[4] = frame[5] = 0;
        if (frame[1] < 0.5 && frame[1] > -0.5) frame[4] = 1; else frame[5] = 1;
    }
    double t = dot3(frame, frame + 3);
    for (int i = 0; i < 3; ++i) frame[3 + i] -= t * frame[i];
    normalize3(frame + 3);
    cross3(frame + 6, frame, frame + 3);
}

void co_collision(const cm_model_t *m, co_data_t *d) {
    d->ncon = 0;
    const int maxcon = m->maxcon > 0 && m->maxcon < CM_MAXCON ? m->maxcon : CM_MAXCON;
    const int prism = (m->flags & CM_FLAG_HFPRISM) != 0;
    /* CM_FLAG_HFPRISM: the height-field pairs' contacts come FIRST (pair order, grid order within a pair), then the other pairs'
     * in pair order -- the kernel forms them in a pass of its own ahead of its pair loop */
    for (int pass = prism ? 0 : 1; pass < 2; ++pass)
    for (int p = 0; p < m->npair; ++p) {
        int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p];
        int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
        const int hfpair = t1 == CM_GEOM_HFIELD && (t2 == CM_GEOM_SPHERE || t2 == CM_GEOM_CAPSULE);
        if (prism && (pass == 0) != (hfpair != 0)) continue;
        double margin = fmax(m->geom_margin[g1], m->geom_margin[g2]);
        double gap = fmax(m->geom_gap[g1], m->geom_gap[g2]);
        const double *p1 = d->geom_xpos[g1], *p2 = d->geom_xpos[g2];
        const double *m1 = d->geom_xmat[g1], *m2 = d->geom_xmat[g2];
        /* bounding-sphere / plane-distance cull */
        if (m->geom_rbound[g1] > 0 && m->geom_rbound[g2] > 0) {
            double dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
            double bound = m->geom_rbound[g1] + m->geom_rbound[g2] + margin;
            if (dot3(dif, dif) > bound * bound) continue;
        } else if (t1 == CM_GEOM_PLANE && m->geom_rbound[g2] > 0) {
            double n[3] = {m1[2], m1[5], m1[8]};
            double dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
            if (dot3(dif, n) > margin + m->geom_rbound[g2]) continue;
        }
        raw_contact_t rc[CO_RC_MAX > CM_MAXCON ? CO_RC_MAX : CM_MAXCON];
        (void)mulmat3;
        int n = 0;
        if (t1 == CM_GEOM_PLANE && t2 == CM_GEOM_SPHERE) n = plane_sphere(rc, p1, m1, p2, m->geom_size[g2][0], margin);
        else if (t1 == CM_GEOM_PLANE && t2 == CM_GEOM_CAPSULE) n = plane_capsule(rc, p1, m1, p2, m2, m->geom_size[g2], margin);
        else if (t1 == CM_GEOM_SPHERE && t2 == CM_GEOM_SPHERE) n = sphere_sphere(rc, p1, m->geom_size[g1][0], p2, m->geom_size[g2][0], margin);
        else if (t1 == CM_GEOM_SPHERE && t2 == CM_GEOM_CAPSULE) n = sphere_capsule(rc, p1, m->geom_size[g1][0], p2, m2, m->geom_size[g2], margin);
        else if (t1 == CM_GEOM_CAPSULE && t2 == CM_GEOM_CAPSULE) n = capsule_capsule(rc, p1, m1, m->geom_size[g1], p2, m2, m->geom_size[g2], margin);
#ifdef CO_STUDY
        else if (t1 == CM_GEOM_HFIELD && (t2 == CM_GEOM_SPHERE || t2 == CM_GEOM_CAPSULE) && g_study_hfield_mode == 1) {
            const double axw[3] = {m2[2], m2[5], m2[8]};
            n = hfield_capsule_per_triangle(rc, m, g_hfield, p1, m1, p2, axw, m->geom_size[g2][0], t2 == CM_GEOM_CAPSULE ? m->geom_size[g2][1] : 0.0, margin);
        }
#endif
        else if (hfpair && prism) {
            const double axw[3] = {m2[2], m2[5], m2[8]};
            const int room = maxcon - d->ncon;
            n = hfield_prism_contacts(rc, room, m, g_hfield, p1, m1, p2, axw, m->geom_size[g2][0], t2 == CM_GEOM_CAPSULE ? m->geom_size[g2][1] : 0.0, margin);
            if (n > room) { d->warn_contact_full = 1; n = room; }
        }
        else if (t1 == CM_GEOM_HFIELD && t2 == CM_GEOM_SPHERE) n = hfield_sphere(rc, m, g_hfield, p1, m1, p2, m->geom_size[g2][0], margin);
        else if (t1 == CM_GEOM_HFIELD && t2 == CM_GEOM_CAPSULE) n = hfield_capsule(rc, m, g_hfield, p1, m1, p2, m2, m->geom_size[g2], margin);
        else if (t1 == CM_GEOM_SPHERE && t2 == CM_GEOM_BOX) n = sphere_box(rc, p1, m->geom_size[g1][0], p2, m2, m->geom_size[g2], margin);
        else if (t1 == CM_GEOM_CAPSULE && t2 == CM_GEOM_BOX) n = capsule_box(rc, p1, m1, m->geom_size[g1], p2, m2, m->geom_size[g2], margin);
        else if (t1 == CM_GEOM_PLANE && t2 == CM_GEOM_BOX) n = plane_box(rc, p1, m1, p2, m2, m->geom_size[g2], margin);
        else if (t1 == CM_GEOM_BOX && t2 == CM_GEOM_BOX) n = box_box_keep(rc, (m->flags & CM_FLAG_BOX8) ? 8 : 4, p1, m1, m->geom_size[g1], p2, m2, m->geom_size[g2], margin);
        else { d->warn_unsupported_pair = 1; continue; }
        for (int k = 0; k < n; ++k) {
            if (d->ncon >= maxcon) { d->warn_contact_full = 1; break; }
            co_contact_t *c = &d->contact[d->ncon++];
            memset(c, 0, sizeof *c);
            c->dist = rc[k].dist;
            for (int i = 0; i < 3; ++i) { c->pos[i] = rc[k].pos[i]; c->frame[i] = rc[k].normal[i]; c->frame[3 + i] = rc[k].tangent[i]; }
            make_frame(c->frame);
            c->geom1 = g1; c->geom2 = g2;
            c->includemargin = margin - gap;
            /* contact parameters: priority wins, else max condim / max friction / solmix-weighted solref, solimp */
            double fri[3];
            int pa = m->geom_priority[g1], pb = m->geom_priority[g2];
            if (pa != pb) {
                int g = pa > pb ? g1 : g2;
                c->dim = m->geom_condim[g];
                for (int i = 0; i < 2; ++i) c->solref[i] = m->geom_solref[g][i];
                for (int i = 0; i < 5; ++i) c->solimp[i] = m->geom_solimp[g][i];
                for (int i = 0; i < 3; ++i) fri[i] = m->geom_friction[g][i];
            } else {
                c->dim = m->geom_condim[g1] > m->geom_condim[g2] ? m->geom_condim[g1] : m->geom_condim[g2];
                double s1 = m->geom_solmix[g1], s2 = m->geom_solmix[g2], mix;
                if (s1 >= CM_MINVAL && s2 >= CM_MINVAL) mix = s1 / (s1 + s2);
                else if (s1 < CM_MINVAL && s2 < CM_MINVAL) mix = 0.5;
                else mix = s1 < CM_MINVAL ? 0.0 : 1.0;
                if (m->geom_solref[g1][0] > 0 && m->geom_solref[g2][0] > 0)
                    for (int i = 0; i < 2; ++i) c->solref[i] = mix * m->geom_solref[g1][i] + (1 - mix) * m->geom_solref[g2][i];
                else
                    for (int i = 0; i < 2; ++i) c->solref[i] = fmin(m->geom_solref[g1][i], m->geom_solref[g2][i]);
                for (int i = 0; i < 5; ++i) c->solimp[i] = mix * m->geom_solimp[g1][i] + (1 - mix) * m->geom_solimp[g2][i];
                for (int i = 0; i < 3; ++i) fri[i] = fmax(m->geom_friction[g1][i], m->geom_friction[g2][i]);
            }
            c->friction[0] = c->friction[1] = fri[0]; c->friction[2] = fri[1]; c->friction[3] = c->friction[4] = fri[2];
        }
    }
}

/* ------------------------------------------- P5: constraint assembly ------ */
static void impedance(const double *solimp, double pos, double margin, double *imp) {
    double dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
    if (dmin == dmax || width <= CM_MINVAL) { *imp = 0.5 * (dmin + dmax); return; }
    double x = fabs((pos - margin) / width);
    if (x >= 1) { *imp = dmax; return; }
    if (x <= 0) { *imp = dmin; return; }
    double y;
    if (power == 1) y = x;
    else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
    else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
    *imp = dmin + y * (dmax - dmin);
}

static int add_row(co_data_t *d, int maxefc, int type, int id, const double *J, int nv, double pos, double margin, double diag) {
    if (d->nefc >= maxefc) { d->warn_constraint_full = 1; return 0; }
    int r = d->nefc++;
    d->efc_type[r] = type; d->efc_id[r] = id;
    for (int k = 0; k < nv; ++k) d->efc_J[r][k] = J[k];
    d->efc_pos[r] = pos; d->efc_margin[r] = margin; d->efc_diagApprox[r] = diag;
    return 1;
}

void co_make_constraint(const cm_model_t *m, co_data_t *d) {
    int nv = m->nv;
    d->nefc = d->ne = d->nl = 0;
    const int maxefc = m->maxefc > 0 && m->maxefc < CM_MAXEFC ? m->maxefc : CM_MAXEFC;
    double jp1[3][NV_], jr1[3][NV_], jp2[3][NV_], jr2[3][NV_], J[NV_];

    /* equality: connect (3 rows each): residual = anchor1_world - anchor2_world */
    for (int e = 0; e < m->neq; ++e) {
        if (!m->eq_active[e]) continue;
        int b1 = m->eq_body1[e], b2 = m->eq_body2[e];
        double a1[3], a2[3];
        mulmatvec3(a1, d->xmat[b1], m->eq_data[e]);
        mulmatvec3(a2, d->xmat[b2], m->eq_data[e] + 3);
        for (int i = 0; i < 3; ++i) { a1[i] += d->xpos[b1][i]; a2[i] += d->xpos[b2][i]; }
        co_jac(m, d, b1, a1, jp1, jr1);
        co_jac(m, d, b2, a2, jp2, jr2);
        double diag = m->body_invweight0[b1][0] + m->body_invweight0[b2][0];
        if (d->nefc + 3 > maxefc) { d->warn_constraint_full = 1; continue; }
        for (int i = 0; i < 3; ++i) {
            for (int k = 0; k < nv; ++k) J[k] = jp1[i][k] - jp2[i][k];
            add_row(d, maxefc, CM_CNSTR_EQUALITY, e, J, nv, a1[i] - a2[i], 0.0, diag);
        }
    }
    d->ne = d->nefc;

    /* joint limits (hinge / slide) */
    for (int j = 0; j < m->njnt; ++j) {
        if (!m->jnt_limited[j]) continue;
        if (m->jnt_type[j] != CM_JNT_HINGE && m->jnt_type[j] != CM_JNT_SLIDE) continue;
        double q = d->qpos[m->jnt_qposadr[j]], margin = m->jnt_margin[j];
        for (int side = -1; side <= 1; side += 2) {
            double dist = side < 0 ? q - m->jnt_range[j][0] : m->jnt_range[j][1] - q;
            if (dist < margin) {
                for (int k = 0; k < nv; ++k) J[k] = 0;
                J[m->jnt_dofadr[j]] = -(double)side;
                add_row(d, maxefc, CM_CNSTR_LIMIT_JOINT, j, J, nv, dist, margin, m->dof_invweight0[m->jnt_dofadr[j]]);
            }
        }
    }
    d->nl = d->nefc - d->ne;

    /* contacts */
    for (int ci = 0; ci < d->ncon; ++ci) {
        co_contact_t *c = &d->contact[ci];
        int b1 = m->geom_bodyid[c->geom1], b2 = m->geom_bodyid[c->geom2];
        int nrow = c->dim == 1 ? 1 : 2 * (c->dim - 1);
        c->efc_address = -1;
        if (c->dim != 1 && c->dim != 3) { d->warn_unsupported_pair = 1; continue; }
        if (d->nefc + nrow > maxefc) { d->warn_constraint_full = 1; continue; }
        co_jac(m, d, b1, c->pos, jp1, jr1);
        co_jac(m, d, b2, c->pos, jp2, jr2);
        double Jf[3][NV_]; /* relative translational Jacobian in the contact frame */
        for (int a = 0; a < 3; ++a)
            for (int k = 0; k < nv; ++k) {
                double v = 0;
                for (int i = 0; i < 3; ++i) v += c->frame[3 * a + i] * (jp2[i][k] - jp1[i][k]);
                Jf[a][k] = v;
            }
        double tran = m->body_invweight0[b1][0] + m->body_invweight0[b2][0];
        c->efc_address = d->nefc;
        if (c->dim == 1) {
            add_row(d, maxefc, CM_CNSTR_CONTACT_FRICTIONLESS, ci, Jf[0], nv, c->dist, c->includemargin, tran);
        } else {
            for (int a = 1; a < c->dim; ++a) {
                double mu = c->friction[a - 1];
                double diag = tran + mu * mu * tran;
                for (int k = 0; k < nv; ++k) J[k] = Jf[0][k] + mu * Jf[a][k];
                add_row(d, maxefc, CM_CNSTR_CONTACT_PYRAMIDAL, ci, J, nv, c->dist, c->includemargin, diag);
                for (int k = 0; k < nv; ++k) J[k] = Jf[0][k] - mu * Jf[a][k];
                add_row(d, maxefc, CM_CNSTR_CONTACT_PYRAMIDAL, ci, J, nv, c->dist, c->includemargin, diag);
            }
        }
    }

    /* impedance, regulariser, reference acceleration parameters */
    for (int r = 0; r < d->nefc; ++r) {
        const double *solref, *solimp;
        double pos = d->efc_pos[r], margin = d->efc_margin[r];
        int id = d->efc_id[r];
        switch (d->efc_type[r]) {
            case CM_CNSTR_EQUALITY: {
                solref = m->eq_solref[id]; solimp = m->eq_solimp[id];
                /* all rows of one connect share the impedance of the residual norm */
                int r0 = r;
                while (r0 > 0 && d->efc_type[r0 - 1] == CM_CNSTR_EQUALITY && d->efc_id[r0 - 1] == id) --r0;
                double s = 0;
                for (int k = 0; k < 3; ++k) s += d->efc_pos[r0 + k] * d->efc_pos[r0 + k];
                pos = sqrt(s);
            } break;
            case CM_CNSTR_LIMIT_JOINT: solref = m->jnt_solref[id]; solimp = m->jnt_solimp[id]; break;
            default: solref = d->contact[id].solref; solimp = d->contact[id].solimp; break;
        }
        double imp;
        impedance(solimp, pos, margin, &imp);
        double R = (1 - imp) * d->efc_diagApprox[r] / imp;
        if (R < CM_MINVAL) R = CM_MINVAL;
        d->efc_R[r] = R;
        double dmax = solimp[1], K, B;
        if (solref[0] > 0) {
            double tc = solref[0], dr = solref[1];
            if ((m->flags & CM_FLAG_REFSAFE) && tc < 2 * m->timestep) tc = 2 * m->timestep;
            K = 1 / fmax(CM_MINVAL, dmax * dmax * tc * tc * dr * dr);
            B = 2 / fmax(CM_MINVAL, dmax * tc);
        } else {
            K = -solref[0] / fmax(CM_MINVAL, dmax * dmax);
            B = -solref[1] / fmax(CM_MINVAL, dmax);
        }
        d->efc_KBIP[r][0] = K; d->efc_KBIP[r][1] = B; d->efc_KBIP[r][2] = imp; d->efc_KBIP[r][3] = 0;
    }
    /* pyramidal contacts: all rows share R = 2 mu^2 R(first row) */
    for (int ci = 0; ci < d->ncon; ++ci) {
        co_contact_t *c = &d->contact[ci];
        if (c->efc_address < 0 || c->dim == 1) continue;
        double Rpy = 2 * c->friction[0] * c->friction[0] * d->efc_R[c->efc_address];
        if (Rpy < CM_MINVAL) Rpy = CM_MINVAL;
        for (int k = 0; k < 2 * (c->dim - 1); ++k) d->efc_R[c->efc_address + k] = Rpy;
    }
    for (int r = 0; r < d->nefc; ++r) d->efc_D[r] = 1 / d->efc_R[r];
}

/* --------------------------------------- P6: velocity-dependent forces ---- */
static void com_vel(const cm_model_t *m, co_data_t *d) {
    memset(d->cvel[0], 0, sizeof d->cvel[0]);
    for (int b = 1; b < m->nbody; ++b) {
        double cvel[6];
        memcpy(cvel, d->cvel[m->body_parentid[b]], sizeof cvel);
        for (int jj = 0; jj < m->body_jntnum[b]; ++jj) {
            int j = m->body_jntadr[b] + jj, da = m->jnt_dofadr[j];
            int nrot = 0, r0 = da;
            switch (m->jnt_type[j]) {
                case CM_JNT_FREE:
                    for (int k = 0; k < 3; ++k) {
                        memset(d->cdof_dot[da + k], 0, sizeof d->cdof_dot[0]);
                        for (int i = 0; i < 6; ++i) cvel[i] += d->cdof[da + k][i] * d->qvel[da + k];
                    }
                    nrot = 3; r0 = da + 3;
                    break;
                case CM_JNT_BALL: nrot = 3; r0 = da; break;
                default: nrot = 1; r0 = da; break;
            }
            /* all axes of one joint are differentiated with the velocity before the joint */
            for (int k = 0; k < nrot; ++k) cross_motion(d->cdof_dot[r0 + k], cvel, d->cdof[r0 + k]);
            for (int k = 0; k < nrot; ++k)
                for (int i = 0; i < 6; ++i) cvel[i] += d->cdof[r0 + k][i] * d->qvel[r0 + k];
        }
        memcpy(d->cvel[b], cvel, sizeof cvel);
    }
}
static void passive(const cm_model_t *m, co_data_t *d) {
    for (int k = 0; k < m->nv; ++k) d->qfrc_passive[k] = 0;
    for (int j = 0; j < m->njnt; ++j) {
        if (m->jnt_stiffness[j] == 0) continue;
        if (m->jnt_type[j] != CM_JNT_HINGE && m->jnt_type[j] != CM_JNT_SLIDE) continue; /* subset: no ball/free springs */
        int qa = m->jnt_qposadr[j];
        d->qfrc_passive[m->jnt_dofadr[j]] = -m->jnt_stiffness[j] * (d->qpos[qa] - m->qpos_spring[qa]);
    }
    for (int k = 0; k < m->nv; ++k) d->qfrc_passive[k] -= m->dof_damping[k] * d->qvel[k];
}
/* recursive Newton-Euler with zero qacc: Coriolis, centrifugal and gravity */
static void rne_bias(const cm_model_t *m, co_data_t *d) {
    static const double zero6[6] = {0};
    double cacc[CM_MAXBODY][6], cfrc[CM_MAXBODY][6];
    (void)zero6;
    for (int i = 0; i < 3; ++i) { cacc[0][i] = 0; cacc[0][3 + i] = -m->gravity[i]; }
    memset(cfrc[0], 0, sizeof cfrc[0]);
    for (int b = 1; b < m->nbody; ++b) {
        memcpy(cacc[b], cacc[m->body_parentid[b]], sizeof cacc[b]);
        for (int k = 0; k < m->body_dofnum[b]; ++k) {
            int dd = m->body_dofadr[b] + k;
            for (int i = 0; i < 6; ++i) cacc[b][i] += d->cdof_dot[dd][i] * d->qvel[dd];
        }
        double t1[6], t2[6], t3[6];
        mul_inert_vec(t1, d->cinert[b], cacc[b]);
        mul_inert_vec(t2, d->cinert[b], d->cvel[b]);
        cross_force(t3, d->cvel[b], t2);
        for (int i = 0; i < 6; ++i) cfrc[b][i] = t1[i] + t3[i];
    }
    for (int b = m->nbody - 1; b > 0; --b) {
        int p = m->body_parentid[b];
        if (p > 0) for (int i = 0; i < 6; ++i) cfrc[p][i] += cfrc[b][i];
    }
    for (int k = 0; k < m->nv; ++k) {
        double v = 0;
        for (int i = 0; i < 6; ++i) v += d->cdof[k][i] * cfrc[m->dof_bodyid[k]][i];
        d->qfrc_bias[k] = v;
    }
}

/* ---------------------------------------------- P7/P8: smooth dynamics ---- */
static void fwd_actuation(const cm_model_t *m, co_data_t *d) {
    for (int k = 0; k < m->nv; ++k) d->qfrc_actuator[k] = 0;
    for (int u = 0; u < m->nu; ++u) {
        double c = d->ctrl[u];
        if (m->act_ctrllimited[u]) c = clampd(c, m->act_ctrlrange[u][0], m->act_ctrlrange[u][1]);
        d->actuator_force[u] = c;
        d->qfrc_actuator[m->act_dofid[u]] += m->act_gear[u] * c;
    }
}
static void fwd_acceleration(const cm_model_t *m, co_data_t *d) {
    int nv = m->nv;
    for (int k = 0; k < nv; ++k)
        d->qfrc_smooth[k] = d->qfrc_passive[k] - d->qfrc_bias[k] + d->qfrc_applied[k] + d->qfrc_actuator[k];
    /* Cartesian perturbations: force then torque, applied at the body's inertial frame origin */
    for (int b = 1; b < m->nbody; ++b) {
        const double *f = d->xfrc_applied[b];
        if (f[0] == 0 && f[1] == 0 && f[2] == 0 && f[3] == 0 && f[4] == 0 && f[5] == 0) continue;
        double jp[3][NV_], jr[3][NV_];
        co_jac(m, d, b, d->xipos[b], jp, jr);
        for (int k = 0; k < nv; ++k)
            for (int i = 0; i < 3; ++i) d->qfrc_smooth[k] += jp[i][k] * f[i] + jr[i][k] * f[3 + i];
    }
    for (int k = 0; k < nv; ++k) d->qacc_smooth[k] = d->qfrc_smooth[k];
    co_solve_m(m, d->qLD, d->qacc_smooth);
}

/* ------------------------------------------- P9/P10: projection + PGS ----- */
static void fwd_constraint(const cm_model_t *m, co_data_t *d) {
    int nv = m->nv, n = d->nefc;
    for (int k = 0; k < nv; ++k) d->qfrc_constraint[k] = 0;
    if (n == 0) {
        for (int k = 0; k < nv; ++k) d->qacc[k] = d->qacc_smooth[k];
        d->solver_iter = 0;
        return;
    }
    /* reference acceleration and b = J qacc_smooth - aref */
    for (int r = 0; r < n; ++r) {
        double vel = 0, ja = 0;
        for (int k = 0; k < nv; ++k) { vel += d->efc_J[r][k] * d->qvel[k]; ja += d->efc_J[r][k] * d->qacc_smooth[k]; }
        d->efc_vel[r] = vel;
        d->efc_aref[r] = -d->efc_KBIP[r][1] * vel - d->efc_KBIP[r][0] * d->efc_KBIP[r][2] * (d->efc_pos[r] - d->efc_margin[r]);
        d->efc_b[r] = ja - d->efc_aref[r];
    }
    /* AR = J M^-1 J^T + diag(R) */
    double MinvJT[CM_MAXEFC][NV_];
    for (int r = 0; r < n; ++r) {
        for (int k = 0; k < nv; ++k) MinvJT[r][k] = d->efc_J[r][k];
        co_solve_m(m, d->qLD, MinvJT[r]);
    }
    for (int r = 0; r < n; ++r)
        for (int s = 0; s < n; ++s) {
            double v = 0;
            for (int k = 0; k < nv; ++k) v += d->efc_J[r][k] * MinvJT[s][k];
            d->efc_AR[r][s] = v;
        }
    for (int r = 0; r < n; ++r) d->efc_AR[r][r] += d->efc_R[r];

    /* warm start: forces implied by qacc_warmstart, kept only if they beat f = 0 in the dual cost */
    double *f = d->efc_force;
    if (m->flags & CM_FLAG_WARMSTART) {
        for (int r = 0; r < n; ++r) {
            double jar = 0;
            for (int k = 0; k < nv; ++k) jar += d->efc_J[r][k] * d->qacc_warmstart[k];
            jar -= d->efc_aref[r];
            double fr = -d->efc_D[r] * jar;
            if (d->efc_type[r] != CM_CNSTR_EQUALITY && fr < 0) fr = 0;
            f[r] = fr;
        }
        double cost = 0;
        for (int r = 0; r < n; ++r) {
            double arf = 0;
            for (int s = 0; s < n; ++s) arf += d->efc_AR[r][s] * f[s];
            cost += f[r] * (d->efc_b[r] + 0.5 * arf);
        }
        if (cost > 0) for (int r = 0; r < n; ++r) f[r] = 0;
    } else {
        for (int r = 0; r < n; ++r) f[r] = 0;
    }

    /* projected Gauss-Seidel */
    double scale = 1 / (m->meaninertia * (nv > 1 ? nv : 1));
    int iter = 0;
    while (iter < m->iterations) {
        double improvement = 0;
        for (int r = 0; r < n; ++r) {
            double res = d->efc_b[r];
            for (int s = 0; s < n; ++s) res += d->efc_AR[r][s] * f[s];
            double old = f[r];
            double fn = old - res / d->efc_AR[r][r];
            if (d->efc_type[r] != CM_CNSTR_EQUALITY && fn < 0) fn = 0;
            double delta = fn - old;
            double change = 0.5 * delta * delta * d->efc_AR[r][r] + delta * res;
            if (change > 1e-10) { fn = old; change = 0; } /* never accept a cost increase */
            f[r] = fn;
            improvement -= change;
        }
        improvement *= scale;
        ++iter;
        if (improvement < m->tolerance) break;
    }
    d->solver_iter = iter;

    /* map back to joint space */
    for (int k = 0; k < nv; ++k) {
        double v = 0;
        for (int r = 0; r < n; ++r) v += d->efc_J[r][k] * f[r];
        d->qfrc_constraint[k] = v;
    }
    double tmp[NV_];
    for (int k = 0; k < nv; ++k) tmp[k] = d->qfrc_constraint[k];
    co_solve_m(m, d->qLD, tmp);
    for (int k = 0; k < nv; ++k) d->qacc[k] = d->qacc_smooth[k] + tmp[k];
}

/* ------------------------------------------------------ P11: sensors ------ */
static void sensors(const cm_model_t *m, co_data_t *d) {
    for (int u = 0; u < m->nu; ++u) {
        d->actuator_length[u] = m->act_gear[u] * d->qpos[m->act_qposadr[u]];
        d->actuator_velocity[u] = m->act_gear[u] * d->qvel[m->act_dofid[u]];
    }
    for (int s = 0; s < m->nsensor; ++s) {
        double *out = d->sensordata + m->sensor_adr[s];
        int id = m->sensor_objid[s];
        switch (m->sensor_type[s]) {
            case CM_SENS_ACTUATORPOS: out[0] = d->actuator_length[id]; break;
            case CM_SENS_JOINTPOS: out[0] = d->qpos[m->jnt_qposadr[id]]; break;
            case CM_SENS_FRAMEQUAT: mulquat(out, d->xquat[m->site_bodyid[id]], m->site_quat[id]); break;
            case CM_SENS_GYRO: mulmatTvec3(out, d->site_xmat[id], d->cvel[m->site_bodyid[id]]); break;
            case CM_SENS_MAGNETOMETER: mulmatTvec3(out, d->site_xmat[id], m->magnetic); break;
            case CM_SENS_ACCELEROMETER: {
                /* com-frame acceleration of the site's body, including the constraint solution */
                int b = m->site_bodyid[id];
                int chain[CM_MAXBODY], nc = 0;
                for (int a = b; a > 0; a = m->body_parentid[a]) chain[nc++] = a;
                double cacc[6] = {0, 0, 0, -m->gravity[0], -m->gravity[1], -m->gravity[2]};
                for (int c = nc - 1; c >= 0; --c) {
                    int a = chain[c];
                    for (int k = 0; k < m->body_dofnum[a]; ++k) {
                        int dd = m->body_dofadr[a] + k;
                        for (int i = 0; i < 6; ++i) cacc[i] += d->cdof_dot[dd][i] * d->qvel[dd] + d->cdof[dd][i] * d->qacc[dd];
                    }
                }
                memcpy(d->cacc_imu, cacc, sizeof cacc);
                const double *com = d->subtree_com[m->body_rootid[b]];
                double dif[3] = {d->site_xpos[id][0] - com[0], d->site_xpos[id][1] - com[1], d->site_xpos[id][2] - com[2]};
                double t[3], lin[3], vlin[3], corr[3];
                cross3(t, dif, cacc);
                for (int i = 0; i < 3; ++i) lin[i] = cacc[3 + i] - t[i];
                cross3(t, dif, d->cvel[b]);
                for (int i = 0; i < 3; ++i) vlin[i] = d->cvel[b][3 + i] - t[i];
                cross3(corr, d->cvel[b], vlin); /* omega x v: acceleration of the moving frame origin */
                for (int i = 0; i < 3; ++i) lin[i] += corr[i];
                mulmatTvec3(out, d->site_xmat[id], lin);
            } break;
            default: for (int i = 0; i < m->sensor_dim[s]; ++i) out[i] = 0;
        }
        if (m->sensor_cutoff[s] > 0 && m->sensor_type[s] != CM_SENS_FRAMEQUAT) {
            double c = m->sensor_cutoff[s];
            for (int i = 0; i < m->sensor_dim[s]; ++i) out[i] = clampd(out[i], -c, c);
        }
    }
}

/* ------------------------------------------------------- P12: Euler ------- */
void co_integrate_pos(const cm_model_t *m, double *qpos, const double *qvel, double dt) {
    for (int j = 0; j < m->njnt; ++j) {
        int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
        switch (m->jnt_type[j]) {
            case CM_JNT_FREE:
                for (int i = 0; i < 3; ++i) qpos[qa + i] += dt * qvel[da + i];
                qa += 3; da += 3;
                /* fall through */
            case CM_JNT_BALL: {
                double ax[3] = {qvel[da], qvel[da + 1], qvel[da + 2]};
                double ang = dt * normalize3(ax), qr[4];
                axisangle2quat(qr, ax, ang);
                normalize4(qpos + qa);
                mulquat(qpos + qa, qpos + qa, qr);
            } break;
            default: qpos[qa] += dt * qvel[da];
        }
    }
}

static int bad(double x) { return !(x == x) || x > 1e10 || x < -1e10; }

static void fwd_position(const cm_model_t *m, co_data_t *d) {
    co_kinematics(m, d);
    co_com_pos(m, d);
    co_crb(m, d);
    co_factor_m(m, d);
    co_collision(m, d);
    co_make_constraint(m, d);
}
static void fwd_velocity(const cm_model_t *m, co_data_t *d) {
    com_vel(m, d);
    passive(m, d);
    rne_bias(m, d);
}

void co_forward(const cm_model_t *m, co_data_t *d) {
    fwd_position(m, d);
    fwd_velocity(m, d);
    fwd_actuation(m, d);
    fwd_acceleration(m, d);
    fwd_constraint(m, d);
    sensors(m, d);
}

void co_step(const cm_model_t *m, co_data_t *d) {
    int nv = m->nv;
    for (int i = 0; i < m->nq; ++i) if (bad(d->qpos[i])) d->diverged = 1;
    for (int i = 0; i < nv; ++i) if (bad(d->qvel[i])) d->diverged = 1;
    if (d->diverged) return; /* sticky: state is left untouched (failure detection) */
    co_forward(m, d);
    for (int i = 0; i < nv; ++i) if (bad(d->qacc[i])) d->diverged = 1;
    if (d->diverged) return;

    double qacc[NV_];
    int damped = 0;
    for (int k = 0; k < nv; ++k) if (m->dof_damping[k] > 0) damped = 1;
    if (damped && (m->flags & CM_FLAG_EULERDAMP)) {
        /* (M + h B) qacc* = qfrc_smooth + qfrc_constraint */
        double MH[NV_][NV_];
        for (int i = 0; i < nv; ++i) for (int j = 0; j < nv; ++j) MH[i][j] = j <= i ? d->qM[i][j] : 0.0;
        for (int k = 0; k < nv; ++k) MH[k][k] += m->timestep * m->dof_damping[k];
        factor_ld(m, MH);
        for (int k = 0; k < nv; ++k) qacc[k] = d->qfrc_smooth[k] + d->qfrc_constraint[k];
        co_solve_m(m, MH, qacc);
    } else {
        for (int k = 0; k < nv; ++k) qacc[k] = d->qacc[k];
    }
    for (int k = 0; k < nv; ++k) d->qvel[k] += m->timestep * qacc[k];
    co_integrate_pos(m, d->qpos, d->qvel, m->timestep);
    d->time += m->timestep;
    for (int k = 0; k < nv; ++k) d->qacc_warmstart[k] = d->qacc[k];
}

/* ---------------------------------------------- batched helpers (bench / tests) ---- */
void co_pd_ctrl(const cm_model_t *m, co_data_t *d, const double *ptarget, const double *kp, const double *kd) {
    for (int u = 0; u < m->nu; ++u) {
        double ratio = m->act_gear[u], tmax = m->act_ctrlrange[u][1];
        double q = d->qpos[m->act_qposadr[u]], qd = d->qvel[m->act_dofid[u]];
        double tau = kp[u] * (ptarget[u] - q) - kd[u] * qd;
        double wmax = m->act_maxrpm[u] * (2.0 * M_PI / 60.0);
        double tlim = clampd(2 * tmax * (1 - fabs(ratio * qd) / wmax), 0.0, tmax);
        d->ctrl[u] = copysign(fmin(fabs(tau / ratio), tlim), tau);
    }
}

#ifdef _OPENMP
#include <omp.h>
int co_max_threads(void) { return omp_get_max_threads(); }
#else
int co_max_threads(void) { return 1; }
#endif

void co_step_batch(const cm_model_t *m, co_data_t *d, int n, int nsteps, const double *ptarget, const double *kp,
                   const double *kd, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int e = 0; e < n; ++e)
        for (int s = 0; s < nsteps; ++s) {
            if (ptarget) co_pd_ctrl(m, &d[e], ptarget + (size_t)e * m->nu, kp + (size_t)e * m->nu, kd + (size_t)e * m->nu);
            co_step(m, &d[e]);
        }
}
