/*
 * physics_kernel.h -- the batched Cassie physics step for gfx950 (MI355X):
 * ONE WAVEFRONT (64 lanes) PER ENVIRONMENT, one single-wave workgroup per env.
 *
 * This is the hot path of the reference -- the mj_step1_fp + mj_step2_fp pair at
 * reference src/cassiemujoco.c:1130-1134 (arithmetic inside MuJoCo 2.1.0; stage
 * list SURVEY.md 8a P1..P12, semantics SURVEY.md App. B) -- redesigned for a
 * CDNA4 wave instead of a CPU thread:
 *
 *   lanes = bodies   kinematics / velocity recursion level by level, inertia
 *   lanes = dofs     motion axes, CRBA rows, bias forces, single-RHS tree solves
 *   lanes = pairs    narrow-phase collision, ballot-compacted contact list
 *   lanes = rows     constraint Jacobian rows, the (<=63)-RHS half solve
 *                    Y = D^-1/2 L^-T [J^T | qfrc_smooth], A = Y^T Y + R, and PGS
 *                    with the residual vector distributed one row per lane and a
 *                    v_readlane broadcast of each row's force update
 *
 * All per-env tiles (poses, spatial inertias, M, the two LDL factors, Y and A)
 * live in LDS; the batched qpos/qvel/ctrl/sensordata arrays are env-major in HBM
 * so a wave's loads and stores are contiguous.  The constant model is read
 * through wave-uniform loads.
 *
 * Numerically this follows the same algorithm as oracle/cassie_oracle.c but with
 * its own operation order (half solves instead of full solves, reciprocal
 * multiplies, wave reductions), so parity is to a tolerance, not bitwise.
 */
#ifndef CASSIE_PHYSICS_KERNEL_H
#define CASSIE_PHYSICS_KERNEL_H

#include "cm_model.h"
#include <wave.h> /* csrc/wave.h in the product build; tests/emu/wave.h under the CPU wave emulator */

namespace ck {

constexpr int NB = CM_MAXBODY;
constexpr int NG = CM_MAXGEOM;
constexpr int NROW = 64;       /* rows 0..62 constraints, column 63 = qfrc_smooth */
constexpr int AP = 65;         /* padded leading dimension of A and Y^T (bank-conflict free columns) */

/* warning bits reported per env */
constexpr int NSTAMP = 16;
#define CK_STAMP(i) do { if (io.prof && lane == 0) io.prof[(size_t)env * NSTAMP + (i)] = wv::clock(); } while (0)
enum { WARN_CONTACT_FULL = 1, WARN_CONSTRAINT_FULL = 2, WARN_UNSUPPORTED_PAIR = 4, WARN_DIVERGED = 8 };

struct PhysIO {
    const cm_model_t *models;   /* one shared model, or one per env */
    int model_stride;           /* 0 = shared, 1 = per-env */
    int nenv, nsub;             /* nsub physics steps per launch (ctrl held) */
    int integrate;              /* 1 = step (Euler), 0 = forward only (mj_forward role) */
    int sq, sv, su, ssd, sb;    /* row strides: nq, nv, nu, nsensordata, nbody */
    double *qpos, *qvel, *qacc_warmstart, *time;
    const double *ctrl, *qfrc_applied, *xfrc_applied; /* the last two may be null */
    double *qacc, *sensordata, *actuator_velocity;
    int *warn;                  /* [nenv] sticky warning bits */
    int *info;                  /* [nenv][4]: ncon, nefc, solver iterations, reserved (may be null) */
    double *xpos_out;           /* optional [nenv][nbody][3] (may be null) */
    double *xquat_out;          /* optional [nenv][nbody][4] (may be null) */
    const float *hfield;        /* shared heightfield samples (may be null) */
    /* optional on-device joint PD (all three null = torque mode): every substep
     * ctrl_u = motor-side torque of  kp (ptarget - q) - kd qdot  after the motor's
     * speed-torque limit -- the motor law of pd_input_step followed by
     * motor() (reference src/cassiemujoco.c:638-664) on the exact joint state */
    const double *pd_ptarget, *pd_kp, *pd_kd; /* [nenv][nu] each */
    long long *prof;            /* optional [nenv][NSTAMP] shader-clock stamps of the last substep (may be null) */
};

template <int NVP>
struct EnvShared {
    union {
        struct {
            double xpos[NB][3], xquat[NB][4], xmat[NB][9], xipos[NB][3];
            double xanchor[CM_MAXJNT][3], xaxis[CM_MAXJNT][3];
            double cinert[NB][10], crb[NB][10];
            double cvel[NB][6], cacc[NB][6], cfrc[NB][6];
            double cdof_dot[NVP][6];
            double geom_xpos[NG][3], geom_xmat[NG][9];
            double M[NVP][NVP + 1];
        } s;
        double A[NROW][AP];
    } x;
    double Yt[NVP][AP];
    double LD[NVP][NVP + 1], LDH[NVP][NVP + 1];
    double rsd[NVP];            /* 1/sqrt(D) of LD */
    double cdof[NVP][6];
    double com[NB][3];          /* subtree com, valid at root bodies */
    double qpos[CM_MAXQ], qvel[NVP], qacc_ws[NVP], qacc[NVP], ctrl[CM_MAXU];
    double qfrc_smooth[NVP];
    double xfrc[NB][6];
    /* contacts */
    double c_dist[CM_MAXCON], c_pos[CM_MAXCON][3], c_frame[CM_MAXCON][9], c_fri[CM_MAXCON][3];
    double c_solref[CM_MAXCON][2], c_solimp[CM_MAXCON][5], c_margin[CM_MAXCON];
    int c_dim[CM_MAXCON], c_g1[CM_MAXCON], c_g2[CM_MAXCON];
    /* sensor site frames (kept past the A overlay) */
    double site_xpos[CM_MAXSITE][3], site_xmat[CM_MAXSITE][9], site_cvel[CM_MAXSITE][6];
};

/* ------------------------------------------------------------ small math --- */
WV_DEVICE double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
WV_DEVICE void cross3(double *r, const double *a, const double *b) {
    double t0 = a[1] * b[2] - a[2] * b[1], t1 = a[2] * b[0] - a[0] * b[2], t2 = a[0] * b[1] - a[1] * b[0];
    r[0] = t0; r[1] = t1; r[2] = t2;
}
WV_DEVICE double normalize3(double *a) {
    double n = sqrt(dot3(a, a));
    if (n < CM_MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; }
    else { double s = 1.0 / n; a[0] *= s; a[1] *= s; a[2] *= s; }
    return n;
}
WV_DEVICE void normalize4(double *q) {
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n < CM_MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
    else { double s = 1.0 / n; q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s; }
}
WV_DEVICE void mulquat(double *r, const double *a, const double *b) {
    double t0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    double t1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    double t2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    double t3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
    r[0] = t0; r[1] = t1; r[2] = t2; r[3] = t3;
}
WV_DEVICE void quat2mat(double *m, const double *q) {
    double q00 = q[0] * q[0], q11 = q[1] * q[1], q22 = q[2] * q[2], q33 = q[3] * q[3];
    double q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
    double q12 = q[1] * q[2], q13 = q[1] * q[3], q23 = q[2] * q[3];
    m[0] = q00 + q11 - q22 - q33; m[1] = 2 * (q12 - q03);       m[2] = 2 * (q13 + q02);
    m[3] = 2 * (q12 + q03);       m[4] = q00 - q11 + q22 - q33; m[5] = 2 * (q23 - q01);
    m[6] = 2 * (q13 - q02);       m[7] = 2 * (q23 + q01);       m[8] = q00 - q11 - q22 + q33;
}
WV_DEVICE void mulmatvec3(double *r, const double *m, const double *v) {
    double t0 = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
    double t1 = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
    double t2 = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
    r[0] = t0; r[1] = t1; r[2] = t2;
}
WV_DEVICE void mulmatTvec3(double *r, const double *m, const double *v) {
    double t0 = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
    double t1 = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
    double t2 = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
    r[0] = t0; r[1] = t1; r[2] = t2;
}
WV_DEVICE void rotvecquat(double *r, const double *v, const double *q) {
    double m[9];
    quat2mat(m, q);
    mulmatvec3(r, m, v);
}
WV_DEVICE void axisangle2quat(double *q, const double *axis, double angle) {
    double s = sin(angle * 0.5);
    q[0] = cos(angle * 0.5); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
WV_DEVICE double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* spatial algebra, [rotational; translational] */
WV_DEVICE void cross_motion(double *r, const double *vel, const double *v) {
    double a[3], b[3], c[3];
    cross3(a, vel, v); cross3(b, vel, v + 3); cross3(c, vel + 3, v);
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2];
    r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
WV_DEVICE void cross_force(double *r, const double *vel, const double *f) {
    double a[3], b[3], c[3];
    cross3(a, vel, f); cross3(b, vel + 3, f + 3); cross3(c, vel, f + 3);
    r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2];
    r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}
WV_DEVICE void mul_inert_vec(double *r, const double *I, const double *v) {
    r[0] = I[0] * v[0] + I[3] * v[1] + I[4] * v[2] - I[8] * v[4] + I[7] * v[5];
    r[1] = I[3] * v[0] + I[1] * v[1] + I[5] * v[2] + I[8] * v[3] - I[6] * v[5];
    r[2] = I[4] * v[0] + I[5] * v[1] + I[2] * v[2] - I[7] * v[3] + I[6] * v[4];
    r[3] = I[8] * v[1] - I[7] * v[2] + I[9] * v[3];
    r[4] = I[6] * v[2] - I[8] * v[0] + I[9] * v[4];
    r[5] = I[7] * v[0] - I[6] * v[1] + I[9] * v[5];
}

/* ------------------------------------------------------ narrow phase ------ */
struct RawContact { double dist, pos[3], normal[3], tangent[3]; };

WV_DEVICE int plane_sphere(RawContact &c, const double *ppos, const double *pmat, const double *spos, double r,
                           double margin) {
    double n[3] = {pmat[2], pmat[5], pmat[8]};
    double dif[3] = {spos[0] - ppos[0], spos[1] - ppos[1], spos[2] - ppos[2]};
    double dist = dot3(dif, n) - r;
    if (dist > margin) return 0;
    c.dist = dist;
    for (int i = 0; i < 3; ++i) { c.normal[i] = n[i]; c.pos[i] = spos[i] - n[i] * (r + 0.5 * dist); c.tangent[i] = 0; }
    return 1;
}
WV_DEVICE int sphere_sphere(RawContact &c, const double *p1, double r1, const double *p2, double r2, double margin) {
    double dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
    double cd = sqrt(dot3(dif, dif));
    double dist = cd - r1 - r2;
    if (dist > margin) return 0;
    double n[3];
    if (cd < CM_MINVAL) { n[0] = 1; n[1] = 0; n[2] = 0; }
    else { n[0] = dif[0] / cd; n[1] = dif[1] / cd; n[2] = dif[2] / cd; }
    c.dist = dist;
    for (int i = 0; i < 3; ++i) { c.normal[i] = n[i]; c.pos[i] = p1[i] + n[i] * (r1 + 0.5 * dist); c.tangent[i] = 0; }
    return 1;
}
WV_DEVICE void segment_closest(const double *p1, const double *a1, double l1, const double *p2, const double *a2,
                               double l2, double &x1, double &x2) {
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
        double s = -mb, c2 = v;
        double lo = fmax(-l2, c2 - l1), hi = fmin(l2, c2 + l1);
        if (lo <= hi) t2 = 0.5 * (lo + hi);
        else t2 = clampd(c2, -l2, l2);
        t1 = clampd((t2 - c2) * (s >= 0 ? 1.0 : -1.0), -l1, l1);
    }
    x1 = t1; x2 = t2;
}
WV_DEVICE void make_frame(double *frame) {
    normalize3(frame);
    if (sqrt(dot3(frame + 3, frame + 3)) < 0.5) {
        frame[3] = frame[4] = frame[5] = 0;
        if (frame[1] < 0.5 && frame[1] > -0.5) frame[4] = 1; else frame[5] = 1;
    }
    double t = dot3(frame, frame + 3);
    for (int i = 0; i < 3; ++i) frame[3 + i] -= t * frame[i];
    normalize3(frame + 3);
    cross3(frame + 6, frame, frame + 3);
}

WV_DEVICE double impedance(const double *solimp, double pos, double margin) {
    double dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
    if (dmin == dmax || width <= CM_MINVAL) return 0.5 * (dmin + dmax);
    double x = fabs((pos - margin) / width);
    if (x >= 1) return dmax;
    if (x <= 0) return dmin;
    double y;
    if (power == 1) y = x;
    else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
    else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
    return dmin + y * (dmax - dmin);
}

/* L^T D L factorisation of the tree-sparse matrix stored densely in LDS.
 * Lane j owns column j (lanes 0..31 -> first matrix, 32..63 -> second one when
 * two are factored at once).  Zeros outside the sparsity pattern stay zero. */
template <int NVP>
WV_DEVICE void factor_columns(const cm_model_t *m, double (*A)[NVP + 1], int j, int nv, bool active) {
    for (int k = nv - 1; k >= 0; --k) {
        double invD = 1.0 / A[k][k];
        double akj = (active && j < k) ? A[k][j] : 0.0;
        for (int i = m->dof_parentid[k]; i >= 0; i = m->dof_parentid[i]) {
            double t = A[k][i] * invD;
            if (active && j <= i) A[i][j] -= t * akj;
        }
        wv::sync();
        if (active && j < k) A[k][j] = akj * invD;
        wv::sync();
    }
}

/* x <- A^-1 x for one right-hand side distributed one entry per lane (lane k holds x[k]);
 * A given by its L^T D L factor.  Dense sweeps: zeros off the tree pattern are harmless. */
template <int NVP>
WV_DEVICE double solve_single(double (*LDm)[NVP + 1], double x, int nv, int lane) {
    for (int k = nv - 1; k > 0; --k) { /* L^-T */
        double xk = wv::readlane(x, k);
        if (lane < k) x -= LDm[k][lane] * xk;
    }
    if (lane < nv) x /= LDm[lane][lane];
    for (int i = 0; i < nv - 1; ++i) { /* L^-1 */
        double xi = wv::readlane(x, i);
        if (lane > i && lane < nv) x -= LDm[lane][i] * xi;
    }
    return x;
}

/* ======================================================== the env step ==== */
template <int NVP>
WV_DEVICE void env_step(const PhysIO &io, EnvShared<NVP> &S, int env) {
    const cm_model_t *m = io.models + (size_t)env * io.model_stride;
    const int lane = wv::lane();
    const int nq = m->nq, nv = m->nv, nu = m->nu, nbody = m->nbody, njnt = m->njnt;
    const double h = m->timestep;
    int warn = 0;

    /* ---------------- load state (coalesced, env-major) ---------------- */
    if (lane < nq) S.qpos[lane] = io.qpos[(size_t)env * io.sq + lane];
    if (lane < nv) {
        S.qvel[lane] = io.qvel[(size_t)env * io.sv + lane];
        S.qacc_ws[lane] = io.qacc_warmstart[(size_t)env * io.sv + lane];
    }
    if (lane < nu) S.ctrl[lane] = io.ctrl[(size_t)env * io.su + lane];
    double time = io.time[env];
    wv::sync();

    for (int sub = 0; sub < io.nsub; ++sub) {
        /* divergence guard (mj_checkPos/mj_checkVel role): sticky flag, state left alone */
        {
            bool badv = false;
            if (lane < nq) { double v = S.qpos[lane]; badv |= !(v == v) || fabs(v) > 1e10; }
            if (lane < nv) { double v = S.qvel[lane]; badv |= !(v == v) || fabs(v) > 1e10; }
            if (wv::ballot(badv) != 0ull) { warn |= WARN_DIVERGED; break; }
        }

        if (io.pd_ptarget) {
            if (lane < nu) {
                const size_t o = (size_t)env * io.su + lane;
                const double ratio = m->act_gear[lane], tmax = m->act_ctrlrange[lane][1];
                const double q = S.qpos[m->act_qposadr[lane]], qd = S.qvel[m->act_dofid[lane]];
                const double tau = io.pd_kp[o] * (io.pd_ptarget[o] - q) - io.pd_kd[o] * qd;
                const double wmax = m->act_maxrpm[lane] * (2.0 * 3.14159265358979323846 / 60.0);
                const double tlim = clampd(2 * tmax * (1 - fabs(ratio * qd) / wmax), 0.0, tmax);
                S.ctrl[lane] = copysign(fmin(fabs(tau / ratio), tlim), tau);
            }
            wv::sync();
        }

        CK_STAMP(0);
        /* ================= P1 kinematics: lane = body, level by level ================= */
        const int b = lane;
        const bool isbody = b < nbody;
        const int depth = isbody ? m->body_depth[b] : -1;
        double ximat[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (b == 0) {
            for (int i = 0; i < 3; ++i) { S.x.s.xpos[0][i] = 0; S.x.s.xipos[0][i] = 0; }
            S.x.s.xquat[0][0] = 1; S.x.s.xquat[0][1] = S.x.s.xquat[0][2] = S.x.s.xquat[0][3] = 0;
            for (int i = 0; i < 9; ++i) S.x.s.xmat[0][i] = (i % 4 == 0) ? 1.0 : 0.0;
        }
        wv::sync();
        for (int d = 1; d <= m->maxdepth; ++d) {
            if (depth == d) {
                const int p = m->body_parentid[b];
                const int j0 = m->body_jntadr[b], jn = m->body_jntnum[b];
                double pos[3], quat[4];
                if (jn == 1 && m->jnt_type[j0] == CM_JNT_FREE) {
                    const int qa = m->jnt_qposadr[j0];
                    for (int i = 0; i < 3; ++i) pos[i] = S.qpos[qa + i];
                    for (int i = 0; i < 4; ++i) quat[i] = S.qpos[qa + 3 + i];
                    normalize4(quat);
                    for (int i = 0; i < 3; ++i) { S.x.s.xanchor[j0][i] = pos[i]; S.x.s.xaxis[j0][i] = (i == 2) ? 1.0 : 0.0; }
                } else {
                    double bp[3] = {m->body_pos[b][0], m->body_pos[b][1], m->body_pos[b][2]};
                    double bq[4] = {m->body_quat[b][0], m->body_quat[b][1], m->body_quat[b][2], m->body_quat[b][3]};
                    mulmatvec3(pos, S.x.s.xmat[p], bp);
                    for (int i = 0; i < 3; ++i) pos[i] += S.x.s.xpos[p][i];
                    mulquat(quat, S.x.s.xquat[p], bq);
                    for (int jj = 0; jj < jn; ++jj) {
                        const int j = j0 + jj, qa = m->jnt_qposadr[j], jt = m->jnt_type[j];
                        double jp[3] = {m->jnt_pos[j][0], m->jnt_pos[j][1], m->jnt_pos[j][2]};
                        double ja[3] = {m->jnt_axis[j][0], m->jnt_axis[j][1], m->jnt_axis[j][2]};
                        double anchor[3], axis[3];
                        rotvecquat(anchor, jp, quat);
                        for (int i = 0; i < 3; ++i) anchor[i] += pos[i];
                        rotvecquat(axis, ja, quat);
                        for (int i = 0; i < 3; ++i) { S.x.s.xanchor[j][i] = anchor[i]; S.x.s.xaxis[j][i] = axis[i]; }
                        if (jt == CM_JNT_SLIDE) {
                            double s = S.qpos[qa] - m->qpos0[qa];
                            for (int i = 0; i < 3; ++i) pos[i] += axis[i] * s;
                        } else {
                            double ql[4];
                            if (jt == CM_JNT_BALL) {
                                for (int i = 0; i < 4; ++i) ql[i] = S.qpos[qa + i];
                                normalize4(ql);
                            } else {
                                axisangle2quat(ql, ja, S.qpos[qa] - m->qpos0[qa]);
                            }
                            mulquat(quat, quat, ql);
                            double r[3];
                            rotvecquat(r, jp, quat);
                            for (int i = 0; i < 3; ++i) pos[i] = anchor[i] - r[i];
                        }
                    }
                }
                normalize4(quat);
                double xm[9];
                quat2mat(xm, quat);
                for (int i = 0; i < 3; ++i) S.x.s.xpos[b][i] = pos[i];
                for (int i = 0; i < 4; ++i) S.x.s.xquat[b][i] = quat[i];
                for (int i = 0; i < 9; ++i) S.x.s.xmat[b][i] = xm[i];
                double ip[3] = {m->body_ipos[b][0], m->body_ipos[b][1], m->body_ipos[b][2]};
                double iq[4] = {m->body_iquat[b][0], m->body_iquat[b][1], m->body_iquat[b][2], m->body_iquat[b][3]};
                double xi[3];
                mulmatvec3(xi, xm, ip);
                for (int i = 0; i < 3; ++i) S.x.s.xipos[b][i] = pos[i] + xi[i];
                double qi[4];
                mulquat(qi, quat, iq);
                quat2mat(ximat, qi);
            }
            wv::sync();
        }

        CK_STAMP(1);
        /* geoms (lane = collision geom) and sites (lane = site) */
        if (lane < m->ngeom) {
            const int g = lane, gb = m->geom_bodyid[g];
            double gp[3] = {m->geom_pos[g][0], m->geom_pos[g][1], m->geom_pos[g][2]};
            double gq[4] = {m->geom_quat[g][0], m->geom_quat[g][1], m->geom_quat[g][2], m->geom_quat[g][3]};
            double t[3], q[4], mm[9];
            mulmatvec3(t, S.x.s.xmat[gb], gp);
            for (int i = 0; i < 3; ++i) S.x.s.geom_xpos[g][i] = t[i] + S.x.s.xpos[gb][i];
            mulquat(q, S.x.s.xquat[gb], gq);
            quat2mat(mm, q);
            for (int i = 0; i < 9; ++i) S.x.s.geom_xmat[g][i] = mm[i];
        }
        if (lane < m->nsite) {
            const int s = lane, sb = m->site_bodyid[s];
            double sp[3] = {m->site_pos[s][0], m->site_pos[s][1], m->site_pos[s][2]};
            double sq[4] = {m->site_quat[s][0], m->site_quat[s][1], m->site_quat[s][2], m->site_quat[s][3]};
            double t[3], q[4], mm[9];
            mulmatvec3(t, S.x.s.xmat[sb], sp);
            for (int i = 0; i < 3; ++i) S.site_xpos[s][i] = t[i] + S.x.s.xpos[sb][i];
            mulquat(q, S.x.s.xquat[sb], sq);
            quat2mat(mm, q);
            for (int i = 0; i < 9; ++i) S.site_xmat[s][i] = mm[i];
        }

        /* ================= com of every kinematic tree (wave reduction per root) ================= */
        const double bmass = (isbody && b > 0) ? m->body_mass[b] : 0.0;
        const int broot = isbody ? m->body_rootid[b] : -1;
        {
            double wx = 0, wy = 0, wz = 0;
            if (isbody && b > 0) { wx = bmass * S.x.s.xipos[b][0]; wy = bmass * S.x.s.xipos[b][1]; wz = bmass * S.x.s.xipos[b][2]; }
            for (int r = 1; r < nbody; ++r) {
                if (m->body_parentid[r] != 0) continue;
                const bool in = broot == r;
                double sm = wv::wave_sum(in ? bmass : 0.0);
                double sx = wv::wave_sum(in ? wx : 0.0), sy = wv::wave_sum(in ? wy : 0.0), sz = wv::wave_sum(in ? wz : 0.0);
                if (lane == 0) {
                    if (sm < CM_MINVAL) { S.com[r][0] = S.x.s.xipos[r][0]; S.com[r][1] = S.x.s.xipos[r][1]; S.com[r][2] = S.x.s.xipos[r][2]; }
                    else { double inv = 1.0 / sm; S.com[r][0] = sx * inv; S.com[r][1] = sy * inv; S.com[r][2] = sz * inv; }
                }
            }
        }
        wv::sync();

        /* ================= cinert (lane = body), cdof (lane = dof) ================= */
        if (isbody) {
            double ci[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            if (b > 0) {
                const double I0 = m->body_inertia[b][0], I1 = m->body_inertia[b][1], I2 = m->body_inertia[b][2];
                const double *c = S.com[broot];
                double dif[3] = {S.x.s.xipos[b][0] - c[0], S.x.s.xipos[b][1] - c[1], S.x.s.xipos[b][2] - c[2]};
                double d2 = dot3(dif, dif);
                const double *R = ximat;
                double W00 = R[0] * I0 * R[0] + R[1] * I1 * R[1] + R[2] * I2 * R[2];
                double W11 = R[3] * I0 * R[3] + R[4] * I1 * R[4] + R[5] * I2 * R[5];
                double W22 = R[6] * I0 * R[6] + R[7] * I1 * R[7] + R[8] * I2 * R[8];
                double W01 = R[0] * I0 * R[3] + R[1] * I1 * R[4] + R[2] * I2 * R[5];
                double W02 = R[0] * I0 * R[6] + R[1] * I1 * R[7] + R[2] * I2 * R[8];
                double W12 = R[3] * I0 * R[6] + R[4] * I1 * R[7] + R[5] * I2 * R[8];
                ci[0] = W00 + bmass * (d2 - dif[0] * dif[0]);
                ci[1] = W11 + bmass * (d2 - dif[1] * dif[1]);
                ci[2] = W22 + bmass * (d2 - dif[2] * dif[2]);
                ci[3] = W01 - bmass * dif[0] * dif[1];
                ci[4] = W02 - bmass * dif[0] * dif[2];
                ci[5] = W12 - bmass * dif[1] * dif[2];
                ci[6] = bmass * dif[0]; ci[7] = bmass * dif[1]; ci[8] = bmass * dif[2]; ci[9] = bmass;
            }
            for (int i = 0; i < 10; ++i) S.x.s.cinert[b][i] = ci[i];
        }
        const int k_ = lane; /* dof owned by this lane in dof-parallel stages */
        const bool isdof = k_ < nv;
        const int kjnt = isdof ? m->dof_jntid[k_] : 0;
        const int kbody = isdof ? m->dof_bodyid[k_] : 0;
        double cd[6] = {0, 0, 0, 0, 0, 0};
        if (isdof) {
            const int jt = m->jnt_type[kjnt], da = m->jnt_dofadr[kjnt];
            const double *c = S.com[m->body_rootid[kbody]];
            double off[3] = {c[0] - S.x.s.xanchor[kjnt][0], c[1] - S.x.s.xanchor[kjnt][1], c[2] - S.x.s.xanchor[kjnt][2]};
            const int sub_k = k_ - da;
            if (jt == CM_JNT_SLIDE) {
                for (int i = 0; i < 3; ++i) cd[3 + i] = S.x.s.xaxis[kjnt][i];
            } else if (jt == CM_JNT_HINGE) {
                for (int i = 0; i < 3; ++i) cd[i] = S.x.s.xaxis[kjnt][i];
                cross3(cd + 3, cd, off);
            } else if (jt == CM_JNT_FREE && sub_k < 3) {
                cd[3 + sub_k] = 1.0;
            } else {
                const int a = (jt == CM_JNT_FREE) ? sub_k - 3 : sub_k;
                cd[0] = S.x.s.xmat[kbody][a]; cd[1] = S.x.s.xmat[kbody][3 + a]; cd[2] = S.x.s.xmat[kbody][6 + a];
                cross3(cd + 3, cd, off);
            }
            for (int i = 0; i < 6; ++i) S.cdof[k_][i] = cd[i];
        }
        /* zero the dense mass matrix while the inertias land */
        for (int e = lane; e < NVP * (NVP + 1); e += WV_WAVE) (&S.x.s.M[0][0])[e] = 0.0;
        wv::sync();

        CK_STAMP(2);
        /* ================= P2 CRBA ================= */
        if (isbody) {
            double acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            const int e = m->body_subtreeend[b];
            if (b > 0) for (int c = b; c < e; ++c) for (int i = 0; i < 10; ++i) acc[i] += S.x.s.cinert[c][i];
            for (int i = 0; i < 10; ++i) S.x.s.crb[b][i] = acc[i];
        }
        wv::sync();
        if (isdof) {
            double buf[6];
            mul_inert_vec(buf, S.x.s.crb[kbody], cd);
            for (int j = k_; j >= 0; j = m->dof_parentid[j]) {
                double v = 0;
                for (int i = 0; i < 6; ++i) v += S.cdof[j][i] * buf[i];
                if (j == k_) v += m->dof_armature[k_];
                S.x.s.M[k_][j] = v;
                S.x.s.M[j][k_] = v;
            }
        }
        wv::sync();

        CK_STAMP(3);
        /* ================= P3 factor M and M + h*diag(damping) ================= */
        for (int e = lane; e < NVP * (NVP + 1); e += WV_WAVE) {
            const int r = e / (NVP + 1), c = e % (NVP + 1);
            double v = (&S.x.s.M[0][0])[e];
            if (c > r) v = 0.0; /* factors live in the lower triangle */
            S.LD[r][c] = v;
            S.LDH[r][c] = (r == c && r < nv) ? v + h * m->dof_damping[r] : v;
        }
        wv::sync();
        if (NVP <= 32) {
            const bool second = lane >= 32;
            factor_columns<NVP>(m, second ? S.LDH : S.LD, lane & 31, nv, (lane & 31) < nv);
        } else {
            factor_columns<NVP>(m, S.LD, lane, nv, lane < nv);
            factor_columns<NVP>(m, S.LDH, lane, nv, lane < nv);
        }
        if (isdof) S.rsd[k_] = 1.0 / sqrt(S.LD[k_][k_]);

        CK_STAMP(4);
        /* ================= P4 collision: lane = candidate pair ================= */
        int ncon = 0;
        for (int p0 = 0; p0 < m->npair; p0 += WV_WAVE) {
            const int p = p0 + lane;
            int n = 0;
            RawContact rc[2];
            int g1 = 0, g2 = 0;
            double margin = 0, gap = 0;
            if (p < m->npair) {
                g1 = m->pair_geom1[p]; g2 = m->pair_geom2[p];
                const int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
                margin = fmax(m->geom_margin[g1], m->geom_margin[g2]);
                gap = fmax(m->geom_gap[g1], m->geom_gap[g2]);
                const double *p1 = S.x.s.geom_xpos[g1], *p2 = S.x.s.geom_xpos[g2];
                const double *m1 = S.x.s.geom_xmat[g1], *m2 = S.x.s.geom_xmat[g2];
                const double rb1 = m->geom_rbound[g1], rb2 = m->geom_rbound[g2];
                double dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
                bool cull = false;
                if (rb1 > 0 && rb2 > 0) {
                    double bound = rb1 + rb2 + margin;
                    cull = dot3(dif, dif) > bound * bound;
                } else if (t1 == CM_GEOM_PLANE && rb2 > 0) {
                    double nn[3] = {m1[2], m1[5], m1[8]};
                    cull = dot3(dif, nn) > margin + rb2;
                }
                if (!cull) {
                    const double s10 = m->geom_size[g1][0], s11 = m->geom_size[g1][1];
                    const double s20 = m->geom_size[g2][0], s21 = m->geom_size[g2][1];
                    if (t1 == CM_GEOM_PLANE && t2 == CM_GEOM_SPHERE) {
                        n = plane_sphere(rc[0], p1, m1, p2, s20, margin);
                    } else if (t1 == CM_GEOM_PLANE && t2 == CM_GEOM_CAPSULE) {
                        double axis[3] = {m2[2], m2[5], m2[8]};
                        for (int s = 0; s < 2; ++s) {
                            double sg = s == 0 ? s21 : -s21;
                            double e[3] = {p2[0] + sg * axis[0], p2[1] + sg * axis[1], p2[2] + sg * axis[2]};
                            if (plane_sphere(rc[n], p1, m1, e, s20, margin)) {
                                for (int i = 0; i < 3; ++i) rc[n].tangent[i] = axis[i];
                                ++n;
                            }
                        }
                    } else if (t1 == CM_GEOM_SPHERE && t2 == CM_GEOM_SPHERE) {
                        n = sphere_sphere(rc[0], p1, s10, p2, s20, margin);
                    } else if (t1 == CM_GEOM_SPHERE && t2 == CM_GEOM_CAPSULE) {
                        double a2[3] = {m2[2], m2[5], m2[8]};
                        double d12[3] = {-dif[0], -dif[1], -dif[2]};
                        double x = clampd(dot3(a2, d12), -s21, s21);
                        double q2[3] = {p2[0] + a2[0] * x, p2[1] + a2[1] * x, p2[2] + a2[2] * x};
                        n = sphere_sphere(rc[0], p1, s10, q2, s20, margin);
                    } else if (t1 == CM_GEOM_CAPSULE && t2 == CM_GEOM_CAPSULE) {
                        double a1[3] = {m1[2], m1[5], m1[8]}, a2[3] = {m2[2], m2[5], m2[8]};
                        double x1, x2;
                        segment_closest(p1, a1, s11, p2, a2, s21, x1, x2);
                        double q1[3] = {p1[0] + a1[0] * x1, p1[1] + a1[1] * x1, p1[2] + a1[2] * x1};
                        double q2[3] = {p2[0] + a2[0] * x2, p2[1] + a2[1] * x2, p2[2] + a2[2] * x2};
                        n = sphere_sphere(rc[0], q1, s10, q2, s20, margin);
                    } else {
                        warn |= WARN_UNSUPPORTED_PAIR;
                    }
                }
            }
            /* ballot-compact in pair order */
            const unsigned long long m1b = wv::ballot(n >= 1), m2b = wv::ballot(n >= 2);
            const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
            int slot = ncon + wv::popc64(m1b & below) + wv::popc64(m2b & below);
            for (int c = 0; c < n; ++c, ++slot) {
                if (slot >= CM_MAXCON) continue;
                double fr[9];
                for (int i = 0; i < 3; ++i) { fr[i] = rc[c].normal[i]; fr[3 + i] = rc[c].tangent[i]; fr[6 + i] = 0; }
                make_frame(fr);
                S.c_dist[slot] = rc[c].dist;
                for (int i = 0; i < 3; ++i) S.c_pos[slot][i] = rc[c].pos[i];
                for (int i = 0; i < 9; ++i) S.c_frame[slot][i] = fr[i];
                S.c_g1[slot] = g1; S.c_g2[slot] = g2;
                S.c_margin[slot] = margin - gap;
                const int pa = m->geom_priority[g1], pb = m->geom_priority[g2];
                if (pa != pb) {
                    const int g = pa > pb ? g1 : g2;
                    S.c_dim[slot] = m->geom_condim[g];
                    for (int i = 0; i < 2; ++i) S.c_solref[slot][i] = m->geom_solref[g][i];
                    for (int i = 0; i < 5; ++i) S.c_solimp[slot][i] = m->geom_solimp[g][i];
                    for (int i = 0; i < 3; ++i) S.c_fri[slot][i] = m->geom_friction[g][i];
                } else {
                    S.c_dim[slot] = m->geom_condim[g1] > m->geom_condim[g2] ? m->geom_condim[g1] : m->geom_condim[g2];
                    const double s1 = m->geom_solmix[g1], s2 = m->geom_solmix[g2];
                    double mix;
                    if (s1 >= CM_MINVAL && s2 >= CM_MINVAL) mix = s1 / (s1 + s2);
                    else if (s1 < CM_MINVAL && s2 < CM_MINVAL) mix = 0.5;
                    else mix = s1 < CM_MINVAL ? 0.0 : 1.0;
                    if (m->geom_solref[g1][0] > 0 && m->geom_solref[g2][0] > 0)
                        for (int i = 0; i < 2; ++i) S.c_solref[slot][i] = mix * m->geom_solref[g1][i] + (1 - mix) * m->geom_solref[g2][i];
                    else
                        for (int i = 0; i < 2; ++i) S.c_solref[slot][i] = fmin(m->geom_solref[g1][i], m->geom_solref[g2][i]);
                    for (int i = 0; i < 5; ++i) S.c_solimp[slot][i] = mix * m->geom_solimp[g1][i] + (1 - mix) * m->geom_solimp[g2][i];
                    for (int i = 0; i < 3; ++i) S.c_fri[slot][i] = fmax(m->geom_friction[g1][i], m->geom_friction[g2][i]);
                }
            }
            ncon += wv::popc64(m1b) + wv::popc64(m2b);
        }
        if (ncon > CM_MAXCON) { ncon = CM_MAXCON; warn |= WARN_CONTACT_FULL; }

        CK_STAMP(5);
        /* ================= P6 velocity recursion: cvel, cdof_dot, bias acceleration ================= */
        if (b == 0) for (int i = 0; i < 6; ++i) { S.x.s.cvel[0][i] = 0; S.x.s.cacc[0][i] = (i < 3) ? 0.0 : -m->gravity[i - 3]; }
        wv::sync();
        double mycvel[6] = {0, 0, 0, 0, 0, 0}, mycacc[6] = {0, 0, 0, 0, 0, 0};
        for (int d = 1; d <= m->maxdepth; ++d) {
            if (depth == d) {
                const int p = m->body_parentid[b];
                for (int i = 0; i < 6; ++i) { mycvel[i] = S.x.s.cvel[p][i]; mycacc[i] = S.x.s.cacc[p][i]; }
                const int j0 = m->body_jntadr[b], jn = m->body_jntnum[b];
                for (int jj = 0; jj < jn; ++jj) {
                    const int j = j0 + jj, jt = m->jnt_type[j];
                    int da = m->jnt_dofadr[j], nrot = 1;
                    if (jt == CM_JNT_FREE) {
                        for (int kk = 0; kk < 3; ++kk) {
                            for (int i = 0; i < 6; ++i) S.x.s.cdof_dot[da + kk][i] = 0;
                            const double qv = S.qvel[da + kk];
                            for (int i = 0; i < 6; ++i) mycvel[i] += S.cdof[da + kk][i] * qv;
                        }
                        da += 3; nrot = 3;
                    } else if (jt == CM_JNT_BALL) nrot = 3;
                    double cdd[3][6];
                    for (int kk = 0; kk < nrot; ++kk) {
                        double c6[6];
                        for (int i = 0; i < 6; ++i) c6[i] = S.cdof[da + kk][i];
                        cross_motion(cdd[kk], mycvel, c6);
                        for (int i = 0; i < 6; ++i) S.x.s.cdof_dot[da + kk][i] = cdd[kk][i];
                    }
                    for (int kk = 0; kk < nrot; ++kk) {
                        const double qv = S.qvel[da + kk];
                        for (int i = 0; i < 6; ++i) { mycvel[i] += S.cdof[da + kk][i] * qv; mycacc[i] += cdd[kk][i] * qv; }
                    }
                }
                for (int i = 0; i < 6; ++i) { S.x.s.cvel[b][i] = mycvel[i]; S.x.s.cacc[b][i] = mycacc[i]; }
            }
            wv::sync();
        }
        /* body force = I*cacc + cvel x* (I*cvel); then subtree sums; then project on the dofs */
        if (isbody) {
            double f6[6] = {0, 0, 0, 0, 0, 0};
            if (b > 0) {
                double ci[10], t1[6], t2[6], t3[6];
                for (int i = 0; i < 10; ++i) ci[i] = S.x.s.cinert[b][i];
                mul_inert_vec(t1, ci, mycacc);
                mul_inert_vec(t2, ci, mycvel);
                cross_force(t3, mycvel, t2);
                for (int i = 0; i < 6; ++i) f6[i] = t1[i] + t3[i];
            }
            for (int i = 0; i < 6; ++i) S.x.s.cfrc[b][i] = f6[i];
        }
        /* sensor sites: remember their body's com-frame velocity before the tiles are recycled */
        if (lane < m->nsite) for (int i = 0; i < 6; ++i) S.site_cvel[lane][i] = S.x.s.cvel[m->site_bodyid[lane]][i];
        wv::sync();
        double qfrc_bias = 0;
        if (isdof) {
            double acc[6] = {0, 0, 0, 0, 0, 0};
            const int e = m->body_subtreeend[kbody];
            for (int c = kbody; c < e; ++c) for (int i = 0; i < 6; ++i) acc[i] += S.x.s.cfrc[c][i];
            for (int i = 0; i < 6; ++i) qfrc_bias += cd[i] * acc[i];
        }

        CK_STAMP(6);
        /* ================= P6/P7/P8 passive + actuation -> qfrc_smooth (lane = dof) ================= */
        if (isdof) {
            const int jt = m->jnt_type[kjnt];
            double f = -m->dof_damping[k_] * S.qvel[k_];
            if ((jt == CM_JNT_HINGE || jt == CM_JNT_SLIDE) && m->jnt_stiffness[kjnt] != 0) {
                const int qa = m->jnt_qposadr[kjnt];
                f += -m->jnt_stiffness[kjnt] * (S.qpos[qa] - m->qpos_spring[qa]);
            }
            f -= qfrc_bias;
            if (io.qfrc_applied) f += io.qfrc_applied[(size_t)env * io.sv + k_];
            for (int u = 0; u < nu; ++u) {
                if (m->act_dofid[u] != k_) continue;
                double c = S.ctrl[u];
                if (m->act_ctrllimited[u]) c = clampd(c, m->act_ctrlrange[u][0], m->act_ctrlrange[u][1]);
                f += m->act_gear[u] * c;
            }
            S.qfrc_smooth[k_] = f;
        }
        if (io.xfrc_applied) {
            /* Cartesian perturbations: [force, torque] at the body's inertial origin */
            for (int e = lane; e < nbody * 6; e += WV_WAVE) S.xfrc[e / 6][e % 6] = io.xfrc_applied[((size_t)env * io.sb) * 6 + e];
            wv::sync();
            if (isdof) {
                double f = 0;
                for (int bb = 1; bb < nbody; ++bb) {
                    if (!((m->body_dofmask[bb] >> k_) & 1ull)) continue;
                    const double *xf = S.xfrc[bb];
                    if (xf[0] == 0 && xf[1] == 0 && xf[2] == 0 && xf[3] == 0 && xf[4] == 0 && xf[5] == 0) continue;
                    const double *c = S.com[m->body_rootid[bb]];
                    double off[3] = {S.x.s.xipos[bb][0] - c[0], S.x.s.xipos[bb][1] - c[1], S.x.s.xipos[bb][2] - c[2]};
                    double t[3];
                    cross3(t, cd, off);
                    for (int i = 0; i < 3; ++i) f += (cd[3 + i] + t[i]) * xf[i] + cd[i] * xf[3 + i];
                }
                S.qfrc_smooth[k_] += f;
            }
        }
        wv::sync();

        CK_STAMP(7);
        /* ================= P5 constraint rows: lane = row ================= */
        /* row descriptor assignment is wave-uniform bookkeeping; every lane keeps its own row */
        const int r_ = lane;
        int rtype = -1, rid = 0, rsub = 0;
        int nefc = 0;
        for (int e = 0; e < m->neq; ++e) {
            if (!m->eq_active[e]) continue;
            if (nefc + 3 > CM_MAXEFC) { warn |= WARN_CONSTRAINT_FULL; continue; }
            if (r_ >= nefc && r_ < nefc + 3) { rtype = CM_CNSTR_EQUALITY; rid = e; rsub = r_ - nefc; }
            nefc += 3;
        }
        for (int j = 0; j < njnt; ++j) {
            if (!m->jnt_limited[j]) continue;
            const int jt = m->jnt_type[j];
            if (jt != CM_JNT_HINGE && jt != CM_JNT_SLIDE) continue;
            const double q = S.qpos[m->jnt_qposadr[j]], mg = m->jnt_margin[j];
            for (int side = 0; side < 2; ++side) {
                const double dist = side == 0 ? q - m->jnt_range[j][0] : m->jnt_range[j][1] - q;
                if (dist < mg) {
                    if (nefc >= CM_MAXEFC) { warn |= WARN_CONSTRAINT_FULL; continue; }
                    if (r_ == nefc) { rtype = CM_CNSTR_LIMIT_JOINT; rid = j; rsub = side; }
                    ++nefc;
                }
            }
        }
        for (int c = 0; c < ncon; ++c) {
            const int dim = S.c_dim[c];
            if (dim != 1 && dim != 3) { warn |= WARN_UNSUPPORTED_PAIR; continue; }
            const int nrow = dim == 1 ? 1 : 2 * (dim - 1);
            if (nefc + nrow > CM_MAXEFC) { warn |= WARN_CONSTRAINT_FULL; continue; }
            if (r_ >= nefc && r_ < nefc + nrow) {
                rtype = dim == 1 ? CM_CNSTR_CONTACT_FRICTIONLESS : CM_CNSTR_CONTACT_PYRAMIDAL;
                rid = c; rsub = r_ - nefc;
            }
            nefc += nrow;
        }

        /* per-row geometry: J_rk = plus_k (u.lin_k + wp.ang_k) - minus_k (u.lin_k + wm.ang_k) (+ sgn at one dof) */
        double u3[3] = {0, 0, 0}, wp[3] = {0, 0, 0}, wm[3] = {0, 0, 0};
        unsigned long long maskp = 0, maskm = 0;
        int limdof = -1;
        double limsgn = 0, rpos = 0, rmargin = 0, rdiag = 0, imp_pos = 0, rRscale = 1.0;
        double solref0 = 0.02, solref1 = 1, solimp[5] = {0.9, 0.95, 0.001, 0.5, 2};
        if (rtype == CM_CNSTR_EQUALITY) {
            const int b1 = m->eq_body1[rid], b2 = m->eq_body2[rid];
            double l1[3] = {m->eq_data[rid][0], m->eq_data[rid][1], m->eq_data[rid][2]};
            double l2[3] = {m->eq_data[rid][3], m->eq_data[rid][4], m->eq_data[rid][5]};
            double a1[3], a2[3];
            mulmatvec3(a1, S.x.s.xmat[b1], l1);
            mulmatvec3(a2, S.x.s.xmat[b2], l2);
            for (int i = 0; i < 3; ++i) { a1[i] += S.x.s.xpos[b1][i]; a2[i] += S.x.s.xpos[b2][i]; }
            u3[rsub] = 1.0;
            const double *c1 = S.com[m->body_rootid[b1]], *c2 = S.com[m->body_rootid[b2]];
            double o1[3] = {a1[0] - c1[0], a1[1] - c1[1], a1[2] - c1[2]};
            double o2[3] = {a2[0] - c2[0], a2[1] - c2[1], a2[2] - c2[2]};
            cross3(wp, o1, u3);
            cross3(wm, o2, u3);
            maskp = m->body_dofmask[b1]; maskm = m->body_dofmask[b2];
            double res[3] = {a1[0] - a2[0], a1[1] - a2[1], a1[2] - a2[2]};
            rpos = res[rsub]; rmargin = 0;
            imp_pos = sqrt(dot3(res, res));
            rdiag = m->body_invweight0[b1][0] + m->body_invweight0[b2][0];
            solref0 = m->eq_solref[rid][0]; solref1 = m->eq_solref[rid][1];
            for (int i = 0; i < 5; ++i) solimp[i] = m->eq_solimp[rid][i];
        } else if (rtype == CM_CNSTR_LIMIT_JOINT) {
            const double q = S.qpos[m->jnt_qposadr[rid]];
            rpos = rsub == 0 ? q - m->jnt_range[rid][0] : m->jnt_range[rid][1] - q;
            rmargin = m->jnt_margin[rid];
            imp_pos = rpos;
            limdof = m->jnt_dofadr[rid];
            limsgn = rsub == 0 ? 1.0 : -1.0;
            rdiag = m->dof_invweight0[limdof];
            solref0 = m->jnt_solref[rid][0]; solref1 = m->jnt_solref[rid][1];
            for (int i = 0; i < 5; ++i) solimp[i] = m->jnt_solimp[rid][i];
        } else if (rtype >= 0) {
            const int c = rid;
            const int b1 = m->geom_bodyid[S.c_g1[c]], b2 = m->geom_bodyid[S.c_g2[c]];
            const double *fr = S.c_frame[c];
            double mu = 0;
            if (rtype == CM_CNSTR_CONTACT_PYRAMIDAL) {
                const int a = 1 + rsub / 2;
                mu = a <= 2 ? S.c_fri[c][0] : (a == 3 ? S.c_fri[c][1] : S.c_fri[c][2]);
                const double sg = (rsub & 1) ? -mu : mu;
                for (int i = 0; i < 3; ++i) u3[i] = fr[i] + sg * fr[3 * a + i];
                const double mu0 = S.c_fri[c][0];
                rRscale = 2 * mu0 * mu0; /* all pyramid rows use R of the first row, times 2 mu^2 */
            } else {
                for (int i = 0; i < 3; ++i) u3[i] = fr[i];
            }
            const double *c1 = S.com[m->body_rootid[b1]], *c2 = S.com[m->body_rootid[b2]];
            double o1[3] = {S.c_pos[c][0] - c1[0], S.c_pos[c][1] - c1[1], S.c_pos[c][2] - c1[2]};
            double o2[3] = {S.c_pos[c][0] - c2[0], S.c_pos[c][1] - c2[1], S.c_pos[c][2] - c2[2]};
            cross3(wp, o2, u3);
            cross3(wm, o1, u3);
            maskp = b2 > 0 ? m->body_dofmask[b2] : 0ull;
            maskm = b1 > 0 ? m->body_dofmask[b1] : 0ull;
            rpos = S.c_dist[c]; rmargin = S.c_margin[c]; imp_pos = rpos;
            const double tran = m->body_invweight0[b1][0] + m->body_invweight0[b2][0];
            /* the regulariser of every pyramid row derives from the FIRST row's diagApprox */
            const double mu_first = S.c_fri[c][0];
            rdiag = rtype == CM_CNSTR_CONTACT_PYRAMIDAL ? tran + mu_first * mu_first * tran : tran;
            solref0 = S.c_solref[c][0]; solref1 = S.c_solref[c][1];
            for (int i = 0; i < 5; ++i) solimp[i] = S.c_solimp[c][i];
        }
        double rR = 1.0, rK = 0, rB = 0, rimp = 1.0;
        if (rtype >= 0) {
            rimp = impedance(solimp, imp_pos, rmargin);
            rR = fmax(CM_MINVAL, (1 - rimp) * rdiag / rimp);
            if (rtype == CM_CNSTR_CONTACT_PYRAMIDAL) rR = fmax(CM_MINVAL, rRscale * rR);
            const double dmax = solimp[1];
            if (solref0 > 0) {
                double tc = solref0;
                if ((m->flags & CM_FLAG_REFSAFE) && tc < 2 * h) tc = 2 * h;
                rK = 1.0 / fmax(CM_MINVAL, dmax * dmax * tc * tc * solref1 * solref1);
                rB = 2.0 / fmax(CM_MINVAL, dmax * tc);
            } else {
                rK = -solref0 / fmax(CM_MINVAL, dmax * dmax);
                rB = -solref1 / fmax(CM_MINVAL, dmax);
            }
        }
        /* fill Y^T (still holding J^T) one dof at a time; accumulate J.qvel and J.qacc_warmstart */
        double jvel = 0, jws = 0;
        for (int k = 0; k < nv; ++k) {
            double v = 0;
            if (rtype >= 0) {
                const double c0 = S.cdof[k][0], c1 = S.cdof[k][1], c2 = S.cdof[k][2];
                const double c3 = S.cdof[k][3], c4 = S.cdof[k][4], c5 = S.cdof[k][5];
                const double ul = u3[0] * c3 + u3[1] * c4 + u3[2] * c5;
                if ((maskp >> k) & 1ull) v += ul + wp[0] * c0 + wp[1] * c1 + wp[2] * c2;
                if ((maskm >> k) & 1ull) v -= ul + wm[0] * c0 + wm[1] * c1 + wm[2] * c2;
                if (k == limdof) v = limsgn;
                jvel += v * S.qvel[k];
                jws += v * S.qacc_ws[k];
            } else if (r_ == NROW - 1) {
                v = S.qfrc_smooth[k];
            }
            S.Yt[k][r_] = v;
        }
        for (int k = nv; k < NVP; ++k) S.Yt[k][r_] = 0.0;
        const double raref = rtype >= 0 ? -rB * jvel - rK * rimp * (rpos - rmargin) : 0.0;

        /* ---- sensors, part 1 (lane = sensor): everything that does not need qacc ---- */
        const bool issens = lane < m->nsensor;
        const int stype = issens ? m->sensor_type[lane] : -1;
        const int sobj = issens ? m->sensor_objid[lane] : 0;
        double sout[4] = {0, 0, 0, 0};
        double acc_lin[3] = {0, 0, 0}, acc_ang[3] = {0, 0, 0}, acc_dif[3] = {0, 0, 0};
        if (issens) {
            if (stype == CM_SENS_ACTUATORPOS) sout[0] = m->act_gear[sobj] * S.qpos[m->act_qposadr[sobj]];
            else if (stype == CM_SENS_JOINTPOS) sout[0] = S.qpos[m->jnt_qposadr[sobj]];
            else if (stype == CM_SENS_FRAMEQUAT) {
                double sq[4] = {m->site_quat[sobj][0], m->site_quat[sobj][1], m->site_quat[sobj][2], m->site_quat[sobj][3]};
                mulquat(sout, S.x.s.xquat[m->site_bodyid[sobj]], sq);
            } else if (stype == CM_SENS_GYRO) mulmatTvec3(sout, S.site_xmat[sobj], S.site_cvel[sobj]);
            else if (stype == CM_SENS_MAGNETOMETER) {
                double mg[3] = {m->magnetic[0], m->magnetic[1], m->magnetic[2]};
                mulmatTvec3(sout, S.site_xmat[sobj], mg);
            } else if (stype == CM_SENS_ACCELEROMETER) {
                const int sb = m->site_bodyid[sobj];
                /* velocity-product part of the body's com-frame acceleration (incl. -gravity) */
                for (int i = 0; i < 3; ++i) { acc_ang[i] = S.x.s.cacc[sb][i]; acc_lin[i] = S.x.s.cacc[sb][3 + i]; }
                const double *c = S.com[m->body_rootid[sb]];
                for (int i = 0; i < 3; ++i) acc_dif[i] = S.site_xpos[sobj][i] - c[i];
            }
        }
        if (io.xpos_out && isbody) {
            for (int i = 0; i < 3; ++i) io.xpos_out[((size_t)env * io.sb + b) * 3 + i] = S.x.s.xpos[b][i];
            if (io.xquat_out) for (int i = 0; i < 4; ++i) io.xquat_out[((size_t)env * io.sb + b) * 4 + i] = S.x.s.xquat[b][i];
        }
        wv::sync(); /* every reader of the body-stage tiles is done: region x becomes A */

        CK_STAMP(8);
        /* ================= half solve: Y = D^-1/2 L^-T [J^T | qfrc_smooth], lane = column ================= */
        for (int k = nv - 1; k >= 0; --k) {
            const double xk = S.Yt[k][r_];
            for (int i = m->dof_parentid[k]; i >= 0; i = m->dof_parentid[i]) S.Yt[i][r_] -= S.LD[k][i] * xk;
            S.Yt[k][r_] = xk * S.rsd[k];
        }
        wv::sync();

        CK_STAMP(9);
        /* ================= P9: A = Y^T Y + diag(R), b = Y^T y63 - aref (lane = column s) ================= */
        double yown[NVP];
#pragma unroll
        for (int k = 0; k < NVP; ++k) yown[k] = S.Yt[k][r_];
        for (int r = 0; r < nefc; ++r) {
            double acc = 0;
#pragma unroll
            for (int k = 0; k < NVP; ++k) acc += S.Yt[k][r] * yown[k];
            if (r == r_) acc += rR;
            S.x.A[r][r_] = acc;
        }
        double rb = 0;
        {
            double acc = 0;
#pragma unroll
            for (int k = 0; k < NVP; ++k) acc += S.Yt[k][NROW - 1] * yown[k];
            rb = acc - raref;
        }
        wv::sync();

        CK_STAMP(10);
        /* ================= P10: warm start + projected Gauss-Seidel, one row per lane ================= */
        const bool isrow = rtype >= 0;
        const bool clampf = isrow && rtype != CM_CNSTR_EQUALITY;
        double Aii = isrow ? S.x.A[r_][r_] : 1.0;
        const double invAii = 1.0 / Aii;
        double f = 0, res = isrow ? rb : 0.0;
        int iters = 0;
        if (nefc > 0) {
            if (m->flags & CM_FLAG_WARMSTART) {
                if (isrow) {
                    f = -(jws - raref) / rR;
                    if (clampf && f < 0) f = 0;
                }
                double af = 0;
                for (int t = 0; t < nefc; ++t) af += S.x.A[t][r_] * wv::readlane(f, t);
                double cost = wv::wave_sum(isrow ? f * (rb + 0.5 * af) : 0.0);
                if (cost > 0) f = 0;
                else if (isrow) res = rb + af;
            }
            const double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
            while (iters < m->iterations) {
                double improvement = 0;
                for (int i = 0; i < nefc; ++i) {
                    const double arow = S.x.A[i][r_];
                    double fn = f - res * invAii;
                    if (clampf && fn < 0) fn = 0;
                    double delta = fn - f;
                    double change = 0.5 * delta * delta * Aii + delta * res;
                    if (change > 1e-10) { delta = 0; change = 0; }
                    if (r_ != i) { delta = 0; change = 0; }
                    f += delta;
                    improvement -= change;
                    const double dlt = wv::readlane(delta, i);
                    if (isrow) res += arow * dlt;
                }
                improvement = wv::wave_sum(improvement) * scale;
                ++iters;
                if (improvement < m->tolerance) break;
            }
        }

        CK_STAMP(11);
        /* ================= qacc = L^-1 D^-1/2 (y63 + Y f)  (lane = dof) ================= */
        double qacc;
        {
            double z = isdof ? S.Yt[k_][NROW - 1] : 0.0;
            for (int r = 0; r < nefc; ++r) {
                const double fr = wv::readlane(f, r);
                if (isdof) z += S.Yt[k_][r] * fr;
            }
            if (isdof) z *= S.rsd[k_];
            for (int i = 0; i < nv - 1; ++i) {
                const double zi = wv::readlane(z, i);
                if (isdof && k_ > i) z -= S.LD[k_][i] * zi;
            }
            qacc = z;
        }
        {
            const bool badv = isdof && (!(qacc == qacc) || fabs(qacc) > 1e10);
            if (wv::ballot(badv) != 0ull) { warn |= WARN_DIVERGED; break; }
        }
        if (isdof) S.qacc[k_] = qacc;
        wv::sync();

        CK_STAMP(12);
        /* ---- sensors, part 2: accelerometer needs qacc; cutoffs; store ---- */
        if (issens) {
            if (stype == CM_SENS_ACCELEROMETER) {
                const int sb = m->site_bodyid[sobj];
                const unsigned long long mk = m->body_dofmask[sb];
                for (int k = 0; k < nv; ++k) {
                    if (!((mk >> k) & 1ull)) continue;
                    const double qa = S.qacc[k];
                    for (int i = 0; i < 3; ++i) { acc_ang[i] += S.cdof[k][i] * qa; acc_lin[i] += S.cdof[k][3 + i] * qa; }
                }
                double t[3], lin[3], vlin[3], corr[3];
                cross3(t, acc_dif, acc_ang);
                for (int i = 0; i < 3; ++i) lin[i] = acc_lin[i] - t[i];
                const double *cv = S.site_cvel[sobj];
                cross3(t, acc_dif, cv);
                for (int i = 0; i < 3; ++i) vlin[i] = cv[3 + i] - t[i];
                cross3(corr, cv, vlin);
                for (int i = 0; i < 3; ++i) lin[i] += corr[i];
                mulmatTvec3(sout, S.site_xmat[sobj], lin);
            }
            const double cut = m->sensor_cutoff[lane];
            const int dim = m->sensor_dim[lane], adr = m->sensor_adr[lane];
            for (int i = 0; i < dim; ++i) {
                double v = sout[i];
                if (cut > 0 && stype != CM_SENS_FRAMEQUAT) v = clampd(v, -cut, cut);
                io.sensordata[(size_t)env * io.ssd + adr + i] = v;
            }
        }
        if (lane < nu) io.actuator_velocity[(size_t)env * io.su + lane] = m->act_gear[lane] * S.qvel[m->act_dofid[lane]];
        if (io.info && lane == 0) {
            io.info[(size_t)env * 4 + 0] = ncon; io.info[(size_t)env * 4 + 1] = nefc;
            io.info[(size_t)env * 4 + 2] = iters; io.info[(size_t)env * 4 + 3] = 0;
        }
        if (isdof) io.qacc[(size_t)env * io.sv + k_] = qacc;
        if (!io.integrate) break;

        CK_STAMP(13);
        /* ================= P12 semi-implicit Euler with implicit joint damping ================= */
        double qacc_int = qacc;
        if (m->flags & CM_FLAG_EULERDAMP) {
            /* (M + hB) x = M qacc  <=>  x = qacc - (M + hB)^-1 (hB qacc) */
            double w = isdof ? h * m->dof_damping[k_] * qacc : 0.0;
            w = solve_single<NVP>(S.LDH, w, nv, lane);
            qacc_int = qacc - w;
        }
        if (isdof) {
            S.qvel[k_] += h * qacc_int;
            S.qacc_ws[k_] = qacc;
        }
        wv::sync();
        if (lane < njnt) {
            const int j = lane, jt = m->jnt_type[j];
            int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
            if (jt == CM_JNT_HINGE || jt == CM_JNT_SLIDE) {
                S.qpos[qa] += h * S.qvel[da];
            } else {
                if (jt == CM_JNT_FREE) {
                    for (int i = 0; i < 3; ++i) S.qpos[qa + i] += h * S.qvel[da + i];
                    qa += 3; da += 3;
                }
                double ax[3] = {S.qvel[da], S.qvel[da + 1], S.qvel[da + 2]};
                const double ang = h * normalize3(ax);
                double qr[4], q[4] = {S.qpos[qa], S.qpos[qa + 1], S.qpos[qa + 2], S.qpos[qa + 3]};
                if (ang == 0) { qr[0] = 1; qr[1] = qr[2] = qr[3] = 0; }
                else axisangle2quat(qr, ax, ang);
                normalize4(q);
                mulquat(q, q, qr);
                for (int i = 0; i < 4; ++i) S.qpos[qa + i] = q[i];
            }
        }
        time += h;
        wv::sync();
        CK_STAMP(14);
    }

    /* ---------------- store state ---------------- */
    if (io.integrate) {
        if (lane < nq) io.qpos[(size_t)env * io.sq + lane] = S.qpos[lane];
        if (lane < nv) {
            io.qvel[(size_t)env * io.sv + lane] = S.qvel[lane];
            io.qacc_warmstart[(size_t)env * io.sv + lane] = S.qacc_ws[lane];
        }
        if (lane == 0) io.time[env] = time;
    }
    {
        int w = 0;
        for (int bit = 1; bit <= 8; bit <<= 1)
            if (wv::ballot((warn & bit) != 0) != 0ull) w |= bit;
        if (lane == 0 && w) io.warn[env] |= w;
    }
}

/* one single-wave workgroup per environment */
template <int NVP>
WV_GLOBAL void __launch_bounds__(WV_WAVE) cassie_step_kernel(PhysIO io) {
    WV_SHARED EnvShared<NVP> S;
    const int env = wv::env_id();
    if (env >= io.nenv) return;
    env_step<NVP>(io, S, env);
}

}  // namespace ck
#endif
