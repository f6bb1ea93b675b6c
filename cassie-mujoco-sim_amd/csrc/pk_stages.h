/*
 * pk_stages.h -- the stage functions both wave forms share: drive-level I/O (H6 / H7), mass-matrix group, bias / passive forces, sensor stage, outputs, Euler step
 * (part of the step kernel: included by physics_kernel.h, in this order, inside nothing; see there for the design)
 */
#ifndef CASSIE_PK_STAGES_H
#define CASSIE_PK_STAGES_H

namespace ck {

/* ---------------------------------------------------- drive-level I/O (H6 / H7) ---- */
/* sensordata slots of the ten drive encoders and the six joint encoders (reference src/cassiemujoco.c:754-755) */
WV_DEVICE int drive_sensor_slot(int i) { return i < 5 ? i : i + 3; }   /* 0 1 2 3 4 8 9 10 11 12 */
WV_DEVICE int joint_sensor_slot(int j) { return j < 3 ? j + 5 : j + 10; } /* 5 6 7 13 14 15 */

/* One cassie_motor_data + cassie_sensor_data pass for one env, lanes = drives (0..9), joint encoders (10..15), IMU
 * words (16..28).  Every floating-point operation is individually rounded in the order the reference's C performs it
 * (drive_encoder :558-593, joint_encoder :596-635, motor :638-664), and the FIR runs in 32-bit integers, so with
 * identical sensordata / actuator_velocity in, the measurement block, the filter histories, the delay lines and the
 * ctrl values are bit for bit those of the host chain (csrc/cassie_hostpath.c, itself pinned to the reference's own
 * compiled code by tests/test_hostpath.py). */
#define WV_DRIVE_FN WV_DEVICE
/* the env's drive-level state between HBM (cm_drive_state_t + the measurement block) and the launch's LDS copy */
template <class SH>
WV_DEVICE void drive_state_load(const PhysIO &io, SH &S, int env, int lane) {
    const cm_drive_state_t *ds = io.drive_state + env;
    const double *meas = io.meas + (size_t)env * CM_MEAS_DIM;
    if (lane < CM_NUM_DRIVES) {
        for (int k = 0; k < CM_DRIVE_FILTER_NB; ++k) S.drv_x[lane][k] = ds->drive_x[lane][k];
        for (int k = 0; k < CM_TORQUE_DELAY_CYCLES; ++k) S.drv_delay[lane][k] = ds->torque_delay[lane][k];
        S.drv_pos[lane] = meas[CM_MEAS_DRIVE_POS + lane]; S.drv_vel[lane] = meas[CM_MEAS_DRIVE_VEL + lane];
        if (lane == 0) S.drv_msg[0] = ds->safety_msg;
    } else if (lane < CM_NUM_DRIVES + CM_NUM_JOINTS) {
        const int j = lane - CM_NUM_DRIVES;
        for (int k = 0; k < CM_JOINT_FILTER_NB; ++k) S.drv_jx[j][k] = ds->joint_x[j][k];
        for (int k = 0; k < CM_JOINT_FILTER_NA; ++k) S.drv_jy[j][k] = ds->joint_y[j][k];
    }
}
template <class SH>
WV_DEVICE void drive_state_store(const PhysIO &io, SH &S, int env, int lane) {
    cm_drive_state_t *ds = io.drive_state + env;
    if (lane < CM_NUM_DRIVES) {
        for (int k = 0; k < CM_DRIVE_FILTER_NB; ++k) ds->drive_x[lane][k] = S.drv_x[lane][k];
        for (int k = 0; k < CM_TORQUE_DELAY_CYCLES; ++k) ds->torque_delay[lane][k] = S.drv_delay[lane][k];
        if (lane == 0) ds->safety_msg = S.drv_msg[0];
    } else if (lane < CM_NUM_DRIVES + CM_NUM_JOINTS) {
        const int j = lane - CM_NUM_DRIVES;
        for (int k = 0; k < CM_JOINT_FILTER_NB; ++k) ds->joint_x[j][k] = S.drv_jx[j][k];
        for (int k = 0; k < CM_JOINT_FILTER_NA; ++k) ds->joint_y[j][k] = S.drv_jy[j][k];
    }
}

/* Once per launch: the constants of the env's drive-level passes, into LDS.  The derived ones (no-load speed in rad/s, encoder
 * scale) are computed here by the same individually rounded operations, in the same order, as the reference computes them on
 * every call -- so the passes read the very bits they used to compute, without three divisions and a trip to the model and
 * to the command arrays per substep. */
enum { DRVC_RATIO = 0, DRVC_TMAX, DRVC_WMAX, DRVC_COUNTS, DRVC_SCALE, DRVC_U_OR_PT, DRVC_STO_OR_DT, DRVC_FF, DRVC_KP, DRVC_KD };
template <class SH>
WV_DEVICE void drive_consts_load(const PhysIO &io, SH &S, ModelPtr m, int env, int lane) {
    const double TWO_PI = 2 * 3.14159265358979323846, PI = 3.14159265358979323846;
    const int nu = m->nu;
    if (lane < CM_NUM_DRIVES) {
        const int i = lane, bits = m->sensor_bits[drive_sensor_slot(i)];
        const double ratio = m->act_gear[i], counts = (double)(1 << bits);
        double *c = S.drv_c[i];
        c[DRVC_RATIO] = ratio; c[DRVC_TMAX] = m->act_ctrlrange[i][1];
        c[DRVC_WMAX] = wv::div_rn(wv::mul_rn(wv::mul_rn(m->act_maxrpm[i], 2.0), PI), 60.0);
        c[DRVC_COUNTS] = counts; c[DRVC_SCALE] = wv::div_rn(wv::div_rn(TWO_PI, counts), ratio);
        if (io.drive_mode == CM_DRIVE_TORQUE) {
            c[DRVC_U_OR_PT] = io.drive_cmd[(size_t)env * (nu + 1) + i];
            c[DRVC_STO_OR_DT] = io.drive_cmd[(size_t)env * (nu + 1) + nu] != 0.0 ? 1.0 : 0.0;
            c[DRVC_FF] = 0.0; c[DRVC_KP] = 0.0; c[DRVC_KD] = 0.0;
        } else {
            const size_t o = (size_t)env * nu + i;
            c[DRVC_U_OR_PT] = io.pd_ptarget[o]; c[DRVC_STO_OR_DT] = io.pd_dtarget ? io.pd_dtarget[o] : 0.0;
            c[DRVC_FF] = io.pd_torque ? io.pd_torque[o] : 0.0; c[DRVC_KP] = io.pd_kp[o]; c[DRVC_KD] = io.pd_kd[o];
            /* CM_DRIVE_PD_SAFE: the STO switch (radio channel 8 of cassie_out_t) travels in the last word of the env's drive command */
            if (i == 0) S.drv_msg[1] = (io.drive_mode == CM_DRIVE_PD_SAFE && io.drive_cmd && io.drive_cmd[(size_t)env * (nu + 1) + nu] != 0.0) ? 1 : 0;
        }
    } else if (lane < CM_NUM_DRIVES + CM_NUM_JOINTS) {
        const int j = lane - CM_NUM_DRIVES, bits = m->sensor_bits[joint_sensor_slot(j)];
        const double counts = (double)(1 << bits);
        S.drv_jc[j][0] = counts; S.drv_jc[j][1] = wv::div_rn(TWO_PI, counts);
    }
}

template <class SH>
WV_DRIVE_FN void drive_level_io(const PhysIO &io, SH &S, ModelPtr m, int env, int lane, bool write_meas) {
    const double TWO_PI = 2 * 3.14159265358979323846, PI = 3.14159265358979323846;
    double *meas = io.meas + (size_t)env * CM_MEAS_DIM;
    /* CM_DRIVE_PD_SAFE: cassie_core_sim's safety layer between pd_input's PD law and the motor model (pk_safety.h).  A drive's
     * torque depends on ALL ten measured positions (the 22 limit constraints) and the message bits on all ten torques: every
     * drive's lane reads the ten positions from LDS and takes its own torque, a ballot collects the torque-limit bit, and the wave
     * meets once before any lane overwrites the measurements the others have just read. */
    double u_safe = 0.0;
    if (io.drive_mode == CM_DRIVE_PD_SAFE) {
        int msg = 0;
        const bool sto = S.drv_msg[1] != 0;
        if (lane < CM_NUM_DRIVES) {
            double qq[CM_NUM_DRIVES];
            for (int k = 0; k < CM_NUM_DRIVES; ++k) qq[k] = S.drv_pos[k];
            const double *c = S.drv_c[lane];
            const double wk = S.drv_vel[lane];
            const double uk = wv::add_rn(wv::add_rn(c[DRVC_FF], wv::mul_rn(c[DRVC_KP], wv::sub_rn(c[DRVC_U_OR_PT], S.drv_pos[lane]))), wv::mul_rn(c[DRVC_KD], wv::sub_rn(c[DRVC_STO_OR_DT], wk)));
            u_safe = safety::drive_torque(lane, uk, qq, wk, safety::torque_limit(lane), sto, &msg);
        }
        const bool lim = wv::ballot((msg & safety::MSG_LIMIT) != 0) != 0ull, trq = wv::ballot((msg & safety::MSG_TORQUE) != 0) != 0ull;
        wv::sync();
        if (lane == 0 && io.integrate) S.drv_msg[0] |= (lim ? safety::MSG_LIMIT : 0) | (trq ? safety::MSG_TORQUE : 0);
    }
    if (lane < CM_NUM_DRIVES) {
        const int i = lane;
        double cst[10];
        for (int k = 0; k < 10; ++k) cst[k] = S.drv_c[i][k];
        const double ratio = cst[DRVC_RATIO], tmax = cst[DRVC_TMAX], wmax = cst[DRVC_WMAX];
        /* the command: a drive torque from the caller, or pd_input's motor PD on the measurements of the previous step */
        double u;
        bool sto = false;
        if (io.drive_mode == CM_DRIVE_TORQUE) {
            u = cst[DRVC_U_OR_PT];
            sto = cst[DRVC_STO_OR_DT] != 0.0;
        } else if (io.drive_mode == CM_DRIVE_PD_SAFE) {
            u = u_safe;
            sto = S.drv_msg[1] != 0;      /* (cassie_motor_data reads the same radio channel, reference :784) */
        } else {
            const double p = S.drv_pos[i], v = S.drv_vel[i];
            const double pt = cst[DRVC_U_OR_PT], dt = cst[DRVC_STO_OR_DT], ff = cst[DRVC_FF];
            u = wv::add_rn(wv::add_rn(ff, wv::mul_rn(cst[DRVC_KP], wv::sub_rn(pt, p))), wv::mul_rn(cst[DRVC_KD], wv::sub_rn(dt, v)));
        }
        /* motor(): speed-torque curve, STO, delay line (reference :638-664) */
        const double w = S.actvel[i];
        double tlim = wv::mul_rn(wv::mul_rn(2.0, tmax), wv::sub_rn(1.0, wv::div_rn(fabs(w), wmax)));
        tlim = fmax(fmin(tlim, tmax), 0.0);
        if (sto) u = 0.0;
        const double tau = copysign(fmin(fabs(wv::div_rn(u, ratio)), tlim), u);
        double dl[CM_TORQUE_DELAY_CYCLES];
        for (int k = 0; k < CM_TORQUE_DELAY_CYCLES; ++k) dl[k] = S.drv_delay[i][k];
        const double ctrl_i = dl[CM_TORQUE_DELAY_CYCLES - 1];
        for (int k = CM_TORQUE_DELAY_CYCLES - 1; k > 0; --k) S.drv_delay[i][k] = dl[k - 1];
        S.drv_delay[i][0] = tau;
        S.ctrl[i] = ctrl_i;
        /* drive_encoder(): truncation to encoder counts, 9-tap integer FIR (reference :558-593) */
        const int slot = drive_sensor_slot(i);
        const double counts = cst[DRVC_COUNTS], scale = cst[DRVC_SCALE];
        const int ev = (int)wv::mul_rn(wv::div_rn(S.sens[slot], TWO_PI), counts);
        const double pos = wv::mul_rn((double)ev, scale);
        int x[CM_DRIVE_FILTER_NB];
        bool allzero = true;
        for (int k = 0; k < CM_DRIVE_FILTER_NB; ++k) { x[k] = S.drv_x[i][k]; allzero &= x[k] == 0; }
        if (allzero) for (int k = 0; k < CM_DRIVE_FILTER_NB; ++k) x[k] = ev;
        for (int k = CM_DRIVE_FILTER_NB - 1; k > 0; --k) x[k] = x[k - 1];
        x[0] = ev;
        const int fir[CM_DRIVE_FILTER_NB] = {2727, 534, -2658, -795, 72, 110, 19, -6, -3};
        int y = 0;
        for (int k = 0; k < CM_DRIVE_FILTER_NB; ++k) { y += x[k] * fir[k]; S.drv_x[i][k] = x[k]; }
        const double vel = wv::div_rn(wv::mul_rn((double)y, scale), PI);
        S.drv_pos[i] = pos; S.drv_vel[i] = vel;
        if (write_meas) {
            meas[CM_MEAS_DRIVE_POS + i] = pos; meas[CM_MEAS_DRIVE_VEL + i] = vel;
            meas[CM_MEAS_DRIVE_TORQUE + i] = wv::mul_rn(ctrl_i, ratio);
        }
    } else if (lane < CM_NUM_DRIVES + CM_NUM_JOINTS) {
        /* joint_encoder(): IIR on the quantised position (reference :596-635) */
        const int j = lane - CM_NUM_DRIVES, slot = joint_sensor_slot(j);
        const double counts = S.drv_jc[j][0], scale = S.drv_jc[j][1];
        const int ev = (int)wv::mul_rn(wv::div_rn(S.sens[slot], TWO_PI), counts);
        const double pos = wv::mul_rn((double)ev, scale);
        double x[CM_JOINT_FILTER_NB], yv[CM_JOINT_FILTER_NA];
        bool allzero = true;
        for (int k = 0; k < CM_JOINT_FILTER_NB; ++k) { x[k] = S.drv_jx[j][k]; allzero &= x[k] == 0; }
        for (int k = 0; k < CM_JOINT_FILTER_NA; ++k) yv[k] = S.drv_jy[j][k];
        if (allzero) for (int k = 0; k < CM_JOINT_FILTER_NB; ++k) x[k] = pos;
        for (int k = CM_JOINT_FILTER_NB - 1; k > 0; --k) x[k] = x[k - 1];
        x[0] = pos;
        for (int k = CM_JOINT_FILTER_NA - 1; k > 0; --k) yv[k] = yv[k - 1];
        const double fb[CM_JOINT_FILTER_NB] = {12.348, 12.348, -12.348, -12.348}, fa[CM_JOINT_FILTER_NA] = {1.0, -1.7658, 0.79045};
        double y0 = 0.0;
        for (int k = 0; k < CM_JOINT_FILTER_NB; ++k) y0 = wv::add_rn(y0, wv::mul_rn(x[k], fb[k]));
        for (int k = 1; k < CM_JOINT_FILTER_NA; ++k) y0 = wv::sub_rn(y0, wv::mul_rn(yv[k], fa[k]));
        yv[0] = y0;
        for (int k = 0; k < CM_JOINT_FILTER_NB; ++k) S.drv_jx[j][k] = x[k];
        for (int k = 0; k < CM_JOINT_FILTER_NA; ++k) S.drv_jy[j][k] = yv[k];
        if (write_meas) { meas[CM_MEAS_JOINT_POS + j] = pos; meas[CM_MEAS_JOINT_VEL + j] = y0; }
    } else if (lane < 29) {
        /* IMU words: orientation, angular velocity, linear acceleration, magnetic field (reference :769-773) */
        if (write_meas) meas[CM_MEAS_ORIENTATION + (lane - 16)] = S.sens[lane];
    }
}

/* FEAT selects the collision code a model needs (see env_step) */
enum { FEAT_HFIELD = 1, FEAT_WAVEPAIRS = 2, FEAT_ALL = 3 };

/* what a lane is, as a body and as a dof: model indices read once per launch and handed to the stage functions that both waves of
 * the two-wave form call */
struct LaneIds { int nbody, nv, broot, bend, kjnt, kbody, kjt, kda, kroot, kbend; unsigned long long kdesc; };

/* ---------------- the mass-matrix stage group: com of every kinematic tree, cinert, cdof, composite inertias, M's columns.
 * One-wave form: called in line by the substep loop, between the geoms and the factorisations.  Two-wave form: wave 1's
 * program calls it between the barriers F and X.  Reads the pose tiles (xmat, xipos, xanchor, xaxis), writes com, cinert, cdof,
 * crb and the buf tile; leaves the lane's columns of M and M + hB in col / colh.  (A function, not a lambda of env_step: a
 * closure over the lane variables that the stage boundaries re-derive would pin them in memory.) ---------------- */
template <int NVP, class TOPO, int FEAT, int NW, class SH>
WV_DEVICE void mass_matrix_columns(const PhysIO &io, SH &S, ModelPtr m, int env, const LaneIds &ids, const double pf_mass, const double (&pf_iner)[3],
                                   const double (&ximat)[9], double (&col)[NVP], double (&colh)[NVP]) {
    const int nbody = ids.nbody, nv = ids.nv, broot = ids.broot, bend = ids.bend, kjnt = ids.kjnt, kbody = ids.kbody, kjt = ids.kjt, kda = ids.kda, kroot = ids.kroot;
    const unsigned long long kdesc = ids.kdesc;
    int lane = wv::fresh_lane(), b = lane, k_ = lane;
    bool isbody = b < nbody, isdof = k_ < nv;
    /* where crb[body] . cdof goes between the composite inertias and M's columns: the buf tile -- except in the two-wave
     * height-field form, where wave 0's height-field result table lies over that tile at this time: there the joint
     * anchors / axes, which nothing reads once cdof is formed, give their place */
    constexpr bool cbuf_over_anchors = NW == 2 && (FEAT & FEAT_HFIELD) != 0;
    static_assert(!cbuf_over_anchors || NVP <= CM_MAXJNT, "crb . cdof (NVP x 6) must fit the xanchor + xaxis tiles");
    static_assert(offsetof(decltype(S.x.s), xaxis) - offsetof(decltype(S.x.s), xanchor) == sizeof(double) * CM_MAXJNT * 3, "xanchor and xaxis are contiguous");
    double (*const cbuf)[6] = cbuf_over_anchors ? reinterpret_cast<double (*)[6]>(&S.x.s.xanchor[0][0]) : S.x.s.buf;
    /* ================= com of every kinematic tree (wave reduction per root) ================= */
    const double bmass = (isbody && b > 0) ? pf_mass : 0.0;
    {
        /* one masked DPP tree reduction per kinematic tree (wave_sum returns the total in every lane) */
        const double px = isbody ? S.x.s.xipos[b < NB ? b : 0][0] : 0.0, py = isbody ? S.x.s.xipos[b < NB ? b : 0][1] : 0.0,
                     pz = isbody ? S.x.s.xipos[b < NB ? b : 0][2] : 0.0;
        for (int ri = 0; ri < m->nroot; ++ri) {
            const int r = m->root_body[ri], e = m->body_subtreeend[r];
            const double w = (isbody && b >= r && b < e) ? bmass : 0.0;
            const double sm = wv::wave_sum(w), sx = wv::wave_sum(w * px), sy = wv::wave_sum(w * py), sz = wv::wave_sum(w * pz);
            if (lane == 0) {
                if (sm < CM_MINVAL) { S.com[r][0] = S.x.s.xipos[r][0]; S.com[r][1] = S.x.s.xipos[r][1]; S.com[r][2] = S.x.s.xipos[r][2]; }
                else { const double inv = 1.0 / sm; S.com[r][0] = sx * inv; S.com[r][1] = sy * inv; S.com[r][2] = sz * inv; }
            }
        }
    }
    wv::sync();
    CK_STAMP(18);
    /* ================= cinert (lane = body), cdof (lane = dof) ================= */
    if (lane < NB) {
        double ci[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (isbody && b > 0) {
            const double I0 = pf_iner[0], I1 = pf_iner[1], I2 = pf_iner[2];
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
        for (int i = 0; i < 10; ++i) S.x.s.cinert[lane][i] = ci[i];
    }
    {
    double cd[6] = {0, 0, 0, 0, 0, 0};
    if (isdof) {
        const double *c = S.com[kroot];
        double off[3] = {c[0] - S.x.s.xanchor[kjnt][0], c[1] - S.x.s.xanchor[kjnt][1], c[2] - S.x.s.xanchor[kjnt][2]};
        const int sub_k = k_ - kda;
        if (kjt == CM_JNT_SLIDE) {
            for (int i = 0; i < 3; ++i) cd[3 + i] = S.x.s.xaxis[kjnt][i];
        } else if (kjt == CM_JNT_HINGE) {
            for (int i = 0; i < 3; ++i) cd[i] = S.x.s.xaxis[kjnt][i];
            cross3(cd + 3, cd, off);
        } else if (kjt == CM_JNT_FREE && sub_k < 3) {
            cd[3 + sub_k] = 1.0;
        } else {
            const int a = (kjt == CM_JNT_FREE) ? sub_k - 3 : sub_k;
            cd[0] = S.x.s.xmat[kbody][a]; cd[1] = S.x.s.xmat[kbody][3 + a]; cd[2] = S.x.s.xmat[kbody][6 + a];
            cross3(cd + 3, cd, off);
        }
    }
    if (lane < NVP) for (int i = 0; i < 6; ++i) S.cdof[lane][i] = cd[i]; /* zero rows past nv */
    }
    wv::sync();
    CK_STAMP(2);

    /* ================= P2 CRBA: composite inertias, then one COLUMN of M per lane ================= */
    /* composite inertias: crb_b = sum of cinert_c over the contiguous subtree range [b, bend): dense loop over all
     * bodies with a per-lane range predicate, operands staged four bodies at a time */
    /* A 0/1-weighted sum over bodies is a matrix product, W (body x body: c in b's subtree) times cinert (body x 10), and its
     * result layout on the matrix core -- lane l holds rows (l >> 4) + 4 v, column l & 15 -- is a layout the LDS tile can be
     * written in directly: 16 v_mfma_f64_16x16x4_f64 (two blocks of 16 bodies x eight blocks of four summands, even and odd
     * blocks in separate accumulators), the weights built from the subtree masks in registers, the summands single LDS reads.
     * (fma(1, x, acc) is acc + x, fma(0, x, acc) is acc: the sums are plain sums, in body order.) */
    {
        const int mi = lane & 15, mk = lane >> 4;
        const unsigned mine = (isbody && b > 0) ? (unsigned)(((1ull << bend) - 1ull) ^ ((1ull << b) - 1ull)) : 0u; /* bodies [b, bend) */
        const unsigned w0 = (unsigned)wv::shfl_i((int)mine, mi) >> mk, w1 = (unsigned)wv::shfl_i((int)mine, 16 + mi) >> mk;
        double bv[NB / 4];
#pragma unroll
        for (int kb = 0; kb < NB / 4; ++kb) { const double v = S.x.s.cinert[4 * kb + mk][mi < 10 ? mi : 0]; bv[kb] = mi < 10 ? v : 0.0; }
        wv::mfma_acc d0a = {{0, 0, 0, 0}}, d0b = {{0, 0, 0, 0}}, d1a = {{0, 0, 0, 0}}, d1b = {{0, 0, 0, 0}};
#pragma unroll
        for (int kb = 0; kb < NB / 4; kb += 2)
            wv::mfma_f64_16x16x4_x4((double)((w0 >> (4 * kb)) & 1u), bv[kb], d0a, (double)((w1 >> (4 * kb)) & 1u), bv[kb], d1a,
                                    (double)((w0 >> (4 * kb + 4)) & 1u), bv[kb + 1], d0b, (double)((w1 >> (4 * kb + 4)) & 1u), bv[kb + 1], d1b);
        wv::mfma_f64_drain4(d0a, d0b, d1a, d1b);
        if (mi < 10) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                S.x.s.crb[mk + 4 * v][mi] = d0a.c[v] + d0b.c[v];
                S.x.s.crb[16 + mk + 4 * v][mi] = d1a.c[v] + d1b.c[v];
            }
        }
    }
    wv::sync();
    CK_STAMP(19);
    if (lane < NVP) {
        double bf[6] = {0, 0, 0, 0, 0, 0}, cd[6];
        for (int i = 0; i < 6; ++i) cd[i] = S.cdof[lane][i];
        if (isdof) mul_inert_vec(bf, S.x.s.crb[kbody], cd);
        for (int i = 0; i < 6; ++i) cbuf[lane][i] = bf[i];
    }
    wv::sync();
    CK_STAMP(20);
    double cdm[6]; /* this lane's motion axis, fetched where it is used rather than carried in registers */
    /* armature and h * damping sit on the diagonal only: they are added where the pivots are read (wave-uniform
     * scalars there) instead of being selected into one lane-dependent entry of each column here */
    /* M[i][lane] = cdof_lane . (crb[body_i] cdof_i): the buf rows are broadcast reads, staged eight rows at a time so
     * the LDS latency is paid once per group instead of once per row */
    if constexpr (NVP == 32) {
        /* 32 columns on 64 lanes: lanes l and l + 32 both work for column l, on rows [0, 16) and [16, 32); the lower lane
         * takes the upper one's sixteen entries through the lane swap */
        const int hk = lane & 31, roff = lane < 32 ? 0 : 16;
        const unsigned hdesc = (unsigned)wv::shfl_i((int)(unsigned)kdesc, hk) >> roff; /* (kdesc: no bit at or past nv <= 32) */
        for (int i = 0; i < 6; ++i) cdm[i] = S.cdof[hk][i];
        const double (*bufr)[6] = &cbuf[roff];
        double part[16];
#pragma unroll
        for (int i0 = 0; i0 < 16; i0 += 8) {
            double bb[8][6];
#pragma unroll
            for (int ii = 0; ii < 8; ++ii)
#pragma unroll
                for (int t = 0; t < 6; ++t) bb[ii][t] = bufr[i0 + ii][t];
            wv::sched_fence();
#pragma unroll
            for (int ii = 0; ii < 8; ++ii) {
                const double v = (cdm[0] * bb[ii][0] + cdm[1] * bb[ii][1]) + (cdm[2] * bb[ii][2] + cdm[3] * bb[ii][3]) + (cdm[4] * bb[ii][4] + cdm[5] * bb[ii][5]);
                part[i0 + ii] = ((hdesc >> (i0 + ii)) & 1u) ? v : 0.0;
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const double up = wv::from_upper_half(part[i]);
            col[i] = part[i]; colh[i] = part[i];
            col[16 + i] = up; colh[16 + i] = up;
        }
    } else {
    for (int i = 0; i < 6; ++i) cdm[i] = S.cdof[lane < NVP ? lane : 0][i];
#pragma unroll
    for (int i0 = 0; i0 < NVP; i0 += 8) {
        double bb[8][6];
#pragma unroll
        for (int ii = 0; ii < 8; ++ii)
#pragma unroll
            for (int t = 0; t < 6; ++t) bb[ii][t] = cbuf[i0 + ii][t];
        wv::sched_fence();
#pragma unroll
        for (int ii = 0; ii < 8; ++ii) {
            const int i = i0 + ii;
            double v = (cdm[0] * bb[ii][0] + cdm[1] * bb[ii][1]) + (cdm[2] * bb[ii][2] + cdm[3] * bb[ii][3]) + (cdm[4] * bb[ii][4] + cdm[5] * bb[ii][5]);
            /* kdesc holds no bit at or past nv; with a compile-time topology the bound is a constant, not a branch */
            if (TOPO::is_static ? (i >= TOPO::nv || !((kdesc >> i) & 1ull)) : !(i < nv && ((kdesc >> i) & 1ull))) v = 0;
            col[i] = v;
            colh[i] = v;
        }
    }
    }
    if (io.ext && isdof) {
        cm_ext_t *ex = io.ext + env;
#pragma unroll
        for (int i = 0; i < NVP; ++i) if (i < nv && i >= k_) { const double v = (i == k_) ? col[i] + m->dof_armature[k_] : col[i]; ex->qM[i][k_] = v; ex->qM[k_][i] = v; }
    }
    CK_STAMP(3);
}

/* ---------------- bias forces projected on the motion axes, passive forces, actuation -> qfrc_smooth (lane = dof).  Reads the
 * cfrc tile the velocity stage left, cdof, qpos / qvel / ctrl; writes S.qfrc_smooth.  One-wave form: in line behind the
 * velocity stage.  Two-wave form: wave 1, behind its factorisations, once wave 0 has published the cfrc tile. ---------------- */
template <int NVP, bool ROLLED = false, class SH>
WV_DEVICE void bias_forces_and_qfrc_smooth(const PhysIO &io, SH &S, ModelPtr m, int env, const LaneIds &ids, const double kdamp, const double kstiff,
                                           const double kref, const double kgear, const double klo, const double khi, const int kq, const int ka) {
    const int nbody = ids.nbody, nv = ids.nv, kbody = ids.kbody, kbend = ids.kbend;
    int lane = wv::fresh_lane(), b = lane, k_ = lane;
    bool isbody = b < nbody, isdof = k_ < nv;
    /* lane = dof: project the subtree's force on the motion axis; subtree = contiguous body range [kbody, kbend) */
    double qfrc_bias = 0;
    if constexpr (NVP == 32) {
        /* (32 dofs on 64 lanes: the two halves of the wave split the bodies of the loop, as in the composite-inertia sums) */
        double acc[6] = {0, 0, 0, 0, 0, 0};
        const int hk = lane & 31, hkbody = wv::shfl_i(kbody, hk), hkbend = wv::shfl_i(kbend, hk), coff = lane < 32 ? 0 : NB / 2;
        const unsigned ksub = hk < nv ? (unsigned)(((1ull << hkbend) - 1ull) ^ ((1ull << hkbody) - 1ull)) >> coff : 0u; /* bodies [kbody, kbend) */
        const double (*cfr)[6] = &S.x.s.cfrc[coff];
#pragma unroll
        for (int c0 = 0; c0 < NB / 2; c0 += 4) {
            double ff[4][6];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int t = 0; t < 6; ++t) ff[cc][t] = cfr[c0 + cc][t];
            wv::sched_fence();
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const double w = bitf(ksub, c0 + cc);
#pragma unroll
                for (int t = 0; t < 6; ++t) acc[t] = fma(w, ff[cc][t], acc[t]);
            }
        }
        for (int i = 0; i < 6; ++i) { acc[i] += wv::from_upper_half(acc[i]); qfrc_bias += S.cdof[lane < NVP ? lane : 0][i] * acc[i]; }
    } else {
        double acc[6] = {0, 0, 0, 0, 0, 0};
        const unsigned ksub = isdof ? (unsigned)(((1ull << kbend) - 1ull) ^ ((1ull << kbody) - 1ull)) : 0u; /* bodies [kbody, kbend) */
        /* (ROLLED: the 40-dof instantiation at 256 registers -- unrolled, the compiler requests all 32 bodies' forces at once, 384
         * registers' worth, and spills a hundred of them around the wait for wave 0's velocity stage) */
#pragma unroll(ROLLED ? 1 : NB / 4)
        for (int c0 = 0; c0 < NB; c0 += 4) {
            double ff[4][6];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int t = 0; t < 6; ++t) ff[cc][t] = S.x.s.cfrc[c0 + cc][t];
            wv::sched_fence();
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const double w = bitf(ksub, c0 + cc);
#pragma unroll
                for (int t = 0; t < 6; ++t) acc[t] = fma(w, ff[cc][t], acc[t]);
            }
        }
        for (int i = 0; i < 6; ++i) qfrc_bias += S.cdof[lane < NVP ? lane : 0][i] * acc[i];
    }
    CK_STAMP(6);

    /* ================= P6/P7/P8 passive + actuation -> qfrc_smooth (lane = dof) ================= */
    {
        if (isdof) {
            double f = -kdamp * S.qvel[k_];
            f -= kstiff * (S.qpos[kq] - kref);
            f -= qfrc_bias;
            if (io.qfrc_applied) f += io.qfrc_applied[(size_t)env * io.sv + k_];
            f += kgear * clampd(S.ctrl[ka], klo, khi);
            S.qfrc_smooth[k_] = f;
        }
    }
    if (io.xfrc_applied) {
        /* Cartesian perturbations: [force, torque] at the body's inertial origin, read straight from HBM (wave-uniform
         * addresses; the perturbation API is not a hot path and its 1.5 KB tile is better spent elsewhere) */
        if (isdof) {
            const double *xfa = io.xfrc_applied + ((size_t)env * io.sb) * 6;
            double f = 0;
            for (int bb = 1; bb < nbody; ++bb) {
                if (!((m->body_dofmask[bb] >> k_) & 1ull)) continue;
                const double xf[6] = {xfa[bb * 6], xfa[bb * 6 + 1], xfa[bb * 6 + 2], xfa[bb * 6 + 3], xfa[bb * 6 + 4], xfa[bb * 6 + 5]};
                if (xf[0] == 0 && xf[1] == 0 && xf[2] == 0 && xf[3] == 0 && xf[4] == 0 && xf[5] == 0) continue;
                const double *c = S.com[m->body_rootid[bb]];
                double off[3] = {S.x.s.xipos[bb][0] - c[0], S.x.s.xipos[bb][1] - c[1], S.x.s.xipos[bb][2] - c[2]};
                double t[3], cdk[6];
                for (int i = 0; i < 6; ++i) cdk[i] = S.cdof[k_][i];
                cross3(t, cdk, off);
                for (int i = 0; i < 3; ++i) f += (cdk[3 + i] + t[i]) * xf[i] + cdk[i] * xf[3 + i];
            }
            S.qfrc_smooth[k_] += f;
        }
    }
    wv::sync();
    CK_STAMP(7);
}

/* ---------------- the stages behind the constraint solve, as functions both forms share: in the one-wave form the substep loop calls
 * them in line, in the two-wave form wave 1 runs them (with the factor rows staged while wave 0 is still in its PGS sweeps). ---------------- */
/* this lane's row of the unit-triangular factor of M (zeros outside its ancestors): the forward substitution's operand */
template <int NVP, class TOPO, class SH>
WV_DEVICE void stage_factor_row(const SH &S, int k_, bool isdof, double (&lrow)[NVP]) {
    typedef LPack<TOPO, NVP> LP;
    const typename LP::Row myrow = LP::row_of(k_);
#pragma unroll
    for (int i = 0; i < NVP; ++i) {
        if constexpr (LP::packed) {
            const bool has = isdof && i < k_ && LP::row_has(myrow, i);
            const double v = S.Lp[has ? LP::row_idx(myrow, i) : 0];
            lrow[i] = has ? v : 0.0;
        } else lrow[i] = (isdof && i < k_) ? S.Lp[CK_TRI(k_, i)] : 0.0;
    }
}
/* this lane's column and row of the factor of M + hB (the Euler step's two substitutions); WHICH: 2 = both, 0 = the column only
 * (the backward substitution's operand, the first of the two), 1 = the row only */
template <int NVP, class TOPO, int WHICH = 2, class SH>
WV_DEVICE void stage_factor_h(const SH &S, int k_, bool isdof, int nv, double (&lcol)[NVP], double (&lrowh)[NVP]) {
    typedef LPack<TOPO, NVP> LP;
    const typename LP::Row myrow = LP::row_of(k_);
#pragma unroll
    for (int k = 0; k < NVP; ++k) {
        const bool inrange = TOPO::is_static ? k < TOPO::nv : k < nv;
        if constexpr (LP::packed) {
            if constexpr (WHICH != 1) {
                const bool hasc = inrange && isdof && k > k_ && LP::col_has(k, k_);
                const double vc = S.LHp[hasc ? LP::col_idx(k, k_) : 0];
                lcol[k] = hasc ? vc : 0.0;
            }
            if constexpr (WHICH != 0) {
                const bool hasr = isdof && k < k_ && LP::row_has(myrow, k);
                const double vr = S.LHp[hasr ? LP::row_idx(myrow, k) : 0];
                lrowh[k] = hasr ? vr : 0.0;
            }
        } else {
            if constexpr (WHICH != 1) lcol[k] = (inrange && isdof && k > k_) ? S.LHp[CK_TRI(k, k_)] : 0.0;
            if constexpr (WHICH != 0) lrowh[k] = (isdof && k < k_) ? S.LHp[CK_TRI(k_, k)] : 0.0;
        }
    }
}
/* ---- sensors, part 1 (lane = sensor): everything that does not need qacc is final here; the accelerometer parks its partial
 *      results in LDS (S.accel) because the body tiles are about to be recycled.  One-wave form: in line behind the Jacobian rows.
 *      Two-wave form: wave 1, in front of the barrier J -- in the time it used to wait there for wave 0's Jacobian rows (its own
 *      drive-level pass has read the previous substep's sensor words by then; poses, body velocities and bias accelerations are
 *      in LDS since F / cmd[1]); behind J, where wave 0 ran it until round 5, it was 4.7 k clocks of wave 0's critical path.
 *      (Placed behind J on wave 1 -- beside wave 0's half solve, with a flag before the staged matrix overwrites the body tiles --
 *      it sits between the mass matrix's columns, which wave 1 keeps for the factorisation of M + hB, and their use: 850 values
 *      went to scratch.) ---- */
/* Who reads a substep's sensors: the launch's caller (the last substep's), and in a drive mode the next substep's
 * encoder models (the joint / actuator positions) and the measurement block the LAST substep's drive pass writes
 * (the IMU words of the substep before it).  The IMU sensors -- frame quaternion, gyro, magnetometer and the
 * accelerometer with its second part after the solve -- are therefore evaluated by the last two substeps only. */
struct SensorConsts { int stype, slot, sqadr, sb, sroot, sdim, sadr; double sgain, scut; };
WV_DEVICE SensorConsts request_sensor_consts(ModelPtr m, int ls) {
    SensorConsts c;
    c.stype = m->sensor_type[ls]; c.slot = m->sensor_slot[ls];
    c.sqadr = m->sensor_qadr[ls]; c.sb = m->sensor_body[ls]; c.sroot = m->sensor_root[ls];
    c.sdim = m->sensor_dim[ls]; c.sadr = m->sensor_adr[ls];
    c.sgain = m->sensor_gain[ls]; c.scut = m->sensor_cutoff[ls];
    return c;
}
/* returns which accelerometer this lane is (-1: none) */
template <class SH>
WV_DEVICE int sensors_before_solve(const PhysIO &io, SH &S, ModelPtr m, int env, bool issens, int ls, SensorConsts sc, bool need_imu, bool lastsub) {
    int stype = sc.stype, slot_ = sc.slot;
    const int sqadr = sc.sqadr, sb = sc.sb, sroot = sc.sroot, sdim = sc.sdim, sadr = sc.sadr;
    const double sgain = sc.sgain, scut = sc.scut;
    wv::keep(stype); wv::keep(slot_);
    if (!issens) stype = -1;
    const int aslot = (stype == CM_SENS_ACCELEROMETER) ? slot_ : -1; /* which accelerometer this lane is */
    if (issens) {
        double sout[4] = {0, 0, 0, 0};
        if (sqadr >= 0) sout[0] = sgain * S.qpos[sqadr]; /* actuatorpos (gear * q) and jointpos */
        else if (need_imu && stype >= CM_SENS_FRAMEQUAT && stype <= CM_SENS_MAGNETOMETER) {
            double sq[4] = {m->sensor_squat[ls][0], m->sensor_squat[ls][1], m->sensor_squat[ls][2], m->sensor_squat[ls][3]};
            double q[4], sxmat[9], scvel[6];
            mulquat(q, S.x.s.xquat[sb], sq);
            quat2mat(sxmat, q);
            for (int i = 0; i < 6; ++i) scvel[i] = S.x.s.cvel[sb][i];
            if (stype == CM_SENS_FRAMEQUAT) { for (int i = 0; i < 4; ++i) sout[i] = q[i]; }
            else if (stype == CM_SENS_GYRO) mulmatTvec3(sout, sxmat, scvel);
            else if (stype == CM_SENS_MAGNETOMETER) {
                double mg[3] = {m->magnetic[0], m->magnetic[1], m->magnetic[2]};
                mulmatTvec3(sout, sxmat, mg);
            } else if (aslot >= 0) {
                /* accelerometer: velocity-product part of the body's com-frame acceleration (incl. -gravity)
                 * = the body's bias acceleration, which the velocity stage left in the buf tile */
                double acc_ang[3] = {S.x.s.buf[sb][0], S.x.s.buf[sb][1], S.x.s.buf[sb][2]};
                double acc_lin[3] = {S.x.s.buf[sb][3], S.x.s.buf[sb][4], S.x.s.buf[sb][5]};
                double sp[3] = {m->sensor_spos[ls][0], m->sensor_spos[ls][1], m->sensor_spos[ls][2]}, t[3];
                mulmatvec3(t, S.x.s.xmat[sb], sp);
                const double *c = S.com[sroot];
                double *pa = S.accel[aslot];
                for (int i = 0; i < 3; ++i) { pa[i] = acc_ang[i]; pa[3 + i] = acc_lin[i]; pa[6 + i] = t[i] + S.x.s.xpos[sb][i] - c[i]; }
                for (int i = 0; i < 9; ++i) pa[9 + i] = sxmat[i];
                for (int i = 0; i < 6; ++i) pa[18 + i] = scvel[i];
            }
        }
        if (stype != CM_SENS_ACCELEROMETER && (need_imu || sqadr >= 0)) {
            for (int i = 0; i < 4; ++i) {
                if (i >= sdim) continue;
                double v = sout[i];
                if (scut > 0 && stype != CM_SENS_FRAMEQUAT) v = clampd(v, -scut, scut);
                if (lastsub) io.sensordata[(size_t)env * io.ssd + sadr + i] = v;
                if (io.drive_mode) S.sens[sadr + i] = v;
            }
        }
    }
    return aslot;
}
/* what a substep still owes once qacc is in LDS: the accelerometers (they need qacc), the actuator velocities, and -- in the last
 * substep of a launch -- the outputs in HBM.  aslot / sb: which accelerometer this lane is (-1: none) and its body. */
template <class SH>
WV_DEVICE void outputs_after_qacc(const PhysIO &io, SH &S, ModelPtr m, int env, int lane, bool isdof, int k_, int nu, double qacc, int aslot, int sb, bool need_imu,
                                  bool lastsub, double av, int ncon, int nefc, int iters, int nguarded) {
    /* av: this lane's actuator velocity, gear * qvel of the state the substep started from (the caller reads it ahead of the Euler
     * step: in the two-wave form this function runs BEHIND the Euler step, beside wave 0's next kinematics stage) */
    if (aslot >= 0 && need_imu) {
        const double *pa = S.accel[aslot];
        double acc_ang[3] = {pa[0], pa[1], pa[2]}, acc_lin[3] = {pa[3], pa[4], pa[5]}, acc_dif[3] = {pa[6], pa[7], pa[8]};
        for (unsigned long long mk = m->body_dofmask[sb]; mk; mk &= mk - 1) {
            const int k = wv::popc64((mk & (0ull - mk)) - 1);
            const double qa = S.qacc[k];
            for (int i = 0; i < 3; ++i) { acc_ang[i] += S.cdof[k][i] * qa; acc_lin[i] += S.cdof[k][3 + i] * qa; }
        }
        double t[3], lin[3], vlin[3], corr[3], outv[3];
        cross3(t, acc_dif, acc_ang);
        for (int i = 0; i < 3; ++i) lin[i] = acc_lin[i] - t[i];
        cross3(t, acc_dif, pa + 18);
        for (int i = 0; i < 3; ++i) vlin[i] = pa[21 + i] - t[i];
        cross3(corr, pa + 18, vlin);
        for (int i = 0; i < 3; ++i) lin[i] += corr[i];
        mulmatTvec3(outv, pa + 9, lin);
        const double cut = m->sensor_cutoff[lane];
        const int adr = m->sensor_adr[lane];
        for (int i = 0; i < 3; ++i) {
            const double v = cut > 0 ? clampd(outv[i], -cut, cut) : outv[i];
            if (lastsub) io.sensordata[(size_t)env * io.ssd + adr + i] = v;
            if (io.drive_mode) S.sens[adr + i] = v;
        }
    }
    if (lane < nu) {
        if (lastsub) io.actuator_velocity[(size_t)env * io.su + lane] = av;
        if (io.drive_mode) S.actvel[lane] = av;
    }
    if (io.info && lane == 0 && lastsub) {
        io.info[(size_t)env * 4 + 0] = ncon; io.info[(size_t)env * 4 + 1] = nefc;
        io.info[(size_t)env * 4 + 2] = iters; io.info[(size_t)env * 4 + 3] = nguarded;
    }
    if (isdof && lastsub) io.qacc[(size_t)env * io.sv + k_] = qacc;
}
/* P12: semi-implicit Euler with implicit joint damping, then the positions (lane = dof, then lane = joint) */
template <int NVP, class TOPO, class SH>
WV_DEVICE void euler_step(SH &S, ModelPtr m, int lane, bool isdof, int k_, int nv, int njnt, double h, double qacc, const double (&lcol)[NVP],
                          const double (&lrowh)[NVP], double dih, double pf_kdamp, int pf_ejt, int pf_eqa, int pf_eda) {
    double qacc_int = qacc;
    if (m->flags & CM_FLAG_EULERDAMP) {
        /* (M + hB) x = M qacc  <=>  x = qacc - (M + hB)^-1 (hB qacc) */
        double w = isdof ? h * pf_kdamp * qacc : 0.0;
        w = solve_backward<NVP, TOPO>(w, lcol, lane, nv); /* L^-T */
        w *= dih;
        w = solve_forward<NVP, TOPO>(w, lrowh, lane, nv);  /* L^-1 */
        qacc_int = qacc - w;
    }
    if (isdof) {
        S.qvel[k_] += h * qacc_int;
        S.qacc_ws[k_] = qacc;
    }
    wv::sync();
    {
        /* lane = joint.  Hinges and slides are one FMA; a ball (or the rotation of a free joint) turns its quaternion by
         * h * |w| about w -- through the stage's own bounded-range sincos and reciprocal-square-root normalisations (the
         * library's sin + cos, a square root and two divisions, run for three lanes, were a tenth of this stage).  The
         * sincos sits outside the lane branches: its range check is a wave vote. */
        const int jt = lane < njnt ? pf_ejt : -1;
        int qa = pf_eqa, da = pf_eda;
        if (jt == CM_JNT_HINGE || jt == CM_JNT_SLIDE) S.qpos[qa] += h * S.qvel[da];
        if (jt == CM_JNT_FREE) {
            for (int i = 0; i < 3; ++i) S.qpos[qa + i] += h * S.qvel[da + i];
            qa += 3; da += 3;
        }
        const bool turns = jt == CM_JNT_FREE || jt == CM_JNT_BALL;
        double ax[3] = {1, 0, 0}, ang = 0;
        if (turns) {
            for (int i = 0; i < 3; ++i) ax[i] = S.qvel[da + i];
            ang = h * normalize3_fast(ax);
        }
        double sn, cs;
        sincos_bounded(0.5 * ang, sn, cs);
        if (turns) {
            double qr[4] = {cs, ax[0] * sn, ax[1] * sn, ax[2] * sn};
            double q[4] = {S.qpos[qa], S.qpos[qa + 1], S.qpos[qa + 2], S.qpos[qa + 3]};
            normalize4_fast(q);
            mulquat(q, q, qr);
            for (int i = 0; i < 4; ++i) S.qpos[qa + i] = q[i];
        }
    }
    wv::sync();
}

}  // namespace ck
#endif
