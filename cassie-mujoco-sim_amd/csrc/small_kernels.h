/*
 * small_kernels.h -- the kernels around the step kernel (physics_kernel.h): the longest-job-first launch order, the
 * drive-level pass on its own, the batched derived getters.  Included by phys_batch.hip (which launches them) and by the
 * CPU wave emulator; the step kernel's own translation units (kernels_*.hip) do not need them.
 */
#ifndef CASSIE_SMALL_KERNELS_H
#define CASSIE_SMALL_KERNELS_H

#include "physics_kernel.h"

namespace ck {

/* Longest-job-first launch order.  A launch is nenv independent jobs (one env x nsub substeps each) handed to the
 * chip's wave slots in workgroup order; with only a few jobs per slot -- 4096 envs on 1024 SIMDs -- the launch ends
 * when the unluckiest slot does, measured ~15 % after the average one.  An env's cost persists from launch to launch
 * (it is its contact situation), so the next launch starts the expensive envs first and lets the cheap ones fill the
 * tail.  One workgroup: counting sort of the envs by the cost of their last launch into NBIN bins, descending; the
 * order inside a bin is arbitrary, which is harmless -- envs are independent and every env is stepped exactly once. */
/* one wave: on a GPU saturated by another stream's step kernel (whose workgroups hold every register of their SIMD and all
 * but 0.8 KB of a CU's LDS) a workgroup is dispatched only when a wave slot frees, and a 1024-thread workgroup only when a
 * whole CU does -- which, measured, took milliseconds and stalled the launching stream (round 3, two-stream stepping) */
constexpr int ORDER_THREADS = 64, ORDER_NBIN = 256;
WV_GLOBAL void __launch_bounds__(ORDER_THREADS) cassie_order_kernel(const unsigned *cost, int *order, int nenv, int base, int *inplace_count, volatile int *inplace_seen) {
    /* sorts the env range [base, base + nenv): cost / order are indexed by the absolute env, the order entries are absolute */
    cost += base; order += base;
#ifndef CK_EMULATED
    /* (round 6: the kernel that runs behind the stepping launches of a range also reports, through host memory, how many env-launches
     * of the in-place fast kernel finished a substep in place since the last report -- or, negated, how many reports in a row had none:
     * the launcher picks the range's next form by it.  The run length is counted HERE, in stream order: the launcher may be hundreds
     * of launches ahead of the device and would count the same stale word again and again) */
    if (inplace_count && threadIdx.x == 0) {
        const int c = inplace_count[0], quiet = c > 0 ? 0 : inplace_count[1] + 1;
        inplace_count[0] = 0; inplace_count[1] = quiet;
        *inplace_seen = c > 0 ? c : -quiet;
    }
    __shared__ unsigned lo_s, hi_s, count[ORDER_NBIN], start[ORDER_NBIN];
    const int t = threadIdx.x;
    if (t == 0) { lo_s = 0xffffffffu; hi_s = 0; }
    for (int b = t; b < ORDER_NBIN; b += ORDER_THREADS) count[b] = 0;
    __syncthreads();
    unsigned lo = 0xffffffffu, hi = 0;
    for (int e = t; e < nenv; e += ORDER_THREADS) { const unsigned c = cost[e]; lo = c < lo ? c : lo; hi = c > hi ? c : hi; }
    atomicMin(&lo_s, lo); atomicMax(&hi_s, hi);
    __syncthreads();
    lo = lo_s; hi = hi_s;
    const float scale = hi > lo ? (float)(ORDER_NBIN - 1) / (float)(hi - lo) : 0.0f;
    auto bin_of = [&](unsigned c) { return ORDER_NBIN - 1 - (int)((float)(c - lo) * scale); }; /* bin 0 = most expensive */
    for (int e = t; e < nenv; e += ORDER_THREADS) atomicAdd(&count[bin_of(cost[e])], 1u);
    __syncthreads();
    if (t == 0) { unsigned acc = 0; for (int b = 0; b < ORDER_NBIN; ++b) { start[b] = acc; acc += count[b]; } }
    __syncthreads();
    for (int e = t; e < nenv; e += ORDER_THREADS) order[atomicAdd(&start[bin_of(cost[e])], 1u)] = base + e;
#endif
}

/* The drive-level pass on its own, one wave per env: cassie_motor_data + cassie_sensor_data for every env on the
 * sensordata / actuator_velocity the last physics step left in HBM; writes ctrl (for the next physics launch), the
 * measurement block and the drive state.  The batched host API launches it ahead of the physics kernel so that the
 * measurements reach the host -- and the state estimators start -- while the physics is still running. */
struct DriveShared {
    double sens[CM_MAXSENSORDATA], actvel[CM_MAXU], ctrl[CM_MAXU];
    int drv_x[CM_NUM_DRIVES][CM_DRIVE_FILTER_NB];
    double drv_jx[CM_NUM_JOINTS][CM_JOINT_FILTER_NB], drv_jy[CM_NUM_JOINTS][CM_JOINT_FILTER_NA];
    double drv_delay[CM_NUM_DRIVES][CM_TORQUE_DELAY_CYCLES];
    double drv_pos[CM_NUM_DRIVES], drv_vel[CM_NUM_DRIVES];
    double drv_c[CM_NUM_DRIVES][10], drv_jc[CM_NUM_JOINTS][2];
    int drv_msg[2];
};

WV_GLOBAL void __launch_bounds__(WV_WAVE) cassie_drive_kernel(PhysIO io, double *ctrl_out) {
    WV_SHARED DriveShared S;
    const int env = wv::env_id();
    if (env >= io.nenv) return;
    const ModelPtr m = (ModelPtr)(io.models + (size_t)env * io.model_stride);
    const int lane = wv::lane(), nu = m->nu;
    if (lane < m->nsensordata) S.sens[lane] = io.sensordata[(size_t)env * io.ssd + lane];
    if (lane < nu) S.actvel[lane] = io.actuator_velocity[(size_t)env * io.su + lane];
    drive_state_load(io, S, env, lane);
    drive_consts_load(io, S, m, env, lane);
    wv::sync();
    drive_level_io(io, S, m, env, lane, true);
    wv::sync();
    drive_state_store(io, S, env, lane);
    if (lane < nu) ctrl_out[(size_t)env * io.su + lane] = S.ctrl[lane];
}

/* ------------------------------------------------ derived getters, batched ---- */
/* What the reference's reward-side getters compute from mjData, for one env from the read-out (cm_ext_t) a forward
 * pass left in HBM: whole-model centre of mass, its velocity, angular momentum about it (reference
 * src/cassiemujoco.c:1632-1700), foot positions / velocities (:1604-1630, :1752-1770), foot and heel / toe contact
 * forces (:1812-1898), the feet's Jacobians (:1254-1301) and the dense mass matrix (:1702-1712).  Same arithmetic as
 * the single-simulator getters in csrc/cassiemujoco.c.  ids: left / right foot body, left / right heel site, left /
 * right toe site (-1 = the model has none). */
struct DeriveIO {
    const cm_model_t *models; int model_stride; int nenv;
    const cm_envparams_t *envparams; /* null, or one block per env (PhysIO::envparams) */
    const cm_ext_t *ext;
    const double *xpos, *xquat;   /* [nenv][nbody][3], [nenv][nbody][4] */
    double *derived;              /* [nenv][CM_DRV_DIM] */
    double *qM;                   /* [nenv][nv][nv] or null */
    int ids[6];
};

WV_DEVICE void derive_env(const DeriveIO &io, int env, int lane) {
    const ModelPtr m = (ModelPtr)(io.models + (size_t)env * io.model_stride);
    const ParamPtr P = io.envparams ? (ParamPtr)(io.envparams + (size_t)env) : (ParamPtr)&m->params;
    const cm_ext_t *ex = io.ext + env;
    const int nb = m->nbody, nv = m->nv;
    double *out = io.derived + (size_t)env * CM_DRV_DIM;
    const double *xpos = io.xpos + (size_t)env * nb * 3, *xquat = io.xquat + (size_t)env * nb * 4;
    /* lane = body: mass-weighted sums */
    const bool isb = lane > 0 && lane < nb;
    const int b = isb ? lane : 0;
    const double mb = isb ? P->body_mass[b] : 0.0;
    double xi[3], vb[3] = {0, 0, 0}, w[3] = {0, 0, 0};
    for (int i = 0; i < 3; ++i) xi[i] = ex->xipos[b][i];
    if (isb) {
        const double *cv = ex->cvel[b], *rc = ex->subtree_com[m->body_rootid[b]];
        double off[3] = {xi[0] - rc[0], xi[1] - rc[1], xi[2] - rc[2]}, t[3];
        for (int i = 0; i < 3; ++i) w[i] = cv[i];
        cross3(t, w, off);
        for (int i = 0; i < 3; ++i) vb[i] = cv[3 + i] + t[i];
    }
    const double M = wv::wave_sum(mb), Mi = M > 0 ? 1.0 / M : 0.0;
    double com[3], vcom[3];
    for (int i = 0; i < 3; ++i) { com[i] = wv::wave_sum(mb * xi[i]) * Mi; vcom[i] = wv::wave_sum(mb * vb[i]) * Mi; }
    /* angular momentum about the whole-model com: spin R diag(I) R^T w + orbital r x m (v - vcom) */
    double L[3] = {0, 0, 0};
    if (isb) {
        double q[4], R[9], iq[4] = {m->body_iquat[b][0], m->body_iquat[b][1], m->body_iquat[b][2], m->body_iquat[b][3]};
        double xq[4] = {xquat[4 * b], xquat[4 * b + 1], xquat[4 * b + 2], xquat[4 * b + 3]};
        mulquat(q, xq, iq);
        quat2mat(R, q);
        double wl[3];
        mulmatTvec3(wl, R, w);
        for (int i = 0; i < 3; ++i) wl[i] *= P->body_inertia[b][i];
        mulmatvec3(L, R, wl);
        double r[3], mv[3], t[3];
        for (int i = 0; i < 3; ++i) { r[i] = xi[i] - com[i]; mv[i] = mb * (vb[i] - vcom[i]); }
        cross3(t, r, mv);
        for (int i = 0; i < 3; ++i) L[i] += t[i];
    }
    for (int i = 0; i < 3; ++i) L[i] = wv::wave_sum(L[i]);
    if (lane == 0) {
        for (int i = 0; i < 3; ++i) { out[CM_DRV_COM_POS + i] = com[i]; out[CM_DRV_COM_VEL + i] = vcom[i]; out[CM_DRV_ANGMOM + i] = L[i]; }
        out[CM_DRV_MASS] = M;
    }
    /* lane = side: foot kinematics and contact forces */
    if (lane < 2) {
        const int side = lane, foot = io.ids[side], heel = io.ids[2 + side], toe = io.ids[4 + side];
        const double off = sqrt(0.01762 * 0.01762 + 0.05219 * 0.05219); /* foot joint to mid-foot (reference :1612) */
        for (int i = 0; i < 3; ++i) out[CM_DRV_FOOT_POS + 3 * side + i] = (foot > 0 ? xpos[3 * foot + i] : 0.0) - (i == 2 ? off : 0.0);
        for (int i = 0; i < 6; ++i) out[CM_DRV_FOOT_VEL + 6 * side + i] = foot > 0 ? ex->cvel[foot][i] : 0.0;
        double ff[3] = {0, 0, 0}, tf[3] = {0, 0, 0}, hf[3] = {0, 0, 0};
        for (int c = 0; c < ex->ncon; ++c) {
            const int b1 = ex->con_body1[c], b2 = ex->con_body2[c];
            if (b1 != foot && b2 != foot) continue;
            const double *fr = ex->con_frame[c], *f = ex->con_force[c];
            const double sgn = (b1 == foot) ? -1.0 : 1.0;
            double fw[3];
            for (int k = 0; k < 3; ++k) fw[k] = fr[k] * f[0] + fr[3 + k] * f[1] + fr[6 + k] * f[2];
            for (int k = 0; k < 3; ++k) ff[k] += sgn * fw[k];
            if (heel >= 0 && toe >= 0 && heel < CM_MAXSITE && toe < CM_MAXSITE) {
                const double *p = ex->con_pos[c], *tp = ex->site_xpos[toe], *hp = ex->site_xpos[heel];
                const double td = sqrt((tp[0] - p[0]) * (tp[0] - p[0]) + (tp[1] - p[1]) * (tp[1] - p[1]));
                const double hd = sqrt((hp[0] - p[0]) * (hp[0] - p[0]) + (hp[1] - p[1]) * (hp[1] - p[1]));
                double *dst = td < hd ? tf : hf;
                for (int k = 0; k < 3; ++k) dst[k] += sgn * fw[k];
            }
        }
        for (int k = 0; k < 3; ++k) {
            out[CM_DRV_FOOT_FORCE + 6 * side + k] = ff[k]; out[CM_DRV_FOOT_FORCE + 6 * side + 3 + k] = 0.0;
            out[CM_DRV_TOE_FORCE + 3 * side + k] = tf[k]; out[CM_DRV_HEEL_FORCE + 3 * side + k] = hf[k];
        }
    }
    /* lane = dof: Jacobian columns of the two foot origins, and this dof's column of the mass matrix */
    if (lane < CM_MAXV) {
        const int k = lane;
        for (int side = 0; side < 2; ++side) {
            const int foot = io.ids[side];
            double jp[3] = {0, 0, 0}, jr[3] = {0, 0, 0};
            if (k < nv && foot > 0 && ((m->body_dofmask[foot] >> k) & 1ull)) {
                const double *cm = ex->subtree_com[m->body_rootid[foot]];
                double off[3] = {xpos[3 * foot] - cm[0], xpos[3 * foot + 1] - cm[1], xpos[3 * foot + 2] - cm[2]}, t[3];
                double cd[6];
                for (int i = 0; i < 6; ++i) cd[i] = ex->cdof[k][i];
                cross3(t, cd, off);
                for (int i = 0; i < 3; ++i) { jp[i] = cd[3 + i] + t[i]; jr[i] = cd[i]; }
            }
            for (int i = 0; i < 3; ++i) {
                out[CM_DRV_FOOT_JACP + (side * 3 + i) * CM_MAXV + k] = jp[i];
                out[CM_DRV_FOOT_JACR + (side * 3 + i) * CM_MAXV + k] = jr[i];
            }
        }
        if (io.qM && k < nv) {
            double *Mo = io.qM + (size_t)env * nv * nv;
            for (int i = 0; i < nv; ++i) Mo[(size_t)i * nv + k] = ex->qM[i][k];
        }
    }
}

WV_GLOBAL void __launch_bounds__(WV_WAVE) cassie_derive_kernel(DeriveIO io) {
    const int env = wv::env_id();
    if (env >= io.nenv) return;
    derive_env(io, env, wv::lane());
}

/* ------------------------------------------- per-env physical parameters ---- */
/* phys_batch_randomize: rows [n][dim] of one input array of cm_envparams_t (cm_model.h: CM_P_*) into the blocks of envs
 * env0 .. env0 + n - 1.  off = the array's offset in the block in doubles.  A few workgroups walk the rows. */
WV_GLOBAL void __launch_bounds__(WV_WAVE) cassie_param_scatter_kernel(cm_envparams_t *params, int off, int dim, const double *src, int env0, int n) {
#ifndef CK_EMULATED
    for (int i = (int)blockIdx.x; i < n; i += (int)gridDim.x) {
        double *dst = (double *)(params + env0 + i) + off;
        for (int k = (int)threadIdx.x; k < dim; k += WV_WAVE) dst[k] = src[(size_t)i * dim + k];
    }
#endif
}

/* The mj_setConst role on the device (reference src/cassiemujoco.c:952, :976 after the mass / inertial-offset setters of
 * :1367-1412), one wave per env: from the env's masses, inertial offsets and principal inertias and the model's kinematics at
 * qpos0 (cm_model_t::body_xpos0 ...: no randomised parameter moves a body frame) it builds the dense joint-space inertia at qpos0,
 * its Cholesky factor, and from it the mean inertia, the bodies' translational / rotational inverse weights
 * (1/3 tr J M^-1 J^T at the inertial origin) and the dofs' (diag M^-1, averaged over ball / free joints); then the values the
 * constraint stages read per joint limit / equality / contact pair, and the pairs' mixed friction.
 *
 * Every floating-point operation is an individually rounded IEEE operation in the order of HostModel::set_const /
 * HostModel::compile (csrc/mjcf_loader.cpp), so an env's block is BIT FOR BIT what compiling a host model with the same
 * parameters gives (tests/test_domain_randomisation.py on the emulator, tests/test_randomise_gpu.py on the device) -- the
 * oracle, which is handed such per-env models, and the kernel then solve the same problem.
 * Lanes: dofs (Jacobian columns, columns of M, rows of the factor), then (body, translational | rotational) for the bodies'
 * solves, then dofs again for the unit-vector solves; the right-hand sides live one to a lane in LDS. */
struct SetConstShared {
    double M[CM_MAXV][CM_MAXV + 1];
    double Jp[3][CM_MAXV], Jl[3][CM_MAXV];
    double X[WV_WAVE][CM_MAXV + 1], X0[WV_WAVE][CM_MAXV + 1];
    double xipos[CM_MAXBODY][3];
    double bw[CM_MAXBODY][2], dinv[CM_MAXV], dw[CM_MAXV];
    double piv;
    int ok;
};
struct SetConstIO {
    const cm_model_t *model;      /* the shared model (topology, kinematics at qpos0, armature, pair / equality tables) */
    cm_envparams_t *params;       /* [nenv] blocks, indexed by the absolute env */
    int env0, nenv;               /* the range to (re)derive */
    int derive_inertial;          /* 0: only the friction / pair tables (phys_batch_randomize of CM_P_GEOM_FRICTION); 1: all */
};

/* Jacobian column of dof d for a world point p attached to body b (HostKin::jac): translational part jp, rotational jr */
WV_DEVICE void setconst_jac_col(ModelPtr m, int b, int d, const double *p, double (&jp)[3], double (&jr)[3]) {
    for (int i = 0; i < 3; ++i) { jp[i] = 0.0; jr[i] = 0.0; }
    if (!((m->body_dofmask[b] >> d) & 1ull)) return;
    double ax[3] = {m->dof_axis0[d][0], m->dof_axis0[d][1], m->dof_axis0[d][2]};
    if (m->dof_trans0[d]) { for (int i = 0; i < 3; ++i) jp[i] = ax[i]; return; }
    const double off[3] = {wv::sub_rn(p[0], m->dof_anchor0[d][0]), wv::sub_rn(p[1], m->dof_anchor0[d][1]), wv::sub_rn(p[2], m->dof_anchor0[d][2])};
    jp[0] = wv::sub_rn(wv::mul_rn(ax[1], off[2]), wv::mul_rn(ax[2], off[1]));
    jp[1] = wv::sub_rn(wv::mul_rn(ax[2], off[0]), wv::mul_rn(ax[0], off[2]));
    jp[2] = wv::sub_rn(wv::mul_rn(ax[0], off[1]), wv::mul_rn(ax[1], off[0]));
    for (int i = 0; i < 3; ++i) jr[i] = ax[i];
}
/* x = (L L^T)^-1 x with L in the lower triangle of S.M (chol_solve of the host compile), x in LDS */
WV_DEVICE void setconst_chol_solve(const SetConstShared &S, int n, double *x) {
    for (int i = 0; i < n; ++i) {
        double s = x[i];
        for (int k = 0; k < i; ++k) s = wv::sub_rn(s, wv::mul_rn(S.M[i][k], x[k]));
        x[i] = wv::div_rn(s, S.M[i][i]);
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = x[i];
        for (int k = i + 1; k < n; ++k) s = wv::sub_rn(s, wv::mul_rn(S.M[k][i], x[k]));
        x[i] = wv::div_rn(s, S.M[i][i]);
    }
}

WV_GLOBAL void __launch_bounds__(WV_WAVE) cassie_setconst_kernel(SetConstIO io) {
    WV_SHARED SetConstShared S;
    const int lane = wv::lane();
    for (int idx = wv::env_id(); idx < io.nenv; idx += wv::grid_size()) {
    const int env = io.env0 + idx;
    const ModelPtr m = (ModelPtr)io.model;
    cm_envparams_t *P = io.params + env;
    const int nv = m->nv, nbody = m->nbody, njnt = m->njnt;
    if (io.derive_inertial) {
    /* inertial origins at qpos0: xipos = xmat0 ipos + xpos0 */
    if (lane < nbody) {
        const int b = lane;
        double v[3] = {0, 0, 0};
        if (b > 0) {
            const double *ip = P->body_ipos[b];
            for (int i = 0; i < 3; ++i)
                v[i] = wv::add_rn(wv::add_rn(wv::add_rn(wv::mul_rn(m->body_xmat0[b][3 * i], ip[0]), wv::mul_rn(m->body_xmat0[b][3 * i + 1], ip[1])), wv::mul_rn(m->body_xmat0[b][3 * i + 2], ip[2])), m->body_xpos0[b][i]);
        }
        for (int i = 0; i < 3; ++i) S.xipos[b][i] = v[i];
    }
    if (lane < CM_MAXV) for (int r = 0; r < CM_MAXV; ++r) S.M[r][lane] = 0.0;
    if (lane == 0) S.ok = 1;
    wv::sync();
    /* M(qpos0) = sum over bodies of J^T diag(m, I) J (lane = column), + armature */
    for (int b = 1; b < nbody; ++b) {
        const double mass = P->body_mass[b];
        if (mass <= 0) continue;
        if (lane < nv) {
            double jp[3], jr[3];
            setconst_jac_col(m, b, lane, S.xipos[b], jp, jr);
            for (int i = 0; i < 3; ++i) {
                S.Jp[i][lane] = jp[i];
                S.Jl[i][lane] = wv::add_rn(wv::add_rn(wv::mul_rn(m->body_ximat0[b][i], jr[0]), wv::mul_rn(m->body_ximat0[b][3 + i], jr[1])), wv::mul_rn(m->body_ximat0[b][6 + i], jr[2]));
            }
        }
        wv::sync();
        if (lane < nv) {
            const int c = lane;
            for (int r = 0; r < nv; ++r) {
                double s = 0;
                for (int i = 0; i < 3; ++i)
                    s = wv::add_rn(s, wv::add_rn(wv::mul_rn(wv::mul_rn(mass, S.Jp[i][r]), S.Jp[i][c]),
                                                 wv::mul_rn(wv::mul_rn(P->body_inertia[b][i], S.Jl[i][r]), S.Jl[i][c])));
                S.M[r][c] = wv::add_rn(S.M[r][c], s);
            }
        }
        wv::sync();
    }
    if (lane < nv) S.M[lane][lane] = wv::add_rn(S.M[lane][lane], m->dof_armature[lane]);
    wv::sync();
    if (lane == 0) {
        double mean = 0;
        for (int d = 0; d < nv; ++d) mean = wv::add_rn(mean, S.M[d][d]);
        P->meaninertia = nv > 0 ? wv::div_rn(mean, (double)nv) : 1.0;
    }
    /* Cholesky factor in place, column by column (lane = row) */
    for (int j = 0; j < nv; ++j) {
        double s = 0;
        if (lane >= j && lane < nv) {
            s = S.M[lane][j];
            for (int k = 0; k < j; ++k) s = wv::sub_rn(s, wv::mul_rn(S.M[lane][k], S.M[j][k]));
            if (lane == j) { if (s <= 0) S.ok = 0; S.piv = wv::sqrt_rn(s > 0 ? s : 1.0); }
        }
        wv::sync();
        if (lane >= j && lane < nv) S.M[lane][j] = lane == j ? S.piv : wv::div_rn(s, S.piv);
        wv::sync();
    }
    if (S.ok) { /* (a mass matrix that is not positive definite leaves the weights as they were, like the host compile) */
    /* bodies: lane = 2 body + (0 translational | 1 rotational); three solves each, summed in order */
    {
        const int b = lane >> 1, kind = lane & 1;
        const bool live = b > 0 && b < nbody && m->body_weldid[b] != 0;
        double acc = 0;
        for (int i = 0; i < 3; ++i) {
            if (live) {
                for (int d = 0; d < nv; ++d) {
                    double jp[3], jr[3];
                    setconst_jac_col(m, b, d, S.xipos[b], jp, jr);
                    const double v = kind ? jr[i] : jp[i];
                    S.X[lane][d] = v; S.X0[lane][d] = v;
                }
                setconst_chol_solve(S, nv, S.X[lane]);
                for (int d = 0; d < nv; ++d) acc = wv::add_rn(acc, wv::mul_rn(S.X0[lane][d], S.X[lane][d]));
            }
        }
        if (b < nbody) S.bw[b][kind] = live ? wv::div_rn(acc, 3.0) : 0.0;
    }
    wv::sync();
    /* dofs: diag(M^-1) by unit-vector solves, averaged over the dofs of ball / free joints */
    if (lane < nv) {
        for (int d = 0; d < nv; ++d) S.X[lane][d] = d == lane ? 1.0 : 0.0;
        setconst_chol_solve(S, nv, S.X[lane]);
        S.dinv[lane] = S.X[lane][lane];
    }
    wv::sync();
    if (lane < njnt) {
        const int j = lane, d = m->jnt_dofadr[j], jt = m->jnt_type[j];
        if (jt == CM_JNT_FREE) {
            const double a = wv::div_rn(wv::add_rn(wv::add_rn(S.dinv[d], S.dinv[d + 1]), S.dinv[d + 2]), 3.0);
            const double r = wv::div_rn(wv::add_rn(wv::add_rn(S.dinv[d + 3], S.dinv[d + 4]), S.dinv[d + 5]), 3.0);
            for (int k = 0; k < 3; ++k) { S.dw[d + k] = a; S.dw[d + 3 + k] = r; }
        } else if (jt == CM_JNT_BALL) {
            const double r = wv::div_rn(wv::add_rn(wv::add_rn(S.dinv[d], S.dinv[d + 1]), S.dinv[d + 2]), 3.0);
            for (int k = 0; k < 3; ++k) S.dw[d + k] = r;
        } else S.dw[d] = S.dinv[d];
    }
    wv::sync();
    if (lane < nbody) { P->body_invweight0[lane][0] = S.bw[lane][0]; P->body_invweight0[lane][1] = S.bw[lane][1]; }
    if (lane < nv) P->dof_invweight0[lane] = S.dw[lane];
    wv::sync();
    }
    /* what the constraint stages read: per limited joint, per equality, per candidate pair (HostModel::compile) */
    if (lane < njnt) P->jnt_liminvweight[lane] = P->dof_invweight0[m->jnt_dofadr[lane]];
    if (lane < m->neq) P->eq_invweight[lane] = wv::add_rn(wv::add_rn(0.0, P->body_invweight0[m->eq_body1[lane]][0]), P->body_invweight0[m->eq_body2[lane]][0]);
    for (int p = lane; p < m->npair; p += WV_WAVE) {
        const int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p];
        P->pair_invweight[p] = wv::add_rn(wv::add_rn(0.0, P->body_invweight0[m->geom_bodyid[g1]][0]), P->body_invweight0[m->geom_bodyid[g2]][0]);
    }
    }
    /* contact friction of every candidate pair: the geom with the higher priority wins outright, else the larger coefficient */
    for (int p = lane; p < m->npair; p += WV_WAVE) {
        const int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p], pa = m->geom_priority[g1], pb = m->geom_priority[g2];
        for (int k = 0; k < 3; ++k) {
            const double f1 = P->geom_friction[g1][k], f2 = P->geom_friction[g2][k];
            P->pair_friction[p][k] = pa != pb ? (pa > pb ? f1 : f2) : (f1 > f2 ? f1 : f2);
        }
    }
    wv::sync();
    }
}

/* Episode restarts on the device (the batched form of what a fresh cassie_sim_t / cassie_sim_full_reset leaves, reference
 * src/cassiemujoco.c:1023-1029, :2008-2034): envs first, first + stride, ... (count of them) get qpos = qpos_row, zero
 * qvel / qacc_warmstart / ctrl / qacc / actuator_velocity / time, optionally sensordata = sens_row (what the first drive-level
 * pass of the new episode reads: the init pose's), a zero measurement block and zero drive-level state (encoder filter
 * histories, torque delay lines) -- one launch instead of a scatter per field.  The sticky warning word is left alone. */
struct ResetIO {
    int first, stride, count;
    int nq, nv, nu, nsd, sq, sqv, ssd;
    double *qpos, *qvel, *warm, *ctrl, *qacc, *time, *sens, *actvel, *meas;
    cm_drive_state_t *drive;    /* may be null */
    const double *qpos_row;     /* [nq] */
    const double *sens_row;     /* [nsensordata] or null: sensordata is left alone */
};
WV_GLOBAL void __launch_bounds__(WV_WAVE) cassie_reset_kernel(ResetIO io) {
#ifndef CK_EMULATED
    /* a few workgroups walk all the envs: on a GPU saturated by another stream's step kernel every workgroup waits for a
     * wave slot to free, and one workgroup per env made this kernel a 0.5 ms stall of its stream (rocprofv3, two-range stepping) */
    const int lane = wv::lane();
    for (int i = (int)blockIdx.x; i < io.count; i += (int)gridDim.x) {
    const size_t env = (size_t)io.first + (size_t)i * io.stride;
    if (lane < io.nq) io.qpos[env * io.sq + lane] = io.qpos_row[lane];
    if (lane < io.nv) { io.qvel[env * io.sqv + lane] = 0.0; io.warm[env * io.nv + lane] = 0.0; io.qacc[env * io.nv + lane] = 0.0; }
    if (lane < io.nu) { io.ctrl[env * io.nu + lane] = 0.0; io.actvel[env * io.nu + lane] = 0.0; }
    if (io.sens_row && lane < io.nsd) io.sens[env * io.ssd + lane] = io.sens_row[lane];
    if (io.meas && lane < CM_MEAS_DIM) io.meas[env * CM_MEAS_DIM + lane] = 0.0;
    if (io.drive) {
        int *w = (int *)(io.drive + env);
        for (int k = lane; k < (int)(sizeof(cm_drive_state_t) / sizeof(int)); k += WV_WAVE) w[k] = 0;
    }
    if (lane == 0) io.time[env] = 0.0;
    }
#endif
}

}  // namespace ck
#endif
