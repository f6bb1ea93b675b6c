/*
 * physics_kernel.h -- the batched Cassie physics step for gfx950 (MI355X):
 * ONE WORKGROUP PER ENVIRONMENT, four envs resident per CU (<= 40 KB of LDS each), stepped by one wavefront (64 lanes; every
 * stage below is written for one wave) or -- the Cassie instantiations since round 4 -- by TWO, which split the substep's
 * stage graph between them (env_step, NW = 2: two waves per SIMD at 256 registers each).
 *
 * This is the hot path of the reference -- the mj_step1_fp + mj_step2_fp pair at
 * reference src/cassiemujoco.c:1130-1134 (arithmetic inside MuJoCo 2.1.0; stage
 * list SURVEY.md 8a P1..P12, semantics SURVEY.md App. B) -- redesigned for a
 * CDNA4 wave instead of a CPU thread:
 *
 *   lanes = bodies   every body builds its joint-inclusive local transform; the tree
 *                    recursion is pointer jumping (4 rounds over the 1st/2nd/4th/8th
 *                    ancestors); composite inertias and bias-force projections are dense
 *                    loops over bodies with the subtree predicate applied by multiplication
 *   lanes = dofs     motion axes; body velocities / bias accelerations as in-place
 *                    pointer-jumping prefix sums over the dof tree; one COLUMN of the mass
 *                    matrix per lane in VGPRs; the tree-sparse L^T D L factorisations of M
 *                    and M + h*B go height by height through LDS broadcasts (compile-time
 *                    topology) or by v_readlane pivots (run-time topology)
 *   lanes = pairs    narrow-phase collision from denormalised pair records, ballot-compacted
 *                    contact list, contacts finished one per lane
 *   lanes = rows     one constraint row per lane: its Jacobian row, its column of
 *                    Y = D^-1/2 L^-T [J^T | qfrc_smooth] and its row of A = Y^T Y all live
 *                    in registers; L and Y are broadcast from LDS; PGS keeps the scaled
 *                    residual one row per lane, broadcasts each step with v_readlane and
 *                    evaluates the cost guard once per sweep
 *
 * Loops over dofs / rows are fully unrolled over compile-time sparsity tables
 * (topo_static.h) where the model matches one, so register arrays are statically
 * indexed and LDS offsets are immediates; LDS reads are staged ahead of their use behind
 * scheduling fences (the compiler otherwise pairs every read with its own wait).  The
 * workgroup is one wave: "sync" is a compiler fence, not a barrier.  The batched
 * qpos/qvel/ctrl/sensordata arrays are env-major in HBM so a wave's loads and stores are
 * contiguous; nsub steps per launch keep the state in LDS.
 *
 * Round 2 added, in the same one-wave-per-env style: the drive-level I/O of cassie_sim_step_ethercat (motor model with
 * torque delay, encoder quantisation and velocity filters; reference src/cassiemujoco.c:558-664, :737-803) as a
 * per-substep prologue whose state lives in LDS for the length of a launch -- bit for bit the host chain; height-field
 * contacts over every grid triangle under a sample sphere, the samples of all height-field pairs spread over the lanes of a
 * pre-pass; box-box by separating axes with the clipped-face candidates one to a lane; FEAT_* template flags that keep
 * collision code a model does not need out of its instantiation; outputs stored by the last substep only; a derive kernel
 * for the batched getters; a longest-job-first launch order.
 *
 * Round 4: the two-wave form (wave 1: mass-matrix group, drive-level pass, factorisations, bias / passive stage, the stages
 * behind the solve; four workgroup barriers and four LDS flags per substep; bit for bit the one-wave form); the hand-over list
 * and the list-walking pass behind the row-capped fast instantiation.
 *
 * Numerically this follows the same algorithm as oracle/cassie_oracle.c but with
 * its own operation order (half solves, reciprocal multiplies, wave reductions,
 * chain sums), so parity is to a tolerance, not bitwise.
 */
#ifndef CASSIE_PHYSICS_KERNEL_H
#define CASSIE_PHYSICS_KERNEL_H

#include <wave.h> /* csrc/wave.h in the product build; tests/emu/wave.h under the CPU wave emulator */

#include "cm_model.h"
#include "topo_static.h"

#include "pk_types.h"
#include "pk_math.h"
#include "pk_safety.h"
#include "pk_collision.h"
#include "pk_factor_solve.h"
#include "pk_stages.h"
#include "pk_wide_solve.h"

namespace ck {

/* ======================================================== the env step ==== */
/* FEAT selects the collision code a model needs, so that the instantiation for plain cassie.xml does not carry the
 * register pressure of paths it never takes: FEAT_HFIELD = height-field pairs, FEAT_WAVEPAIRS = plane-box / box-box
 * pairs handled by the whole wave.  The launcher picks the instantiation from the model (phys_batch.hip). */

/* (env_step's return value: the substep it ended in front of, with this bit set when it ended there because the substep fits the
 * fast code again -- PhysIO::down_rows -- and not because it needs more than this code holds) */
constexpr int ENV_STEP_WENT_DOWN = 1 << 20;

template <int NVP, class TOPO, int FEAT, int MAXR, int NW>
WV_DEVICE int env_step(const PhysIO &io, EnvShared<NVP, LPack<TOPO, NVP>::count, MAXR> &S, int env, int sub_start, int nsub) {
    /* nsub: the substep this call ends in front of -- io.nsub, or the end of this workgroup's chunk of the launch (PhysIO::nchunk):
     * the call then ends like a launch of nsub substeps.  Returns (wave 0; wave 1 of the two-wave form returns -1) the substep the
     * call really ended in front of: nsub, or the substep it handed over / found the state diverged at */
    typedef LPack<TOPO, NVP> LP;
    static_assert(NW == 1 || NW == 2, "one or two wavefronts per env");
    static_assert(NW == 1 || TOPO::is_static, "the two-wave form exists for the compile-time topologies");
    /* NW = 2: the env is stepped by TWO wavefronts that share the env's LDS block.  Wave 0 runs the substep as written below
     * except for the stages wave 1 runs beside it (wave 1's program, ahead of the substep loop): the mass-matrix group (centres
     * of mass, cinert, cdof, composite inertias, M's columns), the drive-level pass, the two factorisations and the bias /
     * passive stage beside wave 0's collision, velocity and constraint-row stages; then qacc, the accelerometers, the substep's
     * outputs and the Euler step behind wave 0's solve, with their operands staged while wave 0 solves.  Four workgroup
     * barriers per substep (F: poses in LDS; J: the factors and qfrc_smooth; P: the row forces; E: the substep is complete) and four
     * one-directional flags: cmd[3] / cmd[4] where a barrier X used to be (com / cinert / cdof for wave 0's velocity stage, the
     * collision verdict for wave 1's drive-level pass: its factorisations in between do not wait for wave 0's collision stage),
     * cmd[1] (body forces), cmd[2] (the staged matrix).  Every value is computed by the
     * same instructions from the same operands as in the one-wave form, so the results are bit for bit the same. */
    const int wid = NW == 2 ? wv::wave_id() : 0;
    /* (wave 0 is on its env's critical path from F to P: above a wave 1 in a stretch nobody waits for, below one in its tail -- env_step_wave1.inc) */
    if constexpr (NW == 2) { if (wid == 0) wv::set_priority<CK_PRIO_W0>(); }
    /* (... and higher still up to the barrier F, which its own wave 1 waits for -- measured: +1.0 % on cassie.xml, -1 % with a height-field pre-pass behind F) */
    constexpr int prio_kin = (FEAT & FEAT_HFIELD) != 0 ? CK_PRIO_W0 : CK_PRIO_W0_KIN;

    static_assert(LP::covers() && LP::distinct(), "packed factor rows must hold every ancestor pair, each in its own slot");
    typedef EnvShared<NVP, LPack<TOPO, NVP>::count, MAXR> SH_T;
    constexpr int MAXC = SH_T::MAXC;
    constexpr bool WIDE = SH_T::WIDE;
    static_assert(!WIDE || (NW == 2 && MAXR == WIDE_ROWS), "the 127-row instantiation spreads its solve over the two wavefronts of an env");
    const ModelPtr m_launch = (ModelPtr)(io.models + (size_t)env * io.model_stride);
    ModelPtr m = m_launch;
    /* the env's physical parameters: its own block, or the model's (wave-uniform either way: scalar loads) */
    const ParamPtr P = io.envparams ? (ParamPtr)(io.envparams + (size_t)env) : (ParamPtr)&m_launch->params;
    int lane = wv::lane();
    const int nq = m->nq, nv = m->nv, nu = m->nu, nbody = m->nbody, njnt = m->njnt;
    /* rows / contacts a substep may use: what this instantiation holds, within the model's caps (cm_model_t::maxefc / maxcon) */
    const int capr = m->maxefc < MAXR ? m->maxefc : MAXR, capc = m->maxcon < MAXC ? m->maxcon : MAXC;
    /* an instantiation with more rows / contacts runs behind this one (PhysIO::has_next) and the model may use them: a substep
     * that does not fit is handed over, from its start, instead of being capped */
    const bool can_hand_over = MAXR < WIDE_ROWS && io.has_next != 0 && io.progress != nullptr && (MAXR < m->maxefc || MAXC < m->maxcon);
    const double h = m->timestep;
    int warn = 0;

    /* ---------------- load state (coalesced, env-major) ---------------- */
    if (NW == 1 || wid == 0) {
    if (lane == 0) { S.cmd[0] = 0; S.cmd[1] = 0; S.cmd[2] = 0; S.cmd[3] = 0; S.cmd[4] = 0; }
    /* (the 127-row solve's turn words: wave 1 polls turn[1] for an odd value before anybody wrote it -- whatever an earlier env of
     * this workgroup or raw LDS left there must not look like one) */
    if constexpr (WIDE) { if (lane == 0) { S.x.turn[0] = 0; S.x.turn[1] = 0; S.x.turn[2] = 0; S.x.turn[3] = 0; } }
    if (lane < nq) S.qpos[lane] = io.qpos[(size_t)env * io.sq + lane];
    if (lane < nv) {
        S.qvel[lane] = io.qvel[(size_t)env * io.sqv + lane];
        S.qacc_ws[lane] = io.qacc_warmstart[(size_t)env * io.sv + lane];
    }
    if (lane < nu) S.ctrl[lane] = io.ctrl[(size_t)env * io.su + lane];
    /* centres of mass are computed for tree roots only, but rows of the static world (body 0: the floor's side of every
     * contact) are read too -- multiplied by an empty dof mask, which is harmless only if the value is finite.  LDS is
     * not initialised: give every row a value once per launch. */
    if (lane < NB && !wv::test_skip_com_init()) { S.com[lane][0] = 0.0; S.com[lane][1] = 0.0; S.com[lane][2] = 0.0; }
    if (io.drive_mode) {
        /* what the last step (or forward) of an earlier launch measured: the inputs of this launch's first drive-level pass */
        if (lane < m->nsensordata) S.sens[lane] = io.sensordata[(size_t)env * io.ssd + lane];
        if (lane < nu) S.actvel[lane] = io.actuator_velocity[(size_t)env * io.su + lane];
        drive_state_load(io, S, env, lane);
        drive_consts_load(io, S, m, env, lane);
    }
    }
    double time = io.time[env];

    /* ---------------- per-lane model indices (the fp64 constants are loaded where they are used, to keep
     * register live ranges short: the kernel runs one wave per SIMD and lives on its 512 VGPRs) ---------------- */
    /* lane = body */
    int b = lane;
    bool isbody = b < nbody;
    const int depth = isbody ? m->body_depth[b] : -1;
    const int bparent = isbody ? m->body_parentid[b] : 0;
    const int broot = isbody ? m->body_rootid[b] : -1;
    const int bjn = isbody ? m->body_jntnum[b] : 0;
    const int bj0 = (isbody && bjn > 0) ? m->body_jntadr[b] : 0;
    const int bend = isbody ? m->body_subtreeend[b] : 0;
    const unsigned long long bdofmask = isbody ? m->body_dofmask[b] : 0ull;
    /* first joint of the body (every Cassie body but the pelvis has at most one) */
    const int bjt = (isbody && bjn > 0) ? m->jnt_type[bj0] : -1;
    const int bjq = (isbody && bjn > 0) ? m->jnt_qposadr[bj0] : 0;
    /* lane = dof */
    int k_ = lane;
    bool isdof = k_ < nv;
    const int kjnt = isdof ? m->dof_jntid[k_] : 0;
    const int kbody = isdof ? m->dof_bodyid[k_] : 0;
    const int kjt = isdof ? m->jnt_type[kjnt] : -1;
    const int kda = isdof ? m->jnt_dofadr[kjnt] : 0;
    const int kqa = isdof ? m->jnt_qposadr[kjnt] : 0;
    const int kroot = isdof ? m->body_rootid[kbody] : 0;
    const int kbend = isdof ? m->body_subtreeend[kbody] : 0;
    const unsigned long long kdesc = isdof ? m->dof_descmask[k_] : 0ull;
    const unsigned long long kvelmask = isdof ? m->dof_velmask[k_] : 0ull;
    /* actuator acting on this dof (at most one per dof in the supported subset) */
    int kact = -1;
    for (int u = 0; u < nu; ++u) if (isdof && m->act_dofid[u] == k_) kact = u;
    const LaneIds ids = {nbody, nv, broot, bend, kjnt, kbody, kjt, kda, kroot, kbend, kdesc};
    wv::sync();

#include "env_step_wave1.inc"

    bool bailed = false, bailed_down = false;
    int sub = sub_start;
    for (; sub < nsub; ++sub) {
        /* Outputs that every substep recomputes (sensordata, qacc, actuator_velocity, xpos / xquat, the solver statistics)
         * are stored only by the LAST substep of a launch: the others' values would be overwritten anyway, and on this
         * hardware vector stores share the loads' completion counter (vmcnt), so a store that is still in flight holds up
         * the next stage's first model read. */
        const bool lastsub = sub == nsub - 1 || !io.integrate;
        /* Body quaternions feed only the IMU frame sensor, the site / body orientation read-outs and xquat_out -- all of
         * them values of the last substep (in a drive mode also of the one before it, see the sensors): the other
         * substeps carry rotation matrices only through the kinematic recursion. */
        const bool need_quat = lastsub || io.ext != nullptr || io.all_outputs_every_substep || (io.drive_mode && sub + 2 == nsub);
        /* divergence guard (mj_checkPos/mj_checkVel role): sticky flag, state left alone */
        {
            bool badv = false;
            if (lane < nq) { double v = S.qpos[lane]; badv |= !(v == v) || fabs(v) > 1e10; }
            if (lane < nv) { double v = S.qvel[lane]; badv |= !(v == v) || fabs(v) > 1e10; }
            if (wv::ballot(badv) != 0ull) {
                warn |= WARN_DIVERGED;
                if constexpr (NW == 2) { if (lane == 0) S.cmd[0] = 1; wv::block_barrier(); } /* (F) */
                break;
            }
        }
        if (io.drive_mode) {
            /* (the row-capped instantiation runs this pass after it knows that the substep fits its rows, see below: a
             * substep it hands over must not have advanced the filter histories and delay lines) */
            /* (behind the collision verdict, below: a substep this instantiation hands over must not have advanced the filter
             * histories and delay lines.  Two-wave form: wave 1, once wave 0's collision verdict is in) */
        } else if (io.pd_ptarget) {
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
        if constexpr (NW == 2 && prio_kin != CK_PRIO_W0) wv::set_priority<prio_kin>();

        /* ================= P1 kinematics ================= */
        /* Every body first builds, in parallel, its transform relative to its parent INCLUDING its joints
         * (rotation matrix Rl, offset pl, and the local quaternion for the few consumers of xquat); the recursion
         * over the tree levels is then just R = Rp Rl, p = pp + Rp pl -- no trigonometry, quaternions or square
         * roots on the dependent chain. */
        double Rl[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pl[3] = {0, 0, 0}, qlq[4] = {1, 0, 0, 0};
        /* rounds of the recursion over the tree below (radix-3 pointer jumping, cm_model_t::body_anc3): a compile-time
         * topology knows how deep its tree is (Cassie: 9 levels -> two rounds) */
        constexpr int kin_rounds = [] {
            if constexpr (TOPO::is_static) { int r = 0, n = 1; while (n < TOPO::body_levels) { n *= 3; ++r; } return r < 1 ? 1 : r; }
            else return 3;
        }();
        int kanc[2 * kin_rounds]; /* the jump ancestors, issued with the stage's other model reads */
#pragma unroll
        for (int r = 0; r < 2 * kin_rounds; ++r) kanc[r] = isbody ? m->body_anc3[b][r] : 0;
        const bool isfree = bjt == CM_JNT_FREE;
        if constexpr (TOPO::is_static) {
            /* Compile-time topologies come with kin_simple models (cm_model.h): a body's joints are up to CM_MAXSLIDE slides
             * and then at most ONE rotational joint, described by one record per body.  Every constant of the stage is one
             * level of reads away (all requested together), the slides are unpredicated FMAs with zero axes where a body has
             * none, the rotation is evaluated once, and no lane waits for another body's joint loop (the pelvis of
             * model/cassie.xml:81-84 has four joints: in the general loop below its lane runs three iterations alone, each
             * behind its own chain of body -> joint -> parameter reads). */
            {
                /* (lanes beyond the bodies read the world's record: no joints, identity frame -- what they hold anyway) */
                const auto *kr = &m->body_kin[isbody ? b : 0];
                const int nsl = kr->nslide, j0 = kr->jnt0, jr = kr->rot_jnt, jt = kr->rot_type, qa = kr->rot_qadr;
                int sqa[CM_MAXSLIDE];
                double sref[CM_MAXSLIDE], sax[CM_MAXSLIDE][3], spo[CM_MAXSLIDE][3];
#pragma unroll
                for (int sl = 0; sl < CM_MAXSLIDE; ++sl) {
                    sqa[sl] = kr->slide_qadr[sl]; sref[sl] = kr->slide_ref[sl];
                    for (int i = 0; i < 3; ++i) { sax[sl][i] = kr->slide_axis_p[sl][i]; spo[sl][i] = kr->slide_pos_p[sl][i]; }
                }
                const double jref = kr->rot_ref;
                double jp[3], ja[3], jpp[3], xl[3], q0[4];
                for (int i = 0; i < 3; ++i) { jp[i] = kr->rot_pos[i]; ja[i] = kr->rot_axis[i]; jpp[i] = kr->rot_pos_p[i]; xl[i] = kr->rot_axis_p[i]; }
                for (int i = 0; i < 9; ++i) Rl[i] = kr->mat[i];
                for (int i = 0; i < 3; ++i) pl[i] = kr->pos[i];
                for (int i = 0; i < 4; ++i) q0[i] = kr->quat[i];
#pragma unroll
                for (int sl = 0; sl < CM_MAXSLIDE; ++sl) {
                    const double d = S.qpos[sqa[sl]] - sref[sl];
                    if (sl < nsl) for (int i = 0; i < 3; ++i) { S.x.s.xanchor[j0 + sl][i] = pl[i] + spo[sl][i]; S.x.s.xaxis[j0 + sl][i] = sax[sl][i]; }
                    for (int i = 0; i < 3; ++i) pl[i] += sax[sl][i] * d;
                }
                for (int i = 0; i < 4; ++i) qlq[i] = q0[i];
                double sn, cs; /* (evaluated by every lane: its range check is a wave vote) */
                sincos_bounded(jt == CM_JNT_HINGE ? 0.5 * (S.qpos[qa] - jref) : 0.0, sn, cs);
                if (jr >= 0) {
                    double qj[4];
                    if (jt == CM_JNT_HINGE) {
                        qj[0] = cs; qj[1] = ja[0] * sn; qj[2] = ja[1] * sn; qj[3] = ja[2] * sn;
                    } else { /* ball, or free: position + quaternion (the record's frame constants are the identity) */
                        const int qo = jt == CM_JNT_FREE ? qa + 3 : qa;
                        for (int i = 0; i < 4; ++i) qj[i] = S.qpos[qo + i];
                        normalize4_fast(qj);
                        if (jt == CM_JNT_FREE) for (int i = 0; i < 3; ++i) pl[i] = S.qpos[qa + i];
                    }
                    double al[3], Rq[9], Rn[9], r[3];
                    for (int i = 0; i < 3; ++i) { al[i] = pl[i] + jpp[i]; S.x.s.xanchor[jr][i] = al[i]; S.x.s.xaxis[jr][i] = xl[i]; }
                    quat2mat(Rq, qj);
                    for (int i = 0; i < 3; ++i)
                        for (int c = 0; c < 3; ++c) Rn[3 * i + c] = Rl[3 * i] * Rq[c] + Rl[3 * i + 1] * Rq[3 + c] + Rl[3 * i + 2] * Rq[6 + c];
                    for (int i = 0; i < 9; ++i) Rl[i] = Rn[i];
                    if (need_quat) mulquat(qlq, q0, qj);
                    /* rotation about the anchor: the origin moves so that the anchor stays put */
                    mulmatvec3(r, Rl, jp);
                    for (int i = 0; i < 3; ++i) pl[i] = al[i] - r[i];
                }
            }
        } else
        if (isbody && b > 0) {
            double bpos[3], bq[4];
            for (int i = 0; i < 3; ++i) bpos[i] = m->body_pos[b][i];
            for (int i = 0; i < 4; ++i) bq[i] = m->body_quat[b][i];
            if (isfree) {
                for (int i = 0; i < 3; ++i) pl[i] = S.qpos[bjq + i];
                for (int i = 0; i < 4; ++i) qlq[i] = S.qpos[bjq + 3 + i];
                normalize4(qlq);
                quat2mat(Rl, qlq);
                for (int i = 0; i < 3; ++i) { S.x.s.xanchor[bj0][i] = pl[i]; S.x.s.xaxis[bj0][i] = (i == 2) ? 1.0 : 0.0; }
            } else {
                for (int i = 0; i < 9; ++i) Rl[i] = m->body_mat[b][i];
                for (int i = 0; i < 3; ++i) pl[i] = bpos[i];
                for (int i = 0; i < 4; ++i) qlq[i] = bq[i];
                /* one iteration for every Cassie body but the pelvis (3 slides + ball).  A joint's constants are one level
                 * of model reads, and the next joint's are requested before this one is processed, so the extra
                 * iterations (where a single lane is active) do not each wait out a memory round trip */
                int jt_n = 0, qa_n = 0;
                double jp_n[3] = {0, 0, 0}, ja_n[3] = {0, 0, 0}, ref_n = 0;
                if (bjn > 0) {
                    jt_n = m->jnt_type[bj0]; qa_n = m->jnt_qposadr[bj0]; ref_n = m->jnt_ref[bj0];
                    for (int i = 0; i < 3; ++i) { jp_n[i] = m->jnt_pos[bj0][i]; ja_n[i] = m->jnt_axis[bj0][i]; }
                }
                for (int jj = 0; jj < bjn; ++jj) {
                    const int j = bj0 + jj, jt = jt_n, qa = qa_n;
                    const double jref = ref_n;
                    double jp[3] = {jp_n[0], jp_n[1], jp_n[2]}, ja[3] = {ja_n[0], ja_n[1], ja_n[2]};
                    if (jj + 1 < bjn) {
                        jt_n = m->jnt_type[j + 1]; qa_n = m->jnt_qposadr[j + 1]; ref_n = m->jnt_ref[j + 1];
                        for (int i = 0; i < 3; ++i) { jp_n[i] = m->jnt_pos[j + 1][i]; ja_n[i] = m->jnt_axis[j + 1][i]; }
                    }
                    /* joint anchor and axis in the PARENT frame (turned into world coordinates after the recursion) */
                    double al[3], xl[3];
                    mulmatvec3(al, Rl, jp);
                    for (int i = 0; i < 3; ++i) al[i] += pl[i];
                    mulmatvec3(xl, Rl, ja);
                    for (int i = 0; i < 3; ++i) { S.x.s.xanchor[j][i] = al[i]; S.x.s.xaxis[j][i] = xl[i]; }
                    if (jt == CM_JNT_SLIDE) {
                        const double sl = S.qpos[qa] - jref;
                        for (int i = 0; i < 3; ++i) pl[i] += xl[i] * sl;
                    } else {
                        double qj[4];
                        if (jt == CM_JNT_BALL) { for (int i = 0; i < 4; ++i) qj[i] = S.qpos[qa + i]; normalize4(qj); }
                        else {
                            const double ang = S.qpos[qa] - jref;
                            const double sn = sin(0.5 * ang);
                            qj[0] = cos(0.5 * ang); qj[1] = ja[0] * sn; qj[2] = ja[1] * sn; qj[3] = ja[2] * sn;
                        }
                        double Rq[9], Rn[9], r[3];
                        quat2mat(Rq, qj);
                        for (int i = 0; i < 3; ++i)
                            for (int c = 0; c < 3; ++c) Rn[3 * i + c] = Rl[3 * i] * Rq[c] + Rl[3 * i + 1] * Rq[3 + c] + Rl[3 * i + 2] * Rq[6 + c];
                        for (int i = 0; i < 9; ++i) Rl[i] = Rn[i];
                        if (need_quat) mulquat(qlq, qlq, qj);
                        /* rotation about the anchor: the origin moves so that the anchor stays put */
                        mulmatvec3(r, Rl, jp);
                        for (int i = 0; i < 3; ++i) pl[i] = al[i] - r[i];
                    }
                }
            }
        }
        CK_STAMP(16);
        /* Model constants of the stages behind the recursion (inertial frames, joint anchors to the world frame, geoms, centres
         * of mass, cinert) are requested before it: their round trips through the memory system then run under the
         * recursion's four LDS rounds instead of in front of each of those stages. */
        const int pf_b = isbody ? b : 0, pf_g = lane < m->ngeom ? lane : -1, pf_gs = pf_g >= 0 ? pf_g : 0;
        const int pf_jpb = lane < njnt ? m->jnt_parentbody[lane] : -1, pf_gb = m->geom_bodyid[pf_gs];
        const double pf_mass = P->body_mass[pf_b];
        double pf_ipos[3], pf_imat[9], pf_iner[3], pf_gpos[3], pf_gmat[9];
        for (int i = 0; i < 3; ++i) { pf_ipos[i] = P->body_ipos[pf_b][i]; pf_iner[i] = P->body_inertia[pf_b][i]; pf_gpos[i] = m->geom_pos[pf_gs][i]; }
        for (int i = 0; i < 9; ++i) { pf_imat[i] = m->body_imat[pf_b][i]; pf_gmat[i] = m->geom_mat[pf_gs][i]; }
        /* recursion over the tree by radix-3 pointer jumping: in round r every body composes its partial transform with
         * those of its 3^r-th and 2 * 3^r-th ancestors, after which it holds the product of the local transforms of its
         * 3^(r+1) nearest ancestors-or-self -- two LDS round trips for Cassie's nine levels where doubling needed four, for
         * the same four compositions.  The partial products ping-pong between the pose tiles and a second buffer laid over the
         * (still unused) cinert / crb tiles, arranged so that the last round lands in the pose tiles. */
        double xm[9], xp[3], xq[4];
        for (int i = 0; i < 9; ++i) xm[i] = Rl[i];
        for (int i = 0; i < 3; ++i) xp[i] = pl[i];
        for (int i = 0; i < 4; ++i) xq[i] = qlq[i];
        {
            double *bufB = &S.x.s.cinert[0][0]; /* 16 doubles per body, R(9) p(3) q(4), at a stride of 17: a 128-byte stride
                                                   would put all lanes on two LDS banks */
            auto park = [&](bool in_pose_tiles) {
                if (lane < NB) {
                    if (in_pose_tiles) {
                        for (int i = 0; i < 9; ++i) S.x.s.xmat[lane][i] = xm[i];
                        for (int i = 0; i < 3; ++i) S.x.s.xpos[lane][i] = xp[i];
                        if (need_quat) for (int i = 0; i < 4; ++i) S.x.s.xquat[lane][i] = xq[i];
                    } else {
                        for (int i = 0; i < 9; ++i) bufB[lane * 17 + i] = xm[i];
                        for (int i = 0; i < 3; ++i) bufB[lane * 17 + 9 + i] = xp[i];
                        if (need_quat) for (int i = 0; i < 4; ++i) bufB[lane * 17 + 12 + i] = xq[i];
                    }
                }
                wv::sync();
            };
            /* compose an ancestor's partial transform (Ra, pa, qa) in front of this body's */
            auto compose = [&](const double *Ra, const double *pa, const double *qa) {
                double Rn[9], pn[3];
                for (int i = 0; i < 3; ++i) {
                    for (int c = 0; c < 3; ++c) Rn[3 * i + c] = Ra[3 * i] * xm[c] + Ra[3 * i + 1] * xm[3 + c] + Ra[3 * i + 2] * xm[6 + c];
                    pn[i] = pa[i] + (Ra[3 * i] * xp[0] + Ra[3 * i + 1] * xp[1] + Ra[3 * i + 2] * xp[2]);
                }
                for (int i = 0; i < 9; ++i) xm[i] = Rn[i];
                for (int i = 0; i < 3; ++i) xp[i] = pn[i];
                if (need_quat) mulquat(xq, qa, xq);
            };
            constexpr bool first_in_pose_tiles = (kin_rounds & 1) == 0;
            park(first_in_pose_tiles);
#pragma unroll
            for (int r = 0; r < kin_rounds; ++r) {
                const bool from_pose_tiles = first_in_pose_tiles == ((r & 1) == 0);
                const int a1 = kanc[2 * r], a2 = kanc[2 * r + 1];
                double R1[9], p1[3], q1[4] = {1, 0, 0, 0}, R2[9], p2[3], q2[4] = {1, 0, 0, 0};
                if (from_pose_tiles) {
                    for (int i = 0; i < 9; ++i) { R1[i] = S.x.s.xmat[a1][i]; R2[i] = S.x.s.xmat[a2][i]; }
                    for (int i = 0; i < 3; ++i) { p1[i] = S.x.s.xpos[a1][i]; p2[i] = S.x.s.xpos[a2][i]; }
                    if (need_quat) for (int i = 0; i < 4; ++i) { q1[i] = S.x.s.xquat[a1][i]; q2[i] = S.x.s.xquat[a2][i]; }
                } else {
                    for (int i = 0; i < 9; ++i) { R1[i] = bufB[a1 * 17 + i]; R2[i] = bufB[a2 * 17 + i]; }
                    for (int i = 0; i < 3; ++i) { p1[i] = bufB[a1 * 17 + 9 + i]; p2[i] = bufB[a2 * 17 + 9 + i]; }
                    if (need_quat) for (int i = 0; i < 4; ++i) { q1[i] = bufB[a1 * 17 + 12 + i]; q2[i] = bufB[a2 * 17 + 12 + i]; }
                }
                /* the world's transform is the identity: nothing to compose beyond the root */
                if (a1 > 0) compose(R1, p1, q1);
                if (a2 > 0) compose(R2, p2, q2);
                park(!from_pose_tiles);
            }
        }
        if (b == 0) for (int i = 0; i < 3; ++i) S.x.s.xipos[0][i] = 0;
        /* inertial frames, and joint anchors / axes from the parent frame to the world frame */
        double ximat[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (isbody && b > 0) {
            double xi[3];
            const double *ip = pf_ipos, *im = pf_imat;
            mulmatvec3(xi, xm, ip);
            for (int i = 0; i < 3; ++i) S.x.s.xipos[b][i] = xp[i] + xi[i];
            for (int i = 0; i < 3; ++i)
                for (int c = 0; c < 3; ++c) ximat[3 * i + c] = xm[3 * i] * im[c] + xm[3 * i + 1] * im[3 + c] + xm[3 * i + 2] * im[6 + c];
        }
        const int jpb = pf_jpb;
        if (jpb >= 0) {
            const int pb = jpb;
            double al[3] = {S.x.s.xanchor[lane][0], S.x.s.xanchor[lane][1], S.x.s.xanchor[lane][2]};
            double xl[3] = {S.x.s.xaxis[lane][0], S.x.s.xaxis[lane][1], S.x.s.xaxis[lane][2]}, aw[3], xw[3];
            mulmatvec3(aw, S.x.s.xmat[pb], al);
            mulmatvec3(xw, S.x.s.xmat[pb], xl);
            for (int i = 0; i < 3; ++i) { S.x.s.xanchor[lane][i] = aw[i] + S.x.s.xpos[pb][i]; S.x.s.xaxis[lane][i] = xw[i]; }
        }
        if constexpr (NW == 2) wv::block_barrier(); /* F */
        if constexpr (NW == 2 && prio_kin != CK_PRIO_W0) wv::set_priority<CK_PRIO_W0>();
        CK_STAMP(1);

        /* geoms (lane = collision geom) */
        if (pf_g >= 0) {
            const int g = pf_g, gb = pf_gb;
            const double *gp = pf_gpos, *gm = pf_gmat;
            double t[3];
            const double *R = S.x.s.xmat[gb];
            mulmatvec3(t, R, gp);
            for (int i = 0; i < 3; ++i) S.x.s.geom_xpos[g][i] = t[i] + S.x.s.xpos[gb][i];
            for (int i = 0; i < 3; ++i)
                for (int c = 0; c < 3; ++c) S.x.s.geom_xmat[g][3 * i + c] = R[3 * i] * gm[c] + R[3 * i + 1] * gm[3 + c] + R[3 * i + 2] * gm[6 + c];
        }
        if constexpr (NW == 2) wv::sync(); /* (the collision stage is next in this wave: its lanes read other lanes' geoms) */

        CK_STAMP(17);
        /* (com of every kinematic tree, cinert, cdof, composite inertias, M's columns: mass_matrix_columns, defined ahead of
         * the loop -- in the two-wave form wave 1 runs them, and the factorisations, beside this wave's collision, velocity
         * and constraint-row stages) */
        double col[NVP], colh[NVP]; /* col[i] = M[i][lane] (i >= lane); colh: same for M + h*diag(damping) */
        if constexpr (NW == 1) mass_matrix_columns<NVP, TOPO, FEAT, NW>(io, S, m, env, ids, pf_mass, pf_iner, ximat, col, colh);

        /* ================= P3 factor M and M + hB in registers; park the factors in LDS ================= */
        constexpr bool by_height = TOPO::is_static;
        if constexpr (NW == 2) {
            /* (wave 1) */
        } else if constexpr (by_height) {
            factor_pair_by_height<NVP, TOPO>(m, P, h, S, col, colh, lane);
        } else {
            factor_pair_in_registers<NVP, TOPO>(m, P, h, col, colh, lane, nv, S.dinv, S.rsd, S.dinvH);
            if (isdof) {
#pragma unroll
                for (int k = 1; k < NVP; ++k) {
                    if (TOPO::is_static ? k >= TOPO::nv : k >= nv) continue;
                    if (k > k_) { S.Lp[CK_TRI(k, k_)] = col[k]; S.LHp[CK_TRI(k, k_)] = colh[k]; }
                }
            }
        }
        if constexpr (NW == 1) CK_STAMP(4);

#include "env_step_collision.inc"

        /* ================= P6 velocities and bias forces ================= */
        /* Spatial vectors are all taken about the tree's centre of mass, so the velocity at the end of dof k's chain is
         * the plain sum of cdof_a * qvel_a over k and its ancestor dofs.  Lane = dof: the terms go to an LDS tile and
         * the chain sums are a pointer-jumping prefix over the 1st/2nd/4th/8th/16th ancestor dofs, in place (one wave:
         * every lane's read of a round is issued before any lane's write).  A body's velocity is the sum at its last
         * dof; the velocity entering a joint is the sum at dof_vinsrc. */
        double mycvel[6], mycacc[6];
        /* (danc / blast / kvin: requested ahead of the collision stage)  The per-dof records of the passive / actuation stage
         * behind this one (cm_model_t::dof_*: damping, the joint's spring, the actuator on the dof -- one level of
         * unconditional reads; dofs without a spring / actuator carry zero stiffness / gear) are requested now. */
#pragma unroll
        for (int r = 0; r < 3 * chain_rounds; ++r) { wv::keep(danc[r]); if (!isdof) danc[r] = -1; }
        wv::keep(blast); wv::keep(kvin);
        if (!isbody) blast = -1;
        if (!isdof) kvin = -1;
        const int kd = isdof ? k_ : 0;
        const double kdamp = P->dof_damping[kd], kstiff = m->dof_stiffness[kd], kref = m->dof_springref[kd];
        const double kgear = m->dof_gear[kd], klo = m->dof_ctrl_lo[kd], khi = m->dof_ctrl_hi[kd];
        const int kq = m->dof_qadr[kd], ka = m->dof_act[kd];
        auto chain_sums = [&](double (&acc)[6]) { /* acc: this dof's term in, its chain sum out; tile: buf */
            if (lane < NVP) for (int t = 0; t < 6; ++t) S.x.s.buf[lane][t] = acc[t];
            wv::sync();
#pragma unroll
            for (int r = 0; r < chain_rounds; ++r) {
                double up[3][6];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int a = danc[3 * r + i] >= 0 ? danc[3 * r + i] : 0;
                    for (int t = 0; t < 6; ++t) up[i][t] = S.x.s.buf[a][t];
                }
                wv::sync();
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    if (danc[3 * r + i] >= 0) for (int t = 0; t < 6; ++t) acc[t] += up[i][t];
                if (lane < NVP) for (int t = 0; t < 6; ++t) S.x.s.buf[lane][t] = acc[t];
                wv::sync();
            }
        };
        {
            double term[6];
            const double qd = isdof ? S.qvel[lane < NVP ? lane : 0] : 0.0;
            for (int t = 0; t < 6; ++t) term[t] = (lane < NVP ? S.cdof[lane < NVP ? lane : 0][t] : 0.0) * qd;
            chain_sums(term);
            for (int t = 0; t < 6; ++t) mycvel[t] = blast >= 0 ? S.x.s.buf[blast >= 0 ? blast : 0][t] : 0.0;
            double vin[6];
            for (int t = 0; t < 6; ++t) vin[t] = kvin >= 0 ? S.x.s.buf[kvin >= 0 ? kvin : 0][t] : 0.0;
            if (lane < NB) for (int t = 0; t < 6; ++t) S.x.s.cvel[lane][t] = mycvel[t];
            /* lane = dof: time derivative of the motion axis = (velocity entering the joint) x axis; zero for the
             * translational dofs of a free joint */
            if (lane < NVP) {
                double cdd[6] = {0, 0, 0, 0, 0, 0}, cd[6];
                for (int i = 0; i < 6; ++i) cd[i] = S.cdof[lane][i];
                if (isdof && !(kjt == CM_JNT_FREE && k_ - kda < 3)) cross_motion(cdd, vin, cd);
                for (int i = 0; i < 6; ++i) S.x.s.cdof_dot[lane][i] = cdd[i];
                /* next: the chain sums of cdof_dot * qvel give the bias accelerations */
                for (int t = 0; t < 6; ++t) term[t] = cdd[t] * qd;
            } else {
                for (int t = 0; t < 6; ++t) term[t] = 0;
            }
            wv::sync();
            CK_STAMP(23);
            chain_sums(term);
            for (int t = 0; t < 6; ++t) mycacc[t] = blast >= 0 ? S.x.s.buf[blast >= 0 ? blast : 0][t] : 0.0;
            mycacc[3] -= m->gravity[0]; mycacc[4] -= m->gravity[1]; mycacc[5] -= m->gravity[2]; /* -g on the world */
            wv::sync();
            if (lane < NB) for (int t = 0; t < 6; ++t) S.x.s.buf[lane][t] = mycacc[t]; /* kept for the accelerometers */
            wv::sync();
        }
        if (lane < NB) {
            double f6[6] = {0, 0, 0, 0, 0, 0};
            if (isbody && b > 0) {
                double t1[6], t2[6], t3[6], ci[10];
                for (int i = 0; i < 10; ++i) ci[i] = S.x.s.cinert[b][i];
                mul_inert_vec(t1, ci, mycacc);
                mul_inert_vec(t2, ci, mycvel);
                cross_force(t3, mycvel, t2);
                for (int i = 0; i < 6; ++i) f6[i] = t1[i] + t3[i];
            }
            for (int i = 0; i < 6; ++i) S.x.s.cfrc[lane][i] = f6[i];
        }
        wv::sync();
        CK_STAMP(24);
        /* The equality rows' constants are requested here, two stages ahead of the rows' geometry: with every equality active (the
         * usual case, closed form in the row assignment below) row r belongs to equality r / 3 */
        if constexpr (NW == 2) {
            /* the body forces are in LDS: wave 1 projects them on the motion axes and forms qfrc_smooth behind its
             * factorisations (bias_forces_and_qfrc_smooth), while this wave goes on to the constraint rows */
            wv::publish(&S.cmd[1], sub + 1);
        } else bias_forces_and_qfrc_smooth<NVP>(io, S, m, env, ids, kdamp, kstiff, kref, kgear, klo, khi, kq, ka);
        const int pf_eq = lane < 3 * m->neq ? lane / 3 : 0;
        /* (only what the rows' LDS reads hang on: the bodies and their roots; the anchors, masks and solver parameters are read in place,
         * where their trip to memory runs under those LDS reads -- carrying them too pushes launch-long values into scratch) */
        int pf_eb1 = m->eq_body1[pf_eq], pf_eb2 = m->eq_body2[pf_eq], pf_er1 = m->eq_root[pf_eq][0], pf_er2 = m->eq_root[pf_eq][1];

#include "env_step_rows.inc"

#include "env_step_solve.inc"

    /* ---------------- store state ---------------- */
    if (io.progress && lane == 0 && (!io.resume || bailed)) {
        /* (a pass behind the fast kernel leaves the record of an env it completes; one that hands the env on -- the 63-row pass of a
         * model that may use 127 -- moves it to the substep the next pass starts from) */
        io.progress[env] = bailed ? sub : nsub;
        if (bailed && !bailed_down && io.handover_out_list) io.handover_out_list[io.env0 + wv::atomic_add(io.handover_out_count, 1)] = env;
    }
    if (io.integrate && io.drive_mode) {
        drive_state_store(io, S, env, lane);
        if (lane < nu) io.ctrl[(size_t)env * io.su + lane] = S.ctrl[lane]; /* the applied torque: d->ctrl of the reference */
        if (bailed) {
            /* handed over in the middle of a launch: what the last completed substep measured is the next drive-level pass's
             * input, and it lives in LDS only */
            if (lane < m->nsensordata) io.sensordata[(size_t)env * io.ssd + lane] = S.sens[lane];
            if (lane < nu) io.actuator_velocity[(size_t)env * io.su + lane] = S.actvel[lane];
            if (lane < CM_NUM_DRIVES) { /* ... and the drive positions / velocities CM_DRIVE_PD's law reads come back from the measurement block */
                double *meas = io.meas + (size_t)env * CM_MEAS_DIM;
                meas[CM_MEAS_DRIVE_POS + lane] = S.drv_pos[lane]; meas[CM_MEAS_DRIVE_VEL + lane] = S.drv_vel[lane];
            }
        }
    }
    if (io.integrate) {
        if (lane < nq) io.qpos[(size_t)env * io.sq + lane] = S.qpos[lane];
        if (lane < nv) {
            io.qvel[(size_t)env * io.sqv + lane] = S.qvel[lane];
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
    return (warn & WARN_DIVERGED) ? -2 : (bailed ? (bailed_down ? sub | ENV_STEP_WENT_DOWN : sub) : nsub);
}

/* one workgroup (NW wavefronts) per environment.
 *
 * MAXR < CM_MAXEFC is the row-capped FAST instantiation: the constraint stages hold MAXR rows (31: a Cassie on its feet uses
 * 20 .. 28), which halves the register arrays of the solve (A's rows, the PGS row chain) -- 112 instead of 316 bytes of
 * scratch per lane -- and shortens every unrolled row loop.  An env whose substep needs more rows is handed over to the full
 * instantiation through PhysIO::progress (see there); results are bit for bit those of the full instantiation alone, because
 * the arithmetic of a substep that fits is the same in both. */
/* WALK: the instantiation is the pass behind the fast kernel in its list-walking form (PhysIO::handover_list): a small grid whose
 * workgroups each finish the handed-over envs blockIdx, blockIdx + gridDim, ... of the list.  (A template parameter and not a
 * run-time branch: env_step is inlined, and two call sites would be two copies of it in one kernel.) */
/* WPS: wavefronts per SIMD the registers are budgeted for (NW: 512 / NW registers a lane; 1 with NW = 2: two wavefronts per env with
 * 512 registers each, for batches that cannot fill the chip anyway -- a single simulator) */
/* INROWS (round 6; 0 = none): rows of the instantiation whose code finishes, inside this workgroup, a substep that needs more rows
 * than this one holds -- the fast kernel then goes on with the next substep itself (PhysIO::inplace_*).  A handed-over env used to
 * take ALL its remaining substeps to the pass behind the kernel, one serial chain per env while the range's stream waited (23 % of
 * the stress workload's time for 0.3 % of the env-launches); in place it costs that one substep's longer code path.  Results are
 * the same bit for bit: a substep is computed by the same instructions whichever instantiation holds it. */
template <int NVP, class TOPO, int FEAT = FEAT_ALL, int MAXR = MID_ROWS, int NW = 1, bool WALK = false, int WPS = NW, int INROWS = 0>
WV_GLOBAL void __launch_bounds__(WV_WAVE * NW) WV_OCC WV_WAVES_PER_SIMD(WPS) cassie_step_kernel(PhysIO io) {
    WV_SHARED EnvShared<NVP, LPack<TOPO, NVP>::count, MAXR> S;
    const int slot = wv::env_id();
    if constexpr (WALK) {
        /* the pass behind the fast kernel, as a small grid walking the hand-over list (PhysIO::handover_list) */
        wv::test_launch_hook(&S, sizeof S);
        const int count = wv::opaque(wv::shfl_i(wv::lane() == 0 ? wv::atomic_add(io.handover_count, 0) : 0, 0));
        for (int idx = slot; idx < count; idx += wv::grid_size()) {
            const int env = io.handover_list[io.env0 + idx];
            const long long t0 = io.cost ? wv::clock() : 0, t0w = io.cost_wall ? wv::wall_clock() : 0;
            env_step<NVP, TOPO, FEAT, MAXR, NW>(io, S, env, io.progress[env], io.nsub);
            if (io.cost && wv::lane() == 0 && (NW == 1 || wv::wave_id() == 0)) {
                io.cost[env] += (unsigned)((wv::clock() - t0) >> 6);
                if (io.cost_wall) io.cost_wall[env] += (unsigned)(wv::wall_clock() - t0w);
            }
            if constexpr (NW == 2) wv::block_barrier(); /* both waves are done with this env before either starts the next */
            else wv::sync();
        }
        /* every workgroup has read the count before it draws its ticket: the last one to draw may clear it */
        if (wv::lane() == 0 && (NW == 1 || wv::wave_id() == 0) && wv::atomic_add(io.handover_count + 1, 1) == wv::grid_size() - 1) {
            io.handover_count[0] = 0; io.handover_count[1] = 0;
            if (io.handover_seen) *io.handover_seen = count;
        }
    } else {
    /* a launch in chunks (PhysIO::nchunk): workgroup w = chunk w / nenv of env slot w % nenv */
    const int chunk = io.nchunk > 1 ? wv::env_id() / io.nenv : 0;
    const int eslot = slot - chunk * io.nenv;
    if (chunk >= (io.nchunk > 1 ? io.nchunk : 1)) return;
    wv::test_launch_hook(&S, sizeof S); /* CPU emulator only (poisons LDS so that a read-before-write shows); empty on the device */
    const int env = io.order ? io.order[io.env0 + eslot] : io.env0 + eslot; /* order holds absolute env ids, sorted range by range */
    int sub_start = io.resume ? io.progress[env] : 0, sub_end = io.nsub;
    if (io.nchunk > 1) {
        const int per = (io.nsub + io.nchunk - 1) / io.nchunk;
        sub_start = chunk * per;
        sub_end = sub_start + per < io.nsub ? sub_start + per : io.nsub;
    }
    if (sub_start >= io.nsub) return; /* resume pass: the fast instantiation finished this env; chunks: none left for this one */
    if (chunk > 0) {
        /* the chunk before this one has stored the env's state (and, had it met a substep with too many rows, handed the env over) */
        if (!wv::wait_global<NW>(io.chunk_flag + env, io.chunk_seq, chunk)) {
            /* the producer's stores sit in ANOTHER XCD's L2 (wave.h): what this chunk is about to load may be stale.  The env is
             * flagged -- its results are not to be trusted -- and the launcher told, which launches in one piece from then on */
            if (wv::lane() == 0) { wv::atomic_or(io.warn + env, WARN_CHUNK_PLACEMENT); if (io.chunk_fault) *io.chunk_fault = 1; }
        }
        if (io.progress[env] != sub_start) {
            if (sub_end < io.nsub) wv::publish_global<NW>(io.chunk_flag + env, io.chunk_seq, chunk + 1);
            return;
        }
    }
    const long long t0 = io.cost ? wv::clock() : 0, t0w = io.cost_wall ? wv::wall_clock() : 0;
    if constexpr (INROWS > 0) {
        static_assert(NW == 2 && INROWS > MAXR && INROWS <= MID_ROWS, "the in-place form: a two-wave fast instantiation with the 63-row code behind it");
        static_assert(sizeof(EnvShared<NVP, LPack<TOPO, NVP>::count, MAXR>) == sizeof(EnvShared<NVP, LPack<TOPO, NVP>::count, INROWS>), "both instantiations use the env's LDS block alike");
        auto &S2 = reinterpret_cast<EnvShared<NVP, LPack<TOPO, NVP>::count, INROWS> &>(S);
        PhysIO io2 = io;            /* what the substep's inner call sees: it hands on (if at all) to the 127-row pass's list */
        io2.has_next = io.inplace_has_next; io2.handover_out_list = io.inplace_out_list; io2.handover_out_count = io.inplace_out_count;
        io2.down_rows = io.inplace_stay_rows;
        const bool stay = io.inplace_stay_rows > 0;
        bool did = false;
        for (int s = sub_start;;) {
            int at = env_step<NVP, TOPO, FEAT, MAXR, NW>(io, S, env, s, sub_end);
            /* (wave 0 knows where the call ended; both waves' stores of the state are out before either loads it again) */
            if (wv::lane() == 0 && wv::wave_id() == 0) S.cmd[5] = at;
            wv::drain_vmem(); wv::block_barrier();
            at = wv::opaque(S.cmd[5]);
            if (at < 0 || at >= sub_end) break;                 /* done (or the state diverged: flagged, left alone) */
            did = true;
            /* the 63-row code: for this one substep, or (PhysIO::inplace_stay_rows) until a substep fits the fast code again */
            int at2 = env_step<NVP, TOPO, FEAT, INROWS, NW>(io2, S2, env, at, stay ? sub_end : at + 1);
            if (wv::lane() == 0 && wv::wave_id() == 0) S.cmd[5] = at2;
            wv::drain_vmem(); wv::block_barrier();
            at2 = wv::opaque(S.cmd[5]);
            if (at2 < 0) break;                                 /* diverged */
            const bool down = (at2 & ENV_STEP_WENT_DOWN) != 0;
            at2 &= ENV_STEP_WENT_DOWN - 1;
            if (at2 >= sub_end) break;                          /* that was the last substep */
            if (stay ? !down : at2 != at + 1) break;            /* handed on to the 127-row pass */
            s = at2;
        }
        if (did && io.inplace_count && wv::lane() == 0 && wv::wave_id() == 0) wv::atomic_add(io.inplace_count, 1);
    } else env_step<NVP, TOPO, FEAT, MAXR, NW>(io, S, env, sub_start, sub_end);
    if (sub_end < io.nsub) wv::publish_global<NW>(io.chunk_flag + env, io.chunk_seq, chunk + 1);
    if (io.prof && wv::lane() == 0) io.prof[(size_t)env * NSTAMP + 40 + (NW == 2 ? wv::wave_id() : 0)] = wv::hw_id(); /* (profiling aid: the CU / SIMD of the wave) */
    if (io.prof && wv::lane() == 0 && (NW == 1 || wv::wave_id() == 0)) { /* (profiling aid: the shader clock against the 100 MHz wall clock at the env's end) */
        io.prof[(size_t)env * NSTAMP + 42] = wv::clock(); io.prof[(size_t)env * NSTAMP + 43] = wv::wall_clock();
    }
    if (io.cost && wv::lane() == 0 && (NW == 1 || wv::wave_id() == 0)) { /* 64-clock units: 32 bits hold minutes */
        const unsigned c = (unsigned)((wv::clock() - t0) >> 6);
        io.cost[env] = io.resume || chunk > 0 ? io.cost[env] + c : c;
        if (io.cost_wall) { const unsigned cw = (unsigned)(wv::wall_clock() - t0w); io.cost_wall[env] = io.resume || chunk > 0 ? io.cost_wall[env] + cw : cw; }
    }
    }
}

}  // namespace ck
#endif
