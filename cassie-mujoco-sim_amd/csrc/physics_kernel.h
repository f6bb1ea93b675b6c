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

namespace ck {

constexpr int NB = CM_MAXBODY;
constexpr int NG = CM_MAXGEOM;
constexpr int NROW = 64;       /* rows 0..62 constraints, column 63 = qfrc_smooth */
constexpr int MID_ROWS = CM_MAXEFC_NARROW;  /* 63: one constraint row per lane of one wavefront (+ the qfrc_smooth column in lane 63) */
constexpr int WIDE_ROWS = CM_MAXEFC;        /* 127: the solve spread over both wavefronts of an env (rows 64 .. 126 + the qfrc_smooth column on wave 1) */
constexpr int FAST_ROWS = 31;  /* rows of the row-capped fast instantiation (+ the qfrc_smooth column: half of the full tile) */
constexpr int FAST_ROWS_TRAY = 47; /* the 40-dof model's fast instantiation: Cassie + tray + cube at rest use 32 .. 40 rows (+ the qfrc_smooth row: three blocks of 16) */
constexpr int NSTAMP = 48;   /* 0..15 stage boundaries, 16..32 sub-stage stamps, 33..39 the two-wave form's barrier arrivals / departures,
                                40..41 where the hardware placed the env's wave(s), 42..43 shader clock and 100 MHz clock at the env's end,
                                44..46 the height-field pre-pass, 47 wave 1's sensor stage (tools/stage_profile.py names them) */
#define CK_TRI(k, i) ((k) * ((k) + 1) / 2 + (i))
#define CK_STAMP(i) do { if (io.prof && lane == 0) io.prof[(size_t)env * NSTAMP + (i)] = wv::clock(); CK_FRESH(); } while (0)
/* stage boundary: re-derive the lane index and its aliases (see wv::fresh_lane) */
#define CK_FRESH() do { lane = wv::fresh_lane(); b = lane; k_ = lane; isbody = b < nbody; isdof = k_ < nv; } while (0)

/* warning bits reported per env */
enum { WARN_CONTACT_FULL = 1, WARN_CONSTRAINT_FULL = 2, WARN_UNSUPPORTED_PAIR = 4, WARN_DIVERGED = 8,
       WARN_CHUNK_PLACEMENT = 16 /* a chunk of a launch found the chunk before it on another XCD (cassie_step_kernel): state possibly stale */ };

#ifndef WV_OCC
#define WV_OCC
#endif
typedef const WV_CONST_AS cm_model_t *ModelPtr;

struct PhysIO {
    const cm_model_t *models;   /* one shared model, or one per env */
    int model_stride;           /* 0 = shared, 1 = per-env */
    int nenv, nsub;             /* envs of this launch; nsub physics steps per launch (ctrl / PD targets held) */
    int env0;                   /* first env of this launch: a launch may cover the env range [env0, env0 + nenv) of the batch (all
                                   per-env arrays are indexed by the absolute env) */
    int integrate;              /* 1 = step (Euler), 0 = forward only (mj_forward role) */
    int sq, sqv, sv, su, ssd, sb; /* row strides in doubles: qpos, qvel, the other nv-sized fields, nu-sized fields, sensordata;
                                     sb = nbody.  qpos / qvel / sensordata have strides of their own so that the three can be
                                     columns of one caller-owned [nenv][nq + nv + nsensordata] observation block */
    double *qpos, *qvel, *qacc_warmstart, *time;
    double *ctrl;               /* read in torque / exact-PD mode; in a drive mode the kernel WRITES the torque its last substep
                                   applied (the delay line's output), so that a later forward pass -- mj_forward reads d->ctrl --
                                   sees the motor torques of the state it evaluates */
    const double *qfrc_applied, *xfrc_applied; /* may be null */
    double *qacc, *sensordata, *actuator_velocity;
    int *warn;                  /* [nenv] sticky warning bits */
    int *info;                  /* [nenv][4]: ncon, nefc, solver iterations, reserved (may be null) */
    double *xpos_out;           /* optional [nenv][nbody][3] (may be null) */
    double *xquat_out;          /* optional [nenv][nbody][4] (may be null) */
    double *body_cfrc;          /* optional [nenv][nbody][3] net contact force per body (world frame), last substep only */
    const float *hfield;        /* heightfield samples (may be null): one grid shared by all envs, or one per env */
    size_t hfield_stride;       /* floats between consecutive envs' grids (0 = shared) */
    /* optional on-device joint PD (all three null = torque mode): every substep
     * ctrl_u = motor-side torque of  kp (ptarget - q) - kd qdot  after the motor's
     * speed-torque limit -- the motor law of pd_input_step followed by
     * motor() (reference src/cassiemujoco.c:638-664) on the exact joint state */
    const double *pd_ptarget, *pd_kp, *pd_kd; /* [nenv][nu] each */
    /* optional drive-level I/O on the device (reference src/cassiemujoco.c:558-664, :737-803):
     * encoder quantisation + integer FIR / IIR velocity filters, motor speed-torque curve + STO + six-cycle torque
     * delay, every substep, bit for bit the host chain of csrc/cassie_hostpath.c.  drive_mode is a CM_DRIVE_* value */
    int drive_mode;
    cm_drive_state_t *drive_state;  /* [nenv] filter histories and delay lines */
    const double *drive_cmd;        /* CM_DRIVE_TORQUE: [nenv][nu + 1] commanded drive torques (cassie_in_t) and the STO flag */
    const double *pd_dtarget, *pd_torque; /* CM_DRIVE_PD: optional [nenv][nu] velocity targets and feed-forward torques */
    double *meas;                   /* [nenv][CM_MEAS_DIM] the cassie_out_t measurement fields of the step */
    cm_ext_t *ext;              /* optional [nenv] extended outputs (may be null) */
    long long *prof;            /* optional [nenv][NSTAMP] shader-clock stamps of the last substep (may be null) */
    /* load balancing across launches (may both be null): workgroup i steps env order[i], and every env reports the
     * shader clocks its launch took; the launcher sorts the next launch's order by that cost, most expensive first */
    const int *order;
    unsigned *cost;
    unsigned *cost_wall;        /* (may be null) the same span in ticks of the constant 100 MHz clock: cost / cost_wall = the shader clock under load */
    /* non-zero: every substep of a launch evaluates every output (IMU sensors, body quaternions) although only the last
     * substep's can be read -- a measurement aid (bench.py reports the rate with it as a side figure) */
    int all_outputs_every_substep;
    /* The row-capped fast instantiation (cassie_step_kernel<..., MAXR < CM_MAXEFC>) steps an env until a substep needs more
     * constraint rows than it holds; it then stores the state as of the start of that substep and records how many substeps
     * it completed in progress[env].  The full instantiation, launched behind it with resume != 0, finishes those envs from
     * there (and returns at once for the others).  progress may be null (then resume must be 0). */
    int *progress;
    int resume;
    /* The hand-over list (may be null: then the pass behind the fast kernel is one workgroup per env of the launch, each looking
     * up its env's record).  The fast instantiation appends every env it hands over to handover_list[env0 ...] (one atomic add on
     * *handover_count per env); the pass behind it is then a SMALL fixed grid whose workgroups walk the list -- entry blockIdx,
     * blockIdx + gridDim, ... -- so that a launch that handed nothing over costs a few workgroup placements, not one per env.
     * The last workgroup of the pass to finish (a ticket on handover_count[1]) zeroes the count for the next launch and reports
     * it to *handover_seen (host memory: the launcher sizes the next pass's grid by it). */
    /* A stepping launch in CHUNKS (nchunk > 1; row-capped fast instantiations only): workgroup w steps env slot w % nenv through
     * substeps [c (w / nenv), c (w / nenv + 1)), c = ceil(nsub / nchunk) -- an env's launch is nchunk jobs instead of one, so what
     * the slots wait for at the end of a launch (the last-started jobs running alone) is a quarter as long.  A chunk is a
     * launch of its own as far as the env is concerned: it loads the state the chunk before it stored and ends like a launch
     * of c substeps.  chunk_flag[env] = 64 chunk_seq + 8 (XCD of the chunk that wrote the word) + (chunks of this launch
     * complete); a chunk waits for the one before it (which has a lower workgroup number, so it was dispatched earlier) and checks
     * that it ran on the same XCD (wave.h: publish_global / wait_global); *chunk_fault (host memory, may be null) is set if not. */
    int nchunk, chunk_seq;
    int *chunk_flag;
    volatile int *chunk_fault;
    int *handover_list, *handover_count;
    volatile int *handover_seen;
    /* Three tiers since round 5: fast (31 / 47 rows) -> mid (63 rows, 16 contacts) -> wide (127 rows, 32 contacts; models whose
     * cm_model_t::maxefc allows it).  has_next: an instantiation with more rows runs behind this one -- a substep that needs more
     * rows or contacts than this one holds is handed over instead of being capped; handover_out_list / handover_out_count: where this
     * pass appends the envs it hands over (the list the pass behind it walks; same layout as handover_list / handover_count). */
    int has_next;
    int *handover_out_list, *handover_out_count;
};

/* MAXR: constraint rows this instantiation can hold (WIDE_ROWS, MID_ROWS, or fewer in the row-capped fast instantiations, see
 * cassie_step_kernel); the Y tile has one more row, the qfrc_smooth column */
template <int NVP>
struct BodyTiles { /* position / velocity stage tiles */
    double xpos[NB][3], xquat[NB][4], xmat[NB][9], xipos[NB][3];
    double xanchor[CM_MAXJNT][3], xaxis[CM_MAXJNT][3];
    double cinert[NB][10], crb[NB][10];
    double cvel[NB][6], cfrc[NB][6];
    double cdof_dot[NVP][6], buf[NVP][6];
    double geom_xpos[NG][3], geom_xmat[NG][9];
};
/* the body tiles and the staged matrix Y share their LDS (the tiles are dead once the Jacobian rows are formed) -- except in the
 * 127-row instantiation, whose rows 64 .. 126 are formed in a second pass while the staged rows of the first are already parked */
template <int NVP, int MAXR, bool SEPARATE> struct TilesAndY {
    static constexpr int YP = NVP + 2;
    union { BodyTiles<NVP> s; double Yr[MAXR + 1][YP]; };
};
template <int NVP, int MAXR> struct TilesAndY<NVP, MAXR, true> {
    static constexpr int YP = NVP + 2;
    BodyTiles<NVP> s;
    double Yr[MAXR + 1][YP];
    /* what the row stages hand to the solve per row (the rows of the second pass are wave 1's in the solve): regulariser R,
     * reference acceleration, J . qacc_warmstart, and 1.0 for rows that are clamped at zero / 0.0 for equality rows / -1.0 for no row */
    double rowt[MAXR + 1][4];
    /* the exchange between the two waves' halves of a Gauss-Seidel sweep: v[w] = sum over wave w's rows of (row of Y) x (its step),
     * the joint-space image of the steps -- the other wave's residuals take it in through their own rows of Y; sums[] = the waves'
     * parts of the warm start's cost and of a sweep's cost change, verdict words */
    double vx[2][NVP];
    double sums[8];
    int turn[4];
};
template <int NVP, int NL = NVP * (NVP + 1) / 2, int MAXR = MID_ROWS>
struct EnvShared {
    static constexpr int YP = NVP + 2; /* leading dimension of the Y staging tile: 16-byte aligned rows, conflict-free */
    static constexpr bool WIDE = MAXR > MID_ROWS;
    static constexpr int MAXC = WIDE ? CM_MAXCON : CM_MAXCON_NARROW; /* contacts the instantiation's list holds */
    /* x.s: the body-stage tiles; x.Yr: Y staged row-major by constraint row for broadcast reads, row MAXR = the qfrc_smooth column */
    TilesAndY<NVP, MAXR, WIDE> x;
    /* L^T D L factors of M and of M + hB, rows stored as LPack<TOPO, NVP> says (NL entries): a full lower triangle,
     * entry (k, i <= k) at k(k+1)/2 + i, or block-dense rows for a compile-time topology that asks for them */
    double Lp[NL], LHp[NL];
    double accel[2][28];            /* accelerometer partial results that must outlive the body tiles */
    double dinv[NVP], rsd[NVP], dinvH[NVP]; /* 1/D, 1/sqrt(D) of M; 1/D of M + hB */
    double cdof[NVP][6];
    double com[NB][3];              /* subtree com, valid at root bodies */
    double qpos[CM_MAXQ], qvel[NVP], qacc_ws[NVP], qacc[NVP], ctrl[CM_MAXU];
    double qfrc_smooth[NVP];
    double sens[CM_MAXSENSORDATA], actvel[CM_MAXU]; /* sensordata / actuator_velocity of the previous step (inputs of the drive-level models) */
    /* drive-level state of the env for the length of a launch (cm_drive_state_t in HBM between launches) and the drive
     * positions / velocities last measured (what CM_DRIVE_PD's law reads) */
    int drv_x[CM_NUM_DRIVES][CM_DRIVE_FILTER_NB];
    double drv_jx[CM_NUM_JOINTS][CM_JOINT_FILTER_NB], drv_jy[CM_NUM_JOINTS][CM_JOINT_FILTER_NA];
    double drv_delay[CM_NUM_DRIVES][CM_TORQUE_DELAY_CYCLES];
    double drv_pos[CM_NUM_DRIVES], drv_vel[CM_NUM_DRIVES];
    /* what a launch's drive-level passes read and no substep changes (drive_consts_load): gear ratio, torque limit, no-load
     * speed in rad/s, encoder counts and scale; the launch's command (torque + STO, or PD targets and gains) */
    double drv_c[CM_NUM_DRIVES][10], drv_jc[CM_NUM_JOINTS][2];
    /* contacts */
    double c_dist[MAXC], c_pos[MAXC][3], c_frame[MAXC][9], c_fri[MAXC][3];
    double c_solref[MAXC][2], c_solimp[MAXC][5], c_margin[MAXC];
    int c_dim[MAXC], c_g1[MAXC], c_g2[MAXC], c_pair[MAXC];
    int c_root[MAXC][2];               /* tree roots of the two bodies, their dof chains, summed inverse weights */
    unsigned long long c_dofmask[MAXC][2];
    double c_tran[MAXC];
    /* two-wave form (NW = 2): cmd[0] = what the waves tell each other at the workgroup barriers -- 0 = carry on, 1 = this env's
     * launch ends here (wave 0: diverged state, or the row-capped instantiation hands the substep over), 2 = wave 1 found a
     * diverged qacc; cmd[1] = the substep (+ 1) whose body forces wave 0's velocity stage has put in LDS, cmd[2] = the substep
     * (+ 1) whose staged matrix Y wave 0 has put in LDS (wave 1 waits for either); cmd[3] = the substep (+ 1) whose mass-matrix group
     * wave 1 has finished (com, cinert, cdof in LDS, the buf tile free again: wave 0's velocity stage waits for it), cmd[4] = the
     * substep (+ 1) whose collision verdict wave 0 has reached (cmd[0] = 1: handed over; wave 1's drive-level pass waits for it) */
    int cmd[6];
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
/* the same through the hardware reciprocal-square-root estimate and two Newton steps (kinematics of the
 * compile-time-topology kernels: no IEEE square root + division sequence on the stage's chain) */
WV_DEVICE void normalize4_fast(double *q) {
    const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (n2 < CM_MINVAL * CM_MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
    else {
        double y = wv::rsq_estimate(n2);
        y = fma(0.5 * y, fma(-n2 * y, y, 1.0), y);
        y = fma(0.5 * y, fma(-n2 * y, y, 1.0), y);
        q[0] *= y; q[1] *= y; q[2] *= y; q[3] *= y;
    }
}
/* normalize3 the same way; returns the norm */
WV_DEVICE double normalize3_fast(double *a) {
    const double n2 = dot3(a, a);
    if (n2 < CM_MINVAL * CM_MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; return sqrt(n2); }
    double y = wv::rsq_estimate(n2);
    y = fma(0.5 * y, fma(-n2 * y, y, 1.0), y);
    y = fma(0.5 * y, fma(-n2 * y, y, 1.0), y);
    a[0] *= y; a[1] *= y; a[2] *= y;
    const double n = n2 * y;
    return fma(0.5 * y, fma(-n, n, n2), n); /* one Newton step on the norm itself: n2 * y carries y's rounding */
}
/* sin and cos of a joint's half angle.  |x| < 2^19: three-part Cody-Waite reduction by pi/2 (the first two parts carry
 * 33 bits each, so k * part is exact for |k| < 2^20) and the classic degree-13 / degree-14 minimax polynomials on
 * [-pi/4, pi/4] (the coefficients of fdlibm's __kernel_sin / __kernel_cos, evaluated as two interleaved chains): about 1 ulp,
 * a third of the instructions of the library routine and no branch.  Beyond (a joint that has spun 80 000 turns) the
 * library's sincos, taken by the whole wave. */
WV_DEVICE void sincos_reduced(double x, double &sn, double &cs) { /* |x| < 2^19 */
    const double k = rint(x * 6.36619772367581382433e-01);
    double r = fma(-k, 1.57079632673412561417e+00, x);
    r = fma(-k, 6.07710050630396597660e-11, r);
    r = fma(-k, 2.02226624879595063154e-21, r);
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double s = fma(r * z, ps, r);
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double c = w + fma(z * z, pc, (1.0 - w) - hz);
    const int n = (int)k;
    const double a = (n & 1) ? c : s, bq = (n & 1) ? s : c;
    sn = (n & 2) ? -a : a;
    cs = ((n + 1) & 2) ? -bq : bq;
}
WV_DEVICE void sincos_bounded(double x, double &sn, double &cs) {
    if (wv::ballot(!(fabs(x) < 524288.0)) != 0ull) sincos(x, &sn, &cs);
    else sincos_reduced(x, sn, cs);
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


/* ---- boxes (same definitions as oracle/cassie_oracle.c) ---- */
WV_DEVICE double point_box(const double *q, const double *pb, const double *mb, const double *sb, double *nworld) {
    double d[3] = {q[0] - pb[0], q[1] - pb[1], q[2] - pb[2]}, loc[3], cl[3];
    mulmatTvec3(loc, mb, d);
    bool inside = true;
    for (int k = 0; k < 3; ++k) { cl[k] = clampd(loc[k], -sb[k], sb[k]); if (cl[k] != loc[k]) inside = false; }
    double nl[3] = {0, 0, 0}, dist;
    if (!inside) {
        double dif[3] = {loc[0] - cl[0], loc[1] - cl[1], loc[2] - cl[2]};
        dist = sqrt(dot3(dif, dif));
        for (int k = 0; k < 3; ++k) nl[k] = dif[k] / dist;
    } else {
        const double d0 = sb[0] - fabs(loc[0]), d1 = sb[1] - fabs(loc[1]), d2 = sb[2] - fabs(loc[2]);
        double best = d0;
        int kb = 0;
        if (d1 < best) { best = d1; kb = 1; }
        if (d2 < best) { best = d2; kb = 2; }
        const double sg = (kb == 0 ? loc[0] : (kb == 1 ? loc[1] : loc[2])) >= 0 ? 1.0 : -1.0;
        nl[0] = kb == 0 ? sg : 0.0; nl[1] = kb == 1 ? sg : 0.0; nl[2] = kb == 2 ? sg : 0.0;
        dist = -best;
    }
    mulmatvec3(nworld, mb, nl);
    return dist;
}
WV_DEVICE int sphere_box(RawContact &c, const double *ps, double r, const double *pb, const double *mb, const double *sb, double margin) {
    double nw[3];
    const double dist = point_box(ps, pb, mb, sb, nw) - r;
    if (dist > margin) return 0;
    c.dist = dist;
    for (int i = 0; i < 3; ++i) { c.normal[i] = -nw[i]; c.pos[i] = ps[i] - nw[i] * (r + 0.5 * dist); c.tangent[i] = 0; }
    return 1;
}
WV_DEVICE int capsule_box(RawContact &c0, RawContact &c1, const double *pc, const double *mc, double rad, double h, const double *pb,
                          const double *mb, const double *sb, double margin) {
    const double ax[3] = {mc[2], mc[5], mc[8]}, gr = 0.6180339887498949;
    double lo = -h, hi = h, nw[3];
    double t1 = hi - gr * (hi - lo), t2 = lo + gr * (hi - lo);
    double q1[3] = {pc[0] + ax[0] * t1, pc[1] + ax[1] * t1, pc[2] + ax[2] * t1}, q2[3] = {pc[0] + ax[0] * t2, pc[1] + ax[1] * t2, pc[2] + ax[2] * t2};
    double f1 = point_box(q1, pb, mb, sb, nw), f2 = point_box(q2, pb, mb, sb, nw);
    for (int it = 0; it < 32; ++it) {
        if (f1 <= f2) { hi = t2; t2 = t1; f2 = f1; t1 = hi - gr * (hi - lo); for (int i = 0; i < 3; ++i) q1[i] = pc[i] + ax[i] * t1; f1 = point_box(q1, pb, mb, sb, nw); }
        else { lo = t1; t1 = t2; f1 = f2; t2 = lo + gr * (hi - lo); for (int i = 0; i < 3; ++i) q2[i] = pc[i] + ax[i] * t2; f2 = point_box(q2, pb, mb, sb, nw); }
    }
    double ts = 0.5 * (lo + hi);
    if (ts > h - 1e-9 * (1 + h)) ts = h;
    if (ts < -h + 1e-9 * (1 + h)) ts = -h;
    const double tf = ts >= 0 ? -h : h;
    int n = 0;
    double qa[3] = {pc[0] + ax[0] * ts, pc[1] + ax[1] * ts, pc[2] + ax[2] * ts};
    if (sphere_box(c0, qa, rad, pb, mb, sb, margin)) { for (int i = 0; i < 3; ++i) c0.tangent[i] = ax[i]; n = 1; }
    if (!(fabs(tf - ts) < 1e-6 + 1e-3 * h)) {
        double qb[3] = {pc[0] + ax[0] * tf, pc[1] + ax[1] * tf, pc[2] + ax[2] * tf};
        RawContact t;
        if (sphere_box(t, qb, rad, pb, mb, sb, margin)) { for (int i = 0; i < 3; ++i) t.tangent[i] = ax[i]; if (n == 0) c0 = t; else c1 = t; ++n; }
    }
    return n;
}


/* ---- box vs box (same definition, same arithmetic order and the same tie rules as oracle/cassie_oracle.c box_box):
 * separating-axis test over 15 axes, then either the incident face clipped against the reference face -- lane =
 * candidate vertex of the clipped polygon: 0-3 incident vertices, 4-7 rectangle corners, 8-23 edge crossings; at most
 * the 4 deepest are kept, in candidate order -- or one edge-edge contact (lane 0).  Everything up to the candidates is
 * wave-uniform and computed redundantly by every lane.  Returns whether this lane holds a contact. ---- */
/* run-time picks out of three / four register values by compare-and-select: an array indexed by a run-time value would be
 * placed in scratch memory (box_box_lane used to keep ~200 bytes of such arrays there: 13 scratch stores and 15 loads per
 * box pair, 1.9 GB of HBM writes per 4096-env launch of the tray model) */
WV_DEVICE double sel3(int i, double a0, double a1, double a2) { return i == 0 ? a0 : (i == 1 ? a1 : a2); }
WV_DEVICE double sel4(int i, double a0, double a1, double a2, double a3) { return i == 0 ? a0 : (i == 1 ? a1 : (i == 2 ? a2 : a3)); }
WV_DEVICE void row3(double (&r)[3], const double (&M)[3][3], int i) {
    for (int x = 0; x < 3; ++x) r[x] = sel3(i, M[0][x], M[1][x], M[2][x]);
}

WV_DEVICE bool box_box_lane(RawContact &rc, int lane, const double *p1, const double *m1, const double *s1, const double *p2, const double *m2,
                            const double *s2, double margin) {
    const double BB_TIE = 1e-10;
    double A[3][3], B[3][3], d[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, ta[3], tb[3], C[3][3], Q[3][3];
    const double sa[3] = {s1[0], s1[1], s1[2]}, sb[3] = {s2[0], s2[1], s2[2]};
    for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) { A[i][k] = m1[3 * k + i]; B[i][k] = m2[3 * k + i]; }
    for (int i = 0; i < 3; ++i) { ta[i] = dot3(d, A[i]); tb[i] = dot3(d, B[i]); }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { C[i][j] = dot3(A[i], B[j]); Q[i][j] = fabs(C[i][j]); }
    int best = -1;
    double bestscore = 0;
    bool separated = false;
#pragma unroll
    for (int k = 0; k < 15; ++k) { /* fully unrolled: every index below is a compile-time constant */
        double sep, sc;
        if (k < 3) {
            sep = fabs(ta[k]) - (sa[k] + (sb[0] * Q[k][0] + sb[1] * Q[k][1] + sb[2] * Q[k][2]));
            sc = sep;
        } else if (k < 6) {
            const int j = k - 3;
            sep = fabs(tb[j]) - (sb[j] + (sa[0] * Q[0][j] + sa[1] * Q[1][j] + sa[2] * Q[2][j]));
            sc = sep - BB_TIE;
        } else {
            const int i = (k - 6) / 3, j = (k - 6) % 3, i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            const double len2 = 1.0 - C[i][j] * C[i][j];
            if (len2 < 1e-6) continue;
            const double proj = ta[i2] * C[i1][j] - ta[i1] * C[i2][j];
            const double ra = sa[i1] * Q[i2][j] + sa[i2] * Q[i1][j], rb = sb[j1] * Q[i][j2] + sb[j2] * Q[i][j1];
            sep = (fabs(proj) - (ra + rb)) / sqrt(len2);
            sc = (sep < 0 ? 1.05 * sep : sep) - 2 * BB_TIE;
        }
        if (sep > margin) separated = true;
        if (best < 0 || sc > bestscore) { best = k; bestscore = sc; }
    }
    if (separated) return false;

    if (best >= 6) {
        const int i = (best - 6) / 3, j = (best - 6) % 3;
        double Ai[3], Bj[3];
        row3(Ai, A, i); row3(Bj, B, j);
        double n[3], pa[3] = {p1[0], p1[1], p1[2]}, pb[3] = {p2[0], p2[1], p2[2]};
        cross3(n, Ai, Bj);
        normalize3(n);
        if (dot3(n, d) < 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k != i) { const double sg = dot3(n, A[k]) > 0 ? 1.0 : -1.0; for (int x = 0; x < 3; ++x) pa[x] += sg * sa[k] * A[k][x]; }
            if (k != j) { const double sg = dot3(n, B[k]) > 0 ? -1.0 : 1.0; for (int x = 0; x < 3; ++x) pb[x] += sg * sb[k] * B[k][x]; }
        }
        const double ab[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
        double Ci[3];
        row3(Ci, C, i);
        const double uaub = sel3(j, Ci[0], Ci[1], Ci[2]), q1 = dot3(Ai, ab), q2 = -dot3(Bj, ab), den = 1.0 - uaub * uaub;
        const double s1i = sel3(i, sa[0], sa[1], sa[2]), s2j = sel3(j, sb[0], sb[1], sb[2]);
        const double al = clampd((q1 + uaub * q2) / den, -s1i, s1i), be = clampd((uaub * q1 + q2) / den, -s2j, s2j);
        double ca[3], cb[3];
        for (int x = 0; x < 3; ++x) { ca[x] = pa[x] + al * Ai[x]; cb[x] = pb[x] + be * Bj[x]; }
        const double cd[3] = {cb[0] - ca[0], cb[1] - ca[1], cb[2] - ca[2]};
        rc.dist = dot3(cd, n);
        for (int x = 0; x < 3; ++x) { rc.normal[x] = n[x]; rc.tangent[x] = 0; rc.pos[x] = 0.5 * (ca[x] + cb[x]); }
        return lane == 0 && !(rc.dist > margin);
    }

    const bool refA = best < 3;
    const int a = best % 3, a1 = (a + 1) % 3, a2 = (a + 2) % 3;
    double R[3][3], I[3][3];
    for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) { R[i][k] = refA ? A[i][k] : B[i][k]; I[i][k] = refA ? B[i][k] : A[i][k]; }
    double pr[3], pi[3], sr[3], si[3];
    for (int x = 0; x < 3; ++x) { pr[x] = refA ? p1[x] : p2[x]; pi[x] = refA ? p2[x] : p1[x]; sr[x] = refA ? sa[x] : sb[x]; si[x] = refA ? sb[x] : sa[x]; }
    /* the reference face's normal axis and its two in-plane axes, picked once (Ra, Ra1, Ra2) */
    double Ra[3], Ra1[3], Ra2[3];
    row3(Ra, R, a); row3(Ra1, R, a1); row3(Ra2, R, a2);
    const double sra = sel3(a, sr[0], sr[1], sr[2]);
    const double dri[3] = {pi[0] - pr[0], pi[1] - pr[1], pi[2] - pr[2]};
    const double sgn = dot3(dri, Ra) >= 0 ? 1.0 : -1.0;
    double n[3] = {sgn * Ra[0], sgn * Ra[1], sgn * Ra[2]};
    int kf = 0;
    double kbest = fabs(dot3(n, I[0]));
#pragma unroll
    for (int k = 1; k < 3; ++k) { const double v = fabs(dot3(n, I[k])); if (v > kbest + 1e-9) { kbest = v; kf = k; } }
    const int k1 = (kf + 1) % 3, k2 = (kf + 2) % 3;
    double If[3], I1[3], I2[3];
    row3(If, I, kf); row3(I1, I, k1); row3(I2, I, k2);
    const double sif = sel3(kf, si[0], si[1], si[2]), si1 = sel3(k1, si[0], si[1], si[2]), si2 = sel3(k2, si[0], si[1], si[2]);
    const double isg = dot3(n, If) > 0 ? -1.0 : 1.0;
    double cr[3], ci[3];
    for (int x = 0; x < 3; ++x) { cr[x] = pr[x] + sgn * sra * Ra[x]; ci[x] = pi[x] + isg * sif * If[x]; }
    const double h1 = sel3(a1, sr[0], sr[1], sr[2]), h2 = sel3(a2, sr[0], sr[1], sr[2]);
    const double su[4] = {1, -1, -1, 1}, sv[4] = {1, 1, -1, -1};
    double pu[4], pv[4], pw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double x[3];
        for (int t = 0; t < 3; ++t) x[t] = ci[t] + su[q] * si1 * I1[t] + sv[q] * si2 * I2[t] - cr[t];
        pu[q] = dot3(x, Ra1); pv[q] = dot3(x, Ra2); pw[q] = dot3(x, n);
    }
    const double tol = 1e-12;
    /* lane = candidate */
    double cu = 0, cv = 0, cw = 0;
    bool valid = false;
    if (lane < 4) {
        const int q = lane;
        cu = sel4(q, pu[0], pu[1], pu[2], pu[3]); cv = sel4(q, pv[0], pv[1], pv[2], pv[3]); cw = sel4(q, pw[0], pw[1], pw[2], pw[3]);
        valid = fabs(cu) <= h1 + tol && fabs(cv) <= h2 + tol;
    } else if (lane < 8) {
        const int q = lane - 4;
        const double e1u = pu[1] - pu[0], e1v = pv[1] - pv[0], e1w = pw[1] - pw[0], e2u = pu[3] - pu[0], e2v = pv[3] - pv[0], e2w = pw[3] - pw[0];
        const double det = e1u * e2v - e1v * e2u;
        const double gu = (e1w * e2v - e1v * e2w) / det, gv = (e1u * e2w - e1w * e2u) / det;
        const double u = sel4(q, 1.0, -1.0, -1.0, 1.0) * h1, v = sel4(q, 1.0, 1.0, -1.0, -1.0) * h2;
        int pos = 0, neg = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int f = (e + 1) & 3;
            const double cr2 = (pu[f] - pu[e]) * (v - pv[e]) - (pv[f] - pv[e]) * (u - pu[e]);
            if (cr2 > tol) ++pos; else if (cr2 < -tol) ++neg;
        }
        cu = u; cv = v; cw = pw[0] + gu * (u - pu[0]) + gv * (v - pv[0]);
        valid = !(pos && neg);
    } else if (lane < 24) {
        const int e = (lane - 8) >> 2, l = (lane - 8) & 3, f = (e + 1) & 3;
        const bool along_u = l < 2;
        const double pue = sel4(e, pu[0], pu[1], pu[2], pu[3]), puf = sel4(f, pu[0], pu[1], pu[2], pu[3]);
        const double pve = sel4(e, pv[0], pv[1], pv[2], pv[3]), pvf = sel4(f, pv[0], pv[1], pv[2], pv[3]);
        const double pwe = sel4(e, pw[0], pw[1], pw[2], pw[3]), pwf = sel4(f, pw[0], pw[1], pw[2], pw[3]);
        const double lim = (l & 1) ? -(along_u ? h1 : h2) : (along_u ? h1 : h2);
        const double x0 = along_u ? pue : pve, x1 = along_u ? puf : pvf;
        const double y0 = along_u ? pve : pue, y1 = along_u ? pvf : puf, hy = along_u ? h2 : h1;
        const double dx = x1 - x0;
        if (!(fabs(dx) < 1e-14)) {
            const double sp = (lim - x0) / dx;
            if (sp > 0 && sp < 1) {
                const double y = y0 + sp * (y1 - y0);
                if (!(fabs(y) > hy)) {
                    cu = along_u ? lim : y; cv = along_u ? y : lim; cw = pwe + sp * (pwf - pwe);
                    valid = true;
                }
            }
        }
    }
    if (valid && cw > margin) valid = false;
    const unsigned long long vmask = wv::ballot(valid);
    bool keep = valid;
    if (wv::popc64(vmask) > 4) {
        int rank = 0;
        for (int r = 0; r < 24; ++r) {
            const double wr = wv::readlane(cw, r);
            if (r == lane || !((vmask >> r) & 1ull)) continue;
            if (wr < cw - 1e-9 || (fabs(wr - cw) <= 1e-9 && r < lane)) ++rank;
        }
        if (rank >= 4) keep = false;
    }
    if (keep) {
        rc.dist = cw;
        for (int x = 0; x < 3; ++x) {
            const double px = cr[x] + cu * Ra1[x] + cv * Ra2[x] + cw * n[x];
            rc.pos[x] = px - 0.5 * cw * n[x];
            rc.normal[x] = refA ? n[x] : -n[x];
            rc.tangent[x] = 0;
        }
    }
    return keep;
}

/* ---- height field: closest feature of the terrain surface over every grid triangle under the sample sphere's footprint
 *      (same definition, same candidate order as oracle/cassie_oracle.c) ---- */
WV_DEVICE void closest_on_triangle(const double *p, const double *a, const double *b, const double *c, double *q) {
    double ab[3], ac[3], ap[3], bp[3], cp[3];
    for (int i = 0; i < 3; ++i) { ab[i] = b[i] - a[i]; ac[i] = c[i] - a[i]; ap[i] = p[i] - a[i]; bp[i] = p[i] - b[i]; cp[i] = p[i] - c[i]; }
    const double d1 = dot3(ab, ap), d2 = dot3(ac, ap), d3 = dot3(ab, bp), d4 = dot3(ac, bp), d5 = dot3(ab, cp), d6 = dot3(ac, cp);
    double v = 0, w = 0;
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
WV_DEVICE void hfield_triangle(const double *p, const double *a, const double *b, const double *c, double &best, double *bestn) {
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
        if (len < best) {
            best = len;
            if (len > 1e-12) { bestn[0] = d[0] / len; bestn[1] = d[1] / len; bestn[2] = d[2] / len; }
            else { bestn[0] = n[0]; bestn[1] = n[1]; bestn[2] = n[2]; }
        }
    } else {
        const double e0 = (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0]);
        const double e1 = (c[0] - b[0]) * (p[1] - b[1]) - (c[1] - b[1]) * (p[0] - b[0]);
        const double e2 = (a[0] - c[0]) * (p[1] - c[1]) - (a[1] - c[1]) * (p[0] - c[0]);
        const bool inside = (e0 >= 0 && e1 >= 0 && e2 >= 0) || (e0 <= 0 && e1 <= 0 && e2 <= 0);
        if (inside && s < best) { best = s; bestn[0] = n[0]; bestn[1] = n[1]; bestn[2] = n[2]; }
    }
}
WV_DEVICE int hfield_sphere(RawContact &c, ModelPtr m, const float *data, const double *ph, const double *mh, const double *ps,
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
    /* touch the first and the last sample of every grid row of the footprint before any of them is needed: all the
     * footprint's cache lines are then in flight together (one memory latency instead of one per cell) */
    float touch = 0.0f;
    for (int i = i0; i <= i1 + 1; ++i) touch += data[i * nc + j0] + data[i * nc + j1 + 1];
    if (touch == -1.2345e30f) best = 0;      /* never true for elevations in [0, 1]: keeps the loads alive */
    const double reach2 = reach * reach;
    for (int i = i0; i <= i1; ++i) {
        const double y0 = -sy + i * dy;
        const double ey = p[1] < y0 ? y0 - p[1] : (p[1] > y0 + dy ? p[1] - (y0 + dy) : 0.0);
        for (int j = j0; j <= j1; ++j) {
            const double x0 = -sx + j * dx;
            /* exact culls: a cell whose rectangle is further than the reach in plan, or whose highest corner is more than
             * the reach below the sphere, cannot hold a point within contact distance */
            const double ex = p[0] < x0 ? x0 - p[0] : (p[0] > x0 + dx ? p[0] - (x0 + dx) : 0.0);
            if (ex * ex + ey * ey > reach2) continue;
            const double z00 = sz * data[i * nc + j], z10 = sz * data[i * nc + j + 1], z01 = sz * data[(i + 1) * nc + j], z11 = sz * data[(i + 1) * nc + j + 1];
            if (p[2] - reach > fmax(fmax(z00, z10), fmax(z01, z11))) continue;
            const double v00[3] = {x0, y0, z00}, v10[3] = {x0 + dx, y0, z10}, v01[3] = {x0, y0 + dy, z01}, v11[3] = {x0 + dx, y0 + dy, z11};
            hfield_triangle(p, v00, v10, v01, best, bn);
            hfield_triangle(p, v11, v01, v10, best, bn);
        }
    }
    if (best > 1e299) return 0;
    const double dist = best - r;
    if (dist > margin) return 0;
    double nw[3];
    mulmatvec3(nw, mh, bn);
    c.dist = dist;
    for (int k = 0; k < 3; ++k) { c.normal[k] = nw[k]; c.pos[k] = ps[k] - nw[k] * (r + 0.5 * dist); c.tangent[k] = 0; }
    return 1;
}

/* The same for up to 64 sample spheres at once, one per lane (`mine`: this lane has one), with the WHOLE WAVE sharing the grid
 * cells under all of them: a sphere of a foot capsule covers 15 .. 25 cells and the pelvis sphere none, and with one lane
 * walking each sphere's cells the wave took as long as its slowest lane (26 k clocks, a quarter of the height-field model's
 * substep).  Here the cells of all spheres form one task list (sphere by sphere, a sphere's cells in its scan order), lane t
 * of round q takes task 64 q + t -- looks its sphere up in a table the spheres wrote their lane numbers into, fetches the
 * sphere from that lane, tests the cell's two triangles -- and a segmented minimum scan over the lanes of one sphere hands the round's
 * closest feature to the sphere's record in LDS.  Ties go to the earlier task, and a sphere's earlier rounds win over later
 * ones, which is the strict `<` of the sequential scan: results are those of hfield_sphere bit for bit.
 * work: HF_WINDOW bytes (the sphere of every task of a window) + 4 doubles per lane (closest distance, normal). */
#ifdef CK_EMULATED
constexpr int HF_WINDOW = 128;  /* (the CPU emulator's tests go through several windows per pass; results do not depend on the size) */
#else
constexpr int HF_WINDOW = 1024;
#endif
WV_DEVICE int hfield_spheres_wave(RawContact &c, ModelPtr m, const float *data, const double *ph, const double *mh, bool mine, const double *ps,
                                  double r, double margin, int lane, double *work) {
    const bool grid_ok = data && m->hfield_nrow >= 2 && m->hfield_ncol >= 2;
    const double sx = m->hfield_size[0], sy = m->hfield_size[1], sz = m->hfield_size[2];
    const int nc = m->hfield_ncol, nr = m->hfield_nrow;
    const double dx = 2 * sx / (nc > 1 ? nc - 1 : 1), dy = 2 * sy / (nr > 1 ? nr - 1 : 1);
    double pl[3] = {0, 0, 0};
    const double reach = r + (margin > 0 ? margin : 0);
    int i0 = 0, j0 = 0, wj = 1, ncell = 0;
    if (mine && grid_ok) {
        double d[3] = {ps[0] - ph[0], ps[1] - ph[1], ps[2] - ph[2]};
        mulmatTvec3(pl, mh, d);
        if (!(fabs(pl[0]) > sx + reach || fabs(pl[1]) > sy + reach || pl[2] - r > sz + margin)) {
            int j1 = (int)floor((pl[0] + reach + sx) / dx), i1 = (int)floor((pl[1] + reach + sy) / dy);
            j0 = (int)floor((pl[0] - reach + sx) / dx); i0 = (int)floor((pl[1] - reach + sy) / dy);
            if (j0 < 0) j0 = 0;
            if (i0 < 0) i0 = 0;
            if (j1 > nc - 2) j1 = nc - 2;
            if (i1 > nr - 2) i1 = nr - 2;
            if (j1 >= j0 && i1 >= i0) { wj = j1 - j0 + 1; ncell = wj * (i1 - i0 + 1); }
        }
    }
    /* running cell counts (inclusive), lane by lane */
    int endx = ncell;
#pragma unroll
    for (int dlt = 1; dlt < WV_WAVE; dlt *= 2) { const int t = wv::shfl_i(endx, (lane - dlt) & 63); if (lane >= dlt) endx += t; }
    const int total = wv::shfl_i(endx, WV_WAVE - 1);
    int maxcell = ncell;
#pragma unroll
    for (int msk = WV_WAVE / 2; msk >= 1; msk /= 2) { const int o = wv::shfl_i(maxcell, lane ^ msk); maxcell = o > maxcell ? o : maxcell; }
    unsigned char *const owner = (unsigned char *)work;      /* the sphere (lane) of every task of a window of HF_WINDOW tasks */
    double *const rec = work + HF_WINDOW / 8 + 4 * lane;
    rec[0] = 1e300; rec[1] = 0; rec[2] = 0; rec[3] = 1;
    const int start = endx - ncell;
    for (int win = 0; win < total; win += HF_WINDOW) {
        /* every sphere writes its lane over its tasks of this window */
        for (int cc = 0; cc < maxcell; ++cc) {
            const int at = start + cc - win;
            if (cc < ncell && at >= 0 && at < HF_WINDOW) owner[at] = (unsigned char)lane;
        }
        wv::sync();
        const int wend = total - win < HF_WINDOW ? total - win : HF_WINDOW;
        /* a round's tasks: sphere, cell, the cell's four heights -- requested one round ahead of their use, so that the trip to
         * memory runs under the previous round's triangles */
        struct Task { bool act; int own; double q0, q1, q2, qreach, x0, y0; float h00, h10, h01, h11; bool cull; };
        auto request = [&](int base, Task &t) {
            const int task = base + lane;
            t.act = task < wend;
            t.own = t.act ? (int)owner[task] : lane;
            t.q0 = wv::shfl(pl[0], t.own); t.q1 = wv::shfl(pl[1], t.own); t.q2 = wv::shfl(pl[2], t.own); t.qreach = wv::shfl(reach, t.own);
            const int qi0 = wv::shfl_i(i0, t.own), qj0 = wv::shfl_i(j0, t.own), qwj = wv::shfl_i(wj, t.own), qstart = wv::shfl_i(start, t.own);
            const int cidx = t.act ? win + task - qstart : 0;
            const int ci = (int)(((float)cidx + 0.5f) * (1.0f / (float)qwj)); /* cidx / qwj: the quotient's distance from an integer is at least 0.5 / qwj */
            const int i = qi0 + ci, j = qj0 + (cidx - ci * qwj);
            t.y0 = -sy + i * dy; t.x0 = -sx + j * dx;
            const double ey = t.q1 < t.y0 ? t.y0 - t.q1 : (t.q1 > t.y0 + dy ? t.q1 - (t.y0 + dy) : 0.0);
            const double ex = t.q0 < t.x0 ? t.x0 - t.q0 : (t.q0 > t.x0 + dx ? t.q0 - (t.x0 + dx) : 0.0);
            t.cull = !t.act || ex * ex + ey * ey > t.qreach * t.qreach;
            t.h00 = t.h10 = t.h01 = t.h11 = 0.0f;
            if (!t.cull) { t.h00 = data[i * nc + j]; t.h10 = data[i * nc + j + 1]; t.h01 = data[(i + 1) * nc + j]; t.h11 = data[(i + 1) * nc + j + 1]; }
        };
        Task cur, nxt;
        request(0, cur);
        for (int base = 0; base < wend; base += WV_WAVE) {
            if (base + WV_WAVE < wend) request(base + WV_WAVE, nxt); /* (wave-uniform) */
            else { nxt.act = false; nxt.cull = true; nxt.own = lane; }
            double best = 1e300, bn[3] = {0, 0, 1};
            if (!cur.cull) {
                const double z00 = sz * cur.h00, z10 = sz * cur.h10, z01 = sz * cur.h01, z11 = sz * cur.h11;
                if (!(cur.q2 - cur.qreach > fmax(fmax(z00, z10), fmax(z01, z11)))) {
                    const double q[3] = {cur.q0, cur.q1, cur.q2}, x0 = cur.x0, y0 = cur.y0;
                    const double v00[3] = {x0, y0, z00}, v10[3] = {x0 + dx, y0, z10}, v01[3] = {x0, y0 + dy, z01}, v11[3] = {x0 + dx, y0 + dy, z11};
                    hfield_triangle(q, v00, v10, v01, best, bn);
                    hfield_triangle(q, v11, v01, v10, best, bn);
                }
            }
            /* the closest feature among the lanes of one sphere, earlier tasks first: segmented inclusive scan of (distance, lane) */
            double sv = best;
            int ssrc = lane;
            const int seg = cur.act ? cur.own : -1 - lane;
#pragma unroll
            for (int dlt = 1; dlt < WV_WAVE; dlt *= 2) {
                const int from = (lane - dlt) & 63;
                const double ov = wv::shfl(sv, from);
                const int osrc = wv::shfl_i(ssrc, from), oseg = wv::shfl_i(seg, from);
                if (lane >= dlt && oseg == seg && !(sv < ov)) { sv = ov; ssrc = osrc; }
            }
            const int nseg = wv::shfl_i(seg, (lane + 1) & 63);
            const double w0 = wv::shfl(bn[0], ssrc), w1 = wv::shfl(bn[1], ssrc), w2 = wv::shfl(bn[2], ssrc);
            if (cur.act && (lane == WV_WAVE - 1 || nseg != seg)) {
                double *const o = work + HF_WINDOW / 8 + 4 * cur.own;
                if (sv < o[0]) { o[0] = sv; o[1] = w0; o[2] = w1; o[3] = w2; }
            }
            wv::sync();
            cur = nxt;
        }
    }
    wv::sync();
    const double best = rec[0];
    if (!mine || best > 1e299) return 0;
    const double dist = best - r;
    if (dist > margin) return 0;
    const double bn[3] = {rec[1], rec[2], rec[3]};
    double nw[3];
    mulmatvec3(nw, mh, bn);
    c.dist = dist;
    for (int k = 0; k < 3; ++k) { c.normal[k] = nw[k]; c.pos[k] = ps[k] - nw[k] * (r + 0.5 * dist); c.tangent[k] = 0; }
    return 1;
}

/* parks one detected contact (geometry only) in the contact list; finish_contacts completes the entries */
template <class SH>
WV_DEVICE void write_raw_contact(SH &S, int slot, int pair, const RawContact &r) {
    S.c_dist[slot] = r.dist;
    S.c_pair[slot] = pair;
    for (int i = 0; i < 3; ++i) { S.c_pos[slot][i] = r.pos[i]; S.c_frame[slot][i] = r.normal[i]; S.c_frame[slot][3 + i] = r.tangent[i]; }
}

/* CM_FLAG_HFPRISM (same definition, same candidate order as oracle/cassie_oracle.c hfield_prism_contacts): ONE CONTACT PER
 * PENETRATED GRID TRIANGLE under every sphere / capsule that has a height-field pair.  Lane = pair first (its capsule in the
 * height field's frame, the cells under its bounding rectangle, its sample count -> a record in `hp`), then lane = (pair, cell,
 * triangle) KEY in the oracle's order -- pair order, cells row-major, the triangle (v00, v10, v01) of a cell before (v11, v01, v10) --
 * 64 keys to a round: a key's lane walks the capsule's sample spheres (no further apart than the radius; exact culls by plan
 * distance and by the cell's highest corner skip most), keeps the deepest, and a key whose deepest sample is within the margin is a
 * contact.  Ballots put the contacts into the list in key order, which is the oracle's.  Returns the number of contacts FOUND;
 * those past the list's `room` slots are not written (the caller caps or hands the substep over).
 * hp: scratch, HP_REC doubles per height-field pair of the model (the idle velocity tiles). */
constexpr int HP_REC = 16;
template <class SH>
WV_DEVICE int hfield_prism_wave(SH &S, ModelPtr m, const float *data, int lane, double *hp, int room) {
    const int nhf = m->nhfpair, nc = m->hfield_ncol, nr = m->hfield_nrow;
    if (!data || nr < 2 || nc < 2) return 0;
    const double sx = m->hfield_size[0], sy = m->hfield_size[1], sz = m->hfield_size[2];
    const double dx = 2 * sx / (nc - 1), dy = 2 * sy / (nr - 1);
    /* ---- lane = pair ---- */
    int nkeys = 0;
    if (lane < nhf) {
        const int p = m->hfpair[lane];
        const int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p], t2 = m->pair_type[p] >> 8;
        const double margin = m->pair_margin[p], r = m->pair_size[p][3], h = t2 == CM_GEOM_CAPSULE ? m->pair_size[p][4] : 0.0;
        const double *ph = S.x.s.geom_xpos[g1], *mh = S.x.s.geom_xmat[g1], *pc = S.x.s.geom_xpos[g2], *mc = S.x.s.geom_xmat[g2];
        const double axw[3] = {mc[2], mc[5], mc[8]}, d[3] = {pc[0] - ph[0], pc[1] - ph[1], pc[2] - ph[2]};
        double p0[3], ax[3];
        mulmatTvec3(p0, mh, d);
        mulmatTvec3(ax, mh, axw);
        const double reach = r + (margin > 0 ? margin : 0);
        int i0 = 0, j0 = 0, wj = 1, ncell = 0;
        if (!(p0[2] - h * fabs(ax[2]) - r > sz + margin)) {
            const double xa = p0[0] - h * fabs(ax[0]) - reach, xb = p0[0] + h * fabs(ax[0]) + reach;
            const double ya = p0[1] - h * fabs(ax[1]) - reach, yb = p0[1] + h * fabs(ax[1]) + reach;
            int j1 = (int)floor((xb + sx) / dx), i1 = (int)floor((yb + sy) / dy);
            j0 = (int)floor((xa + sx) / dx); i0 = (int)floor((ya + sy) / dy);
            if (j0 < 0) j0 = 0;
            if (i0 < 0) i0 = 0;
            if (j1 > nc - 2) j1 = nc - 2;
            if (i1 > nr - 2) i1 = nr - 2;
            if (j1 >= j0 && i1 >= i0) { wj = j1 - j0 + 1; ncell = wj * (i1 - i0 + 1); }
        }
        int ns = h > 0 ? 1 + (int)ceil(2 * h / r) : 1;
        if (ns > CM_HP_MAXS) ns = CM_HP_MAXS;
        double *rec = hp + HP_REC * lane;
        for (int i = 0; i < 3; ++i) { rec[i] = p0[i]; rec[3 + i] = ax[i]; }
        rec[6] = r; rec[7] = h; rec[8] = margin; rec[9] = (double)ns; rec[10] = (double)i0; rec[11] = (double)j0; rec[12] = (double)wj;
        rec[13] = (double)p; rec[14] = (double)g2;
        nkeys = 2 * ncell;
    }
    int endx = nkeys;
#pragma unroll
    for (int dlt = 1; dlt < WV_WAVE; dlt *= 2) { const int t = wv::shfl_i(endx, (lane - dlt) & 63); if (lane >= dlt) endx += t; }
    const int total = wv::shfl_i(endx, WV_WAVE - 1);
    const int start = endx - nkeys;
    wv::sync();
    /* ---- lane = key ---- */
    int ncon = 0;
    const int g1h = m->hfield_geom;
    for (int base = 0; base < total; base += WV_WAVE) {
        const int t = base + lane;
        const bool act = t < total;
        int h = 0, hstart = 0;
        for (int hh = 0; hh < nhf; ++hh) {
            const int st = wv::shfl_i(start, hh), en = wv::shfl_i(endx, hh);
            if (t >= st && t < en) { h = hh; hstart = st; }
        }
        const double *rec = hp + HP_REC * h;
        const double p0[3] = {rec[0], rec[1], rec[2]}, ax[3] = {rec[3], rec[4], rec[5]}, r = rec[6], hl = rec[7], margin = rec[8];
        const int ns = act ? (int)rec[9] : 0, i0 = (int)rec[10], j0 = (int)rec[11], wj = (int)rec[12], pidx = (int)rec[13], g2 = (int)rec[14];
        const int q = act ? t - hstart : 0, cell = q >> 1, tri = q & 1;
        const int ci = (int)(((float)cell + 0.5f) * (1.0f / (float)wj)); /* cell / wj: the quotient's distance from an integer is at least 0.5 / wj */
        const int i = i0 + ci, j = j0 + (cell - ci * wj);
        const double x0 = -sx + j * dx, y0 = -sy + i * dy, reach = r + (margin > 0 ? margin : 0);
        double z00 = 0, z10 = 0, z01 = 0, z11 = 0;
        if (act) { z00 = sz * data[i * nc + j]; z10 = sz * data[i * nc + j + 1]; z01 = sz * data[(i + 1) * nc + j]; z11 = sz * data[(i + 1) * nc + j + 1]; }
        const double zmax = fmax(fmax(z00, z10), fmax(z01, z11));
        const double v00[3] = {x0, y0, z00}, v10[3] = {x0 + dx, y0, z10}, v01[3] = {x0, y0 + dy, z01}, v11[3] = {x0 + dx, y0 + dy, z11};
        double best = 1e300, bn[3] = {0, 0, 1}, bt = 0;
        for (int k = 0; k < CM_HP_MAXS; ++k) {
            if (wv::ballot(k < ns) == 0ull) break; /* (wave-uniform: no lane has a k-th sample) */
            if (k >= ns) continue;
            const double tk = ns > 1 ? hl * (1.0 - 2.0 * k / (ns - 1)) : 0.0;
            const double p[3] = {p0[0] + tk * ax[0], p0[1] + tk * ax[1], p0[2] + tk * ax[2]};
            /* exact culls: a sample further from the cell's rectangle than the reach in plan, or more than the reach above the
             * cell's highest corner, is not within contact distance of either of its triangles */
            const double ex = p[0] < x0 ? x0 - p[0] : (p[0] > x0 + dx ? p[0] - (x0 + dx) : 0.0);
            const double ey = p[1] < y0 ? y0 - p[1] : (p[1] > y0 + dy ? p[1] - (y0 + dy) : 0.0);
            if (ex * ex + ey * ey > reach * reach || p[2] - reach > zmax) continue;
            double cur = 1e300, cn[3] = {0, 0, 1};
            if (tri == 0) hfield_triangle(p, v00, v10, v01, cur, cn); else hfield_triangle(p, v11, v01, v10, cur, cn);
            if (cur < best) { best = cur; bt = tk; bn[0] = cn[0]; bn[1] = cn[1]; bn[2] = cn[2]; }
        }
        const double dist = best - r;
        const bool hit = act && best < 1e299 && !(dist > margin);
        const unsigned long long hb = wv::ballot(hit);
        const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        const int slot = ncon + wv::popc64(hb & below);
        if (hit && slot < room) {
            const double *mh = S.x.s.geom_xmat[g1h], *pc = S.x.s.geom_xpos[g2], *mc = S.x.s.geom_xmat[g2];
            const double axw[3] = {mc[2], mc[5], mc[8]};
            double nw[3];
            mulmatvec3(nw, mh, bn);
            RawContact rc;
            rc.dist = dist;
            for (int x = 0; x < 3; ++x) {
                const double psw = pc[x] + bt * axw[x];
                rc.normal[x] = nw[x]; rc.pos[x] = psw - nw[x] * (r + 0.5 * dist); rc.tangent[x] = hl > 0 ? axw[x] : 0.0;
            }
            write_raw_contact(S, slot, pidx, rc);
        }
        ncon += wv::popc64(hb);
    }
    wv::sync();
    return ncon;
}

/* lane = contact: contact frame from (normal, tangent hint) and the pair's pre-mixed parameters (model compile
 * time, cm_model_t::pair_*), once per contact and outside the divergent pair loops */
template <class SH>
WV_DEVICE void finish_contacts(SH &S, ModelPtr m, int lane, int ncon) {
    if (lane < ncon) {
        const int p = S.c_pair[lane];
        double fr[9];
        for (int i = 0; i < 6; ++i) fr[i] = S.c_frame[lane][i];
        fr[6] = fr[7] = fr[8] = 0;
        make_frame(fr);
        for (int i = 0; i < 9; ++i) S.c_frame[lane][i] = fr[i];
        S.c_g1[lane] = m->pair_geom1[p]; S.c_g2[lane] = m->pair_geom2[p];
        S.c_margin[lane] = m->pair_includemargin[p];
        S.c_dim[lane] = m->pair_condim[p];
        for (int i = 0; i < 2; ++i) S.c_solref[lane][i] = m->pair_solref[p][i];
        for (int i = 0; i < 5; ++i) S.c_solimp[lane][i] = m->pair_solimp[p][i];
        for (int i = 0; i < 3; ++i) S.c_fri[lane][i] = m->pair_friction[p][i];
        for (int k = 0; k < 2; ++k) { S.c_root[lane][k] = m->pair_root[p][k]; S.c_dofmask[lane][k] = m->pair_dofmask[p][k]; }
        S.c_tran[lane] = m->pair_invweight[p];
    }
}

WV_DEVICE double impedance(const double *solimp, double pos, double margin) {
    double dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
    if (dmin == dmax || width <= CM_MINVAL) return 0.5 * (dmin + dmax);
    double x = fabs((pos - margin) / width);
    if (x >= 1) return dmax;
    if (x <= 0) return dmin;
    double y;
    if (power == 1) y = x;
    else if (power == 2) y = x <= mid ? x * x / mid : 1 - (1 - x) * (1 - x) / (1 - mid);
    else y = x; /* other exponents are rejected when the model is compiled (mjcf_loader.cpp) */
    return dmin + y * (dmax - dmin);
}

/* dof-tree sparsity: compile-time tables for the in-scope models (topo_static.h), or the model's own
 * masks for anything else */
struct TopoRuntime { static constexpr bool is_static = false; static constexpr bool packed = false; static constexpr int nv = 0; };

/* Where entry (k, i < k) of a factor lives in EnvShared::Lp / LHp.
 *   dense  : the full lower triangle by rows, (k, i) at k(k+1)/2 + i; the diagonal slot of a row is never read and takes
 *            the row stores of the lanes at or past the diagonal.  A lane's row or column index is base + immediate.
 *   packed : (TOPO::packed) a dof's ancestors are trunk dofs or dofs of its own block (TOPO::bstart), so row k keeps
 *            only [its trunk entries | the entries of its block below k]: 392 slots instead of 820 for the 40-dof
 *            tray model, 13.7 KB less LDS for the two factors, which is what lets four of its workgroups share a CU.
 *            Costs a few integer ops per staged entry where a lane addresses its own row / column. */
template <class TOPO, int NVP>
struct LPack {
    static constexpr bool packed = TOPO::packed;
    static constexpr int trunk() { if constexpr (TOPO::packed) return TOPO::trunk; else return 0; }
    static constexpr int bs(int k) { /* first dof of k's block */
        if constexpr (TOPO::packed) { int s = 0; for (int b = 0; b < TOPO::nblock; ++b) if (TOPO::bstart[b] <= k) s = TOPO::bstart[b]; return s; }
        else return 0;
    }
    static constexpr int len(int k) { if constexpr (TOPO::packed) return k < trunk() ? k : trunk() + (k - bs(k)); else return k + 1; }
    static constexpr int base(int k) { int s = 0; for (int j = 0; j < k; ++j) s += len(j); return s; }
    static constexpr int count = base(NVP) + (TOPO::packed ? 1 : 0);
    static constexpr int dump = count - 1; /* packed: the slot that takes the stores of lanes outside the row */
    static constexpr bool has(int k, int i) { return i < k && (!TOPO::packed || i < trunk() || i >= bs(k)); }
    static constexpr int idx(int k, int i) { /* compile-time (k, i), has(k, i) */
        if constexpr (TOPO::packed) return base(k) + (i < trunk() ? i : trunk() + i - bs(k)); else return CK_TRI(k, i);
    }
    static constexpr bool covers() { /* every ancestor pair of the topology has a slot */
        if constexpr (TOPO::packed) {
            for (int k = 0; k < NVP; ++k) for (int i = 0; i < k; ++i) if (((TOPO::table[k] >> i) & 1ull) && !has(k, i)) return false;
        }
        return true;
    }
    static constexpr bool distinct() { /* the slots of all (k, i) pairs a row keeps are 0 .. dump - 1, each used once, in order */
        if constexpr (TOPO::packed) {
            int next = 0;
            for (int k = 0; k < NVP; ++k) for (int i = 0; i < k; ++i) if (has(k, i)) { if (idx(k, i) != next) return false; ++next; }
            return next == dump;
        }
        return true;
    }
    /* slot row k (compile time) offers lane `lane`: its entry (k, lane), else a slot nobody reads */
    static WV_DEVICE int row_slot(int k, int lane) {
        if constexpr (TOPO::packed) {
            const int b = base(k), s = bs(k), T = trunk();
            if (k < T) return lane < k ? b + lane : dump;
            return lane < T ? b + lane : (lane >= s && lane < k) ? b + T - s + lane : dump;
        } else return CK_TRI(k, 0) + (lane < k ? lane : k);
    }
    /* a lane's own row: where it starts and which block it belongs to */
    struct Row { int base, bs; };
    static WV_DEVICE Row row_of(int k_) {
        Row r = {0, 0};
        if constexpr (TOPO::packed) {
            int B = 0;
#pragma unroll
            for (int b = 1; b < TOPO::nblock; ++b) if (k_ >= TOPO::bstart[b]) { r.bs = TOPO::bstart[b]; B = base(TOPO::bstart[b]); }
            const int d = k_ - r.bs;
            r.base = B + (r.bs > 0 ? d * trunk() : 0) + d * (d - 1) / 2;
        } else r.base = CK_TRI(k_, 0);
        return r;
    }
    /* entry (k_, i) of the lane's own row, i compile time: slot, and whether the row has it (i < k_ is the caller's) */
    static WV_DEVICE bool row_has(const Row &r, int i) { if constexpr (TOPO::packed) return i < trunk() || i >= r.bs; else return true; }
    static WV_DEVICE int row_idx(const Row &r, int i) { if constexpr (TOPO::packed) return r.base + (i < trunk() ? i : trunk() - r.bs + i); else return r.base + i; }
    /* entry (k, k_) of the lane's own column, k compile time (k_ < k is the caller's) */
    static WV_DEVICE bool col_has(int k, int k_) { if constexpr (TOPO::packed) return k_ < trunk() || k_ >= bs(k); else return true; }
    static WV_DEVICE int col_idx(int k, int k_) { if constexpr (TOPO::packed) return (k_ < trunk() ? base(k) : base(k) + trunk() - bs(k)) + k_; else return CK_TRI(k, k_); }
};

template <class TOPO>
WV_DEVICE unsigned long long anc_mask(ModelPtr m, int k) {
    if constexpr (TOPO::is_static) return TOPO::table[k];
    else return m->dof_ancmask[k];
}

/* bit `c` of a per-lane mask as 0.0 / 1.0: predicates of the dense tree loops are applied by multiplication (two VALU
 * ops, no compare -> scalar mask -> select round trip, which costs ~30 clocks per use on this hardware) */
WV_DEVICE double bitf(unsigned long long mask, int c) { return (double)(unsigned)((mask >> c) & 1ull); }

/* reciprocal to full fp64 accuracy without the IEEE division sequence: hardware estimate + two Newton steps
 * (the pivots are positive and far from the denormal / overflow ranges) */
WV_DEVICE double fast_rcp(double x) {
    double r = wv::rcp_estimate(x);
    r = r * (2.0 - x * r);
    r = r * (2.0 - x * r);
    return r;
}

/* L^T D L factorisation of two tree-sparse matrices (M and M + hB) held one column per lane in registers
 * (lane j owns col[i] = A[i][j], i >= j).  Pivot-row entries travel by readlane; the (k, i) loop nest is
 * fully unrolled over the ancestor pattern and the two factorisations are interleaved so that each one's
 * dependent chain hides behind the other's.  No lane predication is needed: entries above the diagonal
 * (col[i] in lanes j > i) are never read, so they may absorb harmless updates.  On exit col[k] holds L[k][j]
 * for k > j; the pivots are returned through dinv / rsd / dinvH (wave-uniform, written by lane 0). */
template <int NVP, class TOPO>
WV_DEVICE void factor_pair_in_registers(ModelPtr m, double h, double (&col)[NVP], double (&colh)[NVP], int lane, int nv,
                                        double *dinv, double *rsd, double *dinvH) {
#pragma unroll
    for (int k = NVP - 1; k >= 0; --k) {
        if (TOPO::is_static ? k >= TOPO::nv : k >= nv) continue;
        const unsigned long long anc = anc_mask<TOPO>(m, k);
        const double arm = m->dof_armature[k]; /* diagonal terms, see the mass-matrix stage */
        const double inv = fast_rcp(wv::readlane(col[k], k) + arm), invh = fast_rcp(wv::readlane(colh[k], k) + (arm + h * m->dof_damping[k]));
        if (lane == 0) { dinv[k] = inv; rsd[k] = sqrt(inv); dinvH[k] = invh; }
        if (anc == 0ull) continue;
#pragma unroll
        for (int i = k - 1; i >= 0; --i) {
            if (!((anc >> i) & 1ull)) continue;
            const double t = wv::readlane(col[k], i) * inv, th = wv::readlane(colh[k], i) * invh; /* A[k][i] / D_k */
            col[i] -= t * col[k];
            colh[i] -= th * colh[k];
        }
        col[k] *= inv;
        colh[k] *= invh;
    }
}

/* Compile-time-topology variant: the same two factorisations, eliminated height by height.  All dofs of one
 * elimination height (TOPO::height) are mutually unrelated, so a round scales their pivot rows (one multiply per
 * matrix gives L[k][:] in every lane at once), parks them in the packed LDS factors -- where the solves want them
 * anyway -- and then applies the rank-one updates with L[k][i] fetched back as LDS broadcast reads: two FMAs and two
 * reads per ancestor pair, no scalar registers, one LDS round trip per height instead of one per dof. */
/* WHICH: 2 = both factorisations, interleaved (their chains hide each other's latency); 0 = that of M alone, 1 = that of M + hB alone
 * (the two-wave form runs the second one behind the barrier J, while wave 0 solves: only the Euler step reads it) */
template <int NVP, class TOPO, int WHICH = 2, class SH>
WV_DEVICE void factor_pair_by_height(ModelPtr m, double h, SH &S, double (&col)[NVP], double (&colh)[NVP], int lane) {
    /* The trunk dofs (the floating base: each one's ancestors are all the lower ones) come last and one to a height: a
     * round through LDS for a single dof is all latency.  They are eliminated in registers instead (below), with
     * v_readlane multipliers whose round trips overlap, so the height rounds stop where the trunk begins. */
    constexpr int first_trunk_height = TOPO::height[TOPO::trunk - 1];
#pragma unroll
    for (int s = 0; s < TOPO::nheight; ++s) {
        if (s >= first_trunk_height) continue;
#pragma unroll
        for (int k = NVP - 1; k >= 0; --k) {
            if (TOPO::height[k] != s) continue;
            const double arm = m->dof_armature[k]; /* diagonal terms, see the mass-matrix stage */
            /* every lane holds the same 1/D: an unpredicated same-address store; lanes at or past the diagonal all land on one
             * unused slot of the row: an unpredicated store too */
            const int at = LPack<TOPO, NVP>::row_slot(k, lane);
            if constexpr (WHICH != 1) { const double inv = fast_rcp(wv::readlane(col[k], k) + arm); S.dinv[k] = inv; S.Lp[at] = col[k] * inv; }
            if constexpr (WHICH != 0) { const double invh = fast_rcp(wv::readlane(colh[k], k) + (arm + h * m->dof_damping[k])); S.dinvH[k] = invh; S.LHp[at] = colh[k] * invh; }
        }
        wv::sync();
#pragma unroll
        for (int k = NVP - 1; k >= 0; --k) {
            if (TOPO::height[k] != s) continue;
            /* all of this dof's multipliers are fetched before the first update (the fences keep the scheduler from
             * pairing every LDS read with its own wait): one LDS latency per dof instead of one per ancestor pair */
            double t[NVP], th[NVP];
#pragma unroll
            for (int i = k - 1; i >= 0; --i) {
                if (!((TOPO::table[k] >> i) & 1ull)) continue;
                if constexpr (WHICH != 1) t[i] = S.Lp[LPack<TOPO, NVP>::idx(k, i)];
                if constexpr (WHICH != 0) th[i] = S.LHp[LPack<TOPO, NVP>::idx(k, i)];
            }
            wv::sched_fence();
#pragma unroll
            for (int i = k - 1; i >= 0; --i) {
                if (!((TOPO::table[k] >> i) & 1ull)) continue;
                if constexpr (WHICH != 1) col[i] -= t[i] * col[k];
                if constexpr (WHICH != 0) colh[i] -= th[i] * colh[k];
            }
            wv::sched_fence();
        }
    }
    /* trunk: same arithmetic (multiplier = entry * 1/D, rounded once; update = one FMA), multipliers by v_readlane */
#pragma unroll
    for (int k = TOPO::trunk - 1; k >= 0; --k) {
        const double arm = m->dof_armature[k];
        const int at = LPack<TOPO, NVP>::row_slot(k, lane);
        if constexpr (WHICH != 1) {
            const double inv = fast_rcp(wv::readlane(col[k], k) + arm);
            S.dinv[k] = inv;
            S.Lp[at] = col[k] * inv;
            double t[TOPO::trunk];
#pragma unroll
            for (int i = k - 1; i >= 0; --i) t[i] = wv::readlane(col[k], i) * inv;
#pragma unroll
            for (int i = k - 1; i >= 0; --i) col[i] -= t[i] * col[k];
        }
        if constexpr (WHICH != 0) {
            const double invh = fast_rcp(wv::readlane(colh[k], k) + (arm + h * m->dof_damping[k]));
            S.dinvH[k] = invh;
            S.LHp[at] = colh[k] * invh;
            double th[TOPO::trunk];
#pragma unroll
            for (int i = k - 1; i >= 0; --i) th[i] = wv::readlane(colh[k], i) * invh;
#pragma unroll
            for (int i = k - 1; i >= 0; --i) colh[i] -= th[i] * colh[k];
        }
    }
    wv::sync();
    if constexpr (WHICH != 1) if (lane < TOPO::nv) S.rsd[lane] = sqrt(S.dinv[lane]);
}

/* Projected Gauss-Seidel sweeps, one constraint row per lane.  The per-row state is the SCALED residual
 * s_j = -res_j / A_jj, so a row's unclamped step is s itself and the serial chain per row is max, readlane, FMA:
 *     delta_I = max(s_I, lo_I);   s_j += B_jI * delta_I  for every j,   B_jI = -A_jI / A_jj  (brow, per lane).
 * Every lane evaluates its own candidate each row; only lane I's is consumed, through readlane.
 *
 * pgs_rows: the guarded sweep (MuJoCo's rule `never accept a cost increase`, evaluated row by row).  Nested so that
 * the first row index >= nrows ends the sweep with one wave-uniform branch. */
template <int I, int N>
WV_DEVICE void pgs_rows(const double (&brow)[N], int nrows, int r_, double Aii, double halfAii, double flo, double &f,
                        double &sres, double &improvement) {
    if constexpr (I < N) {
        if (I < nrows) {
            double delta = fmax(sres, flo - f); /* = max(f - res / Aii, flo) - f */
            double change = delta * (halfAii * delta - Aii * sres);
            if (change > 1e-10) { delta = 0; change = 0; } /* never accept a cost increase */
            const double dlt = wv::readlane(delta, I), chg = wv::readlane(change, I);
            if (r_ == I) f += dlt;
            improvement -= chg;
            sres += brow[I] * dlt;
            pgs_rows<I + 1, N>(brow, nrows, r_, Aii, halfAii, flo, f, sres, improvement);
        }
    }
}

/* The same sweep with the guard off the dependent chain: the row's own lane keeps the residual it started from
 * (its step follows from it), so every row's cost change -- hence the guard and the sweep's improvement -- can be
 * evaluated once, after the sweep.  The caller re-runs the sweep through pgs_rows when a guard would have fired. */
template <int I, int N>
WV_DEVICE void pgs_row_fast(const double (&brow)[N], int r_, double lo_f, double &sres, double &mys) {
    if constexpr (I < N) {
        const double delta = wv::max_raw(sres, lo_f); /* one v_max_f64: fmax() adds a canonicalising self-max to the row chain after every branch */
        if (r_ == I) mys = sres; /* the residual this row started from: its step is recomputed from it after the sweep */
        sres += brow[I] * wv::readlane(delta, I);
    }
}
/* rows go four to a (wave-uniform) branch: rows past the last one are inert -- their column of A is zero in every
 * lane and their own lane's step is finite -- so running up to three of them costs less than three more branches */
template <int I, int N>
WV_DEVICE void pgs_rows_fast(const double (&brow)[N], int nrows, int r_, double lo_f, double &sres, double &mys) {
    if constexpr (I < N) {
        if (I < nrows) {
            pgs_row_fast<I, N>(brow, r_, lo_f, sres, mys);
            pgs_row_fast<I + 1, N>(brow, r_, lo_f, sres, mys);
            pgs_row_fast<I + 2, N>(brow, r_, lo_f, sres, mys);
            pgs_row_fast<I + 3, N>(brow, r_, lo_f, sres, mys);
            pgs_rows_fast<I + 4, N>(brow, nrows, r_, lo_f, sres, mys);
        }
    }
}

/* x := L^-1 x (forward) and x := L^-T x (backward) by substitution, lane = dof: lrow / lcol hold the lane's row /
 * column of the unit-triangular factor (zeros outside its ancestors / descendants) and every hop is a v_readlane round
 * trip (~40 clocks).  With a compile-time topology the hops go LEVEL BY LEVEL of the dof tree (a dof's level = the number
 * of its ancestors): dofs of one level are mutually unrelated, so their broadcasts are all read from the same state of
 * the vector and their terms are summed before they touch it -- the dependent chain is as long as the tree is deep (13
 * for Cassie: floating base, hip, knee, shin, tarsus, crank), not as long as the dof list (32), and the forward pass
 * skips the dofs nobody descends from (their column of L is empty). */
template <class TOPO>
struct DofLevels {
    static constexpr int level(int k) { int n = 0; for (int i = 0; i < TOPO::nv; ++i) n += (int)((TOPO::table[k] >> i) & 1ull); return n; }
    static constexpr bool has_descendants(int j) { for (int k = 0; k < TOPO::nv; ++k) if ((TOPO::table[k] >> j) & 1ull) return true; return false; }
    static constexpr int depth() { int d = 0; for (int k = 0; k < TOPO::nv; ++k) if (level(k) > d) d = level(k); return d; }
};
template <int NVP, class TOPO>
WV_DEVICE double solve_forward(double z, const double (&lrow)[NVP], int lane, int nv) {
    if constexpr (TOPO::is_static) {
        typedef DofLevels<TOPO> LV;
#pragma unroll
        for (int d = 0; d < LV::depth(); ++d) {
            double t0 = 0, t1 = 0;
            int n = 0;
#pragma unroll
            for (int j = 0; j < TOPO::nv; ++j) {
                if (LV::level(j) != d || !LV::has_descendants(j)) continue;
                const double bj = wv::readlane(z, j);
                if ((n++ & 1) == 0) t0 = fma(lrow[j], bj, t0); else t1 = fma(lrow[j], bj, t1);
            }
            z -= t0 + t1;
        }
        return z;
    } else {
#pragma unroll
        for (int i = 0; i < NVP - 1; ++i) {
            if (i >= nv - 1) continue;
            z -= lrow[i] * wv::readlane(z, i);
        }
        return z;
    }
}
template <int NVP, class TOPO>
WV_DEVICE double solve_backward(double w, const double (&lcol)[NVP], int lane, int nv) {
    if constexpr (TOPO::is_static) {
        typedef DofLevels<TOPO> LV;
#pragma unroll
        for (int d = LV::depth(); d >= 1; --d) {
            double t0 = 0, t1 = 0;
            int n = 0;
#pragma unroll
            for (int j = 0; j < TOPO::nv; ++j) {
                if (LV::level(j) != d) continue;
                const double bj = wv::readlane(w, j);
                if ((n++ & 1) == 0) t0 = fma(lcol[j], bj, t0); else t1 = fma(lcol[j], bj, t1);
            }
            w -= t0 + t1;
        }
        return w;
    } else {
#pragma unroll
        for (int k = NVP - 1; k >= 1; --k) {
            if (k >= nv) continue;
            w -= lcol[k] * wv::readlane(w, k);
        }
        return w;
    }
}

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

/* ---------------- the solve of the 127-row instantiation (MAXR = WIDE_ROWS, two wavefronts): both waves call it behind the barrier
 * J, wave w for the rows 64 w .. 64 w + 63 (row 127 = the qfrc_smooth column, wave 1's lane 63).  The row stages (wave 0, one or two
 * passes) have parked every row's raw Jacobian row in x.Yr and its parameters in x.rowt; here every lane takes its row from there,
 * runs the half solve, puts the staged row back (barrier Y: the staged matrix is complete), forms ITS WAVE'S 64 x 64 block of
 * A = Y Y^T in registers, and solves:
 *   - a substep of at most 64 rows lives on wave 0 alone: the very chain of operations of the 63-row instantiation (so a substep that
 *     fits both gives the same bits in both), wave 1 returns at once;
 *   - with more rows a Gauss-Seidel sweep is wave 0's rows, then wave 1's, in row order -- the oracle's order.  The waves never run
 *     at the same time, so the chain crosses them twice per sweep, not once per row: a wave that has finished its half hands over
 *     v = sum over its rows of (staged row of Y) x (the row's step), the steps' image in joint space (NVP numbers through LDS), and
 *     the other wave's residuals take it in through their own staged rows, res_j += Y_j . v -- which is A_jI step_I summed over the
 *     other wave's rows I without the off-diagonal blocks of A existing anywhere.  The guard (never accept a cost increase) is
 *     local to a half; the sweep's cost change is summed in row order across both halves, as the oracle sums it.
 * Returns this lane's row force; iters / nguarded: the sweeps taken (valid in both waves when the substep has more than 64 rows,
 * in wave 0 otherwise). ---------------- */
template <int NVP, class TOPO, class SH>
WV_DEVICE double wide_solve(SH &S, ModelPtr m, const int wid, const int nefc, const int sub, int &iters, int &nguarded) {
    constexpr int H = NROW, MAXR = WIDE_ROWS;
    static_assert(TOPO::is_static, "the 127-row instantiation exists for the compile-time topologies");
    const int lane = wv::fresh_lane();
    const int g = H * wid + lane;                    /* this lane's row (127: the qfrc_smooth column) */
    const bool hasrow = g < nefc, qcol = g == MAXR;
    const int nown = wid == 0 ? (nefc < H ? nefc : H) : (nefc > H ? nefc - H : 0);
    iters = 0; nguarded = 0;
    double rR = 1.0, raref = 0.0, jws = 0.0, code = -1.0;
    {
        const double *rt = S.x.rowt[hasrow ? g : 0];
        const double t0 = rt[0], t1 = rt[1], t2 = rt[2], t3 = rt[3];
        if (hasrow) { rR = t0; raref = t1; jws = t2; code = t3; }
    }
    const bool isrow = hasrow && code >= 0.0, clampf = isrow && code > 0.5;
    double ycol[NVP];
#pragma unroll
    for (int k = 0; k < NVP; ++k) {
        const double raw = S.x.Yr[hasrow ? g : 0][k], qs = S.qfrc_smooth[k < TOPO::nv ? k : 0];
        ycol[k] = hasrow ? raw : ((qcol && k < TOPO::nv) ? qs : 0.0);
    }
    /* ---- half solve in registers: Y = D^-1/2 L^-T [J^T | qfrc_smooth] (env_step's, for a compile-time topology) ---- */
    {
        double ta[NVP], tb[NVP], ra = 0, rb = 0;
        auto fetch = [&](int k, double (&t)[NVP], double &rs) {
#pragma unroll
            for (int i = k - 1; i >= 0; --i) if ((TOPO::table[k] >> i) & 1ull) t[i] = S.Lp[LPack<TOPO, NVP>::idx(k, i)];
            rs = S.rsd[k];
        };
        fetch(TOPO::nv - 1, ((TOPO::nv - 1) & 1) ? ta : tb, ((TOPO::nv - 1) & 1) ? ra : rb);
#pragma unroll
        for (int k = NVP - 1; k >= 0; --k) {
            if (k >= TOPO::nv) continue;
            if (k > 0) fetch(k - 1, ((k - 1) & 1) ? ta : tb, ((k - 1) & 1) ? ra : rb);
            wv::sched_fence();
            const double (&t)[NVP] = (k & 1) ? ta : tb;
            const double xk = ycol[k];
#pragma unroll
            for (int i = k - 1; i >= 0; --i) if ((TOPO::table[k] >> i) & 1ull) ycol[i] -= t[i] * xk;
            ycol[k] = xk * ((k & 1) ? ra : rb);
            wv::sched_fence();
        }
    }
    {   /* (every lane: the lanes that hold no row store zeros, so that all 128 rows of the tile are defined) */
#pragma unroll
        for (int k = 0; k < NVP; ++k) S.x.Yr[g][k] = ycol[k];
    }
    wv::block_barrier(); /* Y: the staged matrix is complete -- both waves' rows and the qfrc_smooth column */
    if (nefc <= H && wid == 1) return 0.0; /* (no rows on this wave) */

    /* ---- this lane's row of its wave's block of A = Y Y^T (one FMA chain per product over the dofs in index order, as everywhere),
     *      b = Y y_q - aref, the diagonal ---- */
    double arow[H];
    const int rbase = H * wid;
#pragma unroll
    for (int r = 0; r < H; r += 2) {
        double acc0 = 0, acc1 = 0;
        if (rbase + r < nefc) {
            double ya[NVP], yb[NVP];
#pragma unroll
            for (int k = 0; k < NVP; ++k) ya[k] = S.x.Yr[rbase + r][k];
#pragma unroll
            for (int k = 0; k < NVP; ++k) yb[k] = S.x.Yr[rbase + r + 1][k];
            wv::sched_fence();
#pragma unroll
            for (int k = 0; k < NVP; ++k) { acc0 = fma(ya[k], ycol[k], acc0); acc1 = fma(yb[k], ycol[k], acc1); }
        }
        arow[r] = acc0;
        arow[r + 1] = rbase + r + 1 < nefc ? acc1 : 0.0; /* (row 127 is the qfrc_smooth column, not a row) */
    }
    double rb;
    {
        double acc = 0;
#pragma unroll
        for (int k = 0; k < NVP; ++k) acc = fma(S.x.Yr[MAXR][k], ycol[k], acc);
        rb = acc - raref;
    }
    double Aii = 1.0;
    if (isrow) {
        double d = 0;
#pragma unroll
        for (int k = 0; k < NVP; ++k) d = fma(ycol[k], ycol[k], d);
        Aii = d + rR;
    }
    const double invAii = 1.0 / Aii;
    double f = 0, res = isrow ? rb : 0.0;
    const int r_ = lane; /* the row's index within its wave */
    const int nvs = TOPO::nv;
    const double scale = 1.0 / (m->meaninertia * (nvs > 1 ? nvs : 1));
    const double halfAii = 0.5 * Aii;
    const double flo = clampf ? 0.0 : -1e300;
    const double ninvAii = -invAii;
    const int maxiter = m->iterations;
    const double tolerance = m->tolerance;
    const int kk = lane < NVP ? lane : 0;

    if (nefc <= H) {
        /* ======== at most 64 rows: wave 0 alone, the 63-row instantiation's chain of operations ======== */
        if (m->flags & CM_FLAG_WARMSTART) {
            if (isrow) {
                f = -(jws - raref) / rR;
                if (clampf && f < 0) f = 0;
            }
            double af0 = 0, af1 = 0, af2 = 0, af3 = 0;
#pragma unroll
            for (int t = 0; t < H; t += 4) {
                if (t < nefc) {
                    af0 += arow[t] * wv::readlane(f, t);
                    af1 += arow[t + 1] * wv::readlane(f, t + 1);
                    af2 += arow[t + 2] * wv::readlane(f, t + 2);
                    af3 += arow[t + 3] * wv::readlane(f, t + 3);
                }
            }
            const double af = ((af0 + af1) + (af2 + af3)) + (isrow ? rR * f : 0.0);
            double cost = wv::wave_sum(isrow ? f * (rb + 0.5 * af) : 0.0);
            if (cost > 0) f = 0;
            else if (isrow) res = rb + af;
        }
        double sres = res * ninvAii;
#pragma unroll
        for (int t = 0; t < H; ++t) arow[t] *= ninvAii;
        const double cdiag = isrow ? rR * ninvAii : 0.0;
        const bool shortcut_ok = 0.5 * tolerance > (double)MID_ROWS * 1e-10 * scale;
        while (iters < maxiter) {
            const int nrows = wv::opaque(nefc);
            bool converged;
            {
                const double f0 = f, s0 = sres;
                double mys = 0;
                const double lo_f = flo - f;
                pgs_rows_fast<0, H>(arow, nrows, r_, lo_f, sres, mys);
                const double mydelta = wv::max_raw(mys, lo_f);
                const double change = (r_ < nrows) ? mydelta * (halfAii * mydelta - Aii * mys) : 0.0;
                const float tol = (float)tolerance;
                const bool one_row_decides = shortcut_ok && wv::ballot(-(float)change * (float)scale > 2.5f * tol) != 0ull;
                const float est = one_row_decides ? 4.0f * tol : -wv::wave_sum_f32((float)change) * (float)scale;
                if (wv::ballot(change > 1e-10) != 0ull || wv::debug_force_guarded()) {
                    double improvement = 0;
                    f = f0; sres = s0; ++nguarded;
                    pgs_rows<0, H>(arow, nrows, r_, Aii, halfAii, flo, f, sres, improvement);
                    sres = fma(cdiag, f - f0, sres);
                    converged = improvement * scale < tolerance;
                } else {
                    if (r_ < nrows) { f += mydelta; sres = fma(cdiag, mydelta, sres); }
                    if (est < 0.5f * tol) converged = true;
                    else if (est > 2.0f * tol) converged = false;
                    else {
                        const double tree = -wv::wave_sum(change) * scale, tolv = tolerance;
                        if (fabs(tree - tolv) > 1e-9 * tolv) converged = tree < tolv;
                        else {
                            double improvement = 0;
                            for (int t = 0; t < nrows; ++t) improvement -= wv::readlane(change, t);
                            converged = improvement * scale < tolerance;
                        }
                    }
                }
            }
            ++iters;
            if (converged) break;
        }
        return f;
    }

    /* ======== more than 64 rows: the sweep crosses the waves ======== */
    /* v[wid] = sum over this wave's rows t < nown of (staged row) x val_t, lane = dof (four partial sums, rows four to a branch) */
    auto image_of = [&](double val) {
        double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
#pragma unroll
        for (int t = 0; t < H; t += 4) {
            if (t < nown) {
                v0 = fma(S.x.Yr[rbase + t][kk], wv::readlane(val, t), v0);
                v1 = fma(S.x.Yr[rbase + t + 1][kk], wv::readlane(val, t + 1), v1);
                v2 = fma(S.x.Yr[rbase + t + 2][kk], wv::readlane(val, t + 2), v2);
                v3 = fma(S.x.Yr[rbase + t + 3][kk], wv::readlane(val, t + 3), v3); /* (val is 0 in lanes that are not rows; their rows of the tile are zeros) */
            }
        }
        if (lane < NVP) S.x.vx[wid][lane] = (v0 + v1) + (v2 + v3);
    };
    /* this lane's staged row times the other wave's image: what the other wave's values contribute to this row's A x */
    auto cross = [&]() {
        double d = 0;
#pragma unroll
        for (int k = 0; k < NVP; ++k) d = fma(ycol[k], S.x.vx[1 - wid][k], d);
        return d;
    };
    if (m->flags & CM_FLAG_WARMSTART) {
        if (isrow) {
            f = -(jws - raref) / rR;
            if (clampf && f < 0) f = 0;
        }
        image_of(f);
        wv::block_barrier();
        double af0 = 0, af1 = 0, af2 = 0, af3 = 0;
#pragma unroll
        for (int t = 0; t < H; t += 4) {
            if (t < nown) {
                af0 += arow[t] * wv::readlane(f, t);
                af1 += arow[t + 1] * wv::readlane(f, t + 1);
                af2 += arow[t + 2] * wv::readlane(f, t + 2);
                af3 += arow[t + 3] * wv::readlane(f, t + 3);
            }
        }
        const double af = (((af0 + af1) + (af2 + af3)) + cross()) + (isrow ? rR * f : 0.0);
        const double part = wv::wave_sum(isrow ? f * (rb + 0.5 * af) : 0.0);
        if (lane == 0) S.x.sums[wid] = part;
        wv::block_barrier();
        const double cost = S.x.sums[0] + S.x.sums[1];
        if (cost > 0) f = 0;
        else if (isrow) res = rb + af;
        wv::block_barrier(); /* (vx and sums are free again) */
    }
    double sres = res * ninvAii;
#pragma unroll
    for (int t = 0; t < H; ++t) arow[t] *= ninvAii;
    const double cdiag = isrow ? rR * ninvAii : 0.0;
    /* the turn word: base + 2 s + 1 = wave 0 has finished its half of sweep s, base + 2 s + 2 = wave 1 has (and has left its verdict) */
    const int base = (sub + 1) << 12;
    const int sweeps_max = maxiter < 2000 ? maxiter : 2000;
    for (int sweep = 0;; ++sweep) {
        double carried = 0.0; /* the sweep's cost change summed in row order up to this wave's first row */
        if (wid == 0) {
            if (sweep > 0) {
                wv::wait_for(&S.x.turn[1], base + 2 * sweep);
                if (wv::opaque(S.x.turn[2])) break; /* (wave 1's verdict on the sweep before: converged, or out of sweeps) */
                sres = fma(ninvAii, cross(), sres);
            }
        } else {
            wv::wait_for(&S.x.turn[1], base + 2 * sweep + 1);
            sres = fma(ninvAii, cross(), sres);
            carried = S.x.sums[2];
        }
        const int nrows = wv::opaque(nown);
        double improvement = carried, dstep;
        {
            const double f0 = f, s0 = sres;
            double mys = 0;
            const double lo_f = flo - f;
            pgs_rows_fast<0, H>(arow, nrows, r_, lo_f, sres, mys);
            const double mydelta = wv::max_raw(mys, lo_f);
            const double change = (r_ < nrows) ? mydelta * (halfAii * mydelta - Aii * mys) : 0.0;
            if (wv::ballot(change > 1e-10) != 0ull || wv::debug_force_guarded()) { /* some row of this half would have raised the cost: redo it guarded */
                f = f0; sres = s0;
                ++nguarded;
                pgs_rows<0, H>(arow, nrows, r_, Aii, halfAii, flo, f, sres, improvement);
                sres = fma(cdiag, f - f0, sres);
                dstep = f - f0;
            } else {
                dstep = (r_ < nrows) ? mydelta : 0.0;
                if (r_ < nrows) { f += mydelta; sres = fma(cdiag, mydelta, sres); }
                for (int t = 0; t < nrows; ++t) improvement -= wv::readlane(change, t);
            }
        }
        image_of(dstep);
        if (wid == 0) {
            if (lane == 0) S.x.sums[2] = improvement;
            wv::publish(&S.x.turn[1], base + 2 * sweep + 1);
        } else {
            iters = sweep + 1;
            const bool stop = improvement * scale < tolerance || iters >= sweeps_max;
            if (lane == 0) { S.x.turn[2] = stop ? 1 : 0; S.x.turn[3] = iters; S.x.sums[3] = (double)nguarded; }
            wv::publish(&S.x.turn[1], base + 2 * sweep + 2);
            if (stop) break;
        }
    }
    if (wid == 0) { iters = wv::opaque(S.x.turn[3]); nguarded += (int)S.x.sums[3]; }
    return f;
}

/* ======================================================== the env step ==== */
/* FEAT selects the collision code a model needs, so that the instantiation for plain cassie.xml does not carry the
 * register pressure of paths it never takes: FEAT_HFIELD = height-field pairs, FEAT_WAVEPAIRS = plane-box / box-box
 * pairs handled by the whole wave.  The launcher picks the instantiation from the model (phys_batch.hip). */

template <int NVP, class TOPO, int FEAT, int MAXR, int NW>
WV_DEVICE void env_step(const PhysIO &io, EnvShared<NVP, LPack<TOPO, NVP>::count, MAXR> &S, int env, int sub_start, int nsub) {
    /* nsub: the substep this call ends in front of -- io.nsub, or the end of this workgroup's chunk of the launch (PhysIO::nchunk):
     * the call then ends like a launch of nsub substeps */
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

    static_assert(LP::covers() && LP::distinct(), "packed factor rows must hold every ancestor pair, each in its own slot");
    typedef EnvShared<NVP, LPack<TOPO, NVP>::count, MAXR> SH_T;
    constexpr int MAXC = SH_T::MAXC;
    constexpr bool WIDE = SH_T::WIDE;
    static_assert(!WIDE || (NW == 2 && MAXR == WIDE_ROWS), "the 127-row instantiation spreads its solve over the two wavefronts of an env");
    const ModelPtr m_launch = (ModelPtr)(io.models + (size_t)env * io.model_stride);
    ModelPtr m = m_launch;
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

    /* ---------------- two-wave form: wave 1's program ---------------- */
    if constexpr (NW == 2) {
        if (wid == 1) {
            for (int sub1 = sub_start;;) {
                /* this substep's constants are requested ahead of the barrier, where their trip through memory costs nothing */
                const int pb = isbody ? b : 0;
                const double mass = m->body_mass[pb];
                double iner[3], imat[9];
                for (int i = 0; i < 3; ++i) iner[i] = m->body_inertia[pb][i];
                for (int i = 0; i < 9; ++i) imat[i] = m->body_imat[pb][i];
                CK_STAMP(35);
                wv::block_barrier(); /* F: wave 0 has the poses, the inertial origins and the joint anchors / axes in LDS */
                if (wv::opaque(S.cmd[0])) return;
                CK_STAMP(36);
                double ximat[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
                if (isbody && b > 0) {
                    double xm[9];
                    for (int i = 0; i < 9; ++i) xm[i] = S.x.s.xmat[b][i];
                    for (int i = 0; i < 3; ++i)
                        for (int c = 0; c < 3; ++c) ximat[3 * i + c] = xm[3 * i] * imat[c] + xm[3 * i + 1] * imat[3 + c] + xm[3 * i + 2] * imat[6 + c];
                }
                double col[NVP], colh[NVP];
                mass_matrix_columns<NVP, TOPO, FEAT, NW>(io, S, m, env, ids, mass, iner, ximat, col, colh);
                /* X as two one-directional flags: com, cinert and cdof are in LDS and the buf tile is free again (wave 0's velocity
                 * stage waits for that); the factorisations -- which touch nothing wave 0's collision stage does -- do not wait for
                 * wave 0's collision verdict, only the drive-level pass (it changes the drive state) does */
                wv::publish(&S.cmd[3], sub1 + 1);
                CK_STAMP(38);
                factor_pair_by_height<NVP, TOPO, 0>(m, h, S, col, colh, lane); /* (that of M + hB: behind the barrier J) */
                /* The columns of M + hB wait for their factorisation behind J.  With 40 dofs and 256 registers they cannot wait in
                 * registers (80 of them, across the drive-level pass, the bias stage and the sensor stage: the model constants those
                 * stages request went to scratch one by one): they wait in LDS, in the slots their factor will take -- entry (i, j)
                 * of the lower triangle in the factor's slot (i, j), the diagonal in dinvH -- and come back behind J. */
                constexpr bool park_colh = NVP > 32;
                if constexpr (park_colh) {
                    if (isdof) {
#pragma unroll
                        for (int i = 0; i < NVP; ++i) {
                            if (i >= TOPO::nv) continue;
                            if (i == k_) S.dinvH[k_] = colh[i];
                            else if (i > k_ && LP::col_has(i, k_)) S.LHp[LP::col_idx(i, k_)] = colh[i]; /* (entries outside the slots are zero) */
                        }
                    }
                    wv::sync();
                }
                wv::wait_for(&S.cmd[4], sub1 + 1);
                if (wv::opaque(S.cmd[0])) return; /* (the row-capped instantiation hands this substep over) */
                if (io.drive_mode) {
                    if (io.integrate) drive_level_io(io, S, m, env, lane, sub1 == nsub - 1);
                    wv::sync();
                }
                CK_STAMP(4);
                {
                    const int kd = isdof ? k_ : 0;
                    const double kdamp = m->dof_damping[kd], kstiff = m->dof_stiffness[kd], kref = m->dof_springref[kd];
                    const double kgear = m->dof_gear[kd], klo = m->dof_ctrl_lo[kd], khi = m->dof_ctrl_hi[kd];
                    const int kq = m->dof_qadr[kd], ka = m->dof_act[kd];
                    wv::wait_for(&S.cmd[1], sub1 + 1); /* wave 0's velocity stage has the body forces (cfrc) in LDS */
                    bias_forces_and_qfrc_smooth<NVP, (NVP > 32)>(io, S, m, env, ids, kdamp, kstiff, kref, kgear, klo, khi, kq, ka);
                }
                /* (the sensor stage's constants: requested ahead of the barrier) */
                const bool lastsub1 = sub1 == nsub - 1 || !io.integrate;
                const bool need_imu1 = lastsub1 || io.all_outputs_every_substep || (io.drive_mode && sub1 + 2 == nsub);
                const bool issens1 = lane < m->nsensor && (lastsub1 || io.drive_mode || io.all_outputs_every_substep);
                const int ls1 = issens1 ? lane : 0;
                const SensorConsts sens_c1 = request_sensor_consts(m, ls1);
                /* the sensor stage, in the time this wave used to wait at J for wave 0's Jacobian rows: everything it reads is in LDS
                 * (poses since F, body velocities and bias accelerations since cmd[1]) and its own drive-level pass has read the
                 * previous substep's sensor words */
                const int aslot1 = sensors_before_solve(io, S, m, env, issens1, ls1, sens_c1, need_imu1, lastsub1);
                CK_STAMP(47);
                wv::block_barrier(); /* J: the factor of M and qfrc_smooth are in LDS, the sensor stage is done with the body tiles */
                CK_STAMP(39);
                if constexpr (WIDE) {
                    /* the 127-row instantiation: this wave's rows 64 .. 126 and the qfrc_smooth column (wide_solve); its row forces
                     * go to LDS for the barrier P like wave 0's */
                    int it1 = 0, ng1 = 0;
                    const double f1 = wide_solve<NVP, TOPO>(S, m, 1, wv::opaque(S.x.turn[0]), sub1, it1, ng1);
                    (&S.c_solimp[0][0])[NROW + lane] = f1;
                }
                /* the factorisation of M + hB, which only this wave's Euler step reads: here, in the time this wave would otherwise
                 * wait for wave 0's solve, instead of on the way to the barrier J, where wave 0 waited for it (+4.6 %) */
                if constexpr (park_colh) {
#pragma unroll
                    for (int i = 0; i < NVP; ++i) {
                        double v = 0.0;
                        if (i < TOPO::nv && isdof) {
                            if (i == k_) v = S.dinvH[k_];
                            else if (i > k_ && LP::col_has(i, k_)) v = S.LHp[LP::col_idx(i, k_)];
                        }
                        colh[i] = v;
                    }
                    wv::sync();
                }
                factor_pair_by_height<NVP, TOPO, 1>(m, h, S, col, colh, lane);
                /* ---- the stages behind wave 0's constraint solve: operands staged now, while wave 0 assembles and solves ---- */
                {
                    const bool lastsub = lastsub1, need_imu = need_imu1;
                    const int sb = sens_c1.sb, aslot = aslot1;
                    const int pf_u = lane < nu ? lane : 0, pf_ej = lane < njnt ? lane : 0;
                    const double pf_agear = m->act_gear[pf_u], pf_kdamp = m->dof_damping[isdof ? k_ : 0];
                    const int pf_adof = m->act_dofid[pf_u], pf_ejt = m->jnt_type[pf_ej], pf_eqa = m->jnt_qposadr[pf_ej], pf_eda = m->jnt_dofadr[pf_ej];
                    const double *const fbuf = &S.c_solimp[0][0];
                    double lrow[NVP];
                    constexpr bool stage_h_early = NVP <= 32;
                    if constexpr (stage_h_early) stage_factor_row<NVP, TOPO>(S, k_, isdof, lrow);
                    /* ... and its column of the factor of M + hB (this wave formed it right behind J): the Euler step's first
                     * substitution then starts without the column's 32 address computations and reads in front of it */
                    double lcol[NVP], lrowh[NVP];
                    /* (the 40-dof instantiation at 256 registers cannot hold its row of L, this column and -- from the qacc stage on -- the row of
                     * the factor of M + hB at once: 240 registers, which went to scratch and came back on the tail; there every operand
                     * is fetched right in front of its use, 120 LDS reads on the tail instead) */
                    if constexpr (stage_h_early) stage_factor_h<NVP, TOPO, 0>(S, k_, isdof, nv, lcol, lrowh);
                    const double rsdk = isdof ? S.rsd[k_] : 0.0;
                    const int kk = isdof ? k_ : 0;
                    /* this lane's column of the staged matrix (row MAXR = the qfrc_smooth column), once wave 0 has put it in LDS: the
                     * row-capped instantiation keeps it in registers across wave 0's solve, the full one reads it at its use */
                    constexpr bool stage_y = MAXR <= 31;
                    double ycolk[stage_y ? MAXR + 1 : 1];
                    if constexpr (stage_y) {
                        wv::wait_for(&S.cmd[2], sub1 + 1);
#pragma unroll
                        for (int r = 0; r <= MAXR; ++r) ycolk[r] = S.x.Yr[r][kk];
                    }
                    wv::block_barrier(); /* P: wave 0's row forces and solver statistics are in LDS */
                    const double f = fbuf[lane], f_hi = WIDE ? fbuf[NROW + lane] : 0.0; /* (rows 64 .. 126 of the 127-row instantiation) */
                    constexpr int FB = WIDE ? 2 * NROW : NROW;
                    const int ncon = (int)fbuf[FB], nefc = (int)fbuf[FB + 1], iters = (int)fbuf[FB + 2], nguarded = (int)fbuf[FB + 3];
                    /* ================= qacc = L^-1 D^-1/2 (y63 + Y f)  (lane = dof) ================= */
                    double z, z1 = 0, z2 = 0, z3 = 0;
                    if constexpr (stage_y) {
                        /* (the summation order of the one-wave form's loop: rows four to a group into four partial sums, the rows of
                         * the last, partial group into the first) */
                        z = isdof ? ycolk[MAXR] : 0.0;
#pragma unroll
                        for (int r = 0; r < MAXR; r += 4) {
                            if (r + 4 <= nefc) {
                                z += ycolk[r] * wv::readlane(f, r); z1 += ycolk[r + 1 < MAXR ? r + 1 : r] * wv::readlane(f, r + 1);
                                z2 += ycolk[r + 2 < MAXR ? r + 2 : r] * wv::readlane(f, r + 2); z3 += ycolk[r + 3 < MAXR ? r + 3 : r] * wv::readlane(f, r + 3);
                            } else {
#pragma unroll
                                for (int j = 0; j < 4; ++j) if (r + j < nefc && r + j < MAXR) z += ycolk[r + j < MAXR ? r + j : r] * wv::readlane(f, r + j);
                            }
                        }
                    } else {
                        z = isdof ? S.x.Yr[MAXR][k_] : 0.0;
                        /* (row r's force: lane r of f, or -- rows 64 .. 126 of the 127-row instantiation -- lane r - 64 of f_hi) */
                        auto frow = [&](int r) { if constexpr (WIDE) return r < NROW ? wv::readlane(f, r) : wv::readlane(f_hi, r - NROW); else return wv::readlane(f, r); };
                        int r = 0;
                        for (; r + 4 <= nefc; r += 4) {
                            const double y0 = S.x.Yr[r][kk], y1 = S.x.Yr[r + 1][kk], y2 = S.x.Yr[r + 2][kk], y3 = S.x.Yr[r + 3][kk];
                            z += y0 * frow(r); z1 += y1 * frow(r + 1);
                            z2 += y2 * frow(r + 2); z3 += y3 * frow(r + 3);
                        }
                        for (; r < nefc; ++r) z += S.x.Yr[r][kk] * frow(r);
                    }
                    z = (z + z1) + (z2 + z3);
                    if (!isdof) z = 0.0;
                    if (isdof) z *= rsdk;
                    /* (the Euler step's operands are requested here: their LDS latency runs under the substitution below) */
                    if constexpr (stage_h_early) stage_factor_h<NVP, TOPO, 1>(S, k_, isdof, nv, lcol, lrowh); /* (the row: its registers were the staged column of Y until here) */
                    const double dih = isdof ? S.dinvH[k_] : 0.0;
                    if constexpr (!stage_h_early) stage_factor_row<NVP, TOPO>(S, k_, isdof, lrow);
                    wv::sched_fence();
                    const double qacc = solve_forward<NVP, TOPO>(z, lrow, lane, nv);
                    {
                        const bool badv = isdof && (!(qacc == qacc) || fabs(qacc) > 1e10);
                        if (wv::ballot(badv) != 0ull) { /* diverged: the state stays as it is, wave 0 raises the flag */
                            if (lane == 0) S.cmd[0] = 2;
                            wv::block_barrier(); /* E */
                            return;
                        }
                    }
                    if (isdof) S.qacc[k_] = qacc;
                    const double av = pf_agear * S.qvel[pf_adof]; /* (the actuator velocity of the state the substep started from) */
                    wv::sync();
                    CK_STAMP(12);
                    /* Euler first: wave 0's next substep waits for qpos / qvel / the warm start only, so the barrier E sits right
                     * behind them and the substep's outputs -- the accelerometers (they read qacc, the partials parked in S.accel and
                     * cdof, none of which wave 0 touches before the next barrier F), the actuator velocities, the last substep's
                     * stores -- run beside wave 0's guard and kinematics stage instead of in front of them */
                    if (io.integrate) {
                        if constexpr (!stage_h_early) stage_factor_h<NVP, TOPO>(S, k_, isdof, nv, lcol, lrowh);
                        euler_step<NVP, TOPO>(S, m, lane, isdof, k_, nv, njnt, h, qacc, lcol, lrowh, dih, pf_kdamp, pf_ejt, pf_eqa, pf_eda);
                    }
                    CK_STAMP(13);
                    wv::block_barrier(); /* E: the state of the next substep is in LDS (qpos / qvel / warm start) */
                    outputs_after_qacc(io, S, m, env, lane, isdof, k_, nu, qacc, aslot, sb, need_imu, lastsub, av, ncon, nefc, iters, nguarded);
                    CK_STAMP(14);
                }
                if (!io.integrate || ++sub1 >= nsub) return;
            }
        }
    }

    bool bailed = false;
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
        const double pf_mass = m->body_mass[pf_b];
        double pf_ipos[3], pf_imat[9], pf_iner[3], pf_gpos[3], pf_gmat[9];
        for (int i = 0; i < 3; ++i) { pf_ipos[i] = m->body_ipos[pf_b][i]; pf_iner[i] = m->body_inertia[pf_b][i]; pf_gpos[i] = m->geom_pos[pf_gs][i]; }
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
            factor_pair_by_height<NVP, TOPO>(m, h, S, col, colh, lane);
        } else {
            factor_pair_in_registers<NVP, TOPO>(m, h, col, colh, lane, nv, S.dinv, S.rsd, S.dinvH);
            if (isdof) {
#pragma unroll
                for (int k = 1; k < NVP; ++k) {
                    if (TOPO::is_static ? k >= TOPO::nv : k >= nv) continue;
                    if (k > k_) { S.Lp[CK_TRI(k, k_)] = col[k]; S.LHp[CK_TRI(k, k_)] = colh[k]; }
                }
            }
        }
        if constexpr (NW == 1) CK_STAMP(4);

        /* ================= P4 collision ================= */
        /* Requested here, read behind the collision passes (joint limits; the dof-chain indices of the velocity stage): the
         * constants' trip through the memory system runs under the pair loop. */
        /* (unconditional reads at clamped indices; the lane predicates are applied where the values are used, behind
         * wv::keep -- a predicate folded into the read would make the compiler wait for the value on the spot) */
        /* rounds of radix-4 pointer jumping along the dof chains (velocity stage, cm_model_t::dof_anc4): after round r a dof
         * holds the sum over itself and its 4^(r+1) - 1 nearest ancestors; a compile-time topology knows how long its longest
         * chain is (Cassie: 14 dofs -> two rounds) */
        constexpr int chain_rounds = [] {
            if constexpr (TOPO::is_static) { int r = 0, n = 1; while (n < DofLevels<TOPO>::depth() + 1) { n *= 4; ++r; } return r < 3 ? r : 3; }
            else return 3;
        }();
        int pf_jlim, pf_jtype, pf_jqadr, danc[9], blast, kvin;
        double pf_jmargin, pf_jlo, pf_jhi;
        auto request_behind_collision = [&](int lane_now) {
            const int pf_lj = lane_now < njnt ? lane_now : 0, pf_kd = lane_now < nv ? lane_now : 0, pf_bb = lane_now < nbody ? lane_now : 0;
            pf_jlim = m->jnt_limited[pf_lj]; pf_jtype = m->jnt_type[pf_lj]; pf_jqadr = m->jnt_qposadr[pf_lj];
            pf_jmargin = m->jnt_margin[pf_lj]; pf_jlo = m->jnt_range[pf_lj][0]; pf_jhi = m->jnt_range[pf_lj][1];
#pragma unroll
            for (int r = 0; r < 3 * chain_rounds; ++r) danc[r] = m->dof_anc4[pf_kd][r];
            blast = m->body_lastdof[pf_bb]; kvin = m->dof_vinsrc[pf_kd];
        };
        /* (the instantiations with the height-field pre-pass have no registers to spare across it and the pair loop: they ask
         * behind the loop) */
        constexpr bool request_early = (FEAT & FEAT_HFIELD) == 0;
        if constexpr (request_early) request_behind_collision(lane);
        /* A pair's constants (one level of lane-coalesced reads of the denormalised pair_* arrays) are requested ahead of
         * their pass: the first 64 pairs' here, the next 64 pairs' between a pass's narrow phase and its contact compaction
         * (into the same registers, which the narrow phase has finished with), so that their trip through the memory system
         * runs under the compaction. */
        const int pair_bound = m->npair_simple;
        struct PairConst { int g1, g2, tt; double margin, rb1, rb2, s10, s11, s20, s21, s22; };
        auto request_pair = [&](int pp, PairConst &c) {
            const int ps = pp < pair_bound ? pp : 0;
            c.g1 = m->pair_geom1[ps]; c.g2 = m->pair_geom2[ps]; c.tt = m->pair_type[ps];
            c.margin = m->pair_margin[ps]; c.rb1 = m->pair_rbound[ps][0]; c.rb2 = m->pair_rbound[ps][1];
            c.s10 = m->pair_size[ps][0]; c.s11 = m->pair_size[ps][1];
            c.s20 = m->pair_size[ps][3]; c.s21 = m->pair_size[ps][4]; c.s22 = m->pair_size[ps][5];
        };
        PairConst pc;
        if constexpr (request_early) request_pair(lane, pc);
        /* pass 1, lane = candidate pair (pair types that give at most two contacts) */
        int ncon = 0;
        const float *const env_hfield = io.hfield ? io.hfield + (size_t)env * io.hfield_stride : nullptr;
        /* block cull: pairs against static non-plane geoms (stairs ...) are visited only if one of those geoms is
         * within reach of a kinematic tree (lane = collision geom) */
        int npass = m->npair_always;
        if (m->npair_simple > m->npair_always) {
            bool nearby = false;
            if (lane < m->ngeom && m->geom_farstatic[lane]) {
                const double rb = m->geom_rbound[lane] + m->geom_margin[lane];
                for (int ri = 0; ri < m->nroot; ++ri) {
                    const int r = m->root_body[ri];
                    double dv[3] = {S.x.s.geom_xpos[lane][0] - S.x.s.xpos[r][0], S.x.s.geom_xpos[lane][1] - S.x.s.xpos[r][1],
                                    S.x.s.geom_xpos[lane][2] - S.x.s.xpos[r][2]};
                    const double bound = m->body_reach[r] + rb + 0.01;
                    if (dot3(dv, dv) < bound * bound) nearby = true;
                }
            }
            if (wv::ballot(nearby) != 0ull) npass = m->npair_simple;
        }
        CK_STAMP(21);
        /* Height-field pairs ahead of the pair loop: their sample spheres -- the sphere itself, or a capsule's two ends and
         * up to four interior samples -- go one to a lane (CM_HF_SLOTS lanes per pair, CM_HF_PASS pairs per pass), so the
         * walks over the grid cells under the samples run side by side instead of one after the other in the pair's lane;
         * the lanes of a pair then apply the capsule rule (oracle hfield_capsule) to the samples' results and the pair's
         * first lane parks the outcome -- contact count and up to two contacts -- in an LDS table laid over the velocity
         * tiles (unused until the velocity stage), where the pair loop below picks it up.  (Round 2 carried the outcome in
         * registers and fetched it with 40 cross-lane moves per pair-loop pass, and kept a second copy of the terrain walk
         * inside the pair loop for models with more pairs than one pass holds: both were what this instantiation spilled.) */
        constexpr int HF_REC = 1 + 10 * CM_HF_MAXC; /* doubles per pair: count, then up to CM_HF_MAXC x (dist, pos[3], normal[3], tangent[3]) */
        static_assert(CM_MAXHFPAIR * HF_REC <= NB * 12 + NVP * 12, "the height-field result table must fit the cvel + cfrc + cdof_dot + buf tiles");
        static_assert(offsetof(decltype(S.x.s), buf) - offsetof(decltype(S.x.s), cvel) == sizeof(double) * (NB * 12 + NVP * 6), "those four tiles are contiguous");
        static_assert(CM_HF_PASS * CM_HF_SLOTS <= WV_WAVE && CM_HF_SLOTS_DENSE <= WV_WAVE, "a pass of the height-field pre-pass is one wave");
        double *const hfres = &S.x.s.cvel[0][0];
        const bool hf_on = (FEAT & FEAT_HFIELD) != 0 && env_hfield != nullptr && m->nhfpair > 0;
        const bool hf_prism = (FEAT & FEAT_HFIELD) != 0 && hf_on && (m->flags & CM_FLAG_HFPRISM) != 0;
        int ncon_prism = 0;
        if constexpr ((FEAT & FEAT_HFIELD) != 0) if (hf_prism) {
            /* CM_FLAG_HFPRISM: one contact per penetrated grid triangle, straight into the contact list ahead of the other pairs' */
            static_assert(CM_MAXHFPAIR * HP_REC <= NB * 12 + NVP * 12, "the per-pair records of the prism pass fit the idle velocity tiles");
            CK_STAMP(44);
            ncon_prism = hfield_prism_wave(S, m, env_hfield, lane, hfres, MAXC);
            CK_STAMP(46);
        }
        if constexpr ((FEAT & FEAT_HFIELD) != 0) if (hf_on && !hf_prism) {
            /* CM_FLAG_HFDENSE: ten sample slots per pair (six pairs per pass) instead of six (ten pairs per pass) */
            const bool dense = (m->flags & CM_FLAG_HFDENSE) != 0;
            const int slots = dense ? CM_HF_SLOTS_DENSE : CM_HF_SLOTS, per_pass = dense ? WV_WAVE / CM_HF_SLOTS_DENSE : CM_HF_PASS;
            for (int h0 = 0; h0 < m->nhfpair; h0 += per_pass) {
                const int hl = dense ? lane / CM_HF_SLOTS_DENSE : lane / CM_HF_SLOTS;
                const int h = h0 + hl, k = lane - hl * slots;
                const bool act = hl < per_pass && h < m->nhfpair;
                const int p = m->hfpair[act ? h : 0];
                const int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p], t2 = m->pair_type[p] >> 8;
                const double margin = m->pair_margin[p], s20 = m->pair_size[p][3], s21 = m->pair_size[p][4];
                const double *p1 = S.x.s.geom_xpos[g1], *m1 = S.x.s.geom_xmat[g1], *p2 = S.x.s.geom_xpos[g2], *m2 = S.x.s.geom_xmat[g2];
                const double axis[3] = {m2[2], m2[5], m2[8]};
                const double cell = 2 * m->hfield_size[0] / (m->hfield_ncol > 1 ? m->hfield_ncol - 1 : 1);
                int ni = 0;
                bool mine = act && k == 0;
                double t = 0;
                if (t2 == CM_GEOM_CAPSULE) {
                    ni = (int)ceil(2 * s21 / cell) - 1;
                    if (ni < 0) ni = 0;
                    if (ni > slots - 2) ni = slots - 2;
                    mine = act && k < 2 + ni;
                    t = k == 0 ? s21 : (k == 1 ? -s21 : s21 * (1.0 - 2.0 * (k - 1) / (ni + 1)));
                }
                RawContact rcs;
                rcs.dist = 1e300;
                bool has = false;
                {
                    /* (all lanes: the wave shares the cells under the pass's samples; the contact tables are scratch until the
                     * pair loop below fills them) */
                    double e[3] = {p2[0] + t * axis[0], p2[1] + t * axis[1], p2[2] + t * axis[2]};
                    static_assert(offsetof(SH_T, c_margin) + sizeof(S.c_margin) - offsetof(SH_T, c_dist) >= HF_WINDOW + sizeof(double) * 4 * WV_WAVE,
                                  "the contact tables hold the work area of hfield_spheres_wave");
                    CK_STAMP(44);
                    has = hfield_spheres_wave(rcs, m, env_hfield, p1, m1, mine, e, s20, margin, lane, &S.c_dist[0]) != 0;
                    CK_STAMP(45);
                }
                /* the samples of this lane's pair: distances (1e300 = no contact) */
                const int lead = lane - k;
                double dk[CM_HF_SLOTS_DENSE];
                const double mydist = has ? rcs.dist : 1e300;
#pragma unroll
                for (int q = 0; q < CM_HF_SLOTS_DENSE; ++q) {
                    if (q >= CM_HF_SLOTS && !dense) { dk[q] = 1e300; continue; } /* (wave-uniform) */
                    const double v = wv::shfl(mydist, (lead + q) & 63);
                    dk[q] = q < slots ? v : 1e300;
                }
                int src0 = 0, src1 = 1;
                bool have0 = dk[0] < 1e299, have1 = dk[1] < 1e299;
                if (t2 == CM_GEOM_CAPSULE) {
                    int kmid = -1;
                    double dmid = 1e300;
#pragma unroll
                    for (int q = 2; q < CM_HF_SLOTS_DENSE; ++q) if (dk[q] < 1e299 && (kmid < 0 || dk[q] < dmid)) { kmid = q; dmid = dk[q]; }
                    if (kmid >= 0 && (!have0 || dmid < dk[0]) && (!have1 || dmid < dk[1])) {
                        const bool drop1 = !have0 ? false : (!have1 ? true : dk[0] <= dk[1]);
                        if (drop1) { src1 = kmid; have1 = true; } else { src0 = kmid; have0 = true; }
                    }
                } else {
                    have1 = false;
                }
                /* the lanes holding the chosen samples write them: first kept contact to record slot 0, second to slot 1 */
                int nkeep = (have0 ? 1 : 0) + (have1 ? 1 : 0);
                double *rec = hfres + (size_t)(act ? h : 0) * HF_REC;
                int slot_of_me = (have0 && k == src0) ? 0 : ((have1 && k == src1) ? (have0 ? 1 : 0) : -1);
                if ((m->flags & CM_FLAG_HFMULTI) && t2 == CM_GEOM_CAPSULE) {
                    /* CM_FLAG_HFMULTI: up to CM_HF_MAXC contacts, the deepest samples first (ties: lower sample index): a
                     * sample's record slot is its rank among the pair's samples */
                    int rank = 0, cnt = 0;
#pragma unroll
                    for (int q = 0; q < CM_HF_SLOTS_DENSE; ++q) {
                        const bool valid = dk[q] < 1e299;
                        cnt += valid ? 1 : 0;
                        rank += (valid && (dk[q] < mydist || (dk[q] == mydist && q < k))) ? 1 : 0;
                    }
                    nkeep = cnt < CM_HF_MAXC ? cnt : CM_HF_MAXC;
                    slot_of_me = (has && rank < CM_HF_MAXC) ? rank : -1;
                }
                if (act && k == 0) rec[0] = (double)nkeep;
                if (act && slot_of_me >= 0) {
                    double *c = rec + 1 + 10 * slot_of_me;
                    c[0] = rcs.dist;
                    for (int i = 0; i < 3; ++i) { c[1 + i] = rcs.pos[i]; c[4 + i] = rcs.normal[i]; c[7 + i] = t2 == CM_GEOM_CAPSULE ? axis[i] : 0.0; }
                }
            }
            wv::sync();
            CK_STAMP(46);
        }
        if constexpr (!request_early) request_pair(lane, pc);
        if constexpr ((FEAT & FEAT_HFIELD) != 0) ncon = ncon_prism; /* (the prism pass's contacts come first) */
        for (int p0 = 0; p0 < npass; p0 += WV_WAVE) {
            const int p = p0 + lane;
            int n = 0;
            RawContact rc0, rc1;
            bool from_spread = false;
            int hfs = -1;
            if constexpr ((FEAT & FEAT_HFIELD) != 0) {
                /* height-field pairs take their result from the table the pre-pass filled (no terrain bound: no contact) */
                const int slot = p < npass ? m->pair_hfslot[p] : -1;
                if (slot >= 0) {
                    from_spread = true;
                    if (hf_on && !hf_prism) {
                        const double *rec = hfres + (size_t)slot * HF_REC;
                        n = (int)rec[0];
                        hfs = slot;     /* (contacts 3 and 4 of a CM_FLAG_HFMULTI pair go from the table straight to the contact list) */
                        rc0.dist = rec[1]; rc1.dist = rec[11];
                        for (int i = 0; i < 3; ++i) {
                            rc0.pos[i] = rec[2 + i]; rc0.normal[i] = rec[5 + i]; rc0.tangent[i] = rec[8 + i];
                            rc1.pos[i] = rec[12 + i]; rc1.normal[i] = rec[15 + i]; rc1.tangent[i] = rec[18 + i];
                        }
                    }
                }
            }
            if (p < npass && !from_spread) {
                const int g1 = pc.g1, g2 = pc.g2, tt = pc.tt;
                const int t1 = tt & 255, t2 = tt >> 8;
                const double margin = pc.margin;
                const double *p1 = S.x.s.geom_xpos[g1], *p2 = S.x.s.geom_xpos[g2];
                const double *m1 = S.x.s.geom_xmat[g1], *m2 = S.x.s.geom_xmat[g2];
                const double rb1 = pc.rb1, rb2 = pc.rb2;
                /* (the sizes come with the pair's other constants, not behind the cull: a second trip to memory for the pairs
                 * that survive it costs the whole wave more than five reads that most lanes do not use) */
                const double s10 = pc.s10, s11 = pc.s11, s20 = pc.s20, s21 = pc.s21, s22 = pc.s22;
                double dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
                CK_STAMP(31);
                bool cull = false;
                if (rb1 > 0 && rb2 > 0) {
                    double bound = rb1 + rb2 + margin;
                    cull = dot3(dif, dif) > bound * bound;
                } else if (t1 == CM_GEOM_PLANE && rb2 > 0) {
                    double nn[3] = {m1[2], m1[5], m1[8]};
                    cull = dot3(dif, nn) > margin + rb2;
                }
                if (!cull) {
                    if (t1 == CM_GEOM_PLANE && t2 == CM_GEOM_SPHERE) {
                        n = plane_sphere(rc0, p1, m1, p2, s20, margin);
                    } else if (t1 == CM_GEOM_PLANE && t2 == CM_GEOM_CAPSULE) {
                        double axis[3] = {m2[2], m2[5], m2[8]};
                        RawContact tmp;
                        double e0[3] = {p2[0] + s21 * axis[0], p2[1] + s21 * axis[1], p2[2] + s21 * axis[2]};
                        if (plane_sphere(tmp, p1, m1, e0, s20, margin)) { rc0 = tmp; n = 1; }
                        double e1[3] = {p2[0] - s21 * axis[0], p2[1] - s21 * axis[1], p2[2] - s21 * axis[2]};
                        if (plane_sphere(tmp, p1, m1, e1, s20, margin)) { if (n == 0) rc0 = tmp; else rc1 = tmp; ++n; }
                        for (int i = 0; i < 3; ++i) { rc0.tangent[i] = axis[i]; rc1.tangent[i] = axis[i]; }
                    } else if (t1 == CM_GEOM_SPHERE && t2 == CM_GEOM_SPHERE) {
                        n = sphere_sphere(rc0, p1, s10, p2, s20, margin);
                    } else if (t1 == CM_GEOM_SPHERE && t2 == CM_GEOM_CAPSULE) {
                        double a2[3] = {m2[2], m2[5], m2[8]};
                        double d12[3] = {-dif[0], -dif[1], -dif[2]};
                        double x = clampd(dot3(a2, d12), -s21, s21);
                        double q2[3] = {p2[0] + a2[0] * x, p2[1] + a2[1] * x, p2[2] + a2[2] * x};
                        n = sphere_sphere(rc0, p1, s10, q2, s20, margin);
                    } else if (t1 == CM_GEOM_CAPSULE && t2 == CM_GEOM_CAPSULE) {
                        double a1[3] = {m1[2], m1[5], m1[8]}, a2[3] = {m2[2], m2[5], m2[8]};
                        double x1, x2;
                        segment_closest(p1, a1, s11, p2, a2, s21, x1, x2);
                        double q1[3] = {p1[0] + a1[0] * x1, p1[1] + a1[1] * x1, p1[2] + a1[2] * x1};
                        double q2[3] = {p2[0] + a2[0] * x2, p2[1] + a2[1] * x2, p2[2] + a2[2] * x2};
                        n = sphere_sphere(rc0, q1, s10, q2, s20, margin);
                    } else if (t1 == CM_GEOM_SPHERE && t2 == CM_GEOM_BOX) {
                        double sb[3] = {s20, s21, s22};
                        n = sphere_box(rc0, p1, s10, p2, m2, sb, margin);
                    } else if (t1 == CM_GEOM_CAPSULE && t2 == CM_GEOM_BOX) {
                        double sb[3] = {s20, s21, s22};
                        n = capsule_box(rc0, rc1, p1, m1, s10, s11, p2, m2, sb, margin);
                    } else {
                        warn |= WARN_UNSUPPORTED_PAIR;
                    }
                }
            }
            CK_STAMP(32);
            if (p0 + WV_WAVE < npass) request_pair(p + WV_WAVE, pc);
            /* ballot-compact in pair order */
            const unsigned long long m1b = wv::ballot(n >= 1), m2b = wv::ballot(n >= 2);
            const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
            int slot = ncon + wv::popc64(m1b & below) + wv::popc64(m2b & below);
            int more = 0;
            if constexpr ((FEAT & FEAT_HFIELD) != 0) {
                /* height-field pairs with CM_FLAG_HFMULTI can hold a third and a fourth contact */
                const unsigned long long m3b = wv::ballot(n >= 3), m4b = wv::ballot(n >= 4);
                slot += wv::popc64(m3b & below) + wv::popc64(m4b & below);
                more = wv::popc64(m3b) + wv::popc64(m4b);
                for (int extra = 2; extra < CM_HF_MAXC; ++extra)
                    if (n > extra && slot + extra < MAXC) {
                        const double *c = hfres + (size_t)hfs * HF_REC + 1 + 10 * extra;
                        RawContact rx;
                        rx.dist = c[0];
                        for (int i = 0; i < 3; ++i) { rx.pos[i] = c[1 + i]; rx.normal[i] = c[4 + i]; rx.tangent[i] = c[7 + i]; }
                        write_raw_contact(S, slot + extra, p, rx);
                    }
            }
            if (n >= 1 && slot < MAXC) write_raw_contact(S, slot, p, rc0);
            if (n >= 2 && slot + 1 < MAXC) write_raw_contact(S, slot + 1, p, rc1);
            ncon += wv::popc64(m1b) + wv::popc64(m2b) + more;
        }
        CK_STAMP(22);
        if constexpr (!request_early) request_behind_collision(lane);
        /* pass 2, one pair at a time with the whole wave: lane = feature (box corner / vertex), first four hits kept */
        if constexpr ((FEAT & FEAT_WAVEPAIRS) != 0) for (int p = m->npair_simple; p < m->npair; ++p) {
            const int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p], tt = m->pair_type[p];
            const int t1 = tt & 255, t2 = tt >> 8;
            const double margin = m->pair_margin[p];
            const double *p1 = S.x.s.geom_xpos[g1], *p2 = S.x.s.geom_xpos[g2];
            const double *m1 = S.x.s.geom_xmat[g1], *m2 = S.x.s.geom_xmat[g2];
            const double rb1 = m->pair_rbound[p][0], rb2 = m->pair_rbound[p][1];
            {
                double dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
                bool cull = false;
                if (rb1 > 0 && rb2 > 0) { const double bound = rb1 + rb2 + margin; cull = dot3(dif, dif) > bound * bound; }
                else if (t1 == CM_GEOM_PLANE && rb2 > 0) { double nn[3] = {m1[2], m1[5], m1[8]}; cull = dot3(dif, nn) > margin + rb2; }
                if (cull) continue; /* wave-uniform: poses come from LDS broadcasts */
            }
            bool hit = false;
            RawContact rc;
            if (t1 == CM_GEOM_PLANE && t2 == CM_GEOM_BOX) {
                if (lane < 8) {
                    const double sb0 = m->pair_size[p][3], sb1 = m->pair_size[p][4], sb2 = m->pair_size[p][5];
                    double nrm[3] = {m1[2], m1[5], m1[8]}, dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
                    const double dist = dot3(dif, nrm);
                    double v[3] = {(lane & 1) ? sb0 : -sb0, (lane & 2) ? sb1 : -sb1, (lane & 4) ? sb2 : -sb2}, corner[3];
                    mulmatvec3(corner, m2, v);
                    const double ld = dot3(nrm, corner);
                    if (!(dist + ld > margin || ld > 0)) {
                        hit = true;
                        rc.dist = dist + ld;
                        for (int k = 0; k < 3; ++k) { rc.normal[k] = nrm[k]; rc.tangent[k] = 0; rc.pos[k] = corner[k] + p2[k] - nrm[k] * 0.5 * rc.dist; }
                    }
                }
            } else if (t1 == CM_GEOM_BOX && t2 == CM_GEOM_BOX) {
                double s1[3] = {m->pair_size[p][0], m->pair_size[p][1], m->pair_size[p][2]};
                double s2[3] = {m->pair_size[p][3], m->pair_size[p][4], m->pair_size[p][5]};
                hit = box_box_lane(rc, lane, p1, m1, s1, p2, m2, s2, margin);
            } else {
                warn |= WARN_UNSUPPORTED_PAIR;
            }
            const unsigned long long hb = wv::ballot(hit);
            const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
            const int rank = wv::popc64(hb & below);
            if (hit && rank < 4 && ncon + rank < MAXC) write_raw_contact(S, ncon + rank, p, rc);
            const int nh = wv::popc64(hb);
            ncon += nh < 4 ? nh : 4;
        }
        const int ncon_found = ncon;
        if (ncon > capc) ncon = capc; /* (the warning bit: below, unless the substep is handed over) */
        wv::sync();
        finish_contacts(S, m, lane, ncon);
        /* joint limits: lane = joint evaluates its own violation (the ballots give the row slots in joint order below) */
        unsigned long long lob, hib;
        {
            bool lo = false, hi = false;
            wv::keep(pf_jlim); wv::keep(pf_jtype);
            if (lane < njnt && pf_jlim && (pf_jtype == CM_JNT_HINGE || pf_jtype == CM_JNT_SLIDE)) {
                const double q = S.qpos[pf_jqadr];
                lo = q - pf_jlo < pf_jmargin;
                hi = pf_jhi - q < pf_jmargin;
            }
            lob = wv::ballot(lo); hib = wv::ballot(hi);
        }
        if constexpr (MAXR < WIDE_ROWS) {
            /* an upper bound of the rows this substep needs (the caps of the row assignment can only lower it): past this
             * instantiation's capacity -- rows or contacts -- the env is handed to the next instantiation, from the start of this substep */
            if (can_hand_over) {
                const int neq = m->neq;
                const bool eact = lane < neq && m->eq_active[lane < neq ? lane : 0] != 0;
                const int need = 3 * wv::popc64(wv::ballot(eact)) + wv::popc64(lob) + wv::popc64(hib) + 4 * ncon;
                if (need > capr || ncon_found > capc) {
                    bailed = true;
                    if constexpr (NW == 2) { if (lane == 0) S.cmd[0] = 1; wv::publish(&S.cmd[4], sub + 1); } /* (the verdict: handed over) */
                    break;
                }
            }
        }
        if (ncon_found > capc) warn |= WARN_CONTACT_FULL;
        if constexpr (NW == 1) if (io.drive_mode) {
            if (io.integrate) drive_level_io(io, S, m, env, lane, lastsub); /* mj_forward leaves the drive-level state alone */
            wv::sync();
        }
        if constexpr (NW == 2) { CK_STAMP(33); wv::publish(&S.cmd[4], sub + 1); wv::wait_for(&S.cmd[3], sub + 1); } /* the verdict for wave 1; wave 1's com / cinert / cdof for the stage below */
        CK_STAMP(5);

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
        const double kdamp = m->dof_damping[kd], kstiff = m->dof_stiffness[kd], kref = m->dof_springref[kd];
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

        /* ================= P5 constraint rows: lane = row ================= */
        /* row descriptor assignment is wave-uniform bookkeeping; every lane keeps its own row */
        /* (The 127-row instantiation forms its rows in TWO PASSES of this wave -- rows 0 .. 63, then, if the substep has them, rows
         * 64 .. 126 -- and parks every row's raw Jacobian row and solver parameters in LDS (x.Yr, x.rowt); behind the barrier J each
         * wave picks up its 64 rows from there: wide_solve.  Every other instantiation runs the loop below once.) */
        int r_ = lane;
        int rtype = -1, rid = 0, rsub = 0;
        int nefc = 0, nefc_before_contacts = 0;
        double rpos = 0, rmargin = 0, rR = 1.0, rK = 0, rB = 0, rimp = 1.0;
        const bool need_imu = lastsub || io.all_outputs_every_substep || (io.drive_mode && sub + 2 == nsub);
        const bool need_pos = lastsub || io.drive_mode || io.all_outputs_every_substep;
        const bool issens = lane < m->nsensor && need_pos;
        const int ls = issens ? lane : 0;
        SensorConsts sens_c;
        double ycol[NVP]; /* this lane's Jacobian row, one dof at a time, straight into registers; lane 63 carries qfrc_smooth */
        double jvel = 0, jws = 0;
        const bool lastcol = !WIDE && lane == NROW - 1;
        for (int pass = 0; pass < (WIDE ? 2 : 1); ++pass) {
        if constexpr (WIDE) {
            if (pass == 1 && nefc <= NROW) break; /* (wave-uniform: no second pass) */
            r_ = NROW * pass + lane; rtype = -1; rid = 0; rsub = 0; nefc = 0; jvel = 0; jws = 0;
            rpos = 0; rmargin = 0; rR = 1.0; rK = 0; rB = 0; rimp = 1.0;
        }
        {
            /* all equalities active and within the cap (the usual case): three rows each, in order, in closed form */
            const int neq = m->neq;
            const bool eact = lane < neq && m->eq_active[lane < neq ? lane : 0] != 0;
            const unsigned long long eall = neq >= 64 ? ~0ull : (1ull << neq) - 1;
            if (wv::ballot(eact) == eall && 3 * neq <= capr) {
                if (r_ < 3 * neq) { rtype = CM_CNSTR_EQUALITY; rid = r_ / 3; rsub = r_ - 3 * rid; }
                nefc = 3 * neq;
            } else
            for (int e = 0; e < neq; ++e) {
                if (!m->eq_active[e]) continue;
                if (nefc + 3 > capr) { warn |= WARN_CONSTRAINT_FULL; continue; }
                if (r_ >= nefc && r_ < nefc + 3) { rtype = CM_CNSTR_EQUALITY; rid = e; rsub = r_ - nefc; }
                nefc += 3;
            }
        }
        {
            /* joint-limit rows in joint order, lower side before upper side (lob / hib: evaluated after the collision stage) */
            for (unsigned long long any = lob | hib; any; any &= any - 1) {
                const int j = wv::popc64((any & (0ull - any)) - 1);
                for (int side = 0; side < 2; ++side) {
                    if (!(((side == 0 ? lob : hib) >> j) & 1ull)) continue;
                    if (nefc >= capr) { warn |= WARN_CONSTRAINT_FULL; continue; }
                    if (r_ == nefc) { rtype = CM_CNSTR_LIMIT_JOINT; rid = j; rsub = side; }
                    ++nefc;
                }
            }
        }
        nefc_before_contacts = nefc;
        /* every contact a friction pyramid of four rows and all of them within the cap (the usual case): closed form */
        const bool cpyr = lane < ncon && S.c_dim[lane < ncon ? lane : 0] == 3;
        if (wv::ballot(cpyr) == (ncon >= 64 ? ~0ull : (1ull << ncon) - 1) && nefc + 4 * ncon <= capr) {
            if (r_ >= nefc && r_ < nefc + 4 * ncon) { rtype = CM_CNSTR_CONTACT_PYRAMIDAL; rid = (r_ - nefc) >> 2; rsub = (r_ - nefc) & 3; }
            nefc += 4 * ncon;
        } else
        for (int c = 0; c < ncon; ++c) {
            const int dim = S.c_dim[c];
            if (dim != 1 && dim != 3) { warn |= WARN_UNSUPPORTED_PAIR; continue; }
            const int nrow = dim == 1 ? 1 : 2 * (dim - 1);
            if (nefc + nrow > capr) { warn |= WARN_CONSTRAINT_FULL; continue; }
            if (r_ >= nefc && r_ < nefc + nrow) {
                rtype = dim == 1 ? CM_CNSTR_CONTACT_FRICTIONLESS : CM_CNSTR_CONTACT_PYRAMIDAL;
                rid = c; rsub = r_ - nefc;
            }
            nefc += nrow;
        }

        CK_STAMP(25);
        /* per-row geometry: J_rk = plus_k (u.lin_k + wp.ang_k) - minus_k (u.lin_k + wm.ang_k) (+ sgn at one dof) */
        double u3[3] = {0, 0, 0}, wp[3] = {0, 0, 0}, wm[3] = {0, 0, 0};
        unsigned long long maskp = 0, maskm = 0;
        int limdof = -1;
        double limsgn = 0, rdiag = 0, imp_pos = 0, rRscale = 1.0;
        double solref0 = 0.02, solref1 = 1, solimp[5] = {0.9, 0.95, 0.001, 0.5, 2};
        if (rtype == CM_CNSTR_EQUALITY) {
            if (rid != pf_eq) { /* (an inactive equality ahead of this one: the constants requested in advance are another row's) */
                pf_eb1 = m->eq_body1[rid]; pf_eb2 = m->eq_body2[rid]; pf_er1 = m->eq_root[rid][0]; pf_er2 = m->eq_root[rid][1];
            }
            maskp = m->eq_dofmask[rid][0]; maskm = m->eq_dofmask[rid][1];
            rdiag = m->eq_invweight[rid];
            solref0 = m->eq_solref[rid][0]; solref1 = m->eq_solref[rid][1];
            for (int i = 0; i < 5; ++i) solimp[i] = m->eq_solimp[rid][i];
            const int b1 = pf_eb1, b2 = pf_eb2;
            double l1[3] = {m->eq_data[rid][0], m->eq_data[rid][1], m->eq_data[rid][2]};
            double l2[3] = {m->eq_data[rid][3], m->eq_data[rid][4], m->eq_data[rid][5]};
            double a1[3], a2[3];
            mulmatvec3(a1, S.x.s.xmat[b1], l1);
            mulmatvec3(a2, S.x.s.xmat[b2], l2);
            for (int i = 0; i < 3; ++i) { a1[i] += S.x.s.xpos[b1][i]; a2[i] += S.x.s.xpos[b2][i]; }
            u3[0] = rsub == 0 ? 1.0 : 0.0; u3[1] = rsub == 1 ? 1.0 : 0.0; u3[2] = rsub == 2 ? 1.0 : 0.0;
            const double *c1 = S.com[pf_er1], *c2 = S.com[pf_er2];
            double o1[3] = {a1[0] - c1[0], a1[1] - c1[1], a1[2] - c1[2]};
            double o2[3] = {a2[0] - c2[0], a2[1] - c2[1], a2[2] - c2[2]};
            cross3(wp, o1, u3);
            cross3(wm, o2, u3);

            double res[3] = {a1[0] - a2[0], a1[1] - a2[1], a1[2] - a2[2]};
            rpos = rsub == 0 ? res[0] : (rsub == 1 ? res[1] : res[2]); rmargin = 0;
            imp_pos = sqrt(dot3(res, res));

        } else if (rtype == CM_CNSTR_LIMIT_JOINT) {
            const double q = S.qpos[m->jnt_qposadr[rid]];
            rpos = rsub == 0 ? q - m->jnt_range[rid][0] : m->jnt_range[rid][1] - q;
            rmargin = m->jnt_margin[rid];
            imp_pos = rpos;
            limdof = m->jnt_dofadr[rid];
            limsgn = rsub == 0 ? 1.0 : -1.0;
            rdiag = m->jnt_liminvweight[rid];
            solref0 = m->jnt_solref[rid][0]; solref1 = m->jnt_solref[rid][1];
            for (int i = 0; i < 5; ++i) solimp[i] = m->jnt_solimp[rid][i];
        } else if (rtype >= 0) {
            const int c = rid;
            const double *fr = S.c_frame[c];
            if (rtype == CM_CNSTR_CONTACT_PYRAMIDAL) {
                const int a = 1 + rsub / 2;
                const double mu = a <= 2 ? S.c_fri[c][0] : (a == 3 ? S.c_fri[c][1] : S.c_fri[c][2]);
                const double sg = (rsub & 1) ? -mu : mu;
                for (int i = 0; i < 3; ++i) u3[i] = fr[i] + sg * fr[3 * a + i];
                const double mu0 = S.c_fri[c][0];
                rRscale = 2 * mu0 * mu0; /* all pyramid rows use R of the first row, times 2 mu^2 */
            } else {
                for (int i = 0; i < 3; ++i) u3[i] = fr[i];
            }
            const double *c1 = S.com[S.c_root[c][0]], *c2 = S.com[S.c_root[c][1]];
            double o1[3] = {S.c_pos[c][0] - c1[0], S.c_pos[c][1] - c1[1], S.c_pos[c][2] - c1[2]};
            double o2[3] = {S.c_pos[c][0] - c2[0], S.c_pos[c][1] - c2[1], S.c_pos[c][2] - c2[2]};
            cross3(wp, o2, u3);
            cross3(wm, o1, u3);
            maskp = S.c_dofmask[c][1];
            maskm = S.c_dofmask[c][0];
            rpos = S.c_dist[c]; rmargin = S.c_margin[c]; imp_pos = rpos;
            const double tran = S.c_tran[c];
            const double mu_first = S.c_fri[c][0];
            rdiag = rtype == CM_CNSTR_CONTACT_PYRAMIDAL ? tran + mu_first * mu_first * tran : tran;
            solref0 = S.c_solref[c][0]; solref1 = S.c_solref[c][1];
            for (int i = 0; i < 5; ++i) solimp[i] = S.c_solimp[c][i];
        }
        CK_STAMP(26);
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
        CK_STAMP(27);
        /* The constants of the sensor stage behind the Jacobian loop are requested here, every one of them in one level of
         * unconditional reads (lane = sensor): their round trip through the memory system runs under the loop instead of
         * in front of the sensors. */
        if constexpr (NW == 1) sens_c = request_sensor_consts(m, ls);
#pragma unroll
        for (int k0 = 0; k0 < NVP; k0 += 4) {
            /* stage four motion axes and the matching qvel / qacc_warmstart / qfrc_smooth entries, then compute */
            double cc[4][6], qv[4], qw[4], qs[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int t = 0; t < 6; ++t) cc[kk][t] = S.cdof[k0 + kk][t];
                qv[kk] = S.qvel[k0 + kk]; qw[kk] = S.qacc_ws[k0 + kk];
                if constexpr (NW == 1) qs[kk] = S.qfrc_smooth[k0 + kk]; else qs[kk] = 0.0; /* (two-wave form: behind the barrier J) */
            }
            wv::sched_fence();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int k = k0 + kk;
                double v = 0;
                if (TOPO::is_static ? k < TOPO::nv : k < nv) {
                    if (rtype >= 0) {
                        const double ul = u3[0] * cc[kk][3] + u3[1] * cc[kk][4] + u3[2] * cc[kk][5];
                        /* the two body-chain predicates applied by multiplication (exact: the factors are 0.0 / 1.0) */
                        const double sp = bitf(maskp, k), sm = bitf(maskm, k);
                        v = sp * (ul + wp[0] * cc[kk][0] + wp[1] * cc[kk][1] + wp[2] * cc[kk][2]) -
                            sm * (ul + wm[0] * cc[kk][0] + wm[1] * cc[kk][1] + wm[2] * cc[kk][2]);
                        if (k == limdof) v = limsgn;
                        jvel += v * qv[kk];
                        jws += v * qw[kk];
                    } else if (lastcol) {
                        v = qs[kk];
                    }
                }
                ycol[k] = v;
            }
        }
        if constexpr (WIDE) {
            /* the row goes to LDS: its raw Jacobian row (the half solve behind the barrier J takes it from there) and what the solve
             * needs of it -- R, the reference acceleration, J . qacc_warmstart, and 1 / 0 / -1 for clamped row / equality row / no row */
            if (r_ < MAXR) {
#pragma unroll
                for (int k = 0; k < NVP; ++k) S.x.Yr[r_][k] = ycol[k];
                double *rt = S.x.rowt[r_];
                rt[0] = rR; rt[1] = rtype >= 0 ? -rB * jvel - rK * rimp * (rpos - rmargin) : 0.0; rt[2] = jws;
                rt[3] = rtype < 0 ? -1.0 : (rtype == CM_CNSTR_EQUALITY ? 0.0 : 1.0);
            }
            if (io.ext && pass == 0 && rtype == CM_CNSTR_EQUALITY && r_ < CM_MAXEQROW) {
                cm_ext_t *ex = io.ext + env;
                ex->eq_pos[r_] = rpos; ex->eq_id[r_] = rid;
#pragma unroll
                for (int k = 0; k < NVP; ++k) if (k < nv) ex->eq_J[r_][k] = ycol[k];
            }
        }
        } /* (pass) */
        const int sb = NW == 1 ? sens_c.sb : 0;
        CK_STAMP(28);
        if constexpr (NW == 2) {
            /* J: wave 1 is done with this substep -- the factors of M and M + hB and qfrc_smooth are in LDS, and its drive-level pass
             * has read the previous substep's sensor words, which the sensor stage below replaces */
            if constexpr (WIDE) { if (lane == 0) S.x.turn[0] = nefc; } /* (wave 1 takes its rows by it) */
            CK_STAMP(34); wv::block_barrier();
            /* lane 63's column of the staged matrix is qfrc_smooth */
            if (lastcol) {
#pragma unroll
                for (int k = 0; k < NVP; ++k) ycol[k] = (TOPO::is_static ? k < TOPO::nv : k < nv) ? S.qfrc_smooth[k] : 0.0;
            }
        }
        const double raref = rtype >= 0 ? -rB * jvel - rK * rimp * (rpos - rmargin) : 0.0;

        /* ---- sensors, part 1 (sensors_before_solve; two-wave form: wave 1 runs it behind the barrier J) ---- */
        int aslot = -1;
        if constexpr (NW == 1) aslot = sensors_before_solve(io, S, m, env, issens, ls, sens_c, need_imu, lastsub);
        CK_STAMP(29);
        if (io.xpos_out && isbody && lastsub) {
            for (int i = 0; i < 3; ++i) io.xpos_out[((size_t)env * io.sb + b) * 3 + i] = S.x.s.xpos[b][i];
            if (io.xquat_out) for (int i = 0; i < 4; ++i) io.xquat_out[((size_t)env * io.sb + b) * 4 + i] = S.x.s.xquat[b][i];
        }
        if (io.ext) {
            cm_ext_t *ex = io.ext + env;
            if (isbody) for (int i = 0; i < 6; ++i) ex->cvel[b][i] = S.x.s.cvel[b][i];
            if (isbody) for (int i = 0; i < 3; ++i) { ex->subtree_com[b][i] = S.com[broot > 0 ? broot : 0][i]; ex->xipos[b][i] = S.x.s.xipos[b][i]; }
            if (lane < m->nsite) {
                const int sb = m->site_bodyid[lane];
                double sp[3] = {m->site_pos[lane][0], m->site_pos[lane][1], m->site_pos[lane][2]};
                double sq[4] = {m->site_quat[lane][0], m->site_quat[lane][1], m->site_quat[lane][2], m->site_quat[lane][3]};
                double t[3], q[4], mm[9];
                mulmatvec3(t, S.x.s.xmat[sb], sp);
                for (int i = 0; i < 3; ++i) ex->site_xpos[lane][i] = t[i] + S.x.s.xpos[sb][i];
                mulquat(q, S.x.s.xquat[sb], sq);
                quat2mat(mm, q);
                for (int i = 0; i < 9; ++i) ex->site_xmat[lane][i] = mm[i];
            }
            if (isdof) for (int i = 0; i < 6; ++i) { ex->cdof[k_][i] = S.cdof[k_][i]; ex->cdof_dot[k_][i] = S.x.s.cdof_dot[k_][i]; }
            if (!WIDE && rtype == CM_CNSTR_EQUALITY && r_ < CM_MAXEQROW) {
                ex->eq_pos[r_] = rpos; ex->eq_id[r_] = rid;
#pragma unroll
                for (int k = 0; k < NVP; ++k) if (k < nv) ex->eq_J[r_][k] = ycol[k];
            }
            if (lane == 0) { int ne = 0; for (int e = 0; e < m->neq; ++e) if (m->eq_active[e] && ne + 3 <= capr) ne += 3; ex->ne = ne; }
            if (lane < ncon) {
                ex->con_geom1[lane] = m->geom_fullid[S.c_g1[lane]]; ex->con_geom2[lane] = m->geom_fullid[S.c_g2[lane]];
                ex->con_body1[lane] = m->geom_bodyid[S.c_g1[lane]]; ex->con_body2[lane] = m->geom_bodyid[S.c_g2[lane]];
                ex->con_dim[lane] = S.c_dim[lane]; ex->con_dist[lane] = S.c_dist[lane];
                for (int i = 0; i < 3; ++i) ex->con_pos[lane][i] = S.c_pos[lane][i];
                for (int i = 0; i < 9; ++i) ex->con_frame[lane][i] = S.c_frame[lane][i];
            }
        }
        /* first constraint row of every contact (lane = contact), for the contact-force read-out */
        int caddr = -1;
        const bool want_cfrc = io.body_cfrc && (sub == nsub - 1 || !io.integrate); /* read out by the last substep only */
        if (io.ext || want_cfrc) {
            int acc = nefc_before_contacts;
            for (int c = 0; c < ncon; ++c) {
                const int dim = S.c_dim[c];
                if (dim != 1 && dim != 3) continue;
                const int nrow = dim == 1 ? 1 : 2 * (dim - 1);
                if (acc + nrow > capr) continue;
                if (c == lane) caddr = acc;
                acc += nrow;
            }
        }
        wv::sync(); /* every reader of the body-stage tiles is done: region x becomes the Y staging tile */
        CK_STAMP(8);

        if constexpr (WIDE) {
            /* the 127-row instantiation: half solve, A and the sweeps for this wave's rows 0 .. 63 (wide_solve; wave 1 is in the same
             * function for rows 64 .. 126 and the qfrc_smooth column), the row forces of both waves through LDS at the barrier P,
             * the read-outs from there */
            int iters = 0, nguarded = 0;
            const double f = wide_solve<NVP, TOPO>(S, m, 0, nefc, sub, iters, nguarded);
            CK_STAMP(11);
            static_assert(sizeof(S.c_solimp) >= (2 * NROW + 4) * sizeof(double), "the row forces are handed over through the contacts' solimp slots");
            double *const fbuf = &S.c_solimp[0][0];
            fbuf[lane] = f;
            if (lane == 0) { fbuf[2 * NROW] = (double)ncon; fbuf[2 * NROW + 1] = (double)nefc; fbuf[2 * NROW + 2] = (double)iters; fbuf[2 * NROW + 3] = (double)nguarded; }
            wv::block_barrier(); /* P */
            if (io.ext || want_cfrc) {
                /* decode the pyramid: normal = sum of the edge forces, tangents = mu (f+ - f-); the rows' forces from LDS */
                const int a0 = caddr >= 0 ? caddr : 0;
                double fn = 0, ft1 = 0, ft2 = 0;
                if (lane < ncon && caddr >= 0) {
                    const double f0 = fbuf[a0];
                    if (S.c_dim[lane] == 1) fn = f0;
                    else { const double f1 = fbuf[a0 + 1], f2 = fbuf[a0 + 2], f3 = fbuf[a0 + 3], mu = S.c_fri[lane][0]; fn = f0 + f1 + f2 + f3; ft1 = mu * (f0 - f1); ft2 = mu * (f2 - f3); }
                }
                if (io.ext) {
                    cm_ext_t *ex = io.ext + env;
                    if (lane < ncon) { ex->con_force[lane][0] = fn; ex->con_force[lane][1] = ft1; ex->con_force[lane][2] = ft2; }
                    if (lane == 0) { ex->ncon = ncon; ex->nefc = nefc; ex->solver_iter = iters; }
                }
                if (want_cfrc) {
                    if (lane < ncon) {
                        const double *fr = S.c_frame[lane];
                        double fw[3];
                        for (int j = 0; j < 3; ++j) fw[j] = fr[j] * fn + fr[3 + j] * ft1 + fr[6 + j] * ft2;
                        for (int j = 0; j < 3; ++j) S.c_pos[lane][j] = fw[j];
                    }
                    wv::sync();
                    if (isbody) {
                        double acc[3] = {0, 0, 0};
                        for (int c = 0; c < ncon; ++c) {
                            const int b1 = m->geom_bodyid[S.c_g1[c]], b2 = m->geom_bodyid[S.c_g2[c]];
                            const double sg = (b2 == b ? 1.0 : 0.0) - (b1 == b ? 1.0 : 0.0);
                            for (int j = 0; j < 3; ++j) acc[j] += sg * S.c_pos[c][j];
                        }
                        for (int j = 0; j < 3; ++j) io.body_cfrc[((size_t)env * io.sb + b) * 3 + j] = acc[j];
                    }
                    wv::sync();
                }
            }
            wv::block_barrier(); /* E */
            CK_STAMP(37);
            if (wv::opaque(S.cmd[0])) { warn |= WARN_DIVERGED; break; }
            if (!io.integrate) break;
            time += h;
            continue;
        }
        /* ================= half solve in registers: Y = D^-1/2 L^-T [J^T | qfrc_smooth], lane = column ================= */
        if constexpr (TOPO::is_static) {
            /* The multipliers do not depend on the solve, so dof k - 1's row of L is fetched (LDS broadcast reads) while
             * dof k's updates run; the fences keep the scheduler from re-pairing every read with its own wait. */
            double ta[NVP], tb[NVP], ra = 0, rb = 0;
            auto fetch = [&](int k, double (&t)[NVP], double &rs) {
#pragma unroll
                for (int i = k - 1; i >= 0; --i) if ((TOPO::table[k] >> i) & 1ull) t[i] = S.Lp[LPack<TOPO, NVP>::idx(k, i)];
                rs = S.rsd[k];
            };
            fetch(TOPO::nv - 1, ((TOPO::nv - 1) & 1) ? ta : tb, ((TOPO::nv - 1) & 1) ? ra : rb);
#pragma unroll
            for (int k = NVP - 1; k >= 0; --k) {
                if (k >= TOPO::nv) continue;
                if (k > 0) fetch(k - 1, ((k - 1) & 1) ? ta : tb, ((k - 1) & 1) ? ra : rb);
                wv::sched_fence();
                const double (&t)[NVP] = (k & 1) ? ta : tb;
                const double xk = ycol[k];
#pragma unroll
                for (int i = k - 1; i >= 0; --i) if ((TOPO::table[k] >> i) & 1ull) ycol[i] -= t[i] * xk;
                ycol[k] = xk * ((k & 1) ? ra : rb);
                wv::sched_fence();
            }
        } else {
#pragma unroll
            for (int k = NVP - 1; k >= 0; --k) {
                if (k >= nv) continue;
                const unsigned long long anc = m->dof_ancmask[k];
                const double xk = ycol[k];
#pragma unroll
                for (int i = k - 1; i >= 0; --i) {
                    if (!((anc >> i) & 1ull)) continue;
                    ycol[i] -= S.Lp[CK_TRI(k, i)] * xk;
                }
                ycol[k] = xk * S.rsd[k];
            }
        }
        if (r_ < MAXR || lastcol) { /* (lanes MAXR .. 62 of a row-capped instantiation hold no row) */
            const int yrow = lastcol ? MAXR : r_;
#pragma unroll
            for (int k = 0; k < NVP; ++k) S.x.Yr[yrow][k] = ycol[k];
        }
        wv::sync();
        if constexpr (NW == 2) wv::publish(&S.cmd[2], sub + 1); /* wave 1 stages its column of Y for the qacc stage while this wave solves */
        CK_STAMP(9);

        /* ================= P9: this lane's row of A = Y^T Y + diag(R), b = Y^T y63 - aref ================= */
        double arow[MAXR];
        double diag_yy = 0; /* this lane's row of Y with itself */
        double rb = 0;
        /* Every product of two staged rows is ONE fused multiply-add chain over the dofs in index order, started from zero --
         * in both instantiations, so that their results agree bit for bit:
         *   - the row-capped instantiation (at most 31 rows + the qfrc_smooth row = a 32 x 32 Gram matrix of the staged tile)
         *     forms it on the matrix core: v_mfma_f64_16x16x4_f64 is exactly that chain per element (wave.h),
         *     three 16 x 16 tiles (the matrix is symmetric) x NVP / 4 blocks of dofs, operands straight from the staged
         *     tile, 64 clocks each with nothing else on the wave's critical path; the tiles go through LDS (the part of the
         *     body-stage region behind the staged tile) so that every lane ends up with its own row in registers;
         *   - the full instantiation forms the same chains on the vector unit, two rows at a time. */
        constexpr bool gram_on_matrix_core = MAXR == 31 && NVP % 4 == 0;
        constexpr bool gram_in_place = MAXR == 47 && NVP == 40; /* (the same on the matrix core, through the staged tile's own LDS: see there) */
        if constexpr (gram_on_matrix_core) {
            constexpr int YP = EnvShared<NVP, LPack<TOPO, NVP>::count, MAXR>::YP;
            static_assert(2 * (MAXR + 1) * YP * sizeof(double) <= sizeof(S.x), "the Gram matrix is parked behind the staged tile");
            double (*Am)[YP] = (double (*)[YP])(&S.x.Yr[0][0] + (MAXR + 1) * YP);
            const int mi = lane & 15, mk = lane >> 4;
            double y0[NVP / 4], y1[NVP / 4];
#pragma unroll
            for (int kb = 0; kb < NVP / 4; ++kb) { y0[kb] = S.x.Yr[mi][4 * kb + mk]; y1[kb] = S.x.Yr[16 + mi][4 * kb + mk]; }
            wv::mfma_acc t00 = {{0, 0, 0, 0}}, t01 = {{0, 0, 0, 0}}, t11 = {{0, 0, 0, 0}};
#pragma unroll
            for (int kb = 0; kb < NVP / 4; ++kb) wv::mfma_f64_16x16x4_x3(y0[kb], y0[kb], t00, y0[kb], y1[kb], t01, y1[kb], y1[kb], t11);
            wv::mfma_f64_drain(t00, t01, t11);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int row = mk + 4 * v;
                Am[row][mi] = t00.c[v];
                Am[row][16 + mi] = t01.c[v];
                Am[16 + mi][row] = t01.c[v];
                Am[16 + row][16 + mi] = t11.c[v];
            }
            wv::sync();
            const int myrow = lane & 31; /* (lanes 32 .. 63 hold no row: they read a row's values and never use them) */
#pragma unroll
            for (int t = 0; t < MAXR; ++t) arow[t] = Am[myrow][t];
            rb = Am[myrow][MAXR] - raref;
            /* (the register allocator's plan for the whole kernel depends on this lane's column of Y staying in registers up to
             * here -- it did when the vector unit formed the products; released at the staging store, 600 values go to
             * scratch all over the kernel.  No instruction is emitted.) */
#pragma unroll
            for (int k = 0; k < NVP; ++k) wv::touch(ycol[k]);
            diag_yy = Am[myrow][myrow];
        } else if constexpr (gram_in_place) {
            /* The 40-dof model's row-capped instantiation, one wave per env (512 registers): the 48 x 48 Gram matrix of the staged
             * tile (47 rows + the qfrc_smooth row) on the matrix core -- six 16 x 16 tiles x NVP / 4 blocks of dofs, the three
             * tiles of rows 32 .. 47 only when the substep has that many rows.  There is no LDS behind the staged tile to turn the
             * tiles into one row per lane, so it happens IN the region of the staged tile: every staged value is an operand of the
             * products and sits in a register by then (row 16 I + (lane & 15), dofs 4 kb + (lane >> 4)), the tiles overwrite the
             * staged tile, every lane takes its row of A, and the lanes store their rows of the tile once more (they still
             * hold them). */
            constexpr int AP = 49, AR = MAXR + 1; /* A at a leading dimension that keeps rows and columns off each other's banks */
            static_assert(AR == 48, "three blocks of sixteen staged rows");
            static_assert(AR * AP * sizeof(double) <= sizeof(S.x), "the matrix fits the region of the staged tile");
            double (*Am)[AP] = (double (*)[AP])&S.x.Yr[0][0];
            const int mi = lane & 15, mk = lane >> 4;
            double y0[NVP / 4], y1[NVP / 4], y2[NVP / 4];
#pragma unroll
            for (int kb = 0; kb < NVP / 4; ++kb) { y0[kb] = S.x.Yr[mi][4 * kb + mk]; y1[kb] = S.x.Yr[16 + mi][4 * kb + mk]; y2[kb] = S.x.Yr[32 + mi][4 * kb + mk]; }
            wv::mfma_acc t00 = {{0, 0, 0, 0}}, t01 = {{0, 0, 0, 0}}, t11 = {{0, 0, 0, 0}}, t02 = {{0, 0, 0, 0}}, t12 = {{0, 0, 0, 0}}, t22 = {{0, 0, 0, 0}};
#pragma unroll
            for (int kb = 0; kb < NVP / 4; ++kb) wv::mfma_f64_16x16x4_x3(y0[kb], y2[kb], t02, y1[kb], y2[kb], t12, y2[kb], y2[kb], t22); /* (row 47: always) */
#pragma unroll
            for (int kb = 0; kb < NVP / 4; ++kb) wv::mfma_f64_16x16x4_x3(y0[kb], y0[kb], t00, y0[kb], y1[kb], t01, y1[kb], y1[kb], t11);
            wv::mfma_f64_drain(t00, t01, t11);
            wv::mfma_f64_drain(t02, t12, t22);
            wv::sync(); /* every operand is in registers: the tile's rows may go */
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int row = mk + 4 * v; /* (a lane's four values of a tile: rows mk + 4 v of column mi, as wave.h lays the result out) */
                Am[row][mi] = t00.c[v];
                Am[row][16 + mi] = t01.c[v]; Am[16 + mi][row] = t01.c[v];
                Am[16 + row][16 + mi] = t11.c[v];
                Am[row][32 + mi] = t02.c[v]; Am[32 + mi][row] = t02.c[v];
                Am[16 + row][32 + mi] = t12.c[v]; Am[32 + mi][16 + row] = t12.c[v];
                Am[32 + row][32 + mi] = t22.c[v];
            }
            wv::sync();
            const int myrow = lane < AR ? lane : AR - 1; /* (lanes past the staged rows hold no row: they take one and never use it) */
            rb = Am[myrow][MAXR] - raref;
            diag_yy = Am[myrow][myrow];
            /* every lane takes its row of A, and the lanes store their rows of the tile once more (they still hold them).  (Leaving A
             * in LDS through the sweeps instead -- one ds_read_b64 per row of the chain, requested eight rows ahead -- frees the 94
             * registers of the row, but the chain is slower by more than the spills cost: round 5, 15.8 against 17.8 M in the two-wave
             * form, 13.7 against 15.6 M with one wave.) */
#pragma unroll
            for (int t = 0; t < MAXR; ++t) arow[t] = Am[myrow][t];
            wv::sync();
            if (r_ < MAXR || lastcol) {
                const int yrow = lastcol ? MAXR : r_;
#pragma unroll
                for (int k = 0; k < NVP; ++k) S.x.Yr[yrow][k] = ycol[k];
            }
            wv::sync();
        }
        else if constexpr (NW == 2 && NVP > 32) {
            /* the 40-dof instantiation at the 256 registers of the two-wave form: one staged row at a time, half a row in flight
             * (the row-pair loop below keeps two staged rows beside this lane's column and its row of A: 330 registers); the same
             * chain per product */
#pragma unroll
            for (int r = 0; r < MAXR; ++r) {
                double acc = 0;
                if (r < nefc) {
#pragma unroll
                    for (int k0 = 0; k0 < NVP; k0 += NVP / 2) {
                        double ya[NVP / 2];
#pragma unroll
                        for (int k = 0; k < NVP / 2; ++k) ya[k] = S.x.Yr[r][k0 + k];
#pragma unroll
                        for (int k = 0; k < NVP / 2; ++k) acc = fma(ya[k], ycol[k0 + k], acc);
                    }
                }
                arow[r] = acc;
            }
            {
                double acc = 0;
#pragma unroll
                for (int k = 0; k < NVP; ++k) acc = fma(S.x.Yr[MAXR][k], ycol[k], acc);
                rb = acc - raref;
            }
        } else {
#pragma unroll
        for (int r = 0; r < MAXR; r += 2) {
            /* rows in pairs: both broadcast rows are requested before the first product, so the second row's LDS latency
             * hides behind the first row's FMAs (rows past the last constraint hold zeros and cost one wasted row at most) */
            double acc0 = 0, acc1 = 0;
            if (r < nefc) {
                /* the 40-dof instantiation requests only the first half of the second row up front (its registers are what
                 * the allocator otherwise spills in this stage) and the rest once the first row's products are under way */
                constexpr int YB1 = NVP > 32 ? NVP / 2 : NVP;
                double ya[NVP], yb[NVP];
#pragma unroll
                for (int k = 0; k < NVP; ++k) ya[k] = S.x.Yr[r][k];
#pragma unroll
                for (int k = 0; k < YB1; ++k) yb[k] = S.x.Yr[r + 1 < MAXR ? r + 1 : r][k];
                wv::sched_fence();
                if constexpr (NVP == 32) {
                    /* (the instantiation with a row-capped twin: the chain of the matrix core, see above) */
#pragma unroll
                    for (int k = 0; k < NVP; ++k) { acc0 = fma(ya[k], ycol[k], acc0); if (r + 1 < MAXR) acc1 = fma(yb[k], ycol[k], acc1); }
                } else if constexpr (NVP == 40) {
                    /* (likewise: its substeps of at most 47 rows go through the matrix core, the in-place form above; the second
                     * row's late half arrives while the chains run over the early halves) */
#pragma unroll
                    for (int k = 0; k < YB1; ++k) { acc0 = fma(ya[k], ycol[k], acc0); if (r + 1 < MAXR) acc1 = fma(yb[k], ycol[k], acc1); }
                    wv::sched_fence();
#pragma unroll
                    for (int k = YB1; k < NVP; ++k) yb[k] = S.x.Yr[r + 1 < MAXR ? r + 1 : r][k];
#pragma unroll
                    for (int k = YB1; k < NVP; ++k) { acc0 = fma(ya[k], ycol[k], acc0); if (r + 1 < MAXR) acc1 = fma(yb[k], ycol[k], acc1); }
                } else {
                double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
                for (int k = 0; k < NVP; k += 4) {
                    a0 += ya[k] * ycol[k]; a1 += ya[k + 1] * ycol[k + 1];
                    a2 += ya[k + 2] * ycol[k + 2]; a3 += ya[k + 3] * ycol[k + 3];
                }
                acc0 = (a0 + a1) + (a2 + a3);
                if constexpr (YB1 < NVP) {
                    wv::sched_fence();
#pragma unroll
                    for (int k = YB1; k < NVP; ++k) yb[k] = S.x.Yr[r + 1 < MAXR ? r + 1 : r][k];
                }
                if (r + 1 < MAXR) {
                    double b0 = 0, b1 = 0, b2 = 0, b3 = 0;
#pragma unroll
                    for (int k = 0; k < NVP; k += 4) {
                        b0 += yb[k] * ycol[k]; b1 += yb[k + 1] * ycol[k + 1];
                        b2 += yb[k + 2] * ycol[k + 2]; b3 += yb[k + 3] * ycol[k + 3];
                    }
                    acc1 = (b0 + b1) + (b2 + b3);
                }
                }
            }
            arow[r] = acc0;
            if (r + 1 < MAXR) arow[r + 1] = acc1;
        }
        {
            double acc = 0;
#pragma unroll
            for (int k = 0; k < NVP; ++k) acc = fma(S.x.Yr[MAXR][k], ycol[k], acc);
            rb = acc - raref;
        }
        }
        CK_STAMP(10);

        /* ================= P10: warm start + projected Gauss-Seidel, one row per lane ================= */
        const bool isrow = rtype >= 0;
        const bool clampf = isrow && rtype != CM_CNSTR_EQUALITY;
        /* The diagonal of A: the lane's own row of Y with itself, plus the regulariser.  arow holds Y Y^T only -- selecting
         * R into the one lane-dependent entry of every row would cost a compare and four selects per row; instead the
         * R part of a row's action on its own residual is applied once per sweep (cdiag below): a row's residual is not
         * read again between its own turn and the end of the sweep. */
        double Aii = 1.0;
        if constexpr (gram_on_matrix_core || gram_in_place) { if (isrow) Aii = diag_yy + rR; }
        else if (isrow) {
            if constexpr (NVP == 32 || NVP == 40) { /* (the instantiations with a matrix-core form: its chain) */
                double d = 0;
#pragma unroll
                for (int k = 0; k < NVP; ++k) d = fma(ycol[k], ycol[k], d);
                Aii = d + rR;
            } else {
                double d0 = 0, d1 = 0, d2 = 0, d3 = 0;
#pragma unroll
                for (int k = 0; k < NVP; k += 4) {
                    d0 += ycol[k] * ycol[k]; d1 += ycol[k + 1] * ycol[k + 1];
                    d2 += ycol[k + 2] * ycol[k + 2]; d3 += ycol[k + 3] * ycol[k + 3];
                }
                Aii = ((d0 + d1) + (d2 + d3)) + rR;
            }
        }
        const double invAii = 1.0 / Aii;
        double f = 0, res = isrow ? rb : 0.0;
        int iters = 0, nguarded = 0; /* sweeps taken, and how many of them through the guarded fallback */
        if (nefc > 0) {
            if (m->flags & CM_FLAG_WARMSTART) {
                if (isrow) {
                    f = -(jws - raref) / rR;
                    if (clampf && f < 0) f = 0;
                }
                /* A f, rows four to a wave-uniform branch and four partial sums (rows past the last one contribute
                 * nothing: their column of A is zero and f is zero in lanes that are not rows) */
                double af0 = 0, af1 = 0, af2 = 0, af3 = 0;
#pragma unroll
                for (int t = 0; t < MAXR; t += 4) {
                    if (t < nefc) {
                        af0 += arow[t] * wv::readlane(f, t);
                        if (t + 1 < MAXR) af1 += arow[t + 1] * wv::readlane(f, t + 1);
                        if (t + 2 < MAXR) af2 += arow[t + 2] * wv::readlane(f, t + 2);
                        if (t + 3 < MAXR) af3 += arow[t + 3] * wv::readlane(f, t + 3);
                    }
                }
                const double af = ((af0 + af1) + (af2 + af3)) + (isrow ? rR * f : 0.0); /* + the diagonal's R */
                double cost = wv::wave_sum(isrow ? f * (rb + 0.5 * af) : 0.0);
                if (cost > 0) f = 0;
                else if (isrow) res = rb + af;
            }
        CK_STAMP(30);
            const double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
            const double halfAii = 0.5 * Aii;
            const double flo = clampf ? 0.0 : -1e300; /* lower bound of this row's force */
            /* to the scaled domain (see pgs_rows): arow becomes B in place */
            const double ninvAii = -invAii;
            double sres = res * ninvAii;
#pragma unroll
            for (int t = 0; t < MAXR; ++t) arow[t] *= ninvAii;
            const double cdiag = isrow ? rR * ninvAii : 0.0; /* the R part of B_jj = -(Y Y^T + R)_jj / A_jj, see above */
            const int maxiter = m->iterations;
            const double tolerance = m->tolerance;
            /* (the one-row shortcut below rests on the other rows raising the cost sum by at most 1e-10 each: that must stay inside
             * the half tolerance between its 2.5 x threshold and the 2 x band -- true of any tolerance a model is likely to ask for,
             * checked because the tolerance is the model's to set) */
            const bool shortcut_ok = 0.5 * tolerance > (double)MAXR * 1e-10 * scale;
            while (iters < maxiter) {
                const int nrows = wv::opaque(nefc); /* keeps the row-bound tests out of loop-invariant hoisting */
                bool converged;
                {
                    const double f0 = f, s0 = sres;
                    double mys = 0;
                    const double lo_f = flo - f;
                    pgs_rows_fast<0, MAXR>(arow, nrows, r_, lo_f, sres, mys);
                    const double mydelta = wv::max_raw(mys, lo_f); /* (the very instruction the row chain took its step with) */
                    const double change = (r_ < nrows) ? mydelta * (halfAii * mydelta - Aii * mys) : 0.0;
                    /* The guarded sweep adds the rows' cost changes in row order, and the sum only feeds the convergence test.
                     * A single-precision tree sum decides it
                     * unless it lands within a factor two of the tolerance -- far outside what precision or the order of
                     * summation can move -- and only then is the ordered double-precision sum formed. */
                    /* One row whose own cost decrease is past 2.5 x the tolerance settles "not converged" without the sum: the other
                     * rows' changes are decreases too (or at most +1e-10 each, else the guard fires), so the sum is past the 2 x band
                     * below whatever they are.  Most sweeps before the last two or three end here: a compare and a ballot instead
                     * of the six-step tree sum on the sweep's dependent chain. */
                    const float tol = (float)tolerance;
                    const bool one_row_decides = shortcut_ok && wv::ballot(-(float)change * (float)scale > 2.5f * tol) != 0ull;
                    const float est = one_row_decides ? 4.0f * tol : -wv::wave_sum_f32((float)change) * (float)scale;
                    if (wv::ballot(change > 1e-10) != 0ull || wv::debug_force_guarded()) { /* some row would have raised the cost: redo guarded */
                        double improvement = 0;
                        f = f0; sres = s0; ++nguarded;
                        pgs_rows<0, MAXR>(arow, nrows, r_, Aii, halfAii, flo, f, sres, improvement);
                        sres = fma(cdiag, f - f0, sres);
                        converged = improvement * scale < tolerance;
                    } else {
                        if (r_ < nrows) { f += mydelta; sres = fma(cdiag, mydelta, sres); }
                        if (est < 0.5f * tol) converged = true;
                        else if (est > 2.0f * tol) converged = false;
                        else {
                            /* inside the band a double-precision tree sum decides: the rows' changes are cost decreases (at
                             * most +1e-10 each, or the guard had fired), so it differs from the ordered sum by rounding only,
                             * and the ordered sum is formed just when the tree sum lands within 1e-9 of the tolerance */
                            const double tree = -wv::wave_sum(change) * scale, tolv = tolerance;
                            if (fabs(tree - tolv) > 1e-9 * tolv) converged = tree < tolv;
                            else
                            {
                                double improvement = 0;
                                for (int t = 0; t < nrows; ++t) improvement -= wv::readlane(change, t);
                                converged = improvement * scale < tolerance;
                            }
                        }
                    }
                }
                ++iters;
                if (converged) break;
            }
        }
        CK_STAMP(11);
        if (io.ext || want_cfrc) {
            /* decode the pyramid: normal = sum of the edge forces, tangents = mu (f+ - f-) */
            const int a0 = caddr >= 0 ? caddr : 0;
            const double f0 = wv::shfl(f, a0), f1 = wv::shfl(f, (a0 + 1) & 63), f2 = wv::shfl(f, (a0 + 2) & 63), f3 = wv::shfl(f, (a0 + 3) & 63);
            double fn = 0, ft1 = 0, ft2 = 0;
            if (lane < ncon && caddr >= 0) {
                if (S.c_dim[lane] == 1) fn = f0;
                else { const double mu = S.c_fri[lane][0]; fn = f0 + f1 + f2 + f3; ft1 = mu * (f0 - f1); ft2 = mu * (f2 - f3); }
            }
            if (io.ext) {
                cm_ext_t *ex = io.ext + env;
                if (lane < ncon) { ex->con_force[lane][0] = fn; ex->con_force[lane][1] = ft1; ex->con_force[lane][2] = ft2; }
                if (lane == 0) { ex->ncon = ncon; ex->nefc = nefc; ex->solver_iter = iters; }
            }
            if (want_cfrc) {
                /* world-frame force of every contact parked over its (no longer needed) position, then lane = body adds
                 * up the contacts it takes part in: + on geom2's body, - on geom1's */
                if (lane < ncon) {
                    const double *fr = S.c_frame[lane];
                    double fw[3];
                    for (int j = 0; j < 3; ++j) fw[j] = fr[j] * fn + fr[3 + j] * ft1 + fr[6 + j] * ft2;
                    for (int j = 0; j < 3; ++j) S.c_pos[lane][j] = fw[j];
                }
                wv::sync();
                if (isbody) {
                    double acc[3] = {0, 0, 0};
                    for (int c = 0; c < ncon; ++c) {
                        const int b1 = m->geom_bodyid[S.c_g1[c]], b2 = m->geom_bodyid[S.c_g2[c]];
                        const double sg = (b2 == b ? 1.0 : 0.0) - (b1 == b ? 1.0 : 0.0);
                        for (int j = 0; j < 3; ++j) acc[j] += sg * S.c_pos[c][j];
                    }
                    for (int j = 0; j < 3; ++j) io.body_cfrc[((size_t)env * io.sb + b) * 3 + j] = acc[j];
                }
                wv::sync();
            }
        }

        if constexpr (NW == 2) {
            /* Two-wave form: the stages behind the solve -- qacc, the accelerometers, the substep's outputs, the Euler step -- are
             * wave 1's, which has staged their operands (its row of L, its column of Y, its row and column of the factor of
             * M + hB) while this wave was in its PGS sweeps.  The row forces and the solver statistics go through LDS -- the contacts'
             * solimp slots, which nothing reads behind the rows' impedances -- at barrier P; barrier E ends the substep. */
            static_assert(sizeof(S.c_solimp) >= 68 * sizeof(double), "the row forces are handed over through the contacts' solimp slots");
            double *const fbuf = &S.c_solimp[0][0];
            fbuf[lane] = f;
            if (lane == 0) { fbuf[64] = (double)ncon; fbuf[65] = (double)nefc; fbuf[66] = (double)iters; fbuf[67] = (double)nguarded; }
            wv::block_barrier(); /* P */
            wv::block_barrier(); /* E */
            CK_STAMP(37);
            if (wv::opaque(S.cmd[0])) { warn |= WARN_DIVERGED; break; }
            if (!io.integrate) break;
            time += h;
            continue;
        }
        /* ================= qacc = L^-1 D^-1/2 (y63 + Y f)  (lane = dof) ================= */
        /* (constants of the stages behind the solves, requested ahead of them: actuator velocities, the Euler step) */
        const int pf_u = lane < nu ? lane : 0, pf_ej = lane < njnt ? lane : 0;
        const double pf_agear = m->act_gear[pf_u], pf_kdamp = m->dof_damping[isdof ? k_ : 0];
        const int pf_adof = m->act_dofid[pf_u], pf_ejt = m->jnt_type[pf_ej], pf_eqa = m->jnt_qposadr[pf_ej], pf_eda = m->jnt_dofadr[pf_ej];
        double qacc;
        {
            double z = isdof ? S.x.Yr[MAXR][k_] : 0.0;
            {
                /* z += Y f : rows in groups of four so LDS reads and broadcasts overlap; rolled (row count is dynamic) */
                double z1 = 0, z2 = 0, z3 = 0;
                const int kk = isdof ? k_ : 0;
                int r = 0;
                for (; r + 4 <= nefc; r += 4) {
                    const double y0 = S.x.Yr[r][kk], y1 = S.x.Yr[r + 1][kk], y2 = S.x.Yr[r + 2][kk], y3 = S.x.Yr[r + 3][kk];
                    z += y0 * wv::readlane(f, r); z1 += y1 * wv::readlane(f, r + 1);
                    z2 += y2 * wv::readlane(f, r + 2); z3 += y3 * wv::readlane(f, r + 3);
                }
                for (; r < nefc; ++r) z += S.x.Yr[r][kk] * wv::readlane(f, r);
                z = (z + z1) + (z2 + z3);
                if (!isdof) z = 0.0;
            }
            if (isdof) z *= S.rsd[k_];
            /* forward substitution with this lane's row of L staged first (all LDS reads in flight together) */
            double lrow[NVP];
            stage_factor_row<NVP, TOPO>(S, k_, isdof, lrow);
            qacc = solve_forward<NVP, TOPO>(z, lrow, lane, nv);
        }
        {
            const bool badv = isdof && (!(qacc == qacc) || fabs(qacc) > 1e10);
            if (wv::ballot(badv) != 0ull) { warn |= WARN_DIVERGED; break; }
        }
        if (isdof) S.qacc[k_] = qacc;
        wv::sync();
        CK_STAMP(12);

        /* ---- sensors, part 2 (the accelerometer needs qacc), actuator velocities, the last substep's outputs ---- */
        outputs_after_qacc(io, S, m, env, lane, isdof, k_, nu, qacc, aslot, sb, need_imu, lastsub, pf_agear * S.qvel[pf_adof], ncon, nefc, iters, nguarded);
        if (!io.integrate) break;
        CK_STAMP(13);

        /* ================= P12 semi-implicit Euler with implicit joint damping ================= */
        {
            double lcol[NVP], lrowh[NVP]; /* this lane's column and row of the factor of M + hB, staged before the chains */
            stage_factor_h<NVP, TOPO>(S, k_, isdof, nv, lcol, lrowh);
            const double dih = isdof ? S.dinvH[k_] : 0.0;
            euler_step<NVP, TOPO>(S, m, lane, isdof, k_, nv, njnt, h, qacc, lcol, lrowh, dih, pf_kdamp, pf_ejt, pf_eqa, pf_eda);
        }
        time += h;
        CK_STAMP(14);
    }

    /* ---------------- store state ---------------- */
    if (io.progress && lane == 0 && (!io.resume || bailed)) {
        /* (a pass behind the fast kernel leaves the record of an env it completes; one that hands the env on -- the 63-row pass of a
         * model that may use 127 -- moves it to the substep the next pass starts from) */
        io.progress[env] = bailed ? sub : nsub;
        if (bailed && io.handover_out_list) io.handover_out_list[io.env0 + wv::atomic_add(io.handover_out_count, 1)] = env;
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
template <int NVP, class TOPO, int FEAT = FEAT_ALL, int MAXR = MID_ROWS, int NW = 1, bool WALK = false, int WPS = NW>
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
    env_step<NVP, TOPO, FEAT, MAXR, NW>(io, S, env, sub_start, sub_end);
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
