/*
 * pk_types.h -- constants, PhysIO (the launch descriptor) and EnvShared (an env's LDS block)
 * (part of the step kernel: included by physics_kernel.h, in this order, inside nothing; see there for the design)
 */
#ifndef CASSIE_PK_TYPES_H
#define CASSIE_PK_TYPES_H

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
typedef const WV_CONST_AS cm_envparams_t *ParamPtr; /* an env's physical parameters (cm_model.h: cm_envparams_t) */

struct PhysIO {
    const cm_model_t *models;   /* one shared model, or one per env */
    int model_stride;           /* 0 = shared, 1 = per-env */
    /* per-env physical parameters (domain randomisation): null = every env reads its model's own block
     * (cm_model_t::params); else one cm_envparams_t per env of the batch (indexed by the absolute env), written by
     * phys_batch_randomize / phys_batch_set_const.  The step kernel reads masses, inertial offsets, principal inertias, joint
     * damping, contact friction and the set_const-derived inverse weights / mean inertia through this pointer ONLY. */
    const cm_envparams_t *envparams;
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
    /* A fast instantiation that finishes, IN PLACE, the substeps it cannot hold (cassie_step_kernel's INROWS, round 6): the substep is
     * run by the 63-row instantiation's code inside the same workgroup and the env goes back to the fast code for the next one --
     * no hand-over list, no pass behind the kernel, no serial chain of the remaining substeps.  What that inner call uses in place
     * of has_next / handover_out_*: whether an instantiation with still more rows runs behind the kernel (models with the wide
     * caps) and the list it walks. */
    int inplace_has_next;
    int *inplace_out_list, *inplace_out_count;
    /* ... and how long the env STAYS in the 63-row code once it is there: 0 = for that one substep; r > 0 = until a substep needs at
     * most r rows again (the inner call then ends in front of that substep the way a hand-over does, and the fast code goes on from
     * there).  An env whose substeps mostly need more rows than the fast code holds -- CM_FLAG_HFPRISM: 82 % of them -- would
     * otherwise pay a fast attempt, a store and a load of its state per substep.  down_rows: what the inner call sees of it. */
    int inplace_stay_rows;
    int down_rows;
    int *inplace_count;         /* (may be null) device word: += 1 per env-launch (chunk) that finished a substep in place -- the launcher's
                                   signal for which form of the fast kernel the range's next launches take (phys_batch.hip) */
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
    /* (round 6) a half's steps, one per row, for the image's two half-wave partial sums; wave 0's per-row cost changes of a sweep, for
     * the rare sweep whose cost change has to be summed in row order (the convergence test within a factor two of the tolerance) */
    double stepv[2][NROW];
    double chg[NROW];
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
    int drv_msg[2];                 /* CM_DRIVE_PD_SAFE: message bits of the safety layer (cm_drive_state_t::safety_msg), the STO switch */
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

}  // namespace ck
#endif
