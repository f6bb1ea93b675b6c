/*
 * cm_model.h -- flat, pointer-free ("POD") compiled Cassie model.
 *
 * One cm_model_t is the constant block a physics step needs: it is produced on
 * the host by the MJCF-subset compiler (mjcf_loader.cpp), memcpy'd verbatim into
 * HBM and read by the HIP kernels through wave-uniform (scalar) loads.  The same
 * struct is what the CPU oracle (oracle/cassie_oracle.c) consumes, so the model
 * compiler is exercised by both sides.
 *
 * It plays the role of the mjModel fields the reference touches
 * (SURVEY.md 8b field census; reference src/cassiemujoco.c:67-122 dlsym table),
 * restricted to what the three in-scope models use.
 *
 * Plain C so that C, C++ and HIP translation units can all include it.
 */
#ifndef CM_MODEL_H
#define CM_MODEL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* capacity limits (cassie.xml: nbody 26, njnt 26, nq 35, nv 32;
 * cassie_tray_box.xml: nbody 28, nq 42, nv 38) */
#define CM_MAXBODY   32
#define CM_MAXJNT    32
#define CM_MAXQ      48
#define CM_MAXV      40
#define CM_MAXU      12
#define CM_MAXEQ     8
#define CM_MAXEQROW  24      /* 3 rows per connect */
#define CM_MAXGEOM   32      /* collision-capable geoms only */
#define CM_MAXPAIR   192     /* statically filtered candidate geom pairs */
#define CM_MAXSITE   16
#define CM_MAXSENSOR 24
#define CM_MAXSENSORDATA 40
#define CM_MAXHFPAIR 18      /* height-field pairs of a model (their results travel through an LDS table of this many records) */
#define CM_HF_PASS   10      /* height-field pairs whose samples fit one wave pass: CM_HF_PASS * CM_HF_SLOTS <= 64 lanes */
#define CM_HF_SLOTS  6       /* sample spheres per height-field pair: two ends + at most four interior ones */
#define CM_HF_MAXC    4      /* contacts per capsule / height-field pair with CM_FLAG_HFMULTI */
#define CM_HF_SLOTS_DENSE 10 /* the same with CM_FLAG_HFDENSE: two ends + at most eight interior ones (six pairs per wave pass) */
/* CAPACITIES of the contact list and of the constraint rows of an env-step (the oracle's arrays, the read-out block, the widest
 * kernel instantiation).  What a MODEL may use of them is cm_model_t::maxcon / maxefc: 32 contacts and 127 rows for models on the
 * 32-dof Cassie dof tree (cassie.xml, cassie_hfield.xml: the step kernel has a 127-row instantiation whose solve is spread over
 * both wavefronts of an env for them), 16 and 63 -- one constraint row per lane of ONE wavefront -- for every other model.  Past a
 * model's caps later contacts / rows are dropped and a warning bit is raised, in oracle and kernel alike.
 * (Both can be raised from the command line for ORACLE-ONLY studies: tests/collision_fidelity_study.py.) */
#ifndef CM_MAXCON
#define CM_MAXCON    32      /* contacts kept per env-step, at most */
#endif
#define CM_MAXSLIDE  3       /* slide joints ahead of a body's rotational joint (kin_simple) */
#ifndef CM_MAXEFC
#define CM_MAXEFC    127     /* constraint rows per env-step, at most (two wavefronts of 64 lanes: 127 rows + the qfrc_smooth column) */
#endif
#define CM_MAXCON_NARROW 16  /* the caps of models without a 127-row kernel instantiation (and of the one-wavefront instantiations) */
#define CM_MAXEFC_NARROW 63
#define CM_HP_MAXS   16      /* CM_FLAG_HFPRISM: sample spheres along a capsule's axis, at most */

/* joint types (same numbering as MuJoCo's mjtJoint) */
enum { CM_JNT_FREE = 0, CM_JNT_BALL = 1, CM_JNT_SLIDE = 2, CM_JNT_HINGE = 3 };
/* geom types (same numbering as mjtGeom) */
enum { CM_GEOM_PLANE = 0, CM_GEOM_HFIELD = 1, CM_GEOM_SPHERE = 2, CM_GEOM_CAPSULE = 3,
       CM_GEOM_ELLIPSOID = 4, CM_GEOM_CYLINDER = 5, CM_GEOM_BOX = 6, CM_GEOM_MESH = 7 };
/* sensor types used by the in-scope models (model/cassie.xml:270-292) */
enum { CM_SENS_ACTUATORPOS = 0, CM_SENS_JOINTPOS = 1, CM_SENS_FRAMEQUAT = 2, CM_SENS_GYRO = 3,
       CM_SENS_ACCELEROMETER = 4, CM_SENS_MAGNETOMETER = 5, CM_SENS_RANGEFINDER = 6 };
/* constraint row types (same order as MuJoCo's mjtConstraint for the ones used) */
enum { CM_CNSTR_EQUALITY = 0, CM_CNSTR_LIMIT_JOINT = 3, CM_CNSTR_CONTACT_FRICTIONLESS = 5,
       CM_CNSTR_CONTACT_PYRAMIDAL = 6 };

/* option flags */
#define CM_FLAG_EULERDAMP  1u   /* implicit joint damping in the Euler step (SURVEY App.B 11) */
#define CM_FLAG_WARMSTART  2u
#define CM_FLAG_REFSAFE    4u
#define CM_FLAG_HFMULTI   16u   /* capsule vs height field: up to CM_HF_MAXC contacts per pair -- its deepest sample spheres, deepest first
                                   (ties: lower sample index) -- instead of the two-contact rule; towards MuJoCo's one contact per
                                   penetrated prism.  Off by default: a Cassie standing on both feet then needs 44 constraint rows
                                   instead of 28 */
#define CM_FLAG_HFPRISM   32u   /* sphere / capsule vs height field: ONE CONTACT PER PENETRATED GRID TRIANGLE (MuJoCo reports one per penetrated
                                   prism, which is why model/cassie_hfield.xml:4 asks for nconmax = 300) instead of at most two per capsule: the
                                   capsule's axis is sampled no further apart than its radius, every grid triangle under the capsule keeps its
                                   deepest sample sphere (ties: the sample nearer the +axis end), and every triangle whose deepest sample is within
                                   the margin gives a contact, in grid order.  Off by default: a Cassie standing on rough terrain
                                   then needs 36 rows on average and up to the 127 of the widest instantiation.  Overrides HFMULTI / HFDENSE. */
#define CM_FLAG_BOX8      64u   /* box vs box: keep up to EIGHT points of the clipped incident face (MuJoCo's mjc_BoxBox returns up to eight: the
                                   corners of the octagon two partly overlapping faces have in common) instead of its four deepest.  A cube resting
                                   flat on a larger face has four candidates either way (model/cassie_tray_box.xml:213-216, :230-237 at rest), so
                                   the option matters for faces that overlap partly.  Off by default */
#define CM_FLAG_HFDENSE    8u   /* capsule vs height field: up to CM_HF_SLOTS_DENSE - 2 interior samples instead of CM_HF_SLOTS - 2 (a grid
                                   cell apart along Cassie's 0.43 m shin on the 5 cm grid of example/test_hfield.py); off by default: -10 % on
                                   BASELINE config 4 */

#define CM_MINVAL 1e-15

/* Kinematics record of a body of a kin_simple model: everything its lane needs to build the body's transform relative
 * to its parent, in ONE level of reads indexed by the body (no body -> joint -> parameter chains, no loop over joints).
 * Axes and anchors of the slides and of the rotational joint are constants in the PARENT frame (*_p = mat * local value),
 * because no rotation of the same body precedes them.  A body with a free joint has mat / quat = identity and pos unused
 * (its frame is qpos itself); absent slides have zero axes (their arithmetic runs unpredicated), absent rotational joint:
 * rot_type = rot_jnt = -1. */
typedef struct cm_kinrec {
    int nslide, jnt0, rot_jnt, rot_type;
    int rot_qadr, slide_qadr[CM_MAXSLIDE];
    double slide_ref[CM_MAXSLIDE], slide_axis_p[CM_MAXSLIDE][3], slide_pos_p[CM_MAXSLIDE][3];
    double rot_ref, rot_axis[3], rot_pos[3];   /* local axis (hinge quaternion) and anchor (0 for free joints) */
    double rot_axis_p[3], rot_pos_p[3];       /* (0, 0, 1) and 0 for free joints */
    double mat[9], quat[4], pos[3];
} cm_kinrec_t;

/* Per-env physical parameters (domain randomisation on the device).  The few fields of the model that
 * the reference's setters change per simulator -- body masses, inertial frame offsets (and principal inertias), joint
 * damping, geom friction (reference src/cassiemujoco.c:1323-1436) -- and everything mj_setConst derives from them
 * (reference :949-977: the inverse weights at qpos0 behind the constraint regularisers, the mean inertia behind the solver's
 * tolerance), already denormalised into the per-joint / per-equality / per-pair values the constraint stages read.  The
 * step kernel reads THESE fields through one pointer per env: cm_model_t::params of the shared model, or the env's own
 * block (PhysIO::envparams, [nenv], 10 KB each) once phys_batch_randomize has been used -- the other 95 KB of the model
 * stay shared.  The first five arrays are the inputs (phys_batch_randomize), the rest is written by the device's set_const
 * kernel (phys_batch_set_const) or, for the model's own block, by the host compile. */
typedef struct cm_envparams {
    double body_mass[CM_MAXBODY];
    double body_ipos[CM_MAXBODY][3];
    double body_inertia[CM_MAXBODY][3];
    double dof_damping[CM_MAXV];
    double geom_friction[CM_MAXGEOM][3];      /* collision geoms in compiled order (cm_model_t::geom_fullid maps to the full list) */
    /* derived */
    double meaninertia, pad;
    double body_invweight0[CM_MAXBODY][2];
    double dof_invweight0[CM_MAXV];
    double jnt_liminvweight[CM_MAXJNT];
    double eq_invweight[CM_MAXEQ];
    double pair_invweight[CM_MAXPAIR];
    double pair_friction[CM_MAXPAIR][3];
} cm_envparams_t;
/* the inputs of a cm_envparams_t, as phys_batch_randomize names them */
enum { CM_P_BODY_MASS = 0, CM_P_BODY_IPOS = 1, CM_P_BODY_INERTIA = 2, CM_P_DOF_DAMPING = 3, CM_P_GEOM_FRICTION = 4, CM_P_COUNT = 5 };

typedef struct cm_model {
    /* sizes */
    int nq, nv, nu, nbody, njnt, ngeom, npair, neq, nsite, nsensor, nsensordata;
    int maxdepth;          /* deepest body level (world = 0) */
    int iterations;        /* PGS sweeps (model/cassie.xml:5 -> 50) */
    unsigned flags;
    int hfield_geom;       /* index into geom_* of the hfield geom, -1 if none */
    int hfield_nrow, hfield_ncol;
    int npair_always;      /* simple pairs [0, npair_always) are always tested; [npair_always, npair_simple) involve a static
                            * non-plane geom (stairs ...) and are skipped as a block while all of those are out of reach */
    int maxcon, maxefc;    /* contacts / constraint rows an env-step of this model may use (<= CM_MAXCON / CM_MAXEFC, see there) */
    int npair_simple;      /* pairs [0, npair_simple) give <= 2 contacts and are tested one per lane; the rest
                            * (plane-box, box-box) are tested by the whole wave, one pair at a time */
    double timestep, tolerance, meaninertia;
    double gravity[3], magnetic[3];
    double hfield_size[4]; /* x half-size, y half-size, z top scale, z bottom */

    /* bodies (index 0 = world) */
    int body_parentid[CM_MAXBODY];
    int body_rootid[CM_MAXBODY];          /* top-level ancestor (child of world); 0 for world */
    int body_weldid[CM_MAXBODY];          /* nearest ancestor-or-self that has joints; 0 = static */
    int body_jntadr[CM_MAXBODY], body_jntnum[CM_MAXBODY];
    int body_dofadr[CM_MAXBODY], body_dofnum[CM_MAXBODY];
    int body_depth[CM_MAXBODY];
    int body_subtreeend[CM_MAXBODY];      /* bodies [b, end) form b's subtree (depth-first ids) */
    uint64_t body_dofmask[CM_MAXBODY];    /* bit k set <=> dof k moves this body */
    int body_nchild[CM_MAXBODY], body_child[CM_MAXBODY][4]; /* direct children (at most 4 in the supported models) */
    int body_anc3[CM_MAXBODY][6];         /* ancestors at distance 1, 2 | 3, 6 | 9, 18 (0 = world): radix-3 pointer jumping for the
                                           * kinematic recursion -- a round composes a body's partial transform with those of two
                                           * ancestors, so two rounds cover trees 9 levels deep (Cassie), three rounds 27 */
    int nroot, root_body[4];              /* kinematic tree roots (children of the world that move) */
    double body_pos[CM_MAXBODY][3], body_quat[CM_MAXBODY][4];
    double body_ipos[CM_MAXBODY][3], body_iquat[CM_MAXBODY][4];
    double body_mass[CM_MAXBODY], body_inertia[CM_MAXBODY][3];
    double body_mat[CM_MAXBODY][9], body_imat[CM_MAXBODY][9]; /* rotation matrices of body_quat / body_iquat */
    /* kin_simple: every body's joints are at most CM_MAXSLIDE slides followed by at most one rotational joint (hinge /
     * ball / free) -- true of the three in-scope models (the pelvis of model/cassie.xml:81-84 is three slides and a
     * ball); body_kin is meaningful only then, and the compile-time-topology kernels require it */
    int kin_simple;
    cm_kinrec_t body_kin[CM_MAXBODY];
    double body_invweight0[CM_MAXBODY][2];
    double body_reach[CM_MAXBODY];        /* for tree roots: radius around the root body's origin that contains every
                                           * collision geom of the tree in any configuration (1e30 = unbounded) */

    /* joints */
    int jnt_type[CM_MAXJNT], jnt_qposadr[CM_MAXJNT], jnt_dofadr[CM_MAXJNT];
    int jnt_bodyid[CM_MAXJNT], jnt_limited[CM_MAXJNT];
    double jnt_liminvweight[CM_MAXJNT];   /* dof_invweight0 of the joint's (first) dof: diagonal approximation of a limit row */
    double jnt_ref[CM_MAXJNT];            /* qpos0 at the joint's qposadr (hinge / slide reference), so kinematics reads it in one level */
    int jnt_parentbody[CM_MAXJNT];        /* parent of the joint's body, -1 for free joints (their anchor is already in world coordinates) */
    double jnt_pos[CM_MAXJNT][3], jnt_axis[CM_MAXJNT][3], jnt_range[CM_MAXJNT][2];
    double jnt_stiffness[CM_MAXJNT], jnt_margin[CM_MAXJNT];
    double jnt_solref[CM_MAXJNT][2], jnt_solimp[CM_MAXJNT][5];
    double qpos0[CM_MAXQ], qpos_spring[CM_MAXQ];

    /* dofs */
    int dof_bodyid[CM_MAXV], dof_jntid[CM_MAXV], dof_parentid[CM_MAXV];
    double dof_armature[CM_MAXV], dof_damping[CM_MAXV], dof_invweight0[CM_MAXV];
    uint64_t dof_ancmask[CM_MAXV];        /* proper ancestors of dof k in the dof tree (M's row pattern) */
    uint64_t dof_descmask[CM_MAXV];       /* dofs that have k as ancestor, plus k itself (M's column pattern) */
    /* denormalised per-dof records of the passive / actuation stage (one level of reads, no branches): spring of the
     * dof's hinge / slide joint (stiffness 0 otherwise), and the actuator on the dof (gear 0, actuator 0 if none) */
    double dof_stiffness[CM_MAXV], dof_springref[CM_MAXV], dof_gear[CM_MAXV], dof_ctrl_lo[CM_MAXV], dof_ctrl_hi[CM_MAXV];
    int dof_qadr[CM_MAXV], dof_act[CM_MAXV];
    int dof_anc4[CM_MAXV][9];             /* ancestor dofs at distance 1, 2, 3 | 4, 8, 12 | 16, 32, 48 (-1 = none): radix-4 pointer jumping
                                           * for the prefix sums along the dof chains -- a round adds the partial sums of three
                                           * ancestors, so two rounds cover chains of 16 dofs, three rounds chains of 64 */
    int dof_vinsrc[CM_MAXV];              /* dof whose chain sum is the velocity entering dof k's joint (-1 = none): the nearest
                                             ancestor of another joint; for the rotational dofs of a free joint its last translational dof */
    int body_lastdof[CM_MAXBODY];         /* last dof on the body's chain to the root (-1 = none) */
    uint64_t dof_velmask[CM_MAXV];        /* ancestors of k that belong to other joints (velocity seen by joint k) */

    /* collision geoms (contype|conaffinity != 0) */
    int geom_type[CM_MAXGEOM], geom_bodyid[CM_MAXGEOM], geom_condim[CM_MAXGEOM];
    int geom_priority[CM_MAXGEOM], geom_contype[CM_MAXGEOM], geom_conaffinity[CM_MAXGEOM];
    int geom_fullid[CM_MAXGEOM];          /* id in the host model's full geom list */
    double geom_pos[CM_MAXGEOM][3], geom_quat[CM_MAXGEOM][4], geom_size[CM_MAXGEOM][3];
    double geom_mat[CM_MAXGEOM][9];       /* rotation matrix of geom_quat */
    double geom_friction[CM_MAXGEOM][3], geom_solref[CM_MAXGEOM][2], geom_solimp[CM_MAXGEOM][5];
    double geom_solmix[CM_MAXGEOM], geom_margin[CM_MAXGEOM], geom_gap[CM_MAXGEOM];
    double geom_rbound[CM_MAXGEOM];       /* bounding-sphere radius, 0 for planes/hfields */
    int geom_farstatic[CM_MAXGEOM];       /* 1: static, not a plane / height field -> its pairs can be block-culled */

    /* candidate pairs after the static bitmask / same-body / parent-child filter;
     * geom1's type <= geom2's type (MuJoCo's narrow-phase convention) and the list
     * is sorted by (body1, body2, geom1, geom2) within the simple and the wave-cooperative groups, so contact
     * order is deterministic */
    int pair_geom1[CM_MAXPAIR], pair_geom2[CM_MAXPAIR];
    /* everything about a pair that does not depend on the state, denormalised so the collision pass reads it with
     * one level of (lane-coalesced) loads instead of chasing pair -> geom -> parameter arrays; the contact
     * parameters are already mixed the way mj_contactParam does (priority, else solmix / max condim / max friction) */
    int pair_type[CM_MAXPAIR];            /* geom1 type | geom2 type << 8 */
    int pair_condim[CM_MAXPAIR];
    double pair_margin[CM_MAXPAIR];       /* max of the two margins */
    double pair_includemargin[CM_MAXPAIR];/* margin - max of the two gaps */
    double pair_rbound[CM_MAXPAIR][2];
    double pair_size[CM_MAXPAIR][6];      /* geom1 size, geom2 size */
    double pair_solref[CM_MAXPAIR][2], pair_solimp[CM_MAXPAIR][5], pair_friction[CM_MAXPAIR][3];
    /* what a contact row needs about the two bodies: tree roots, dof chains, translational inverse weights summed */
    int pair_root[CM_MAXPAIR][2];
    uint64_t pair_dofmask[CM_MAXPAIR][2];
    double pair_invweight[CM_MAXPAIR];

    /* height-field pairs (geom1 is the height field), in pair order: the kernel spreads their sample spheres over the
     * lanes of one wave pass ahead of the pair loop (CM_HF_SLOTS lanes per pair) */
    int nhfpair, hfpair[CM_MAXHFPAIR];
    int pair_hfslot[CM_MAXPAIR];          /* index into hfpair, -1 for the other pairs */

    /* equality constraints (connect only) */
    int eq_body1[CM_MAXEQ], eq_body2[CM_MAXEQ], eq_active[CM_MAXEQ];
    double eq_data[CM_MAXEQ][6];          /* anchor in body1 frame, anchor in body2 frame */
    double eq_solref[CM_MAXEQ][2], eq_solimp[CM_MAXEQ][5];
    int eq_root[CM_MAXEQ][2];             /* same for the two bodies of an equality constraint */
    uint64_t eq_dofmask[CM_MAXEQ][2];
    double eq_invweight[CM_MAXEQ];

    /* actuators (motor on a hinge joint) */
    int act_dofid[CM_MAXU], act_qposadr[CM_MAXU], act_ctrllimited[CM_MAXU];
    double act_gear[CM_MAXU], act_ctrlrange[CM_MAXU][2];
    double act_maxrpm[CM_MAXU];           /* actuator user[0]: no-load motor speed (model/cassie.xml:257-267) */

    /* sites */
    int site_bodyid[CM_MAXSITE];
    double site_pos[CM_MAXSITE][3], site_quat[CM_MAXSITE][4];

    /* sensors */
    int sensor_type[CM_MAXSENSOR], sensor_objid[CM_MAXSENSOR];
    int sensor_adr[CM_MAXSENSOR], sensor_dim[CM_MAXSENSOR];
    double sensor_cutoff[CM_MAXSENSOR];
    /* denormalised per-sensor records, so the sensor stage reads its constants in one level */
    int sensor_qadr[CM_MAXSENSOR];        /* actuatorpos / jointpos: qpos address; -1 otherwise */
    double sensor_gain[CM_MAXSENSOR];     /* actuatorpos: gear; jointpos: 1 */
    int sensor_body[CM_MAXSENSOR], sensor_root[CM_MAXSENSOR]; /* frame sensors: body of the site and that body's root */
    double sensor_squat[CM_MAXSENSOR][4], sensor_spos[CM_MAXSENSOR][3]; /* frame sensors: site frame in its body */
    int sensor_slot[CM_MAXSENSOR];        /* accelerometers: 0, 1, ... in sensor order (-1 otherwise / beyond two) */
    int sensor_bits[CM_MAXSENSOR];        /* sensor user[0]: encoder resolution in bits (model/cassie.xml:272-287), 0 if none */

    /* The pose-dependent constants of mj_setConst (reference src/cassiemujoco.c:952, :976): kinematics at qpos0, which no
     * randomised parameter changes -- body frames, inertial frame orientations, and per dof the motion axis / anchor in
     * world coordinates (dof_trans0: 1 = translational dof, the Jacobian column is the axis itself).  The device's set_const
     * kernel (small_kernels.h) builds M(qpos0) and the inverse weights of an env from these and the env's masses / inertial
     * offsets, in the host compile's order of operations. */
    double body_xpos0[CM_MAXBODY][3], body_xmat0[CM_MAXBODY][9], body_ximat0[CM_MAXBODY][9];
    double dof_axis0[CM_MAXV][3], dof_anchor0[CM_MAXV][3];
    int dof_trans0[CM_MAXV];
    /* the model's own parameter block: what every env uses until it is given one of its own */
    cm_envparams_t params;
} cm_model_t;

/* A cm_model_t handed over by a caller may have been edited field by field (tests and tools do: a heavier pelvis, another
 * damping): its top-level arrays are the authority, and its parameter block is brought in line with them before the model is
 * used (phys_batch_create / phys_batch_set_model do this on their copy). */
static inline void cm_model_sync_params(cm_model_t *m) {
    cm_envparams_t *p = &m->params;
    int i, k;
    for (i = 0; i < CM_MAXBODY; ++i) {
        p->body_mass[i] = m->body_mass[i];
        for (k = 0; k < 3; ++k) { p->body_ipos[i][k] = m->body_ipos[i][k]; p->body_inertia[i][k] = m->body_inertia[i][k]; }
        for (k = 0; k < 2; ++k) p->body_invweight0[i][k] = m->body_invweight0[i][k];
    }
    for (i = 0; i < CM_MAXV; ++i) { p->dof_damping[i] = m->dof_damping[i]; p->dof_invweight0[i] = m->dof_invweight0[i]; }
    for (i = 0; i < CM_MAXGEOM; ++i) for (k = 0; k < 3; ++k) p->geom_friction[i][k] = m->geom_friction[i][k];
    p->meaninertia = m->meaninertia;
    for (i = 0; i < CM_MAXJNT; ++i) p->jnt_liminvweight[i] = m->jnt_liminvweight[i];
    for (i = 0; i < CM_MAXEQ; ++i) p->eq_invweight[i] = m->eq_invweight[i];
    for (i = 0; i < CM_MAXPAIR; ++i) {
        p->pair_invweight[i] = m->pair_invweight[i];
        for (k = 0; k < 3; ++k) p->pair_friction[i][k] = m->pair_friction[i][k];
    }
}

/* Drive-level I/O state of one env: what `struct cassie_sim` keeps beside mjData for cassie_sim_step_ethercat
 * (reference src/cassiemujoco.c:210-217, :255-265): the encoder velocity filters and the motors' torque delay lines.
 * Layout-compatible with the host env's drive_filter_t[10] / joint_filter_t[6] / torque_delay[10][6]. */
#define CM_NUM_DRIVES 10
#define CM_NUM_JOINTS 6
#define CM_DRIVE_FILTER_NB 9
#define CM_JOINT_FILTER_NB 4
#define CM_JOINT_FILTER_NA 3
#define CM_TORQUE_DELAY_CYCLES 6
typedef struct cm_drive_state {
    int drive_x[CM_NUM_DRIVES][CM_DRIVE_FILTER_NB];                    /* integer FIR history of the drive encoders */
    int safety_msg;   /* CM_DRIVE_PD_SAFE: diagnostic messages the safety layer has raised since the state was cleared (bit 0: code 635,
                         a joint-limit constraint violated; bit 1: code 630, a torque at its limit) -- the block's sticky message queue */
    int pad[5];
    double joint_x[CM_NUM_JOINTS][CM_JOINT_FILTER_NB], joint_y[CM_NUM_JOINTS][CM_JOINT_FILTER_NA]; /* IIR history */
    double torque_delay[CM_NUM_DRIVES][CM_TORQUE_DELAY_CYCLES];
} cm_drive_state_t;

/* The measurement block the device-side encoder / motor models write every step: the cassie_out_t fields that
 * cassie_sensor_data and cassie_motor_data fill (reference src/cassiemujoco.c:737-803), as doubles */
enum { CM_MEAS_DRIVE_POS = 0, CM_MEAS_DRIVE_VEL = 10, CM_MEAS_DRIVE_TORQUE = 20, CM_MEAS_JOINT_POS = 30, CM_MEAS_JOINT_VEL = 36,
       CM_MEAS_ORIENTATION = 42, CM_MEAS_ANGVEL = 46, CM_MEAS_LINACC = 49, CM_MEAS_MAG = 52, CM_MEAS_DIM = 56 };
/* The derived block of an env (phys_batch_derive): what the reference's reward-side getters read out of mjData
 * (reference src/cassiemujoco.c:1254-1301 Jacobians, :1604-1700 foot kinematics / centre of mass / momentum,
 * :1812-1898 foot and heel / toe forces), as one row of doubles per env */
enum { CM_DRV_COM_POS = 0, CM_DRV_COM_VEL = 3, CM_DRV_ANGMOM = 6,
       CM_DRV_FOOT_POS = 9,     /* [2][3] cassie_sim_foot_positions (mid-foot offset applied) */
       CM_DRV_FOOT_VEL = 15,    /* [2][6] cassie_sim_foot_velocities (com-frame spatial velocity of the foot bodies) */
       CM_DRV_FOOT_FORCE = 27,  /* [12]   cassie_sim_foot_forces layout: left xyz at 0..2, right xyz at 6..8 */
       CM_DRV_TOE_FORCE = 39, CM_DRV_HEEL_FORCE = 45, /* [2][3] each, cassie_sim_heeltoe_forces */
       CM_DRV_MASS = 51,
       CM_DRV_FOOT_JACP = 52,   /* [2][3][CM_MAXV] translational Jacobians of the foot bodies' origins (cassie_sim_get_jacobian) */
       CM_DRV_FOOT_JACR = 52 + 6 * CM_MAXV, /* [2][3][CM_MAXV] rotational */
       CM_DRV_DIM = 52 + 12 * CM_MAXV };

/* drive modes of the step kernel */
enum { CM_DRIVE_OFF = 0,     /* ctrl (or the exact-state PD of phys_batch_set_pd_mode) goes straight to the actuators */
       CM_DRIVE_TORQUE = 1,  /* cassie_sim_step_ethercat on the device: commanded drive torques -> motor model + delay line */
       CM_DRIVE_PD = 2,      /* pd_input's motor PD on the ENCODER measurements of the previous step, then the same */
       CM_DRIVE_PD_SAFE = 3 }; /* ... with cassie_core_sim's safety layer between the PD law and the motor model (joint-limit attenuation /
                                restoring torques, torque-limit clamp, STO: csrc/pk_safety.h) -- cassie_sim_step_pd's whole torque path,
                                reference src/cassiemujoco.c:1147-1157, on the device; cm_drive_state_t::safety_msg collects the block's
                                diagnostic messages */

/* Optional per-env "extended" outputs of a step (what the reference reads out of mjData for its
 * derived getters: contact list + forces, body velocities, site frames, com; SURVEY.md 8b field census). */
typedef struct cm_ext {
    int ncon, nefc, solver_iter, pad;
    int con_geom1[CM_MAXCON], con_geom2[CM_MAXCON]; /* ids in the FULL geom list */
    int con_dim[CM_MAXCON], con_body1[CM_MAXCON]; /* bodies of geom1 / geom2 */
    int con_body2[CM_MAXCON], con_pad2[CM_MAXCON];
    double con_dist[CM_MAXCON], con_pos[CM_MAXCON][3], con_frame[CM_MAXCON][9];
    double con_force[CM_MAXCON][3];       /* contact-frame force: normal, tangent 1, tangent 2 (mj_contactForce role) */
    double cvel[CM_MAXBODY][6];           /* com-frame spatial velocity [rot; lin] */
    double subtree_com[CM_MAXBODY][3];    /* valid at tree roots */
    double site_xpos[CM_MAXSITE][3], site_xmat[CM_MAXSITE][9];
    double xipos[CM_MAXBODY][3];
    double cdof[CM_MAXV][6], cdof_dot[CM_MAXV][6]; /* motion axes at the tree com and their time derivatives */
    double qM[CM_MAXV][CM_MAXV];          /* dense joint-space inertia (mj_fullM role) */
    double eq_J[CM_MAXEQROW][CM_MAXV], eq_pos[CM_MAXEQROW]; /* equality rows: Jacobian and residual */
    int eq_id[CM_MAXEQROW], ne, pad2;
} cm_ext_t;

#ifdef __cplusplus
}
#endif
#endif /* CM_MODEL_H */
