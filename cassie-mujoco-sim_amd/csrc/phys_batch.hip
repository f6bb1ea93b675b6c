/*
 * phys_batch.hip -- device half of the inner C ABI (include/cassie_phys.h): N
 * Cassie environments resident in HBM and the launch of the one-wave-per-env
 * step kernel (physics_kernel.h).  Replaces the reference's per-sim
 * mj_makeData / mj_step1 / mj_step2 / mj_forward / mj_deleteData calls
 * (reference src/cassiemujoco.c:441-447, :1130-1134, :1029, :452).
 *
 * Layout in HBM: every per-env field is one env-major array [nenv][dim] of fp64,
 * so the wave that owns env e touches one contiguous row per field.  The model
 * is a single cm_model_t (or one per env for domain randomisation).
 */
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "cassie_phys.h"
#include "small_kernels.h"
#include "step_launch.h"

void phys_set_last_error(const char *s);

/* stepping launches of the fast instantiations in chunks (PhysIO::nchunk): chunks per env-launch, for launches of at least
 * CHUNK_MIN_ENVS envs (two jobs per workgroup slot: below that there is no queue whose end could be evened out) and chunks of at
 * least CHUNK_MIN_SUBSTEPS substeps */
/* Defaults by measurement (profiles/round4/chunks_ab.txt): a launch over the whole batch has nothing to fill the end of its queue
 * with: 4 chunks (+7 %); launches over env ranges (phys_batch_step_range: other ranges' launches fill in) gain nothing from more
 * than 2 in steady state, and as much as the whole-batch launch when they stand alone between two synchronisations. */
constexpr int DEFAULT_CHUNKS_WHOLE = 7 /* (round 6; 4 before: jobs of 7 substeps leave the shortest end of a queue, profiles/round6/one_stream_chunks.txt) */, DEFAULT_CHUNKS_RANGE = 2, CHUNK_MIN_ENVS = 2048, CHUNK_MIN_SUBSTEPS = 5;
constexpr int DEFAULT_TRAY_WAVES = 2; /* the 40-dof model's default form (by measurement: round 5, 17.77 against 15.72 M with one wave, profiles/round5/tray_two_waves_ab.txt; round 4 had it at -7.5 %) */

struct phys_batch {
    int nenv = 0, device = 0;
    cm_model_t host_model;          /* copy of the shared model (sizes) */
    cm_model_t *d_models = nullptr; /* 1 or nenv models in HBM */
    cm_envparams_t *d_envparams = nullptr; /* null, or one parameter block per env (phys_batch_randomize: PhysIO::envparams) */
    int model_stride = 0;
    bool generic_kernel = false;   /* validation aid: never pick a compile-time-topology instantiation */
    int dim[PHYS_F_COUNT];
    int stride[PHYS_F_COUNT];       /* doubles between consecutive envs' rows (= dim unless bound with a stride) */
    double *d_field[PHYS_F_COUNT];
    bool owned[PHYS_F_COUNT];
    int *d_warn = nullptr, *d_info = nullptr;
    float *d_hfield = nullptr;
    size_t hfield_stride = 0, hfield_floats = 0; /* stride 0: one grid shared by all envs; else one grid of hfield_floats per env */
    hipStream_t stream = nullptr;
    hipStream_t recent_streams[4] = {nullptr, nullptr, nullptr, nullptr}; /* streams of the most recent launches (callers may pass
                                       their own, and ranges of one batch may be in flight on several at once) */
    int recent_next = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_mark = nullptr;
    bool use_applied = false;       /* qfrc_applied / xfrc_applied are passed only once uploaded */
    bool pd_mode = false;
    int drive_mode = CM_DRIVE_OFF;
    cm_drive_state_t *d_drive = nullptr; /* [nenv], allocated when a drive mode is first selected */
    bool use_pd_dtarget = false, use_pd_torque = false;
    long long *d_prof = nullptr;
    /* launch-order balancing (see ck::cassie_order_kernel) */
    bool all_outputs = false;       /* measurement aid: see PhysIO::all_outputs_every_substep */
    bool balance = true;
    unsigned *d_cost = nullptr, *d_cost_wall = nullptr; /* per-env span of the last launch in 64 shader clocks / in 100 MHz ticks */
    int *d_order = nullptr;
    /* d_order holds a permutation of the env ids of every range it was last sorted for (the order kernel sorts one launch's
     * range [env0, env0 + n) at a time) and the identity everywhere else.  A launch may use the array only for a range that is
     * exactly one of these segments, or that lies wholly in identity territory: any other range would step envs outside itself
     * and skip envs inside it.  launch() keeps the list and puts overlapping segments back to the identity first. */
    struct OrderSeg { int env0, n, launches_since_sort; };
    std::vector<OrderSeg> order_segs;
    std::vector<int> order_ident;   /* 0 .. nenv - 1, the source of those resets */
    cm_ext_t *d_ext = nullptr;
    /* per-kernel timing (phys_batch_kernel_timing): event pairs around the kernel of every stepping launch that does the work */
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    int *d_progress = nullptr;      /* [nenv] substeps completed by the row-capped fast instantiation (PhysIO::progress) */
    /* stepping launches of the fast instantiations in chunks (PhysIO::nchunk): chunks per env-launch asked for (1 = off), the
     * words the chunks of an env hand over through, and the tag of the last chunked launch */
    int chunks = DEFAULT_CHUNKS_WHOLE, chunks_range = DEFAULT_CHUNKS_RANGE; /* (launches over the whole batch / over an env range) */
    bool chunks_default = true;    /* nobody has asked for a chunk count: a range's SHORT launches go as three (see launch) */
    int *d_chunk_flag = nullptr;
    int chunk_seq = 0;
    bool chunks_allowed = true;     /* (false: this device does not place workgroup w on XCD w % 8 -- launches stay in one piece) */
    /* the placement rule is a property of the QUEUE a launch goes to (a CU-masked stream, another partition mode ...): every stream
     * is probed the first time a chunked launch is about to go to it -- with the step kernel's own workgroup shape -- and the
     * kernel checks every hand-over besides (PhysIO::chunk_fault: a word in pinned host memory a consumer sets when it finds its
     * producer on another XCD; launches stay in one piece from then on) */
    std::vector<std::pair<hipStream_t, bool>> probed_streams;
    int *h_chunk_fault = nullptr, *d_chunk_fault = nullptr;
    bool chunk_fault_reported = false;
    /* the hand-over list (PhysIO::handover_list): env ids per range, [count, ticket] pairs indexed by a range's first env, and
     * -- in pinned host memory the device writes -- the number of envs the last launch of a range handed over */
    int *d_handover_list = nullptr, *d_handover_count = nullptr;
    int *h_handover_seen = nullptr, *d_handover_seen = nullptr;
    /* ... and the same for the second list: what the 63-row pass hands on to the 127-row pass (models on the Cassie dof tree) */
    int *d_handover_list2 = nullptr, *d_handover_count2 = nullptr;
    int *h_handover_seen2 = nullptr, *d_handover_seen2 = nullptr;
    bool fast_rows = true;          /* use the row-capped fast instantiation where one exists (phys_batch_set_fast_rows) */
    /* Which form of the two-wave fast kernel a range's launches take (phys_batch_set_inplace): 0 = the kernel + the list-walking pass
     * behind it, 1 = the kernel that finishes the substeps it cannot hold in place, 2 (default) = per range by what its recent launches
     * needed.  The in-place form costs the default workload 1.8 % (both codes share one register allocation) and gains 8 - 24 % where
     * envs leave the fast tier at all (profiles/round6/inplace_ab.txt): a range switches to it once a launch handed envs over
     * and back after INPLACE_QUIET reports in a row in which no env needed the wider code.  h_handover_seen[env0] is the signal in both
     * forms (the pass reports the list's length; in the in-place form the order kernel reports the kernel's count, or the run of
     * reports without one -- counted on the device, in stream order, because the launcher may run far ahead of it). */
    int inplace_mode = 2;
    struct RangeForm { int env0; bool inplace; };
    std::vector<RangeForm> range_forms;
    long long form_launches[2] = {0, 0};   /* stepping launches of the two-wave fast kernel in the plain / the in-place form (diagnostics) */
    int waves_per_env = 2;          /* two-wave form of the fast instantiations (phys_batch_set_waves_per_env) */
    int waves_per_env_tray = DEFAULT_TRAY_WAVES; /* ... of the 40-dof instantiations (CASSIE_TRAY_TWO_WAVES=0/1 overrides the default: A/B aid) */
    double *d_scratch_out = nullptr; /* [nenv][nv + nsensordata + nu]: where phys_batch_forward_kinematics sends qacc / sensordata / actuator_velocity */
};

static bool hip_ok(hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    std::string msg = std::string("HIP error in ") + what + ": " + hipGetErrorString(e);
    phys_set_last_error(msg.c_str());
    fprintf(stderr, "cassie_phys: %s\n", msg.c_str());
    return false;
}

static bool stream_may_chunk(phys_batch *b, hipStream_t s);

static ck::PhysIO make_io(phys_batch *b, int nsub, int integrate) {
    ck::PhysIO io;
    memset(&io, 0, sizeof io);
    io.models = b->d_models;
    io.model_stride = b->model_stride;
    io.envparams = b->d_envparams;
    io.nenv = b->nenv; io.nsub = nsub; io.integrate = integrate;
    io.sq = b->stride[PHYS_F_QPOS]; io.sqv = b->stride[PHYS_F_QVEL]; io.sv = b->host_model.nv; io.su = b->host_model.nu;
    io.ssd = b->stride[PHYS_F_SENSORDATA]; io.sb = b->host_model.nbody;
    io.qpos = b->d_field[PHYS_F_QPOS]; io.qvel = b->d_field[PHYS_F_QVEL];
    io.qacc_warmstart = b->d_field[PHYS_F_QACC_WARMSTART]; io.time = b->d_field[PHYS_F_TIME];
    io.ctrl = b->d_field[PHYS_F_CTRL];
    io.qfrc_applied = b->use_applied ? b->d_field[PHYS_F_QFRC_APPLIED] : nullptr;
    io.xfrc_applied = b->use_applied ? b->d_field[PHYS_F_XFRC_APPLIED] : nullptr;
    io.qacc = b->d_field[PHYS_F_QACC]; io.sensordata = b->d_field[PHYS_F_SENSORDATA];
    io.actuator_velocity = b->d_field[PHYS_F_ACTUATOR_VELOCITY];
    io.warn = b->d_warn; io.info = b->d_info;
    io.xpos_out = b->d_field[PHYS_F_XPOS]; io.xquat_out = b->d_field[PHYS_F_XQUAT];
    io.body_cfrc = b->d_field[PHYS_F_BODY_CFRC];
    io.hfield = b->d_hfield;
    io.hfield_stride = b->hfield_stride;
    if (b->pd_mode) {
        io.pd_ptarget = b->d_field[PHYS_F_PD_PTARGET]; io.pd_kp = b->d_field[PHYS_F_PD_KP]; io.pd_kd = b->d_field[PHYS_F_PD_KD];
    }
    io.drive_mode = b->d_drive ? b->drive_mode : CM_DRIVE_OFF;
    if (io.drive_mode != CM_DRIVE_OFF) {
        io.drive_state = b->d_drive;
        io.drive_cmd = b->d_field[PHYS_F_DRIVE_CMD];
        io.meas = b->d_field[PHYS_F_MEAS];
        if (io.drive_mode == CM_DRIVE_PD || io.drive_mode == CM_DRIVE_PD_SAFE) {
            io.pd_ptarget = b->d_field[PHYS_F_PD_PTARGET]; io.pd_kp = b->d_field[PHYS_F_PD_KP]; io.pd_kd = b->d_field[PHYS_F_PD_KD];
            io.pd_dtarget = b->use_pd_dtarget ? b->d_field[PHYS_F_PD_DTARGET] : nullptr;
            io.pd_torque = b->use_pd_torque ? b->d_field[PHYS_F_PD_TORQUE] : nullptr;
        }
    }
    io.all_outputs_every_substep = b->all_outputs ? 1 : 0;
    io.prof = b->d_prof;
    io.ext = b->d_ext;
    if (b->balance && b->d_order) { io.order = b->d_order; io.cost = b->d_cost; io.cost_wall = b->d_cost_wall; }
    return io;
}

/* Model, terrain and ext-buffer updates are blocking copies that must not overtake (or be overtaken by) a kernel that
 * is still in flight on the batch's stream or on the caller's: wait for both first. */
static bool quiesce(phys_batch *b) {
    bool ok = hip_ok(hipStreamSynchronize(b->stream), "hipStreamSynchronize");
    for (hipStream_t r : b->recent_streams)
        if (r && r != b->stream) ok = hip_ok(hipStreamSynchronize(r), "hipStreamSynchronize(caller stream)") && ok;
    return ok;
}

static void note_stream(phys_batch *b, hipStream_t s) {
    for (hipStream_t r : b->recent_streams) if (r == s) return;
    b->recent_streams[b->recent_next] = s;
    b->recent_next = (b->recent_next + 1) % 4;
}

/* The launch-order array for the range [env0, env0 + n): returns the range's segment record (created if the range lies in
 * identity territory), after resetting every segment that overlaps the range without coinciding with it (stream-ordered
 * copies on the launch's stream; ranges in flight on other streams must not overlap this one anyway -- their state would race). */
static phys_batch::OrderSeg *order_segment_for(phys_batch *b, int env0, int n, hipStream_t s) {
    for (auto &g : b->order_segs) if (g.env0 == env0 && g.n == n) return &g;
    for (size_t i = 0; i < b->order_segs.size();) {
        const auto g = b->order_segs[i];
        if (g.env0 < env0 + n && env0 < g.env0 + g.n) {
            if (!hip_ok(hipMemcpyAsync(b->d_order + g.env0, b->order_ident.data() + g.env0, sizeof(int) * (size_t)g.n, hipMemcpyHostToDevice, s), "hipMemcpy(order reset)")) return nullptr;
            b->order_segs.erase(b->order_segs.begin() + (long)i);
        } else ++i;
    }
    b->order_segs.push_back({env0, n, 0});
    return &b->order_segs.back();
}

static int launch(phys_batch *b, int nsub, int integrate, hipStream_t s, bool scratch_outputs = false, int env0 = 0, int n = -1) {
    ck::PhysIO io = make_io(b, nsub, integrate);
    if (n < 0) n = b->nenv;
    io.env0 = env0; io.nenv = n;
    phys_batch::OrderSeg *seg = nullptr;
    if (io.order && !integrate) { io.order = nullptr; io.cost = nullptr; io.cost_wall = nullptr; } /* forward / read-out passes: one substep, nothing to balance (identity order) */
    if (io.order) {
        seg = order_segment_for(b, env0, n, s);
        if (!seg) { io.order = nullptr; io.cost = nullptr; io.cost_wall = nullptr; }
    }
    if (scratch_outputs) {
        /* a read-out pass: the step outputs the caller's fields hold (sensordata and actuator_velocity of the last STEP feed
         * the encoder / motor models of the next one; qacc) stay as they are */
        const cm_model_t &m = b->host_model;
        io.qacc = b->d_scratch_out; io.sv = m.nv;
        io.sensordata = b->d_scratch_out + (size_t)b->nenv * m.nv; io.ssd = m.nsensordata;
        io.actuator_velocity = io.sensordata + (size_t)b->nenv * m.nsensordata;
    }
    note_stream(b, s);
    const dim3 grid(n);
    /* the compile-time-topology instantiations are used only when the model's dof tree is exactly theirs */
    const cm_model_t &hm = b->host_model;
    auto matches = [&](const unsigned long long *table, int nv, int body_levels) {
        /* (kin_simple, and a body tree no deeper than theirs: the record-based local transforms and the round count of the
         * recursion in their kinematics stage) */
        if (b->generic_kernel || hm.nv != nv || !hm.kin_simple || hm.maxdepth > body_levels) return false;
        for (int k = 0; k < nv; ++k) if (hm.dof_ancmask[k] != table[k]) return false;
        return true;
    };
    /* ... and the collision code of an instantiation is what the model's pair list needs (FEAT_*): plain cassie.xml has
     * neither height-field nor whole-wave (plane-box / box-box) pairs */
    const bool hf = hm.nhfpair > 0 || hm.hfield_geom >= 0, wp = hm.npair > hm.npair_simple;
    bool launched;
    hipEvent_t ev_after = nullptr;
    if (b->timing && integrate) {
        if (b->ev_used == b->ev_pool.size() && b->ev_pool.size() < 65536) { /* (launches beyond that between two queries go untimed) */
            hipEvent_t a = nullptr, c = nullptr;
            if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&c) == hipSuccess) b->ev_pool.emplace_back(a, c);
        }
        if (b->ev_used < b->ev_pool.size()) {
            (void)hipEventRecord(b->ev_pool[b->ev_used].first, s);
            ev_after = b->ev_pool[b->ev_used].second;
            ++b->ev_used;
        }
    }
    /* a row-capped fast instantiation with the passes behind it: the fast kernel's record of completed substeps, the launch in
     * chunks, and the hand-over lists the passes walk -> their grids */
    ck::HandoverLists hl = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    ck::TierGrids tg = {grid, grid};
    auto tiers = [&](bool fast) {
        io.progress = fast ? b->d_progress : nullptr;
        io.nchunk = 1;
        if (fast && b->d_chunk_flag && b->chunks_allowed && (n == b->nenv ? b->chunks : b->chunks_range) > 1 && n >= CHUNK_MIN_ENVS && n % 8 == 0 && nsub >= 2 * CHUNK_MIN_SUBSTEPS &&
            stream_may_chunk(b, s)) {
            /* (n % 8: workgroup w runs on XCD w % 8, so the chunks of an env -- workgroups n apart -- share an XCD and its L2) */
            /* the fast kernel's launch as chunks of at least CHUNK_MIN_SUBSTEPS substeps (the launchers size its grid) */
            /* (round 6: a range's launch of 15 .. 25 substeps -- a consumer that fences every few substeps, the driver's 20-step regions --
             * as three chunks instead of two: nothing fills the end of such a launch's queue, finer jobs shorten it, + 1.5 %; at 50
             * substeps between fences three cost 0.6 %, profiles/round6/chunks3_ab.txt) */
            const int range_chunks = b->chunks_default && nsub <= 25 ? 3 : b->chunks_range;
            const int most = nsub / CHUNK_MIN_SUBSTEPS, asked = n == b->nenv ? b->chunks : range_chunks;
            io.nchunk = asked < most ? asked : most;
            if (b->chunk_seq >= (1 << 24)) { /* (the tag has 25 bits: start over once NOTHING is in flight on the device -- the words of
                                               * envs in flight on a stream this batch does not remember must not be cleared under them --
                                               * and the clearing itself is complete before the next chunk can publish) */
                (void)hipDeviceSynchronize();
                (void)hipMemsetAsync(b->d_chunk_flag, 0, sizeof(int) * (size_t)b->nenv, s);
                (void)hipStreamSynchronize(s);
                b->chunk_seq = 0;
            }
            io.chunk_seq = ++b->chunk_seq;
            io.chunk_flag = b->d_chunk_flag;
            io.chunk_fault = b->d_chunk_fault;
        }
        if (fast && b->d_handover_list && b->d_handover_list2) {
            const bool wide_caps = hm.maxefc > CM_MAXEFC_NARROW;
            /* the pass behind the fast kernel walks the hand-over list with a small grid: twice what the range's last launch
             * handed over (the launcher learns that a launch late, through host memory) plus 16, at most one workgroup per env;
             * the 127-row pass behind that one likewise, plus 8 */
            hl.list1 = b->d_handover_list; hl.count1 = b->d_handover_count + 2 * (size_t)env0; hl.seen1 = b->d_handover_seen + env0;
            if (wide_caps) { hl.list2 = b->d_handover_list2; hl.count2 = b->d_handover_count2 + 2 * (size_t)env0; hl.seen2 = b->d_handover_seen2 + env0; }
            const int seen = b->h_handover_seen[env0], seen2 = wide_caps ? b->h_handover_seen2[env0] : 0;
            const int seen12 = seen > seen2 ? seen : seen2; /* (the first pass is never smaller than the second: it feeds it) */
            const long want = 2L * (seen12 > 0 ? seen12 : 0) + 16, want2 = 2L * (seen2 > 0 ? seen2 : 0) + 8;
            /* (a floor of 256 workgroups under both grids was measured: no gain on the prism workload, -0.6 % on config 2, profiles/round5) */
            tg.mid = dim3((unsigned)(want < n ? want : n));
            tg.wide = dim3((unsigned)(want2 < n ? want2 : n));
        }
    };
    bool inplace_launch = false;
    if (matches(ck::TopoCassie32::table, ck::TopoCassie32::nv, ck::TopoCassie32::body_levels)) {
        /* stepping launches of the two Cassie instantiations go through the row-capped fast instantiation first; the 63-row pass
         * behind it finishes the envs that met a substep with more rows, and -- for a model with the wide caps (CM_FLAG_HFPRISM) -- the
         * 127-row pass behind that one what is left; forward / read-out passes and small batches take one instantiation alone */
        const bool fast = b->fast_rows && integrate && !wp && !io.ext && b->d_progress;
        tiers(fast);
        if (fast && hl.list1 && b->waves_per_env == 2 && !wp) {
            /* the form of this range's fast kernel (see phys_batch::inplace_mode) */
            constexpr int INPLACE_QUIET = 8;
            phys_batch::RangeForm *rf = nullptr;
            for (auto &r : b->range_forms) if (r.env0 == env0) rf = &r;
            if (!rf) { b->range_forms.push_back({env0, false}); rf = &b->range_forms.back(); }
            const bool was = rf->inplace;
            /* the range's word in host memory: > 0 = env-launches the last reporting launch handed over (plain form: the pass behind the
             * kernel writes it) or finished in place (the order kernel does); -k = the last k reports of the in-place form had none */
            const int seen = *(volatile int *)(b->h_handover_seen + env0);
            if (b->inplace_mode != 2 || !(io.order && seg)) rf->inplace = b->inplace_mode == 1;   /* (auto needs the order kernel: it reports the in-place count) */
            else if (!rf->inplace) { if (seen > 0) rf->inplace = true; }
            else if (seen <= -INPLACE_QUIET) rf->inplace = false;
            if (was != rf->inplace) {
                /* the first list's count word changes its meaning with the form: start the new form from zero (stream-ordered) */
                (void)hipMemsetAsync(hl.count1, 0, 2 * sizeof(int), s);
                b->h_handover_seen[env0] = 0;
            }
            inplace_launch = rf->inplace;
            ++b->form_launches[inplace_launch ? 1 : 0];
        }
        if (!hf && !wp) { launched = ck::launch_step_cassie(grid, tg, s, io, hl, fast, hm.maxefc > CM_MAXEFC_NARROW, ev_after, b->waves_per_env, inplace_launch); ev_after = nullptr; }
        else if (hf && !wp) { launched = ck::launch_step_cassie_hfield(grid, tg, s, io, hl, fast, hm.maxefc > CM_MAXEFC_NARROW, ev_after, b->waves_per_env, inplace_launch); ev_after = nullptr; }
        else launched = ck::launch_step_cassie_all(grid, s, io);
    } else if (matches(ck::TopoCassieTray38::table, ck::TopoCassieTray38::nv, ck::TopoCassieTray38::body_levels)) {
        /* the 40-dof model: a fast instantiation of 47 rows (the boxes resting on the tray take it to 32 .. 40 routinely) -- one wave
         * per env and the Gram matrix on the matrix core by default, or the two-wave form -- with the 63-row one behind it */
        const bool plain = integrate && !io.ext && !hf;
        const bool two = plain && b->waves_per_env_tray == 2;
        const bool fast = plain && b->fast_rows && b->d_progress;
        tiers(fast);
        launched = ck::launch_step_tray(grid, tg.mid, s, io, hl, hf, fast, ev_after, two ? 2 : 1); ev_after = nullptr;
    }
    else launched = ck::launch_step_generic(grid, s, io, hm.nv > 32);
    if (ev_after) (void)hipEventRecord(ev_after, s);
    if (!launched) { (void)hip_ok(hipErrorLaunchFailure, "cassie_step_kernel launch"); return -1; }
    if (!hip_ok(hipGetLastError(), "cassie_step_kernel launch")) return -1;
    /* the next launch's order from this one's per-env cost: after every long launch, now and then after short ones.  (Round 6: "long" is
     * more than 25 substeps, not 8 -- the sort is 20 - 25 us at the end of the launch's stream, 0.7 % of a fenced 20-substep launch, and the
     * order itself is worth nothing either way since launches go in chunks: profiles/round6/launch_order_ab.txt.  The CASSIE_ORDER_EVERY
     * switch: the A/B.) */
    static const int sort_every_above = getenv("CASSIE_ORDER_EVERY") ? atoi(getenv("CASSIE_ORDER_EVERY")) : 25;
    if (io.order && integrate && (nsub > sort_every_above || ++seg->launches_since_sort >= 16)) {
        seg->launches_since_sort = 0;
        hipLaunchKernelGGL(ck::cassie_order_kernel, dim3(1), dim3(ck::ORDER_THREADS), 0, s, b->d_cost, b->d_order, n, env0,
                           inplace_launch ? hl.count1 : (int *)nullptr, inplace_launch ? hl.seen1 : (volatile int *)nullptr);
        if (!hip_ok(hipGetLastError(), "cassie_order_kernel launch")) return -1;
    }
    return 0;
}

/* optional inputs are handed to the kernel only once somebody uploaded or bound them */
static void note_field_in_use(phys_batch *b, int field) {
    if (field == PHYS_F_QFRC_APPLIED || field == PHYS_F_XFRC_APPLIED) b->use_applied = true;
    if (field == PHYS_F_PD_DTARGET) b->use_pd_dtarget = true;
    if (field == PHYS_F_PD_TORQUE) b->use_pd_torque = true;
}

/* rows [env0, env0 + n) of a field between a dense host array and HBM (dense, or strided when the field is a column
 * block of a caller-owned tensor), asynchronously on the batch's stream */
static bool copy_rows(phys_batch *b, int field, void *host, int env0, int n, bool to_device, const char *what) {
    if (!b->d_field[field]) { phys_set_last_error("this field is allocated by phys_batch_derive; call it first"); return false; }
    const size_t row = (size_t)b->dim[field], st = (size_t)b->stride[field];
    double *dev = b->d_field[field] + st * env0;
    if (n == 0) return true;
    if (st == row)
        return hip_ok(to_device ? hipMemcpyAsync(dev, host, sizeof(double) * row * n, hipMemcpyHostToDevice, b->stream)
                                : hipMemcpyAsync(host, dev, sizeof(double) * row * n, hipMemcpyDeviceToHost, b->stream), what);
    return hip_ok(to_device ? hipMemcpy2DAsync(dev, sizeof(double) * st, host, sizeof(double) * row, sizeof(double) * row, n, hipMemcpyHostToDevice, b->stream)
                            : hipMemcpy2DAsync(host, sizeof(double) * row, dev, sizeof(double) * st, sizeof(double) * row, n, hipMemcpyDeviceToHost, b->stream), what);
}

extern "C" {

/* A launch in chunks hands an env's state from one workgroup to another through the L2 both share (wave.h: publish_global): the
 * chunks of an env are workgroups a multiple of 8 apart, and workgroup w runs on XCD w % 8.  That assignment is checked here, once
 * per batch of a size that could be chunked: a grid of 1024 workgroups reports where it ran. */
__global__ void __launch_bounds__(128) cassie_xcd_probe_kernel(int *xcc) {
    if (threadIdx.x == 0) xcc[blockIdx.x] = (int)(wv::hw_id() >> 32) & 7;
}
/* does the queue behind stream s place workgroup w on XCD w % 8 (in the sense that workgroups 8 apart share an XCD)?  A grid of 1024
 * workgroups of the step kernel's shape (128 threads) reports where it ran. */
static bool workgroups_go_round_the_xcds(hipStream_t s) {
    constexpr int NWG = 1024;
    int *d = nullptr, h[NWG];
    bool ok = hipMalloc((void **)&d, sizeof h) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(cassie_xcd_probe_kernel, dim3(NWG), dim3(128), 0, s, d);
        ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
        for (int w = 8; ok && w < NWG; ++w) ok = h[w] == h[w % 8];
    }
    if (d) (void)hipFree(d);
    return ok;
}
static bool stream_may_chunk(phys_batch *b, hipStream_t s) {
    if (!b->chunks_allowed) return false;
    if (b->h_chunk_fault && *(volatile int *)b->h_chunk_fault) {
        b->chunks_allowed = false;
        if (!b->chunk_fault_reported) {
            b->chunk_fault_reported = true;
            fprintf(stderr, "cassie_phys: a chunk of a stepping launch ran on another XCD than the chunk before it (a CU-masked stream or a partition mode that "
                            "breaks the round-robin placement): the envs concerned carry warning bit 16, launches go in one piece from now on\n");
        }
        return false;
    }
    for (const auto &ps : b->probed_streams) if (ps.first == s) return ps.second;
    const bool ok = workgroups_go_round_the_xcds(s);
    if (b->probed_streams.size() < 64) b->probed_streams.emplace_back(s, ok);
    return ok;
}

phys_batch_t *phys_batch_create(const cm_model_t *model, int nenv, int device) {
    if (!model || nenv <= 0) { phys_set_last_error("phys_batch_create: bad arguments"); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        const char *msg = "phys_batch_create: no HIP device available -- this library has no CPU fallback";
        phys_set_last_error(msg);
        fprintf(stderr, "cassie_phys: %s\n", msg);
        return nullptr;
    }
    if (device < 0 || device >= ndev) { phys_set_last_error("phys_batch_create: bad device index"); return nullptr; }
    if (!hip_ok(hipSetDevice(device), "hipSetDevice")) return nullptr;
    phys_batch *b = new phys_batch;
    b->nenv = nenv; b->device = device;
    if (const char *tw = getenv("CASSIE_TRAY_TWO_WAVES")) b->waves_per_env_tray = atoi(tw) ? 2 : 1;
    b->host_model = *model;
    cm_model_sync_params(&b->host_model);
    model = &b->host_model;
    const int d[PHYS_F_COUNT] = {model->nq, model->nv, model->nv, 1, model->nu, model->nv, model->nbody * 6,
                                 model->nv, model->nsensordata, model->nu, model->nbody * 3, model->nbody * 4,
                                 model->nu, model->nu, model->nu, model->nbody * 3,
                                 model->nu + 1, CM_MEAS_DIM, model->nu, model->nu, CM_DRV_DIM, model->nv * model->nv};
    bool ok = true;
    for (int f = 0; f < PHYS_F_COUNT; ++f) {
        b->dim[f] = d[f]; b->stride[f] = d[f]; b->d_field[f] = nullptr; b->owned[f] = true;
        if (f == PHYS_F_DERIVED || f == PHYS_F_QM) continue; /* large and optional: allocated by the first phys_batch_derive */
        size_t bytes = sizeof(double) * (size_t)nenv * (d[f] > 0 ? d[f] : 1);
        ok = ok && hip_ok(hipMalloc((void **)&b->d_field[f], bytes), "hipMalloc(field)");
        if (ok) ok = hip_ok(hipMemset(b->d_field[f], 0, bytes), "hipMemset(field)");
    }
    ok = ok && hip_ok(hipMalloc((void **)&b->d_models, sizeof(cm_model_t)), "hipMalloc(model)");
    ok = ok && hip_ok(hipMemcpy(b->d_models, model, sizeof(cm_model_t), hipMemcpyHostToDevice), "hipMemcpy(model)");
    ok = ok && hip_ok(hipMalloc((void **)&b->d_warn, sizeof(int) * nenv), "hipMalloc(warn)");
    ok = ok && hip_ok(hipMemset(b->d_warn, 0, sizeof(int) * nenv), "hipMemset(warn)");
    ok = ok && hip_ok(hipMalloc((void **)&b->d_info, sizeof(int) * 4 * nenv), "hipMalloc(info)");
    ok = ok && hip_ok(hipMemset(b->d_info, 0, sizeof(int) * 4 * nenv), "hipMemset(info)");
    if (nenv >= 2048) { /* fewer envs than a couple per wave slot leave nothing to balance */
        b->order_ident.resize((size_t)nenv);
        for (int e = 0; e < nenv; ++e) b->order_ident[(size_t)e] = e;
        ok = ok && hip_ok(hipMalloc((void **)&b->d_order, sizeof(int) * (size_t)nenv), "hipMalloc(order)");
        ok = ok && hip_ok(hipMemcpy(b->d_order, b->order_ident.data(), sizeof(int) * (size_t)nenv, hipMemcpyHostToDevice), "hipMemcpy(order)");
        ok = ok && hip_ok(hipMalloc((void **)&b->d_cost, sizeof(unsigned) * (size_t)nenv), "hipMalloc(cost)");
        ok = ok && hip_ok(hipMemset(b->d_cost, 0, sizeof(unsigned) * (size_t)nenv), "hipMemset(cost)");
        ok = ok && hip_ok(hipMalloc((void **)&b->d_cost_wall, sizeof(unsigned) * (size_t)nenv), "hipMalloc(cost, wall clock)");
        ok = ok && hip_ok(hipMemset(b->d_cost_wall, 0, sizeof(unsigned) * (size_t)nenv), "hipMemset(cost, wall clock)");
    }
    ok = ok && hip_ok(hipMalloc((void **)&b->d_progress, sizeof(int) * (size_t)nenv), "hipMalloc(progress)");
    ok = ok && hip_ok(hipMemset(b->d_progress, 0, sizeof(int) * (size_t)nenv), "hipMemset(progress)");
    ok = ok && hip_ok(hipMalloc((void **)&b->d_chunk_flag, sizeof(int) * (size_t)nenv), "hipMalloc(chunk words)");
    ok = ok && hip_ok(hipMemset(b->d_chunk_flag, 0, sizeof(int) * (size_t)nenv), "hipMemset(chunk words)");
    if (const char *ck = getenv("CASSIE_CHUNKS")) { b->chunks = b->chunks_range = atoi(ck) > 1 ? (atoi(ck) < 7 ? atoi(ck) : 7) : 1; b->chunks_default = false; } /* (A/B switch) */
    ok = ok && hip_ok(hipHostMalloc((void **)&b->h_chunk_fault, sizeof(int), hipHostMallocMapped), "hipHostMalloc(chunk fault word)");
    if (ok) *b->h_chunk_fault = 0;
    ok = ok && hip_ok(hipHostGetDevicePointer((void **)&b->d_chunk_fault, b->h_chunk_fault, 0), "hipHostGetDevicePointer");
    ok = ok && hip_ok(hipMalloc((void **)&b->d_handover_list, sizeof(int) * (size_t)nenv), "hipMalloc(hand-over list)");
    ok = ok && hip_ok(hipMalloc((void **)&b->d_handover_count, sizeof(int) * 2 * (size_t)nenv), "hipMalloc(hand-over counts)");
    ok = ok && hip_ok(hipMemset(b->d_handover_count, 0, sizeof(int) * 2 * (size_t)nenv), "hipMemset(hand-over counts)");
    ok = ok && hip_ok(hipHostMalloc((void **)&b->h_handover_seen, sizeof(int) * (size_t)nenv, hipHostMallocMapped), "hipHostMalloc(hand-over seen)");
    if (ok) memset(b->h_handover_seen, 0, sizeof(int) * (size_t)nenv);
    ok = ok && hip_ok(hipHostGetDevicePointer((void **)&b->d_handover_seen, b->h_handover_seen, 0), "hipHostGetDevicePointer");
    ok = ok && hip_ok(hipMalloc((void **)&b->d_handover_list2, sizeof(int) * (size_t)nenv), "hipMalloc(second hand-over list)");
    ok = ok && hip_ok(hipMalloc((void **)&b->d_handover_count2, sizeof(int) * 2 * (size_t)nenv), "hipMalloc(second hand-over counts)");
    ok = ok && hip_ok(hipMemset(b->d_handover_count2, 0, sizeof(int) * 2 * (size_t)nenv), "hipMemset(second hand-over counts)");
    ok = ok && hip_ok(hipHostMalloc((void **)&b->h_handover_seen2, sizeof(int) * (size_t)nenv, hipHostMallocMapped), "hipHostMalloc(second hand-over seen)");
    if (ok) memset(b->h_handover_seen2, 0, sizeof(int) * (size_t)nenv);
    ok = ok && hip_ok(hipHostGetDevicePointer((void **)&b->d_handover_seen2, b->h_handover_seen2, 0), "hipHostGetDevicePointer");
    ok = ok && hip_ok(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking), "hipStreamCreate");
    ok = ok && hip_ok(hipEventCreate(&b->ev0), "hipEventCreate") && hip_ok(hipEventCreate(&b->ev1), "hipEventCreate");
    ok = ok && hip_ok(hipEventCreateWithFlags(&b->ev_mark, hipEventDisableTiming), "hipEventCreate");
    if (ok) {
        /* every env starts at qpos0 */
        std::vector<double> q0((size_t)nenv * model->nq);
        for (int e = 0; e < nenv; ++e) memcpy(&q0[(size_t)e * model->nq], model->qpos0, sizeof(double) * model->nq);
        ok = hip_ok(hipMemcpy(b->d_field[PHYS_F_QPOS], q0.data(), q0.size() * sizeof(double), hipMemcpyHostToDevice),
                    "hipMemcpy(qpos0)");
    }
    if (!ok) { phys_batch_free(b); return nullptr; }
    return b;
}

void phys_batch_free(phys_batch_t *b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    for (int f = 0; f < PHYS_F_COUNT; ++f)
        if (b->owned[f] && b->d_field[f]) (void)hipFree(b->d_field[f]);
    if (b->d_models) (void)hipFree(b->d_models);
    if (b->d_envparams) (void)hipFree(b->d_envparams);
    if (b->d_warn) (void)hipFree(b->d_warn);
    if (b->d_info) (void)hipFree(b->d_info);
    if (b->d_hfield) (void)hipFree(b->d_hfield);
    if (b->d_ext) (void)hipFree(b->d_ext);
    if (b->d_scratch_out) (void)hipFree(b->d_scratch_out);
    if (b->d_progress) (void)hipFree(b->d_progress);
    if (b->d_chunk_flag) (void)hipFree(b->d_chunk_flag);
    if (b->d_handover_list) (void)hipFree(b->d_handover_list);
    if (b->d_handover_count) (void)hipFree(b->d_handover_count);
    if (b->h_handover_seen) (void)hipHostFree(b->h_handover_seen);
    if (b->d_handover_list2) (void)hipFree(b->d_handover_list2);
    if (b->d_handover_count2) (void)hipFree(b->d_handover_count2);
    if (b->h_handover_seen2) (void)hipHostFree(b->h_handover_seen2);
    if (b->h_chunk_fault) (void)hipHostFree(b->h_chunk_fault);
    for (auto &e : b->ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (b->d_drive) (void)hipFree(b->d_drive);
    if (b->d_order) (void)hipFree(b->d_order);
    if (b->d_cost) (void)hipFree(b->d_cost);
    if (b->d_cost_wall) (void)hipFree(b->d_cost_wall);
    if (b->ev0) (void)hipEventDestroy(b->ev0);
    if (b->ev1) (void)hipEventDestroy(b->ev1);
    if (b->ev_mark) (void)hipEventDestroy(b->ev_mark);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}

int phys_batch_nenv(const phys_batch_t *b) { return b ? b->nenv : 0; }
int phys_batch_field_dim(const phys_batch_t *b, int field) {
    return (b && field >= 0 && field < PHYS_F_COUNT) ? b->dim[field] : 0;
}

int phys_batch_set_model(phys_batch_t *b, const cm_model_t *model, int env) {
    if (!b || !model) return -1;
    if (model->nq != b->host_model.nq || model->nv != b->host_model.nv || model->nbody != b->host_model.nbody ||
        model->nu != b->host_model.nu || model->nsensordata != b->host_model.nsensordata) {
        phys_set_last_error("phys_batch_set_model: model dimensions differ from the batch's");
        return -1;
    }
    (void)hipSetDevice(b->device);
    if (!quiesce(b)) return -1;
    cm_model_t synced = *model;     /* (the caller's top-level arrays are the authority: cm_model_sync_params) */
    cm_model_sync_params(&synced);
    model = &synced;
    if (env < 0) {
        if (b->d_envparams) { (void)hipFree(b->d_envparams); b->d_envparams = nullptr; } /* (the new model's own block again, for every env) */
        if (b->model_stride == 1) { /* back to one shared model */
            (void)hipFree(b->d_models);
            b->d_models = nullptr;
            if (!hip_ok(hipMalloc((void **)&b->d_models, sizeof(cm_model_t)), "hipMalloc(model)")) return -1;
            b->model_stride = 0;
        }
        b->host_model = *model;
        return hip_ok(hipMemcpy(b->d_models, model, sizeof(cm_model_t), hipMemcpyHostToDevice), "hipMemcpy(model)") ? 0 : -1;
    }
    if (env >= b->nenv) return -1;
    if (b->d_envparams) { phys_set_last_error("phys_batch_set_model: per-env models and per-env parameter blocks (phys_batch_randomize) do not mix"); return -1; }
    /* one launch serves every env with the kernel instantiation picked from the shared model: a per-env model may vary
     * parameters, not the dof tree or the kinds of collision pairs */
    if (memcmp(model->dof_ancmask, b->host_model.dof_ancmask, sizeof(model->dof_ancmask[0]) * (size_t)model->nv) != 0 ||
        model->kin_simple != b->host_model.kin_simple || model->maxdepth != b->host_model.maxdepth ||
        (model->nhfpair > 0) != (b->host_model.nhfpair > 0) || (model->hfield_geom >= 0) != (b->host_model.hfield_geom >= 0) ||
        (model->npair > model->npair_simple) != (b->host_model.npair > b->host_model.npair_simple) ||
        /* the tier chain behind a launch (63-row pass only, or 63 + 127) is picked from the SHARED model's caps, the kernel reads the env's */
        model->maxefc != b->host_model.maxefc || model->maxcon != b->host_model.maxcon ||
        ((model->flags ^ b->host_model.flags) & (CM_FLAG_HFPRISM | CM_FLAG_HFMULTI | CM_FLAG_HFDENSE | CM_FLAG_BOX8)) != 0) {
        phys_set_last_error("phys_batch_set_model: a per-env model must keep the shared model's dof tree, collision pair kinds, contact / row caps and height-field contact option");
        return -1;
    }
    if (b->model_stride == 0) { /* expand to one model per env */
        cm_model_t *all = nullptr;
        if (!hip_ok(hipMalloc((void **)&all, sizeof(cm_model_t) * (size_t)b->nenv), "hipMalloc(models)")) return -1;
        std::vector<cm_model_t> tmp((size_t)b->nenv, b->host_model);
        if (!hip_ok(hipMemcpy(all, tmp.data(), sizeof(cm_model_t) * tmp.size(), hipMemcpyHostToDevice), "hipMemcpy(models)")) return -1;
        (void)hipFree(b->d_models);
        b->d_models = all;
        b->model_stride = 1;
    }
    return hip_ok(hipMemcpy(b->d_models + env, model, sizeof(cm_model_t), hipMemcpyHostToDevice), "hipMemcpy(model)") ? 0 : -1;
}

int phys_batch_set_hfield(phys_batch_t *b, const float *data, int n) {
    if (!b || !data || n <= 0) return -1;
    (void)hipSetDevice(b->device);
    if (!quiesce(b)) return -1;
    if (b->d_hfield && (b->hfield_stride != 0 || b->hfield_floats != (size_t)n)) { /* back to one shared grid */
        (void)hipFree(b->d_hfield);
        b->d_hfield = nullptr;
    }
    if (!b->d_hfield && !hip_ok(hipMalloc((void **)&b->d_hfield, sizeof(float) * (size_t)n), "hipMalloc(hfield)")) return -1;
    b->hfield_stride = 0;
    b->hfield_floats = (size_t)n;
    return hip_ok(hipMemcpy(b->d_hfield, data, sizeof(float) * (size_t)n, hipMemcpyHostToDevice), "hipMemcpy(hfield)") ? 0 : -1;
}

int phys_batch_set_hfield_env(phys_batch_t *b, int env, const float *data, int n) {
    if (!b || !data || n <= 0 || env < 0 || env >= b->nenv) return -1;
    (void)hipSetDevice(b->device);
    if (!quiesce(b)) return -1;
    if (b->hfield_stride == 0 || b->hfield_floats != (size_t)n) {
        /* first per-env grid: expand to one grid per env, every env starting from the shared grid (or from zeros) */
        float *all = nullptr;
        const size_t bytes = sizeof(float) * (size_t)n;
        if (!hip_ok(hipMalloc((void **)&all, bytes * (size_t)b->nenv), "hipMalloc(hfield per env)")) return -1;
        const bool seed = b->d_hfield && b->hfield_stride == 0 && b->hfield_floats == (size_t)n;
        if (!seed && !hip_ok(hipMemset(all, 0, bytes * (size_t)b->nenv), "hipMemset(hfield)")) { (void)hipFree(all); return -1; }
        for (int e = 0; seed && e < b->nenv; ++e)
            if (!hip_ok(hipMemcpy(all + (size_t)e * n, b->d_hfield, bytes, hipMemcpyDeviceToDevice), "hipMemcpy(hfield)")) { (void)hipFree(all); return -1; }
        if (b->d_hfield) (void)hipFree(b->d_hfield);
        b->d_hfield = all;
        b->hfield_stride = (size_t)n;
        b->hfield_floats = (size_t)n;
    }
    return hip_ok(hipMemcpy(b->d_hfield + (size_t)env * n, data, sizeof(float) * (size_t)n, hipMemcpyHostToDevice), "hipMemcpy(hfield)") ? 0 : -1;
}

int phys_batch_upload(phys_batch_t *b, int field, const double *host, int env0, int n) {
    if (!b || !host || field < 0 || field >= PHYS_F_COUNT || env0 < 0 || n < 0 || env0 + n > b->nenv) return -1;
    (void)hipSetDevice(b->device);
    note_field_in_use(b, field);
    return copy_rows(b, field, (void *)host, env0, n, true, "upload") && hip_ok(hipStreamSynchronize(b->stream), "upload sync") ? 0 : -1;
}

int phys_batch_download(phys_batch_t *b, int field, double *host, int env0, int n) {
    if (!b || !host || field < 0 || field >= PHYS_F_COUNT || env0 < 0 || n < 0 || env0 + n > b->nenv) return -1;
    (void)hipSetDevice(b->device);
    /* (launches on callers' streams -- step_range, reset_envs -- may still be writing the field) */
    return quiesce(b) && copy_rows(b, field, host, env0, n, false, "download") && hip_ok(hipStreamSynchronize(b->stream), "download sync") ? 0 : -1;
}

int phys_batch_upload_async(phys_batch_t *b, int field, const double *host, int env0, int n) {
    if (!b || !host || field < 0 || field >= PHYS_F_COUNT || env0 < 0 || n < 0 || env0 + n > b->nenv) return -1;
    (void)hipSetDevice(b->device);
    note_field_in_use(b, field);
    return copy_rows(b, field, (void *)host, env0, n, true, "upload_async") ? 0 : -1;
}

int phys_batch_download_async(phys_batch_t *b, int field, double *host, int env0, int n) {
    if (!b || !host || field < 0 || field >= PHYS_F_COUNT || env0 < 0 || n < 0 || env0 + n > b->nenv) return -1;
    (void)hipSetDevice(b->device);
    return copy_rows(b, field, host, env0, n, false, "download_async") ? 0 : -1;
}

void *phys_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    memset(p, 0, bytes);
    return p;
}
void phys_host_free(void *p) { if (p) (void)hipHostFree(p); }

int phys_batch_download_warn(phys_batch_t *b, int *host_warn, int *host_info) {
    if (!b) return -1;
    (void)hipSetDevice(b->device);
    bool ok = quiesce(b);
    if (host_warn) ok = ok && hip_ok(hipMemcpy(host_warn, b->d_warn, sizeof(int) * b->nenv, hipMemcpyDeviceToHost), "warn");
    if (host_info) ok = ok && hip_ok(hipMemcpy(host_info, b->d_info, sizeof(int) * 4 * b->nenv, hipMemcpyDeviceToHost), "info");
    return ok ? 0 : -1;
}

int phys_batch_uses_applied(const phys_batch_t *b) { return (b && b->use_applied) ? 1 : 0; }

int phys_batch_clear_warn(phys_batch_t *b, int env0, int n) {
    if (!b || env0 < 0 || n < 0 || env0 + n > b->nenv) return -1;
    (void)hipSetDevice(b->device);
    if (!quiesce(b)) return -1;
    return hip_ok(hipMemsetAsync(b->d_warn + env0, 0, sizeof(int) * (size_t)n, b->stream), "hipMemset(warn)") &&
                   hip_ok(hipStreamSynchronize(b->stream), "warn sync") ? 0 : -1;
}

void *phys_batch_device_ptr(phys_batch_t *b, int field) {
    return (b && field >= 0 && field < PHYS_F_COUNT) ? (void *)b->d_field[field] : nullptr;
}

int phys_batch_bind(phys_batch_t *b, int field, void *device_ptr) {
    return phys_batch_bind_strided(b, field, device_ptr, (b && field >= 0 && field < PHYS_F_COUNT) ? b->dim[field] : 0);
}

int phys_batch_bind_strided(phys_batch_t *b, int field, void *device_ptr, int row_stride) {
    if (!b || !device_ptr || field < 0 || field >= PHYS_F_COUNT) return -1;
    if (row_stride != b->dim[field]) {
        const bool may = field == PHYS_F_QPOS || field == PHYS_F_QVEL || field == PHYS_F_SENSORDATA;
        if (!may || row_stride < b->dim[field]) {
            phys_set_last_error("phys_batch_bind_strided: only qpos / qvel / sensordata take a row stride, and it must be >= the field's dim");
            return -1;
        }
    }
    (void)hipSetDevice(b->device);
    /* no stream synchronisation: launches already queued keep the pointers they were given, and hipFree of the
     * replaced buffer waits for the device by itself */
    if (b->owned[field] && b->d_field[field]) (void)hipFree(b->d_field[field]);
    b->d_field[field] = (double *)device_ptr;
    b->stride[field] = row_stride;
    b->owned[field] = false;
    note_field_in_use(b, field);
    return 0;
}

int phys_batch_step(phys_batch_t *b, int nsub, void *stream) {
    if (!b || nsub <= 0) return -1;
    (void)hipSetDevice(b->device);
    return launch(b, nsub, 1, stream ? (hipStream_t)stream : b->stream);
}

int phys_batch_step_range(phys_batch_t *b, int env0, int n, int nsub, void *stream) {
    if (!b || nsub <= 0 || env0 < 0 || n <= 0 || env0 + n > b->nenv) return -1;
    (void)hipSetDevice(b->device);
    return launch(b, nsub, 1, stream ? (hipStream_t)stream : b->stream, false, env0, n);
}

int phys_batch_forward(phys_batch_t *b, void *stream) {
    if (!b) return -1;
    (void)hipSetDevice(b->device);
    return launch(b, 1, 0, stream ? (hipStream_t)stream : b->stream);
}

int phys_batch_forward_kinematics(phys_batch_t *b, void *stream) {
    if (!b) return -1;
    (void)hipSetDevice(b->device);
    if (b->drive_mode != CM_DRIVE_OFF) { phys_set_last_error("phys_batch_forward_kinematics: not in a drive mode (the pass reads the sensordata field)"); return -1; }
    if (!b->d_scratch_out) {
        const cm_model_t &m = b->host_model;
        const size_t bytes = sizeof(double) * (size_t)b->nenv * (size_t)(m.nv + m.nsensordata + m.nu);
        if (!hip_ok(hipMalloc((void **)&b->d_scratch_out, bytes), "hipMalloc(scratch outputs)")) return -1;
    }
    return launch(b, 1, 0, stream ? (hipStream_t)stream : b->stream, true);
}

int phys_batch_set_pd_mode(phys_batch_t *b, int on) {
    if (!b) return -1;
    b->pd_mode = on != 0;
    return 0;
}

static bool ensure_drive_state(phys_batch *b) {
    if (b->d_drive) return true;
    if (b->host_model.nu != CM_NUM_DRIVES || b->host_model.nsensordata < 29) {
        phys_set_last_error("the drive-level models need Cassie's 10 drives and 29 sensor words");
        return false;
    }
    if (!quiesce(b)) return false;
    const size_t bytes = sizeof(cm_drive_state_t) * (size_t)b->nenv;
    if (!hip_ok(hipMalloc((void **)&b->d_drive, bytes), "hipMalloc(drive state)")) return false;
    return hip_ok(hipMemsetAsync(b->d_drive, 0, bytes, b->stream), "hipMemset(drive state)") && hip_ok(hipStreamSynchronize(b->stream), "sync");
}

int phys_batch_set_drive_mode(phys_batch_t *b, int mode) {
    if (!b || mode < CM_DRIVE_OFF || mode > CM_DRIVE_PD_SAFE) return -1;
    (void)hipSetDevice(b->device);
    if (mode != CM_DRIVE_OFF && !ensure_drive_state(b)) return -1;
    b->drive_mode = mode;
    return 0;
}

int phys_batch_drive_pass(phys_batch_t *b, int mode, void *stream) {
    if (!b || (mode != CM_DRIVE_TORQUE && mode != CM_DRIVE_PD && mode != CM_DRIVE_PD_SAFE)) return -1;
    (void)hipSetDevice(b->device);
    if (!ensure_drive_state(b)) return -1;
    const int keep = b->drive_mode;
    b->drive_mode = mode;
    ck::PhysIO io = make_io(b, 1, 1);
    b->drive_mode = keep;
    hipStream_t s = stream ? (hipStream_t)stream : b->stream;
    note_stream(b, s);
    hipLaunchKernelGGL(ck::cassie_drive_kernel, dim3(b->nenv), dim3(WV_WAVE), 0, s, io, b->d_field[PHYS_F_CTRL]);
    return hip_ok(hipGetLastError(), "cassie_drive_kernel launch") ? 0 : -1;
}

int phys_batch_mark(phys_batch_t *b) {
    if (!b) return -1;
    (void)hipSetDevice(b->device);
    return hip_ok(hipEventRecord(b->ev_mark, b->stream), "hipEventRecord") ? 0 : -1;
}
int phys_batch_wait_mark(phys_batch_t *b) {
    if (!b) return -1;
    (void)hipSetDevice(b->device);
    return hip_ok(hipEventSynchronize(b->ev_mark), "hipEventSynchronize") ? 0 : -1;
}

int phys_batch_clear_drive_state(phys_batch_t *b, int first, int stride, int count, void *stream) {
    if (!b || first < 0 || stride < 1 || count < 0 || (count > 0 && first + (size_t)(count - 1) * stride >= (size_t)b->nenv)) return -1;
    (void)hipSetDevice(b->device);
    if (!ensure_drive_state(b)) return -1;
    if (count == 0) return 0;
    return hip_ok(hipMemset2DAsync(b->d_drive + first, sizeof(cm_drive_state_t) * (size_t)stride, 0, sizeof(cm_drive_state_t), (size_t)count,
                                   stream ? (hipStream_t)stream : b->stream), "hipMemset2D(drive state)") ? 0 : -1;
}

int phys_batch_upload_drive_state(phys_batch_t *b, const cm_drive_state_t *host, int env0, int n) {
    if (!b || !host || env0 < 0 || n < 0 || env0 + n > b->nenv) return -1;
    (void)hipSetDevice(b->device);
    if (!ensure_drive_state(b)) return -1;
    return hip_ok(hipMemcpyAsync(b->d_drive + env0, host, sizeof(cm_drive_state_t) * (size_t)n, hipMemcpyHostToDevice, b->stream), "drive state upload") &&
                   hip_ok(hipStreamSynchronize(b->stream), "drive state sync") ? 0 : -1;
}

int phys_batch_download_drive_state(phys_batch_t *b, cm_drive_state_t *host, int env0, int n) {
    if (!b || !host || env0 < 0 || n < 0 || env0 + n > b->nenv) return -1;
    (void)hipSetDevice(b->device);
    if (!ensure_drive_state(b)) return -1;
    return hip_ok(hipMemcpyAsync(host, b->d_drive + env0, sizeof(cm_drive_state_t) * (size_t)n, hipMemcpyDeviceToHost, b->stream), "drive state download") &&
                   hip_ok(hipStreamSynchronize(b->stream), "drive state sync") ? 0 : -1;
}

int phys_batch_reset_envs(phys_batch_t *b, int first, int stride, int count, const double *qpos_row, const double *sens_row, void *stream) {
    if (!b || !qpos_row || first < 0 || stride < 1 || count < 0 || (count > 0 && first + (size_t)(count - 1) * stride >= (size_t)b->nenv)) return -1;
    (void)hipSetDevice(b->device);
    if (count == 0) return 0;
    const cm_model_t &m = b->host_model;
    ck::ResetIO io;
    memset(&io, 0, sizeof io);
    io.first = first; io.stride = stride; io.count = count;
    io.nq = m.nq; io.nv = m.nv; io.nu = m.nu; io.nsd = m.nsensordata;
    io.sq = b->stride[PHYS_F_QPOS]; io.sqv = b->stride[PHYS_F_QVEL]; io.ssd = b->stride[PHYS_F_SENSORDATA];
    io.qpos = b->d_field[PHYS_F_QPOS]; io.qvel = b->d_field[PHYS_F_QVEL]; io.warm = b->d_field[PHYS_F_QACC_WARMSTART];
    io.ctrl = b->d_field[PHYS_F_CTRL]; io.qacc = b->d_field[PHYS_F_QACC]; io.time = b->d_field[PHYS_F_TIME];
    io.sens = b->d_field[PHYS_F_SENSORDATA]; io.actvel = b->d_field[PHYS_F_ACTUATOR_VELOCITY];
    io.meas = b->d_drive ? b->d_field[PHYS_F_MEAS] : nullptr;
    io.drive = b->d_drive;
    io.qpos_row = qpos_row; io.sens_row = sens_row;
    hipStream_t s = stream ? (hipStream_t)stream : b->stream;
    note_stream(b, s);
    hipLaunchKernelGGL(ck::cassie_reset_kernel, dim3(count < 4 ? count : 4), dim3(WV_WAVE), 0, s, io);
    return hip_ok(hipGetLastError(), "cassie_reset_kernel launch") ? 0 : -1;
}

int phys_batch_sync(phys_batch_t *b) {
    if (!b) return -1;
    (void)hipSetDevice(b->device);
    return quiesce(b) ? 0 : -1; /* the batch's stream and the callers' streams of the recent launches (step_range / reset_envs) */
}

int phys_batch_enable_ext(phys_batch_t *b, int on) {
    if (!b) return -1;
    (void)hipSetDevice(b->device);
    if (!quiesce(b)) return -1;
    if (on && !b->d_ext) {
        if (!hip_ok(hipMalloc((void **)&b->d_ext, sizeof(cm_ext_t) * (size_t)b->nenv), "hipMalloc(ext)")) return -1;
        if (!hip_ok(hipMemsetAsync(b->d_ext, 0, sizeof(cm_ext_t) * (size_t)b->nenv, b->stream), "hipMemset(ext)") ||
            !hip_ok(hipStreamSynchronize(b->stream), "ext sync")) return -1;
    } else if (!on && b->d_ext) {
        (void)hipFree(b->d_ext);
        b->d_ext = nullptr;
    }
    return 0;
}

int phys_batch_download_ext(phys_batch_t *b, cm_ext_t *host, int env0, int n) {
    if (!b || !host || !b->d_ext || env0 < 0 || n < 0 || env0 + n > b->nenv) return -1;
    (void)hipSetDevice(b->device);
    return hip_ok(hipMemcpyAsync(host, b->d_ext + env0, sizeof(cm_ext_t) * (size_t)n, hipMemcpyDeviceToHost, b->stream), "ext download") &&
                   hip_ok(hipStreamSynchronize(b->stream), "ext sync")
               ? 0 : -1;
}

int phys_batch_download_ext_async(phys_batch_t *b, cm_ext_t *host, int env0, int n) {
    if (!b || !host || !b->d_ext || env0 < 0 || n < 0 || env0 + n > b->nenv) return -1;
    (void)hipSetDevice(b->device);
    return hip_ok(hipMemcpyAsync(host, b->d_ext + env0, sizeof(cm_ext_t) * (size_t)n, hipMemcpyDeviceToHost, b->stream), "ext download") ? 0 : -1;
}

int phys_batch_derive(phys_batch_t *b, const int ids[6], void *stream) {
    if (!b || !ids) return -1;
    (void)hipSetDevice(b->device);
    hipStream_t s = stream ? (hipStream_t)stream : b->stream;
    for (int f : {PHYS_F_DERIVED, PHYS_F_QM}) {
        if (b->d_field[f]) continue;
        const size_t bytes = sizeof(double) * (size_t)b->nenv * b->dim[f];
        if (!hip_ok(hipMalloc((void **)&b->d_field[f], bytes), "hipMalloc(derived)")) return -1;
        if (!hip_ok(hipMemsetAsync(b->d_field[f], 0, bytes, s), "hipMemset(derived)")) return -1;
        b->owned[f] = true;
    }
    const bool had_ext = b->d_ext != nullptr;
    if (!had_ext && phys_batch_enable_ext(b, 1) != 0) return -1;
    int rc = launch(b, 1, 0, s); /* forward: the read-out of the current state */
    ck::DeriveIO io;
    memset(&io, 0, sizeof io);
    io.models = b->d_models; io.model_stride = b->model_stride; io.nenv = b->nenv; io.envparams = b->d_envparams;
    io.ext = b->d_ext; io.xpos = b->d_field[PHYS_F_XPOS]; io.xquat = b->d_field[PHYS_F_XQUAT];
    io.derived = b->d_field[PHYS_F_DERIVED]; io.qM = b->d_field[PHYS_F_QM];
    for (int i = 0; i < 6; ++i) io.ids[i] = ids[i];
    hipLaunchKernelGGL(ck::cassie_derive_kernel, dim3(b->nenv), dim3(WV_WAVE), 0, s, io);
    rc |= hip_ok(hipGetLastError(), "cassie_derive_kernel launch") ? 0 : -1;
    if (!had_ext) {
        /* the read-out buffer is 20 KB per env and makes every step write it: keep it only for the caller who asked for it */
        if (!hip_ok(hipStreamSynchronize(s), "derive sync")) rc = -1;
        rc |= phys_batch_enable_ext(b, 0);
    }
    return rc;
}

/* ------------------------------------------------ per-env physical parameters ---- */
static const struct { size_t off; int per; } PARAM_TABLE[CM_P_COUNT] = {
    {offsetof(cm_envparams_t, body_mass), 1}, {offsetof(cm_envparams_t, body_ipos), 3}, {offsetof(cm_envparams_t, body_inertia), 3},
    {offsetof(cm_envparams_t, dof_damping), 1}, {offsetof(cm_envparams_t, geom_friction), 3}};
static int param_count(const phys_batch *b, int param) {
    const cm_model_t &m = b->host_model;
    return param == CM_P_DOF_DAMPING ? m.nv : param == CM_P_GEOM_FRICTION ? m.ngeom : m.nbody;
}
int phys_batch_param_dim(const phys_batch_t *b, int param) {
    return (b && param >= 0 && param < CM_P_COUNT) ? param_count(b, param) * PARAM_TABLE[param].per : 0;
}
/* the per-env blocks, created on first use: every env starts from the shared model's own block */
static bool ensure_envparams(phys_batch *b) {
    if (b->d_envparams) return true;
    if (b->model_stride != 0) { phys_set_last_error("per-env parameter blocks and per-env models (phys_batch_set_model with env >= 0) do not mix"); return false; }
    if (!quiesce(b)) return false;
    cm_envparams_t *all = nullptr;
    if (!hip_ok(hipMalloc((void **)&all, sizeof(cm_envparams_t) * (size_t)b->nenv), "hipMalloc(env parameters)")) return false;
    std::vector<cm_envparams_t> tmp((size_t)b->nenv, b->host_model.params);
    if (!hip_ok(hipMemcpy(all, tmp.data(), sizeof(cm_envparams_t) * tmp.size(), hipMemcpyHostToDevice), "hipMemcpy(env parameters)")) { (void)hipFree(all); return false; }
    b->d_envparams = all;
    return true;
}
static int launch_setconst(phys_batch *b, int env0, int n, int derive_inertial, hipStream_t s) {
    ck::SetConstIO io;
    io.model = b->d_models; io.params = b->d_envparams; io.env0 = env0; io.nenv = n; io.derive_inertial = derive_inertial;
    note_stream(b, s);
    /* one wave per env, at most a few thousand workgroups walking the range (58 KB of LDS each: two to a CU) */
    hipLaunchKernelGGL(ck::cassie_setconst_kernel, dim3((unsigned)(n < 2048 ? n : 2048)), dim3(WV_WAVE), 0, s, io);
    return hip_ok(hipGetLastError(), "cassie_setconst_kernel launch") ? 0 : -1;
}
int phys_batch_randomize(phys_batch_t *b, int param, const double *values, int on_device, int env0, int n, void *stream) {
    if (!b || !values || param < 0 || param >= CM_P_COUNT || env0 < 0 || n < 0 || env0 + n > b->nenv) { phys_set_last_error("phys_batch_randomize: bad arguments"); return -1; }
    (void)hipSetDevice(b->device);
    if (!ensure_envparams(b)) return -1;
    if (n == 0) return 0;
    hipStream_t s = stream ? (hipStream_t)stream : b->stream;
    const int dim = phys_batch_param_dim(b, param);
    const double *src = values;
    double *staged = nullptr;
    if (!on_device) {
        const size_t bytes = sizeof(double) * (size_t)n * dim;
        if (!hip_ok(hipMalloc((void **)&staged, bytes), "hipMalloc(parameter rows)")) return -1;
        if (!hip_ok(hipMemcpyAsync(staged, values, bytes, hipMemcpyHostToDevice, s), "hipMemcpy(parameter rows)")) { (void)hipFree(staged); return -1; }
        src = staged;
    }
    note_stream(b, s);
    hipLaunchKernelGGL(ck::cassie_param_scatter_kernel, dim3((unsigned)(n < 1024 ? n : 1024)), dim3(WV_WAVE), 0, s, b->d_envparams,
                       (int)(PARAM_TABLE[param].off / sizeof(double)), dim, src, env0, n);
    int rc = hip_ok(hipGetLastError(), "cassie_param_scatter_kernel launch") ? 0 : -1;
    /* friction needs no set_const in the reference (mj_contactParam mixes the geoms' values at every step): the pairs' mixed
     * values are refreshed right away; so is nothing else -- masses, inertial offsets and inertias wait for phys_batch_set_const
     * like mjModel edits wait for mj_setConst, damping takes effect as it is */
    if (rc == 0 && param == CM_P_GEOM_FRICTION) rc = launch_setconst(b, env0, n, 0, s);
    if (staged) { if (!hip_ok(hipStreamSynchronize(s), "randomize sync")) rc = -1; (void)hipFree(staged); }
    return rc;
}
int phys_batch_set_const(phys_batch_t *b, int env0, int n, void *stream) {
    if (!b || env0 < 0 || n < 0 || env0 + n > b->nenv) return -1;
    (void)hipSetDevice(b->device);
    if (!ensure_envparams(b)) return -1;
    if (n == 0) return 0;
    return launch_setconst(b, env0, n, 1, stream ? (hipStream_t)stream : b->stream);
}
int phys_batch_download_params(phys_batch_t *b, cm_envparams_t *host, int env0, int n) {
    if (!b || !host || env0 < 0 || n < 0 || env0 + n > b->nenv) return -1;
    (void)hipSetDevice(b->device);
    if (!quiesce(b)) return -1;
    if (!b->d_envparams) { for (int e = 0; e < n; ++e) host[e] = b->host_model.params; return 0; }
    return hip_ok(hipMemcpy(host, b->d_envparams + env0, sizeof(cm_envparams_t) * (size_t)n, hipMemcpyDeviceToHost), "parameter download") ? 0 : -1;
}
int phys_batch_uses_env_params(const phys_batch_t *b) { return b && b->d_envparams ? 1 : 0; }
size_t phys_sizeof_envparams(void) { return sizeof(cm_envparams_t); }

int phys_batch_set_all_outputs_every_substep(phys_batch_t *b, int on) {
    if (!b) return -1;
    b->all_outputs = on != 0;
    return 0;
}

int phys_batch_download_progress(phys_batch_t *b, int *host) {
    if (!b || !host || !b->d_progress) return -1;
    (void)hipSetDevice(b->device);
    if (!quiesce(b)) return -1;
    return hip_ok(hipMemcpy(host, b->d_progress, sizeof(int) * (size_t)b->nenv, hipMemcpyDeviceToHost), "progress download") ? 0 : -1;
}

int phys_batch_enable_kernel_timing(phys_batch_t *b, int on) {
    if (!b) return -1;
    b->timing = on != 0;
    b->ev_used = 0;
    return 0;
}

int phys_batch_kernel_timing(phys_batch_t *b, int *launches, double *total_ms) {
    if (!b || !launches || !total_ms) return -1;
    (void)hipSetDevice(b->device);
    if (!quiesce(b)) return -1;
    double sum = 0;
    int n = 0;
    for (size_t i = 0; i < b->ev_used; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, b->ev_pool[i].first, b->ev_pool[i].second) == hipSuccess) { sum += ms; ++n; }
    }
    b->ev_used = 0;
    *launches = n; *total_ms = sum;
    return 0;
}

int phys_batch_set_fast_rows(phys_batch_t *b, int on) {
    if (!b) return -1;
    b->fast_rows = on != 0;
    return 0;
}

int phys_batch_debug_inplace_ranges(const phys_batch_t *b) {
    if (!b) return -1;
    int k = 0;
    for (const auto &r : b->range_forms) k += r.inplace ? 1 : 0;
    return k;
}

int phys_batch_debug_form_launches(const phys_batch_t *b, long long *plain, long long *inplace) {
    if (!b) return -1;
    if (plain) *plain = b->form_launches[0];
    if (inplace) *inplace = b->form_launches[1];
    return 0;
}

int phys_batch_set_inplace(phys_batch_t *b, int mode) {
    if (!b || mode < 0 || mode > 2) return -1;
    b->inplace_mode = mode;
    return 0;
}

int phys_batch_set_chunks(phys_batch_t *b, int chunks) {
    if (!b || chunks < 1 || chunks > 7) return -1;
    b->chunks = b->chunks_range = chunks;
    b->chunks_default = false;
    return 0;
}

int phys_batch_set_waves_per_env(phys_batch_t *b, int waves) {
    if (!b || (waves != 1 && waves != 2)) return -1;
    b->waves_per_env = waves;
    b->waves_per_env_tray = waves;
    return 0;
}

int phys_batch_download_cost(phys_batch_t *b, unsigned *host) {
    if (!b || !host || !b->d_cost) return -1;
    (void)hipSetDevice(b->device);
    return quiesce(b) && hip_ok(hipMemcpy(host, b->d_cost, sizeof(unsigned) * (size_t)b->nenv, hipMemcpyDeviceToHost), "cost download") ? 0 : -1;
}

int phys_batch_measured_shader_clock(phys_batch_t *b, double *hz) {
    if (!b || !hz || !b->d_cost || !b->d_cost_wall) return -1;
    (void)hipSetDevice(b->device);
    std::vector<unsigned> c((size_t)b->nenv), w((size_t)b->nenv);
    if (!quiesce(b) || !hip_ok(hipMemcpy(c.data(), b->d_cost, sizeof(unsigned) * c.size(), hipMemcpyDeviceToHost), "cost download") ||
        !hip_ok(hipMemcpy(w.data(), b->d_cost_wall, sizeof(unsigned) * w.size(), hipMemcpyDeviceToHost), "cost download")) return -1;
    double sc = 0, sw = 0;
    for (size_t i = 0; i < c.size(); ++i) if (w[i] > 1000u) { sc += 64.0 * c[i]; sw += w[i]; } /* (envs that spanned at least 10 us) */
    if (sw <= 0) return -1;
    *hz = sc / sw * 1e8;
    return 0;
}

int phys_batch_debug_handover_pending(phys_batch_t *b) {
    if (!b) return -1;
    if (!b->d_handover_count) return 0;
    (void)hipSetDevice(b->device);
    std::vector<int> h(2 * (size_t)b->nenv);
    long total = 0;
    if (!quiesce(b)) return -1;
    for (int *d : {b->d_handover_count, b->d_handover_count2}) {
        if (!d) continue;
        if (!hip_ok(hipMemcpy(h.data(), d, sizeof(int) * h.size(), hipMemcpyDeviceToHost), "hand-over count download")) return -1;
        /* (a range in the in-place form keeps other things in its first list's words: the in-place count since the order kernel's last
         * report and the run of quiet reports) */
        if (d == b->d_handover_count) for (const auto &r : b->range_forms) if (r.inplace) h[2 * (size_t)r.env0] = h[2 * (size_t)r.env0 + 1] = 0;
        for (int v : h) total += v < 0 ? -(long)v : v;
    }
    return total > 0x7fffffff ? 0x7fffffff : (int)total;
}

/* envs the 63-row pass of the last stepping launch over [env0, ...) handed on to the 127-row pass (what that pass reported) */
int phys_batch_wide_pass_envs(phys_batch_t *b, int env0) {
    if (!b || env0 < 0 || env0 >= b->nenv || !b->h_handover_seen2) return -1;
    (void)hipSetDevice(b->device);
    if (!quiesce(b)) return -1;
    return *(volatile int *)(b->h_handover_seen2 + env0);
}

int phys_batch_set_balance(phys_batch_t *b, int on) {
    if (!b) return -1;
    b->balance = on != 0;
    return 0;
}

/* validation aid: fills the LDS of every CU with NaN bit patterns (LDS is not cleared between kernels), so that a step
 * kernel that reads LDS it has not written shows up in the results on any box, not only on a freshly booted one */
__global__ void __launch_bounds__(64) cassie_poison_lds_kernel(int *sink) {
    __shared__ unsigned long long blob[4975];   /* 39 800 B: the step kernel's footprint, so four of these fill a CU */
    for (int i = threadIdx.x; i < 4975; i += 64) blob[i] = 0xffffffffffffffffull;
    __syncthreads();
    if (sink && blob[(threadIdx.x * 77) % 4975] == 1ull) sink[0] = 1; /* keeps the stores alive */
}
int phys_batch_debug_poison_lds(phys_batch_t *b) {
    if (!b) return -1;
    (void)hipSetDevice(b->device);
    hipLaunchKernelGGL(cassie_poison_lds_kernel, dim3(8192), dim3(64), 0, b->stream, (int *)nullptr);
    return hip_ok(hipGetLastError(), "poison launch") && hip_ok(hipStreamSynchronize(b->stream), "poison sync") ? 0 : -1;
}

int phys_batch_set_generic_kernel(phys_batch_t *b, int on) {
    if (!b) return -1;
    b->generic_kernel = on != 0;
    return 0;
}

int phys_batch_profile_step(phys_batch_t *b, long long *host_stamps) { return phys_batch_profile_substeps(b, 1, host_stamps); }

int phys_batch_profile_substeps(phys_batch_t *b, int nsub, long long *host_stamps) {
    if (!b || !host_stamps || nsub < 1) return -1;
    (void)hipSetDevice(b->device);
    const size_t bytes = sizeof(long long) * ck::NSTAMP * (size_t)b->nenv;
    if (!hip_ok(hipMalloc((void **)&b->d_prof, bytes), "hipMalloc(prof)")) return -1;
    (void)hipMemsetAsync(b->d_prof, 0, bytes, b->stream);
    int rc = launch(b, nsub, 1, b->stream);
    bool ok = rc == 0 && hip_ok(hipStreamSynchronize(b->stream), "sync") &&
              hip_ok(hipMemcpy(host_stamps, b->d_prof, bytes, hipMemcpyDeviceToHost), "prof download");
    (void)hipFree(b->d_prof);
    b->d_prof = nullptr;
    return ok ? 0 : -1;
}

int phys_batch_time_steps(phys_batch_t *b, int nsub, int reps, float *mean_ms) {
    if (!b || nsub <= 0 || reps <= 0 || !mean_ms) return -1;
    (void)hipSetDevice(b->device);
    if (!hip_ok(hipEventRecord(b->ev0, b->stream), "hipEventRecord")) return -1;
    for (int r = 0; r < reps; ++r)
        if (launch(b, nsub, 1, b->stream) != 0) return -1;
    if (!hip_ok(hipEventRecord(b->ev1, b->stream), "hipEventRecord")) return -1;
    if (!hip_ok(hipEventSynchronize(b->ev1), "hipEventSynchronize")) return -1;
    float ms = 0;
    if (!hip_ok(hipEventElapsedTime(&ms, b->ev0, b->ev1), "hipEventElapsedTime")) return -1;
    *mean_ms = ms / reps;
    return 0;
}

}  // extern "C"
