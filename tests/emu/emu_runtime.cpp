/*
 * emu_runtime.cpp -- TEST-ONLY wavefront emulator (see tests/emu/wave.h).
 * Runs physics_kernel.h on the CPU: one coroutine per lane, round-robin
 * scheduling with a rendezvous at every cross-lane primitive.  Because the
 * kernel keeps all collectives in wave-uniform control flow, "resume every lane
 * until its next rendezvous" reproduces the SIMT semantics exactly.
 *
 * Workgroups of two waves (the step kernel's NW = 2 form): 128 coroutines; the
 * cross-lane primitives are rendezvous of ONE wave's 64 lanes, wv::block_barrier
 * is a rendezvous of the workgroup.  A wave that has ended does not take part in
 * later barriers (as on the hardware).  How the two waves interleave between
 * barriers is a test parameter (emu_wave_schedule): a program free of LDS races
 * gives the same bits under every schedule.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "small_kernels.h"

namespace {
constexpr int NL = 64, NWMAX = 2, NLMAX = NL * NWMAX;
constexpr size_t STACK_BYTES = 1 << 20;

struct LaneCtx { void *sp; char *stack; bool done; };
LaneCtx g_lane[NLMAX];
void *g_sched_sp;
int g_cur = 0, g_env = 0, g_grid = 1;
double g_xd[NLMAX];
int g_xi[NLMAX];
int g_kind[NLMAX];      /* what the lane yielded at: 0 = a rendezvous of its wave, 1 = the workgroup barrier */
int g_wave_schedule = 0; /* 0: the waves take turns, one rendezvous each; 1: wave 0 runs whenever it can; 2: wave 1 does */
void (*g_body)() = nullptr;
bool g_mismatch = false;
inline int wave_base() { return g_cur & ~(NL - 1); }

extern "C" void emu_switch(void **save_sp, void *load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
)");

void rendezvous() { g_kind[g_cur] = 0; emu_switch(&g_lane[g_cur].sp, g_sched_sp); }
void barrier_rendezvous() { g_kind[g_cur] = 1; emu_switch(&g_lane[g_cur].sp, g_sched_sp); }
void spin_rendezvous() { g_kind[g_cur] = 2; emu_switch(&g_lane[g_cur].sp, g_sched_sp); }

void lane_entry() {
    g_body();
    g_lane[g_cur].done = true;
    emu_switch(&g_lane[g_cur].sp, g_sched_sp);
    abort(); /* a finished lane is never resumed */
}

void run_block(void (*body)(), int nwaves = 1) {
    g_body = body;
    const int nl = NL * nwaves;
    for (int l = 0; l < nl; ++l) {
        if (!g_lane[l].stack) g_lane[l].stack = (char *)aligned_alloc(64, STACK_BYTES);
        uintptr_t top = ((uintptr_t)g_lane[l].stack + STACK_BYTES) & ~(uintptr_t)15;
        void **slot = (void **)(top - 16); /* 16-byte aligned return-address slot */
        *slot = (void *)&lane_entry;
        void **sp = slot - 6;
        for (int i = 0; i < 6; ++i) sp[i] = nullptr;
        g_lane[l].sp = sp;
        g_lane[l].done = false;
    }
    bool wave_done[NWMAX] = {false, false}, at_barrier[NWMAX] = {false, false};
    for (;;) {
        /* one turn: every wave that can run resumes its 64 lanes once, i.e. up to its next rendezvous */
        for (int t = 0; t < nwaves; ++t) {
            const int w = g_wave_schedule == 2 ? nwaves - 1 - t : t;
            if (wave_done[w] || at_barrier[w]) continue;
            int ndone = 0, nbar = 0, nspin = 0;
            for (int l = w * NL; l < (w + 1) * NL; ++l) {
                if (g_lane[l].done) { ++ndone; continue; }
                g_cur = l;
                emu_switch(&g_sched_sp, g_lane[l].sp);
                if (g_lane[l].done) ++ndone;
                else if (g_kind[l] == 1) ++nbar;
                else if (g_kind[l] == 2) ++nspin;
            }
            if (ndone == NL) wave_done[w] = true;
            else if (ndone != 0) { g_mismatch = true; fprintf(stderr, "emu: lanes of wave %d left the kernel at different rendezvous counts (%d done)\n", w, ndone); abort(); }
            else if (nbar == NL) at_barrier[w] = true;
            else if (nbar != 0) { g_mismatch = true; fprintf(stderr, "emu: %d lanes of wave %d are at the workgroup barrier, the others at a wave rendezvous\n", nbar, w); abort(); }
            if (nspin != 0 && nspin != NL) { g_mismatch = true; fprintf(stderr, "emu: %d lanes of wave %d poll a flag, the others do not\n", nspin, w); abort(); }
            if (g_wave_schedule != 0 && nspin == 0) break; /* the preferred wave runs on until it is blocked, polling or done */
        }
        bool all_done = true, all_blocked = true;
        for (int w = 0; w < nwaves; ++w) { all_done = all_done && wave_done[w]; all_blocked = all_blocked && (wave_done[w] || at_barrier[w]); }
        if (all_done) break;
        if (all_blocked) for (int w = 0; w < nwaves; ++w) at_barrier[w] = false; /* the barrier opens: every wave still alive has arrived */
    }
}
}  // namespace

namespace wv {
int lane() { return g_cur & (NL - 1); }
int wave_id() { return g_cur / NL; }
int env_id() { return g_env; }
int grid_size() { return g_grid; }
void sync() { rendezvous(); }
void block_barrier() { barrier_rendezvous(); }
void spin_yield() { spin_rendezvous(); }
double shfl(double v, int src) {
    g_xd[g_cur] = v;
    rendezvous();
    double r = g_xd[wave_base() + (src & 63)];
    rendezvous();
    return r;
}
double shfl_xor(double v, int mask) { return shfl(v, g_cur ^ mask); }
int shfl_i(int v, int src) {
    g_xi[g_cur] = v;
    rendezvous();
    int r = g_xi[wave_base() + (src & 63)];
    rendezvous();
    return r;
}
double readlane(double v, int src) { return shfl(v, src); }
/* the matrix-core instruction as the device performs it: per element the FMA chain over k = 0 .. 3 on top of C */
static double g_xa[NLMAX], g_xb[NLMAX];
void mfma_f64_16x16x4(double a, double b, double (&c)[4]) {
    g_xa[g_cur] = a; g_xb[g_cur] = b;
    rendezvous();
    const int l = g_cur & 63, j = l & 15, wb = wave_base();
    for (int v = 0; v < 4; ++v) {
        const int i = (l >> 4) + 4 * v;
        double acc = c[v];
        for (int k = 0; k < 4; ++k) acc = std::fma(g_xa[wb + i + 16 * k], g_xb[wb + j + 16 * k], acc);
        c[v] = acc;
    }
    rendezvous();
}
unsigned long long ballot(bool p) {
    g_xi[g_cur] = p ? 1 : 0;
    rendezvous();
    unsigned long long m = 0;
    for (int l = 0; l < NL; ++l) if (g_xi[wave_base() + l]) m |= 1ull << l;
    rendezvous();
    return m;
}
int g_force_guarded = 0;
int g_poison_lds = 0;
int g_skip_com_init = 0;
unsigned long g_poison_lo = 0, g_poison_hi = ~0ul;
double wave_sum(double v) { /* same association order as the device's DPP reduction (csrc/wave.h) */
    const int l = lane(), row = l >> 4;
    v += shfl(v, l ^ 1);
    v += shfl(v, l ^ 2);
    v += shfl(v, (l & ~7) | (7 - (l & 7)));
    v += shfl(v, (l & ~15) | (15 - (l & 15)));
    { const double t = shfl(v, ((row > 0 ? row - 1 : 0) << 4) | 15); if (row & 1) v += t; }
    { const double t = shfl(v, 31); if (row >= 2) v += t; }
    return shfl(v, 63);
}
float wave_sum_f32(float v) { /* same tree in single precision */
    const int l = lane(), row = l >> 4;
    v += (float)shfl((double)v, l ^ 1);
    v += (float)shfl((double)v, l ^ 2);
    v += (float)shfl((double)v, (l & ~7) | (7 - (l & 7)));
    v += (float)shfl((double)v, (l & ~15) | (15 - (l & 15)));
    { const float t = (float)shfl((double)v, ((row > 0 ? row - 1 : 0) << 4) | 15); if (row & 1) v += t; }
    { const float t = (float)shfl((double)v, 31); if (row >= 2) v += t; }
    return (float)shfl((double)v, 63);
}
}  // namespace wv

static ck::PhysIO g_io;
extern "C" void emu_force_guarded_pgs(int on) { wv::g_force_guarded = on; }
extern "C" void emu_poison_lds(int on) { wv::g_poison_lds = on; }
extern "C" void emu_skip_com_init(int on) { wv::g_skip_com_init = on; }
extern "C" void emu_poison_range(unsigned long lo, unsigned long hi) { wv::g_poison_lo = lo; wv::g_poison_hi = hi; }
extern "C" unsigned long emu_offsetof32(int which) {
    typedef ck::EnvShared<32> E;
    switch (which) {
    case 0: return offsetof(E, x); case 1: return offsetof(E, Lp); case 2: return offsetof(E, LHp); case 3: return offsetof(E, accel);
    case 4: return offsetof(E, dinv); case 5: return offsetof(E, cdof); case 6: return offsetof(E, com); case 7: return offsetof(E, qpos);
    case 8: return offsetof(E, qfrc_smooth); case 9: return offsetof(E, sens); case 10: return offsetof(E, drv_x); case 11: return offsetof(E, c_dist);
    case 12: return offsetof(E, c_dim); case 13: return offsetof(E, c_root); case 14: return offsetof(E, c_tran); default: return sizeof(E);
    }
}
static int g_force_runtime_topology = 0;
static void body32s() { ck::cassie_step_kernel<32, ck::TopoCassie32>(g_io); }
static void body32s_fast() { ck::cassie_step_kernel<32, ck::TopoCassie32, ck::FEAT_ALL, ck::FAST_ROWS>(g_io); }
/* the two-wave forms (wave 1 runs the mass-matrix stage group beside wave 0's collision / velocity / row stages) */
static void body32s_2w() { ck::cassie_step_kernel<32, ck::TopoCassie32, ck::FEAT_ALL, ck::MID_ROWS, 2>(g_io); }
/* the full instantiation as the list-walking pass behind the fast kernel */
static void body32s_walk() { ck::cassie_step_kernel<32, ck::TopoCassie32, ck::FEAT_ALL, ck::MID_ROWS, 1, true>(g_io); }
static void body32s_2w_walk() { ck::cassie_step_kernel<32, ck::TopoCassie32, ck::FEAT_ALL, ck::MID_ROWS, 2, true>(g_io); }
/* the 127-row instantiation (two wavefronts; the solve of a substep with more than 64 rows is spread over both): alone, and as the
 * pass that walks the list of envs the 63-row pass handed on */
static void body32s_wide() { ck::cassie_step_kernel<32, ck::TopoCassie32, ck::FEAT_ALL, ck::WIDE_ROWS, 2, false, 1>(g_io); }
static void body32s_wide_walk() { ck::cassie_step_kernel<32, ck::TopoCassie32, ck::FEAT_ALL, ck::WIDE_ROWS, 2, true, 1>(g_io); }
static void body32s_fast_2w() { ck::cassie_step_kernel<32, ck::TopoCassie32, ck::FEAT_ALL, ck::FAST_ROWS, 2>(g_io); }
/* ... with the 63-row code behind it in the same kernel (cassie_step_kernel's INROWS): substeps it cannot hold are finished in place */
static void body32s_fast_2w_inplace() { ck::cassie_step_kernel<32, ck::TopoCassie32, ck::FEAT_ALL, ck::FAST_ROWS, 2, false, 2, ck::MID_ROWS>(g_io); }
static int g_inplace = 0;
static int g_inplace_stay = 0;
extern "C" void emu_inplace(int on) { g_inplace = on; }
extern "C" void emu_inplace_stay_rows(int rows) { g_inplace_stay = rows; }   /* PhysIO::inplace_stay_rows (0: one substep at a time) */
static void body40s_2w() { ck::cassie_step_kernel<40, ck::TopoCassieTray38, ck::FEAT_WAVEPAIRS, ck::MID_ROWS, 2>(g_io); } /* (no height-field pairs) */
static int g_two_waves = 0, g_resume_grid = 2;
extern "C" void emu_resume_grid(int n) { g_resume_grid = n > 0 ? n : 1; }
extern "C" void emu_two_waves(int on) { g_two_waves = on; }
extern "C" void emu_wave_schedule(int mode) { g_wave_schedule = mode; }
/* the row-capped fast instantiation ahead of the full one, as phys_batch.hip launches them (PhysIO::progress / resume);
 * g_fast_bails counts the envs the fast instantiation handed over */
namespace wv { void emu_fail(const char *what) { fprintf(stderr, "emu: %s\n", what); abort(); } }
/* a launch in chunks (PhysIO::nchunk): the fast instantiation's workgroups in launch order, chunk by chunk */
static int g_chunks = 1, g_chunk_seq = 0;
extern "C" void emu_chunks(int k) { g_chunks = k > 1 ? k : 1; }
/* test hook: the "XCD" a chunk says it ran on when it publishes (consumers run on 0: anything else makes the consumer's placement
 * check fire), and the word the kernel sets then (PhysIO::chunk_fault) */
static int g_producer_xcc = 0;
static volatile int g_chunk_fault = 0;
namespace wv { int emu_xcc() { return g_producer_xcc; } }
extern "C" void emu_producer_xcc(int x) { g_producer_xcc = x & 7; g_chunk_fault = 0; }
extern "C" int emu_chunk_fault(void) { return g_chunk_fault; }
static int g_fast_rows = 0, g_fast_bails = 0, g_wide_envs = 0;
extern "C" void emu_fast_rows(int on) { g_fast_rows = on; }
extern "C" int emu_fast_bails(void) { return g_fast_bails; }
extern "C" int emu_wide_envs(void) { return g_wide_envs; } /* envs the 63-row pass handed on to the 127-row pass, so far */
static void body40s() { ck::cassie_step_kernel<40, ck::TopoCassieTray38>(g_io); }
/* the 40-dof model's row-capped instantiation (47 rows, one wave per env: the Gram matrix through the staged tile's own LDS) and
 * the full one as the list-walking pass behind it (no height-field pairs) */
static void body40s_fast() { ck::cassie_step_kernel<40, ck::TopoCassieTray38, ck::FEAT_WAVEPAIRS, ck::FAST_ROWS_TRAY>(g_io); }
static void body40s_walk() { ck::cassie_step_kernel<40, ck::TopoCassieTray38, ck::FEAT_WAVEPAIRS, ck::MID_ROWS, 1, true>(g_io); }
static void body40s_2w_walk() { ck::cassie_step_kernel<40, ck::TopoCassieTray38, ck::FEAT_WAVEPAIRS, ck::MID_ROWS, 2, true>(g_io); }
static void body32() { ck::cassie_step_kernel<32, ck::TopoRuntime>(g_io); }
static void body40() { ck::cassie_step_kernel<40, ck::TopoRuntime>(g_io); }
extern "C" void emu_force_runtime_topology(int on) { g_force_runtime_topology = on; }
static bool topo_matches(const cm_model_t *m, const unsigned long long *t, int nv, int body_levels) {
    if (m->nv != nv || !m->kin_simple || m->maxdepth > body_levels) return false;
    for (int k = 0; k < nv; ++k) if (m->dof_ancmask[k] != t[k]) return false;
    return true;
}

/* drive-level I/O of the next emu_phys_run calls (mode = CM_DRIVE_*; all pointers may be null when mode is 0) */
static int g_drive_mode = 0;
static cm_drive_state_t *g_drive_state = nullptr;
static const double *g_drive_cmd = nullptr, *g_pd_dtarget = nullptr, *g_pd_torque = nullptr;
static double *g_meas = nullptr;
extern "C" void emu_set_drive_io(int mode, cm_drive_state_t *state, const double *cmd, double *meas, const double *pd_dtarget, const double *pd_torque) {
    g_drive_mode = mode; g_drive_state = state; g_drive_cmd = cmd; g_meas = meas; g_pd_dtarget = pd_dtarget; g_pd_torque = pd_torque;
}
/* per-env physical parameter blocks of the next emu_phys_run / emu_derive calls ([nenv] cm_envparams_t, or null: the model's own) */
static const cm_envparams_t *g_envparams = nullptr;
extern "C" void emu_set_envparams(const cm_envparams_t *p) { g_envparams = p; }
/* phys_batch_set_const / the friction refresh of phys_batch_randomize on the emulator: the device's set_const kernel, env by env */
static ck::SetConstIO g_scio;
static void body_setconst() { ck::cassie_setconst_kernel(g_scio); }
extern "C" int emu_set_const(const cm_model_t *model, cm_envparams_t *params, int nenv, int derive_inertial) {
    g_scio.model = model; g_scio.params = params; g_scio.env0 = 0; g_scio.nenv = nenv; g_scio.derive_inertial = derive_inertial;
    g_grid = nenv;
    for (int e = 0; e < nenv; ++e) { g_env = e; run_block(body_setconst); }
    g_grid = 1;
    return 0;
}
extern "C" unsigned long emu_sizeof_envparams(void) { return sizeof(cm_envparams_t); }
extern "C" int emu_phys_run(const cm_model_t *model, int nenv, int nsub, int integrate, double *qpos, double *qvel,
                            double *qacc_warmstart, double *time, const double *ctrl, const double *qfrc_applied,
                            const double *xfrc_applied, double *qacc, double *sensordata, double *actuator_velocity,
                            int *warn, int *info, double *xpos_out, double *xquat_out, const double *pd_ptarget,
                            const double *pd_kp, const double *pd_kd, const float *hfield) {
    /* (tests edit compiled models field by field: the top-level arrays are the authority, as in phys_batch_create / _set_model) */
    static cm_model_t synced;
    synced = *model; cm_model_sync_params(&synced); model = &synced;
    memset(&g_io, 0, sizeof g_io);
    g_io.models = model; g_io.model_stride = 0; g_io.envparams = g_envparams;
    g_io.nenv = nenv; g_io.nsub = nsub; g_io.integrate = integrate;
    g_io.sq = model->nq; g_io.sqv = model->nv; g_io.sv = model->nv; g_io.su = model->nu; g_io.ssd = model->nsensordata; g_io.sb = model->nbody;
    g_io.qpos = qpos; g_io.qvel = qvel; g_io.qacc_warmstart = qacc_warmstart; g_io.time = time;
    g_io.ctrl = (double *)ctrl; g_io.qfrc_applied = qfrc_applied; g_io.xfrc_applied = xfrc_applied;
    g_io.qacc = qacc; g_io.sensordata = sensordata; g_io.actuator_velocity = actuator_velocity;
    g_io.warn = warn; g_io.info = info; g_io.xpos_out = xpos_out; g_io.xquat_out = xquat_out;
    g_io.hfield = hfield;
    g_io.hfield_stride = 0;
    g_io.pd_ptarget = pd_ptarget; g_io.pd_kp = pd_kp; g_io.pd_kd = pd_kd;
    g_io.drive_mode = g_drive_mode; g_io.drive_state = g_drive_state; g_io.drive_cmd = g_drive_cmd; g_io.meas = g_meas;
    g_io.pd_dtarget = g_pd_dtarget; g_io.pd_torque = g_pd_torque;
    const bool cassie32 = !g_force_runtime_topology && topo_matches(model, ck::TopoCassie32::table, ck::TopoCassie32::nv, ck::TopoCassie32::body_levels);
    const bool tray38 = !cassie32 && !g_force_runtime_topology && model->nhfpair == 0 && model->hfield_geom < 0 &&
                        topo_matches(model, ck::TopoCassieTray38::table, ck::TopoCassieTray38::nv, ck::TopoCassieTray38::body_levels);
    if ((cassie32 || tray38) && g_fast_rows && integrate && nenv <= (1 << 16)) {
        /* as phys_batch.hip launches them: the row-capped fast instantiation for every env (it appends the envs it hands over to
         * the hand-over list), then the full instantiation as ONE small grid walking that list (here: g_resume_grid workgroups) */
        static int progress[1 << 16], list[1 << 16], count[2], list2[1 << 16], count2[2];
        static volatile int seen, seen2;
        count[0] = count[1] = 0; seen = -1; count2[0] = count2[1] = 0; seen2 = -1;
        /* the fast kernel appends to `list`; the 63-row pass walks it and -- for models that may use 127 rows -- appends what it cannot
         * hold to `list2`, which the 127-row pass walks */
        const bool third = cassie32 && model->maxefc > ck::MID_ROWS;
        g_io.progress = progress; g_io.resume = 0; g_io.has_next = 1;
        g_io.handover_list = nullptr; g_io.handover_count = nullptr; g_io.handover_seen = nullptr;
        g_io.handover_out_list = list; g_io.handover_out_count = count;
        /* the in-place form (cassie_step_kernel's INROWS): no first list, no 63-row pass; the inner 63-row call appends to the second list */
        const bool inplace = g_inplace && cassie32 && g_two_waves;
        if (inplace) {
            g_io.handover_out_list = nullptr; g_io.handover_out_count = nullptr;
            g_io.inplace_has_next = third ? 1 : 0; g_io.inplace_out_list = third ? list2 : nullptr; g_io.inplace_out_count = third ? count2 : nullptr;
            g_io.inplace_stay_rows = g_inplace_stay;
        }
        static int chunk_flag[1 << 16];
        const int nchunk = (g_chunks > 1 && nsub >= 2) ? g_chunks : 1;
        g_io.nchunk = nchunk; g_io.chunk_seq = ++g_chunk_seq; g_io.chunk_flag = chunk_flag; g_io.chunk_fault = &g_chunk_fault;
        g_grid = nenv * nchunk;
        for (int wg = 0; wg < nenv * nchunk; ++wg) {
            g_env = wg;
            if (tray38) run_block(body40s_fast);
            else if (inplace) run_block(body32s_fast_2w_inplace, 2);
            else if (g_two_waves) run_block(body32s_fast_2w, 2); else run_block(body32s_fast);
        }
        for (int e = 0; e < nenv; ++e) if (progress[e] < nsub) ++g_fast_bails;
        g_io.nchunk = 1;
        const int handed = count[0];
        g_io.resume = 1; g_io.has_next = third ? 1 : 0;
        g_io.handover_list = list; g_io.handover_count = count; g_io.handover_seen = &seen;
        g_io.handover_out_list = third ? list2 : nullptr; g_io.handover_out_count = third ? count2 : nullptr;
        g_grid = g_resume_grid;
        for (int wg = 0; wg < g_resume_grid && !inplace; ++wg) {
            g_env = wg;
            if (tray38) { if (g_two_waves) run_block(body40s_2w_walk, 2); else run_block(body40s_walk); }
            else if (g_two_waves) run_block(body32s_2w_walk, 2); else run_block(body32s_walk);
        }
        if (!inplace && (count[0] != 0 || count[1] != 0 || seen != handed)) { fprintf(stderr, "emu: the pass behind the fast kernel left count %d ticket %d seen %d (handed %d)\n", count[0], count[1], (int)seen, handed); abort(); }
        if (third) {
            const int handed2 = count2[0];
            g_wide_envs += handed2;
            g_io.has_next = 0;
            g_io.handover_list = list2; g_io.handover_count = count2; g_io.handover_seen = &seen2;
            g_io.handover_out_list = nullptr; g_io.handover_out_count = nullptr;
            const int grid2 = g_resume_grid > 1 ? g_resume_grid - 1 : 1;
            g_grid = grid2;
            for (int wg = 0; wg < grid2; ++wg) { g_env = wg; run_block(body32s_wide_walk, 2); }
            if (count2[0] != 0 || count2[1] != 0 || seen2 != handed2) { fprintf(stderr, "emu: the 127-row pass left count %d ticket %d seen %d (handed %d)\n", count2[0], count2[1], (int)seen2, handed2); abort(); }
        }
        g_grid = 1;
        g_io.progress = nullptr; g_io.resume = 0; g_io.has_next = 0; g_io.handover_list = nullptr; g_io.handover_count = nullptr; g_io.handover_seen = nullptr;
        return 0;
    }
    for (int e = 0; e < nenv; ++e) {
        g_env = e;
        if (cassie32) {
            /* (alone -- forward / read-out passes, the fast kernel switched off: the 127-row instantiation for models that may use
             * its rows, the 63-row one for per-env models capped there) */
            if (model->maxefc > ck::MID_ROWS) run_block(body32s_wide, 2);
            else if (g_two_waves) run_block(body32s_2w, 2); else run_block(body32s);
        }
        else if (!g_force_runtime_topology && topo_matches(model, ck::TopoCassieTray38::table, ck::TopoCassieTray38::nv, ck::TopoCassieTray38::body_levels)) {
            if (g_two_waves && model->nhfpair == 0 && model->hfield_geom < 0) run_block(body40s_2w, 2); else run_block(body40s);
        }
        else run_block(model->nv <= 32 ? body32 : body40);
    }
    return 0;
}
/* phys_batch_derive on the emulator: a forward pass with the read-out enabled, then the derive kernel, env by env */
static ck::DeriveIO g_dio;
static void body_derive() { ck::cassie_derive_kernel(g_dio); }
extern "C" int emu_derive(const cm_model_t *model, int nenv, double *qpos, double *qvel, double *qacc_warmstart, double *time,
                          const double *ctrl, double *qacc, double *sensordata, double *actuator_velocity, int *warn, int *info,
                          const float *hfield, const int *ids, double *derived, double *qM) {
    static cm_model_t synced;
    synced = *model; cm_model_sync_params(&synced); model = &synced;
    cm_ext_t *ext = (cm_ext_t *)calloc((size_t)nenv, sizeof(cm_ext_t));
    double *xpos = (double *)calloc((size_t)nenv * model->nbody * 3, sizeof(double)), *xquat = (double *)calloc((size_t)nenv * model->nbody * 4, sizeof(double));
    memset(&g_io, 0, sizeof g_io);
    g_io.models = model; g_io.nenv = nenv; g_io.nsub = 1; g_io.integrate = 0; g_io.envparams = g_envparams;
    g_io.sq = model->nq; g_io.sqv = model->nv; g_io.sv = model->nv; g_io.su = model->nu; g_io.ssd = model->nsensordata; g_io.sb = model->nbody;
    g_io.qpos = qpos; g_io.qvel = qvel; g_io.qacc_warmstart = qacc_warmstart; g_io.time = time; g_io.ctrl = (double *)ctrl;
    g_io.qacc = qacc; g_io.sensordata = sensordata; g_io.actuator_velocity = actuator_velocity; g_io.warn = warn; g_io.info = info;
    g_io.xpos_out = xpos; g_io.xquat_out = xquat; g_io.hfield = hfield; g_io.ext = ext;
    memset(&g_dio, 0, sizeof g_dio);
    g_dio.models = model; g_dio.nenv = nenv; g_dio.envparams = g_envparams; g_dio.ext = ext; g_dio.xpos = xpos; g_dio.xquat = xquat; g_dio.derived = derived; g_dio.qM = qM;
    for (int i = 0; i < 6; ++i) g_dio.ids[i] = ids[i];
    for (int e = 0; e < nenv; ++e) {
        g_env = e;
        if (topo_matches(model, ck::TopoCassie32::table, ck::TopoCassie32::nv, ck::TopoCassie32::body_levels)) { if (g_two_waves) run_block(body32s_2w, 2); else run_block(body32s); }
        else if (topo_matches(model, ck::TopoCassieTray38::table, ck::TopoCassieTray38::nv, ck::TopoCassieTray38::body_levels)) run_block(body40s);
        else run_block(model->nv <= 32 ? body32 : body40);
        run_block(body_derive);
    }
    free(ext); free(xpos); free(xquat);
    return 0;
}
/* cassie_core_sim's safety layer as the step kernel computes it (csrc/pk_safety.h), sample by sample and drive by drive: the
 * ten torques and the message bits of n samples (u, q, w, L: [n][10]; sto: [n]) */
extern "C" void emu_core_safety(int n, const double *u, const double *q, const double *w, const double *L, const unsigned char *sto,
                                double *tau_out, int *msg_out) {
    for (int s = 0; s < n; ++s) {
        int msg = 0;
        for (int k = 0; k < 10; ++k)
            tau_out[10 * s + k] = ck::safety::drive_torque(k, u[10 * s + k], q + 10 * s, w[10 * s + k], L[10 * s + k], sto[s] != 0, &msg);
        msg_out[s] = msg;
    }
}
extern "C" double emu_core_safety_torque_limit(int k) { return ck::safety::torque_limit(k); }
/* the kinematics stage's own elementary functions, for direct accuracy tests */
extern "C" void emu_sincos_reduced(double x, double *s, double *c) { ck::sincos_reduced(x, *s, *c); }
extern "C" void emu_normalize4_fast(double *q) { ck::normalize4_fast(q); }
extern "C" double emu_normalize3_fast(double *a) { return ck::normalize3_fast(a); }
extern "C" unsigned long emu_sizeof_shared32(void) { return sizeof(ck::EnvShared<32>); }

/* packed factor rows (ck::LPack): the run-time-lane addressing agrees with the compile-time slots; returns the number of mismatches */
template <class TOPO, int NVP>
static int lpack_mismatches() {
    typedef ck::LPack<TOPO, NVP> LP;
    int bad = 0;
    for (int k = 0; k < NVP; ++k) {
        const typename LP::Row r = LP::row_of(k);
        for (int i = 0; i < NVP; ++i) {
            const bool has = LP::has(k, i);
            const int want = has ? LP::idx(k, i) : -1;
            if (has) {
                bad += LP::row_slot(k, i) != want;
                bad += !(LP::row_has(r, i) && LP::row_idx(r, i) == want);
                bad += !(LP::col_has(k, i) && LP::col_idx(k, i) == want);
            } else {
                if (LP::packed) bad += LP::row_slot(k, i) != LP::dump;
                if (i < k) bad += LP::row_has(r, i) || LP::col_has(k, i);
            }
        }
    }
    return bad;
}
extern "C" int emu_lpack_check(void) {
    return lpack_mismatches<ck::TopoCassieTray38, 40>() + lpack_mismatches<ck::TopoCassie32, 32>() + lpack_mismatches<ck::TopoRuntime, 40>();
}
extern "C" int emu_lpack_count(int which) { return which ? ck::LPack<ck::TopoCassieTray38, 40>::count : ck::LPack<ck::TopoCassie32, 32>::count; }
