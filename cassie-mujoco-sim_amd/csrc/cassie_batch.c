/*
 * cassie_batch.c -- N environments behind the reference's step semantics
 * (cassie_sim_step_pd / cassie_sim_step / cassie_sim_step_ethercat, reference
 * src/cassiemujoco.c:1115-1157), batched:
 *
 *   host threads (one contiguous slice of envs each, so an env's 8 KB of block state stays in one
 *   core's cache):      pd_input -> cassie_core_sim -> motor model -> encoder models      [pre]
 *   HIP stream:         ctrl H2D -> physics kernel (all envs) -> sensordata / actuator_velocity D2H
 *   host threads:       state_output (the estimator only needs the pre-step measurement, so it runs
 *                       while the kernel is in flight)                                       [post]
 *
 * Per env the sequence of operations and therefore every output is identical to a stand-alone
 * cassie_sim_t driven with the same inputs.
 */
#define _GNU_SOURCE
#include <linux/futex.h>
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <sys/syscall.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "cassie_batch.h"

static const double qpos_nominal_joints[28] = {
    0.0045, 0, 0.4973, 0.9785, -0.0164, 0.01787, -0.2049, -1.1997, 0, 1.4267, 0, -1.5244, 1.5244, -1.5968,
    -0.0045, 0, 0.4973, 0.9786, 0.00386, -0.01524, -0.2051, -1.1997, 0, 1.4267, 0, -1.5244, 1.5244, -1.5968};

/* cores this process may really use: the smaller of its affinity mask and its cgroup CPU quota */
int cassie_host_cpu_count(void)
{
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) { int c = CPU_COUNT(&set); if (c > 0 && c < n) n = c; }
    FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (f) {
        char q[64];
        long period = 0;
        if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            long quota = atol(q), c = (quota + period - 1) / period;
            if (c > 0 && c < n) n = c;
        }
        fclose(f);
    } else if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"))) {
        long quota = -1, period = 0;
        if (fscanf(f, "%ld", &quota) != 1) quota = -1;
        fclose(f);
        FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        if (g) { if (fscanf(g, "%ld", &period) != 1) period = 0; fclose(g); }
        if (quota > 0 && period > 0) { long c = (quota + period - 1) / period; if (c > 0 && c < n) n = c; }
    }
    return n > 0 ? (int)n : 1;
}

typedef void (*job_fn)(struct cassie_batch *b, int e0, int e1);

/* Workers wait for the next job by spinning for a short while (the pre / post phases of one step and consecutive steps
 * of a tight loop follow each other within tens of microseconds) and then SLEEP on the generation counter (futex), so
 * an idle batch -- policy inference, a long fused launch -- does not burn nthreads - 1 host cores. */
#define POOL_SPINS 20000
static void futex_wait(atomic_int *addr, int expected) { syscall(SYS_futex, (int *)addr, FUTEX_WAIT_PRIVATE, expected, NULL, NULL, 0); }
static void futex_wake_all(atomic_int *addr) { syscall(SYS_futex, (int *)addr, FUTEX_WAKE_PRIVATE, 0x7fffffff, NULL, NULL, 0); }

struct cassie_batch {
    phys_model_t *m;
    cm_model_t pod;
    cassie_hostmodel_t hm;
    phys_batch_t *pb;
    int nenv, nthreads, nsd, nu;
    int mjsteps;                        /* physics steps per control step: round(5e-4 / timestep), reference :1128-1131 */
    cassie_hostenv_t **env;
    double *sensordata, *actvel, *ctrl; /* pinned host mirrors [nenv][dim] */
    int device_drives;                  /* encoder / motor models on the device (cassie_batch_set_device_drives) */
    double *cmd, *meas;                 /* pinned [nenv][11] commands up, [nenv][CM_MEAS_DIM] measurements down */
    cassie_out_t *ytmp;
    /* fork-join pool */
    pthread_t *threads;
    int *tid_arg;
    atomic_int generation, done, quit, sleepers;
    job_fn job;
    /* per-call arguments */
    const pd_in_t *u_pd;
    const cassie_user_in_t *u_user;
    const cassie_in_t *u_in;
    state_out_t *y_state;
    cassie_out_t *y_out;
};

static void slice(const struct cassie_batch *b, int t, int *e0, int *e1)
{
    long n = b->nenv, T = b->nthreads;
    *e0 = (int)(n * t / T);
    *e1 = (int)(n * (t + 1) / T);
}

static void *worker(void *arg)
{
    struct cassie_batch *b = ((void **)arg)[0];
    int tid = (int)(long)((void **)arg)[1];
    free(arg);
    int seen = 0;
    for (;;) {
        int spins = 0;
        while (atomic_load_explicit(&b->generation, memory_order_acquire) == seen) {
            if (atomic_load_explicit(&b->quit, memory_order_relaxed)) return NULL;
            if (++spins > POOL_SPINS) {
                atomic_fetch_add_explicit(&b->sleepers, 1, memory_order_seq_cst);
                if (atomic_load_explicit(&b->generation, memory_order_seq_cst) == seen && !atomic_load_explicit(&b->quit, memory_order_seq_cst))
                    futex_wait(&b->generation, seen);
                atomic_fetch_sub_explicit(&b->sleepers, 1, memory_order_seq_cst);
                spins = 0;
            } else {
                __builtin_ia32_pause();
            }
        }
        seen = atomic_load_explicit(&b->generation, memory_order_acquire);
        /* teardown bumps the generation too (so that a worker that was about to sleep on the old value does not): it is
         * not a job */
        if (atomic_load_explicit(&b->quit, memory_order_seq_cst)) return NULL;
        int e0, e1;
        slice(b, tid, &e0, &e1);
        b->job(b, e0, e1);
        atomic_fetch_add_explicit(&b->done, 1, memory_order_release);
    }
}

static void run_parallel(struct cassie_batch *b, job_fn fn)
{
    b->job = fn;
    atomic_store_explicit(&b->done, 0, memory_order_relaxed);
    atomic_fetch_add_explicit(&b->generation, 1, memory_order_seq_cst);
    if (atomic_load_explicit(&b->sleepers, memory_order_seq_cst) > 0) futex_wake_all(&b->generation);
    int e0, e1;
    slice(b, 0, &e0, &e1);
    fn(b, e0, e1);
    int spins = 0;
    while (atomic_load_explicit(&b->done, memory_order_acquire) < b->nthreads - 1)
        if (++spins > 4000) { sched_yield(); spins = 0; }
}

static void job_pd_pre(struct cassie_batch *b, int e0, int e1)
{
    for (int e = e0; e < e1; ++e)
        cassie_hostenv_step_pd_pre(b->env[e], &b->hm, &b->u_pd[e], b->sensordata + (size_t)e * b->nsd,
                                   b->actvel + (size_t)e * b->nu, b->ctrl + (size_t)e * b->nu, &b->ytmp[e]);
}
static void job_pd_post(struct cassie_batch *b, int e0, int e1)
{
    for (int e = e0; e < e1; ++e) cassie_hostenv_step_pd_post(b->env[e], &b->ytmp[e], &b->y_state[e]);
}
static void job_user(struct cassie_batch *b, int e0, int e1)
{
    for (int e = e0; e < e1; ++e)
        cassie_hostenv_step(b->env[e], &b->hm, &b->u_user[e], b->sensordata + (size_t)e * b->nsd,
                            b->actvel + (size_t)e * b->nu, b->ctrl + (size_t)e * b->nu, &b->y_out[e]);
}
static void job_ethercat(struct cassie_batch *b, int e0, int e1)
{
    for (int e = e0; e < e1; ++e)
        cassie_hostenv_ethercat(b->env[e], &b->hm, &b->u_in[e], b->sensordata + (size_t)e * b->nsd,
                                b->actvel + (size_t)e * b->nu, b->ctrl + (size_t)e * b->nu, &b->y_out[e]);
}

/* --- device-drive mode: commands up, drive-level pass, measurements down, then the physics --- */
static void job_cmd_pd(struct cassie_batch *b, int e0, int e1)
{
    for (int e = e0; e < e1; ++e) cassie_hostenv_command_pd(b->env[e], &b->u_pd[e], b->cmd + (size_t)e * 11);
}
static void job_cmd_user(struct cassie_batch *b, int e0, int e1)
{
    for (int e = e0; e < e1; ++e) cassie_hostenv_command(b->env[e], &b->u_user[e], b->cmd + (size_t)e * 11);
}
static void job_cmd_ethercat(struct cassie_batch *b, int e0, int e1)
{
    for (int e = e0; e < e1; ++e) cassie_hostenv_command_ethercat(b->env[e], &b->u_in[e], b->cmd + (size_t)e * 11);
}
static void job_meas_pd(struct cassie_batch *b, int e0, int e1)
{
    for (int e = e0; e < e1; ++e) {
        cassie_hostenv_apply_meas(b->env[e], b->meas + (size_t)e * CM_MEAS_DIM, &b->ytmp[e]);
        cassie_hostenv_step_pd_post(b->env[e], &b->ytmp[e], &b->y_state[e]);
    }
}
static void job_meas_out(struct cassie_batch *b, int e0, int e1)
{
    for (int e = e0; e < e1; ++e) cassie_hostenv_apply_meas(b->env[e], b->meas + (size_t)e * CM_MEAS_DIM, &b->y_out[e]);
}
static int launch_device_drives(struct cassie_batch *b)
{
    int rc = phys_batch_upload_async(b->pb, PHYS_F_DRIVE_CMD, b->cmd, 0, b->nenv);
    rc |= phys_batch_drive_pass(b->pb, CM_DRIVE_TORQUE, NULL);
    rc |= phys_batch_download_async(b->pb, PHYS_F_MEAS, b->meas, 0, b->nenv);
    rc |= phys_batch_mark(b->pb);
    rc |= phys_batch_step(b->pb, 1, NULL);          /* reads the ctrl the drive pass left in HBM */
    rc |= phys_batch_wait_mark(b->pb);              /* measurements are on the host; the physics kernel is still running */
    return rc;
}

/* ctrl up, one physics step for every env, measurements for the next step down -- all asynchronous */
static int launch_physics(struct cassie_batch *b)
{
    int rc = phys_batch_upload_async(b->pb, PHYS_F_CTRL, b->ctrl, 0, b->nenv);
    rc |= phys_batch_step(b->pb, b->mjsteps, NULL);
    rc |= phys_batch_download_async(b->pb, PHYS_F_SENSORDATA, b->sensordata, 0, b->nenv);
    rc |= phys_batch_download_async(b->pb, PHYS_F_ACTUATOR_VELOCITY, b->actvel, 0, b->nenv);
    return rc;
}

cassie_batch_t *cassie_batch_create(const char *modelfile, int nenv, int device, int nthreads)
{
    char err[512] = "";
    if (nenv <= 0) return NULL;
    struct cassie_batch *b = calloc(1, sizeof *b);
    if (!b) return NULL;
    b->m = phys_model_load(modelfile, err, sizeof err);
    if (!b->m) { fprintf(stderr, "cassie_batch_create: %s\n", err); free(b); return NULL; }
    if (phys_model_compile(b->m, &b->pod, err, sizeof err) != 0 || cassie_hostmodel_from_model(b->m, &b->hm) != 0) {
        fprintf(stderr, "cassie_batch_create: model not usable: %s\n", err);
        cassie_batch_free(b);
        return NULL;
    }
    b->nenv = nenv; b->nsd = b->pod.nsensordata; b->nu = b->pod.nu;
    b->mjsteps = (int)round(5e-4 / b->pod.timestep);
    if (b->mjsteps < 1) b->mjsteps = 1;
    b->pb = phys_batch_create(&b->pod, nenv, device);
    if (!b->pb) { fprintf(stderr, "cassie_batch_create: %s\n", phys_last_error()); cassie_batch_free(b); return NULL; }
    b->sensordata = phys_host_alloc(sizeof(double) * (size_t)nenv * b->nsd);
    b->actvel = phys_host_alloc(sizeof(double) * (size_t)nenv * b->nu);
    b->ctrl = phys_host_alloc(sizeof(double) * (size_t)nenv * b->nu);
    b->ytmp = calloc((size_t)nenv, sizeof(cassie_out_t));
    b->env = calloc((size_t)nenv, sizeof *b->env);
    if (!b->sensordata || !b->actvel || !b->ctrl || !b->ytmp || !b->env) { cassie_batch_free(b); return NULL; }
    for (int e = 0; e < nenv; ++e)
        if (!(b->env[e] = cassie_hostenv_alloc())) { cassie_batch_free(b); return NULL; }
    /* initial state of every env: what cassie_sim_init leaves (reference :1023-1029) */
    double *q = malloc(sizeof(double) * (size_t)nenv * b->pod.nq);
    if (!q) { cassie_batch_free(b); return NULL; }
    for (int e = 0; e < nenv; ++e) {
        memcpy(q + (size_t)e * b->pod.nq, b->pod.qpos0, sizeof(double) * b->pod.nq);
        memcpy(q + (size_t)e * b->pod.nq + 7, qpos_nominal_joints, sizeof qpos_nominal_joints);
    }
    phys_batch_upload(b->pb, PHYS_F_QPOS, q, 0, nenv);
    free(q);
    phys_batch_forward(b->pb, NULL);
    phys_batch_download(b->pb, PHYS_F_SENSORDATA, b->sensordata, 0, nenv);
    phys_batch_download(b->pb, PHYS_F_ACTUATOR_VELOCITY, b->actvel, 0, nenv);

    if (nthreads <= 0) {
        /* all usable cores, shared fairly between the ranks of a one-process-per-GPU job on this node */
        const char *lws = getenv("LOCAL_WORLD_SIZE");
        int ranks = lws ? atoi(lws) : 1;
        nthreads = cassie_host_cpu_count() / (ranks > 0 ? ranks : 1);
        if (nthreads < 1) nthreads = 1;
    }
    if (nthreads > nenv) nthreads = nenv;
    b->nthreads = nthreads;
    b->threads = calloc((size_t)nthreads, sizeof(pthread_t));
    for (int t = 1; t < nthreads; ++t) {
        void **arg = malloc(2 * sizeof(void *));
        arg[0] = b; arg[1] = (void *)(long)t;
        if (pthread_create(&b->threads[t], NULL, worker, arg) != 0) { b->nthreads = t; free(arg); break; }
    }
    return b;
}

void cassie_batch_free(cassie_batch_t *b)
{
    if (!b) return;
    if (b->threads) {
        /* quit first, then a new generation value, then the wake: a worker that read quit == 0 and has not reached
         * futex_wait yet finds the futex word changed and returns from the wait at once; one that is already asleep is
         * woken; either way it re-reads quit before it would run a job */
        atomic_store(&b->quit, 1);
        atomic_fetch_add_explicit(&b->generation, 1, memory_order_seq_cst);
        futex_wake_all(&b->generation);
        for (int t = 1; t < b->nthreads; ++t) pthread_join(b->threads[t], NULL);
        free(b->threads);
    }
    if (b->pb) { phys_batch_sync(b->pb); phys_batch_free(b->pb); }
    if (b->env) { for (int e = 0; e < b->nenv; ++e) cassie_hostenv_free(b->env[e]); free(b->env); }
    phys_host_free(b->sensordata); phys_host_free(b->actvel); phys_host_free(b->ctrl);
    phys_host_free(b->cmd); phys_host_free(b->meas);
    free(b->ytmp);
    if (b->m) phys_model_free(b->m);
    free(b);
}

int cassie_batch_nenv(const cassie_batch_t *b) { return b ? b->nenv : 0; }
int cassie_batch_nthreads(const cassie_batch_t *b) { return b ? b->nthreads : 0; }
phys_batch_t *cassie_batch_phys(cassie_batch_t *b) { return b ? b->pb : NULL; }
phys_model_t *cassie_batch_model(cassie_batch_t *b) { return b ? b->m : NULL; }
cassie_hostenv_t *cassie_batch_hostenv(cassie_batch_t *b, int env) { return (b && env >= 0 && env < b->nenv) ? b->env[env] : NULL; }

int cassie_batch_set_device_drives(cassie_batch_t *b, int on)
{
    if (!b) return -1;
    on = on != 0;
    if (on == b->device_drives) return 0;
    if (on && b->mjsteps != 1) return -1;
    cm_drive_state_t *st = malloc(sizeof(cm_drive_state_t) * (size_t)b->nenv);
    if (!st) return -1;
    int rc = phys_batch_sync(b->pb);
    if (on) {
        if (!b->cmd) b->cmd = phys_host_alloc(sizeof(double) * (size_t)b->nenv * 11);
        if (!b->meas) b->meas = phys_host_alloc(sizeof(double) * (size_t)b->nenv * CM_MEAS_DIM);
        if (!b->cmd || !b->meas) { free(st); return -1; }
        for (int e = 0; e < b->nenv; ++e) cassie_hostenv_get_drive_state(b->env[e], &st[e]);
        rc |= phys_batch_upload_drive_state(b->pb, st, 0, b->nenv);
    } else {
        rc |= phys_batch_download_drive_state(b->pb, st, 0, b->nenv);
        for (int e = 0; rc == 0 && e < b->nenv; ++e) cassie_hostenv_set_drive_state(b->env[e], &st[e]);
        /* the host models read the physics outputs of the last step from the host mirrors */
        rc |= phys_batch_download(b->pb, PHYS_F_SENSORDATA, b->sensordata, 0, b->nenv);
        rc |= phys_batch_download(b->pb, PHYS_F_ACTUATOR_VELOCITY, b->actvel, 0, b->nenv);
    }
    free(st);
    if (rc == 0) b->device_drives = on;
    return rc;
}

int cassie_batch_step_pd(cassie_batch_t *b, const pd_in_t *u, state_out_t *y)
{
    if (!b || !u || !y) return -1;
    b->u_pd = u; b->y_state = y;
    if (b->device_drives) {
        run_parallel(b, job_cmd_pd);
        int rc = launch_device_drives(b);
        run_parallel(b, job_meas_pd);   /* cassie_out from the device's measurements, then the estimator, over the kernel */
        return rc | phys_batch_sync(b->pb);
    }
    run_parallel(b, job_pd_pre);
    int rc = launch_physics(b);
    run_parallel(b, job_pd_post); /* estimator overlaps the kernel and the copies */
    rc |= phys_batch_sync(b->pb);
    return rc;
}

int cassie_batch_step(cassie_batch_t *b, const cassie_user_in_t *u, cassie_out_t *y)
{
    if (!b || !u || !y) return -1;
    b->u_user = u; b->y_out = y;
    if (b->device_drives) {
        run_parallel(b, job_cmd_user);
        int rc = launch_device_drives(b);
        run_parallel(b, job_meas_out);
        return rc | phys_batch_sync(b->pb);
    }
    run_parallel(b, job_user);
    int rc = launch_physics(b);
    rc |= phys_batch_sync(b->pb);
    return rc;
}

int cassie_batch_step_ethercat(cassie_batch_t *b, const cassie_in_t *u, cassie_out_t *y)
{
    if (!b || !u || !y) return -1;
    b->u_in = u; b->y_out = y;
    if (b->device_drives) {
        run_parallel(b, job_cmd_ethercat);
        int rc = launch_device_drives(b);
        run_parallel(b, job_meas_out);
        return rc | phys_batch_sync(b->pb);
    }
    run_parallel(b, job_ethercat);
    int rc = launch_physics(b);
    rc |= phys_batch_sync(b->pb);
    return rc;
}

int cassie_batch_foot_forces(cassie_batch_t *b, double *cfrc)
{
    if (!b || !cfrc) return -1;
    const int nb = b->pod.nbody;
    const int feet[2] = {phys_model_name2id(b->m, 1 /* mjOBJ_BODY */, "left-foot"), phys_model_name2id(b->m, 1, "right-foot")};
    if (feet[0] < 0 || feet[1] < 0 || feet[0] >= nb || feet[1] >= nb) return -1;
    double *all = malloc(sizeof(double) * (size_t)b->nenv * nb * 3);
    if (!all) return -1;
    int rc = phys_batch_download(b->pb, PHYS_F_BODY_CFRC, all, 0, b->nenv);
    for (int e = 0; rc == 0 && e < b->nenv; ++e) {
        double *out = cfrc + (size_t)e * 12;
        memset(out, 0, 12 * sizeof(double));
        for (int side = 0; side < 2; ++side)
            for (int j = 0; j < 3; ++j) out[6 * side + j] = all[((size_t)e * nb + feet[side]) * 3 + j];
    }
    free(all);
    return rc;
}

int cassie_batch_derive(cassie_batch_t *b, double *derived, double *qM)
{
    if (!b) return -1;
    const int ids[6] = {phys_model_name2id(b->m, 1 /* mjOBJ_BODY */, "left-foot"), phys_model_name2id(b->m, 1, "right-foot"),
                        phys_model_name2id(b->m, 6 /* mjOBJ_SITE */, "left-heel"), phys_model_name2id(b->m, 6, "right-heel"),
                        phys_model_name2id(b->m, 6, "left-toe"), phys_model_name2id(b->m, 6, "right-toe")};
    int rc = phys_batch_derive(b->pb, ids, NULL);
    if (derived) rc |= phys_batch_download(b->pb, PHYS_F_DERIVED, derived, 0, b->nenv);
    if (qM) rc |= phys_batch_download(b->pb, PHYS_F_QM, qM, 0, b->nenv);
    rc |= phys_batch_sync(b->pb);
    return rc;
}

/* rows of one field for the envs that are being reset: the whole field comes down once, the masked rows are rewritten
 * (row = the given values, or zeros) and it goes up once -- two transfers per field whatever the number of envs */
static int reset_field(cassie_batch_t *b, int field, const unsigned char *mask, const double *row, double *scratch)
{
    const int dim = phys_batch_field_dim(b->pb, field);
    int rc = mask ? phys_batch_download(b->pb, field, scratch, 0, b->nenv) : 0;
    for (int e = 0; e < b->nenv; ++e) {
        if (mask && !mask[e]) continue;
        if (row) memcpy(scratch + (size_t)e * dim, row, sizeof(double) * (size_t)dim);
        else memset(scratch + (size_t)e * dim, 0, sizeof(double) * (size_t)dim);
    }
    return rc | phys_batch_upload(b->pb, field, scratch, 0, b->nenv);
}

int cassie_batch_full_reset(cassie_batch_t *b, const unsigned char *mask)
{
    /* cassie_sim_full_reset per env (reference :2008-2033): pose, velocities, controls, perturbations, torque delay,
     * estimator; plus the sticky device-side warning bits of the envs that are reset (mj_resetData clears
     * mjData.warning).  Time, filters and the solver warm start are left alone like the single-env version. */
    if (!b) return -1;
    const int nq = b->pod.nq, nb6 = 6 * b->pod.nbody;
    const int widest = nq > nb6 ? nq : nb6;
    double *q = malloc(sizeof(double) * nq), *scratch = malloc(sizeof(double) * (size_t)b->nenv * (size_t)widest);
    if (!q || !scratch) { free(q); free(scratch); return -1; }
    memcpy(q, b->pod.qpos0, sizeof(double) * nq);
    q[0] = 0; q[1] = 0; q[2] = 1.01; q[3] = 1; q[4] = q[5] = q[6] = 0; /* reference :2010 */
    memcpy(q + 7, qpos_nominal_joints, sizeof qpos_nominal_joints);
    int rc = phys_batch_sync(b->pb);
    rc |= reset_field(b, PHYS_F_QPOS, mask, q, scratch);
    rc |= reset_field(b, PHYS_F_QVEL, mask, NULL, scratch);
    rc |= reset_field(b, PHYS_F_CTRL, mask, NULL, scratch);
    rc |= reset_field(b, PHYS_F_QACC, mask, NULL, scratch);
    if (phys_batch_uses_applied(b->pb)) { /* perturbations exist on the device only once somebody uploaded some */
        rc |= reset_field(b, PHYS_F_QFRC_APPLIED, mask, NULL, scratch);
        rc |= reset_field(b, PHYS_F_XFRC_APPLIED, mask, NULL, scratch);
    }
    cm_drive_state_t *st = NULL;
    if (b->device_drives) { /* the torque delay line that cassie_hostenv_reset clears lives in HBM in this mode */
        st = malloc(sizeof(cm_drive_state_t) * (size_t)b->nenv);
        if (!st) { free(q); free(scratch); return -1; }
        rc |= phys_batch_download_drive_state(b->pb, st, 0, b->nenv);
    }
    for (int e = 0; e < b->nenv; ++e) {
        if (mask && !mask[e]) continue;
        int run = 1; /* warning bits: one clear per run of consecutive reset envs */
        while (e + run < b->nenv && (!mask || mask[e + run])) ++run;
        rc |= phys_batch_clear_warn(b->pb, e, run);
        for (int k = e; k < e + run; ++k) {
            memset(b->ctrl + (size_t)k * b->nu, 0, sizeof(double) * b->nu);
            cassie_hostenv_reset(b->env[k]);
            if (st) memset(st[k].torque_delay, 0, sizeof st[k].torque_delay);
        }
        e += run - 1;
    }
    if (st) { rc |= phys_batch_upload_drive_state(b->pb, st, 0, b->nenv); free(st); }
    free(q); free(scratch);
    return rc;
}
