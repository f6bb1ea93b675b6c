/*
 * mj_harness.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Opportunistic pin of the physics against the TRUE reference.
 *
 * The reference reaches its physics by dlopen'ing MuJoCo 2.1.0 (reference src/cassiemujoco.c:521-555,
 * ~/.mujoco/mujoco210/bin/libmujoco210.so) and calling mj_step1 / mj_step2 (reference :1130-1134).  That binary is
 * not in /root/reference and not installable in the build container, so this file cannot be compiled there: it is
 * compiled AT RUN TIME by tests/mujoco_ref.py, and only when a mujoco210 directory (bin/ + include/mujoco.h) is found
 * on the machine, into the git-ignored oracle/_ref/.  It then drives the genuine mj_step1 + mj_step2 on the inputs
 * the oracle and the HIP kernel get, one simulator per handle.
 */
#include <stdlib.h>
#include <string.h>

#include "mujoco.h"

typedef struct mjh { mjModel *m; mjData *d; } mjh_t;

mjh_t *mjh_load(const char *xml, char *err, int errlen)
{
    mjModel *m = mj_loadXML(xml, NULL, err, errlen);
    if (!m) return NULL;
    mjh_t *h = calloc(1, sizeof *h);
    h->m = m;
    h->d = mj_makeData(m);
    return h;
}
void mjh_free(mjh_t *h) { if (h) { mj_deleteData(h->d); mj_deleteModel(h->m); free(h); } }
/* out[8]: nq nv nu nbody nsensordata njnt ngeom nhfielddata */
void mjh_sizes(const mjh_t *h, int *out)
{
    out[0] = h->m->nq; out[1] = h->m->nv; out[2] = h->m->nu; out[3] = h->m->nbody; out[4] = h->m->nsensordata;
    out[5] = h->m->njnt; out[6] = h->m->ngeom; out[7] = h->m->nhfield > 0 ? h->m->hfield_nrow[0] * h->m->hfield_ncol[0] : 0;
}
const char *mjh_version(void) { static char v[32]; int n = mj_version(); v[0] = 0; strcat(v, n == 210 ? "2.1.0" : "other"); return v; }
int mjh_version_number(void) { return mj_version(); }
void mjh_set_hfield(mjh_t *h, const float *data, int n) { if (h->m->nhfield > 0) memcpy(h->m->hfield_data, data, sizeof(float) * (size_t)n); }
void mjh_reset(mjh_t *h) { mj_resetData(h->m, h->d); }
void mjh_set_state(mjh_t *h, const double *qpos, const double *qvel, const double *qacc_warmstart)
{
    memcpy(h->d->qpos, qpos, sizeof(double) * h->m->nq);
    memcpy(h->d->qvel, qvel, sizeof(double) * h->m->nv);
    if (qacc_warmstart) memcpy(h->d->qacc_warmstart, qacc_warmstart, sizeof(double) * h->m->nv);
}
void mjh_forward(mjh_t *h) { mj_forward(h->m, h->d); }
/* one reference physics step with `ctrl` held: the call pair of reference src/cassiemujoco.c:1132-1133 */
void mjh_step(mjh_t *h, const double *ctrl, const double *qfrc_applied, const double *xfrc_applied)
{
    memcpy(h->d->ctrl, ctrl, sizeof(double) * h->m->nu);
    if (qfrc_applied) memcpy(h->d->qfrc_applied, qfrc_applied, sizeof(double) * h->m->nv);
    if (xfrc_applied) memcpy(h->d->xfrc_applied, xfrc_applied, sizeof(double) * 6 * h->m->nbody);
    mj_step1(h->m, h->d);
    mj_step2(h->m, h->d);
}
/* counts[3]: ncon, nefc, solver iterations of the last step */
void mjh_get(const mjh_t *h, double *qpos, double *qvel, double *qacc, double *sensordata, double *actuator_velocity, int *counts)
{
    if (qpos) memcpy(qpos, h->d->qpos, sizeof(double) * h->m->nq);
    if (qvel) memcpy(qvel, h->d->qvel, sizeof(double) * h->m->nv);
    if (qacc) memcpy(qacc, h->d->qacc, sizeof(double) * h->m->nv);
    if (sensordata) memcpy(sensordata, h->d->sensordata, sizeof(double) * h->m->nsensordata);
    if (actuator_velocity) memcpy(actuator_velocity, h->d->actuator_velocity, sizeof(double) * h->m->nu);
    if (counts) { counts[0] = h->d->ncon; counts[1] = h->d->nefc; counts[2] = h->d->solver_iter; }
}
/* the constants the restatement has to guess ("least certain"): dumped for a field-by-field diff */
void mjh_get_consts(const mjh_t *h, double *body_invweight0 /* [nbody][2] */, double *dof_invweight0 /* [nv] */, double *meaninertia)
{
    memcpy(body_invweight0, h->m->body_invweight0, sizeof(double) * 2 * h->m->nbody);
    memcpy(dof_invweight0, h->m->dof_invweight0, sizeof(double) * h->m->nv);
    *meaninertia = h->m->stat.meaninertia;
}
