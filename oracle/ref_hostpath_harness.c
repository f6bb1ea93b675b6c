/*
 * ref_hostpath_harness.c -- TEST INFRASTRUCTURE.  Thin C-ABI around the reference's own encoder / motor
 * model functions (pulled in at build time by oracle/build_ref.sh as _ref/ref_extract.inc, never
 * committed) so that golden vectors for the host path can be generated from the real reference code.
 * Only the mjModel / mjData members those functions read or write are declared.
 */
#include <math.h>
#include <stdbool.h>
#include <string.h>

#include "cassie_io_types.h" /* layout-identical restatement of the reference's bus structs */

typedef double mjtNum;
typedef struct {
    int nuser_sensor, nuser_actuator;
    double *sensor_user, *actuator_gear, *actuator_ctrlrange, *actuator_user;
    int *sensor_objid;
} mjModel;
typedef struct {
    double *actuator_velocity, *ctrl;
} mjData;

/* the reference declares these two typedef names itself; rename ours (from cassie_io_types.h) out of the way */
#define drive_filter_t ref_drive_filter_t
#define joint_filter_t ref_joint_filter_t
#define drive_filter ref_drive_filter
#define joint_filter ref_joint_filter
#include "ref_extract.inc"

/* model constants of cassie.xml's sensors / actuators, filled by the caller */
static double g_sensor_user[32], g_gear[6 * 16], g_ctrlrange[2 * 16], g_act_user[16], g_actvel[16], g_ctrl[16];
static int g_objid[32];
static mjModel g_m = {1, 1, g_sensor_user, g_gear, g_ctrlrange, g_act_user, g_objid};
static mjData g_d = {g_actvel, g_ctrl};

void ref_set_model(const double *sensor_bits20, const int *sensor_objid20, const double *gear10, const double *tmax10,
                   const double *rpm10)
{
    for (int i = 0; i < 20; ++i) { g_sensor_user[i] = sensor_bits20[i]; g_objid[i] = sensor_objid20[i]; }
    for (int i = 0; i < 10; ++i) { g_gear[6 * i] = gear10[i]; g_ctrlrange[2 * i] = -tmax10[i]; g_ctrlrange[2 * i + 1] = tmax10[i]; g_act_user[i] = rpm10[i]; }
}

void ref_drive_encoder(elmo_out_t *drive, const double *sensordata, int *filter_x9, int isensor)
{
    drive_encoder(&g_m, drive, sensordata, (ref_drive_filter_t *)filter_x9, isensor);
}
void ref_joint_encoder(cassie_joint_out_t *joint, const double *sensordata, double *filter_x4y3, int isensor)
{
    joint_encoder(&g_m, joint, sensordata, (ref_joint_filter_t *)filter_x4y3, isensor);
}
double ref_motor(int i, double u, double actuator_velocity, double *torque_delay6, bool sto, double *ctrl_out)
{
    g_actvel[i] = actuator_velocity;
    double r = motor(&g_m, &g_d, i, u, torque_delay6, sto);
    *ctrl_out = g_ctrl[i];
    return r;
}
int ref_sizes(int which)
{
    switch (which) {
        case 0: return (int)sizeof(cassie_out_t);
        case 1: return (int)sizeof(ref_drive_filter_t);
        case 2: return (int)sizeof(ref_joint_filter_t);
    }
    return 0;
}

/* cassie_core_sim_step of the closed Agility library on n samples (golden vectors / live comparison of the safety-layer
 * restatement csrc/pk_safety.h): user torques u, measured drive positions q / velocities w, torque limits L ([n][10] each, left
 * leg first), radio channel 8 ([n]), telemetry ([n][9] shorts or NULL); fresh != 0: cassie_core_sim_setup before every sample,
 * else one block instance sees the samples in order (the message queue is sticky).  Out: the ten torques, the 14 radio shorts
 * and the (sto, piezoState, piezoTone) bytes of cassie_in_t, and the controlWords. */
static elmo_out_t *ref_drive_out(cassie_out_t *o, int i) {
    cassie_leg_out_t *l = i < 5 ? &o->leftLeg : &o->rightLeg;
    elmo_out_t *d[5] = {&l->hipRollDrive, &l->hipYawDrive, &l->hipPitchDrive, &l->kneeDrive, &l->footDrive};
    return d[i % 5];
}
static elmo_in_t *ref_drive_in(cassie_in_t *o, int i) {
    cassie_leg_in_t *l = i < 5 ? &o->leftLeg : &o->rightLeg;
    elmo_in_t *d[5] = {&l->hipRollDrive, &l->hipYawDrive, &l->hipPitchDrive, &l->kneeDrive, &l->footDrive};
    return d[i % 5];
}
void ref_core_sim_batch(int n, const double *u, const double *q, const double *w, const double *L, const double *ch8,
                        const short *telemetry, int fresh, double *tau_out, short *radio_out, unsigned char *flags_out,
                        unsigned short *cw_out)
{
    cassie_core_sim_t *c = cassie_core_sim_alloc();
    cassie_core_sim_setup(c);
    for (int s = 0; s < n; ++s) {
        cassie_out_t out;
        cassie_user_in_t uin;
        cassie_in_t in;
        memset(&out, 0, sizeof out); memset(&uin, 0, sizeof uin); memset(&in, 0xA5, sizeof in);
        for (int i = 0; i < 10; ++i) {
            elmo_out_t *d = ref_drive_out(&out, i);
            d->position = q[10 * s + i]; d->velocity = w[10 * s + i]; d->torqueLimit = L[10 * s + i];
            uin.torque[i] = u[10 * s + i];
        }
        out.pelvis.radio.channel[8] = ch8[s];
        if (telemetry) for (int i = 0; i < 9; ++i) uin.telemetry[i] = telemetry[9 * s + i];
        if (fresh) cassie_core_sim_setup(c);
        cassie_core_sim_step(c, &uin, &out, &in);
        for (int i = 0; i < 10; ++i) { tau_out[10 * s + i] = ref_drive_in(&in, i)->torque; cw_out[10 * s + i] = ref_drive_in(&in, i)->controlWord; }
        for (int i = 0; i < 14; ++i) radio_out[14 * s + i] = in.pelvis.radio.channel[i];
        flags_out[3 * s] = in.pelvis.sto; flags_out[3 * s + 1] = in.pelvis.piezoState; flags_out[3 * s + 2] = in.pelvis.piezoTone;
    }
    cassie_core_sim_free(c);
}
