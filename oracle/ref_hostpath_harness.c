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
