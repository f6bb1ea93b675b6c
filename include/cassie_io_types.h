/*
 * cassie_io_types.h -- the frozen I/O struct ABI of the Cassie simulator.
 *
 * Layout-identical restatement (x86-64 SysV) of the five bus structs the reference
 * exchanges with user code: cassie_out_t (1336 B), cassie_in_t (192 B),
 * cassie_user_in_t (104 B), pd_in_t (952 B), state_out_t (992 B) -- reference
 * include/cassie_out_t.h:27-109, cassie_in_t.h:24-52, cassie_user_in_t.h:24-27,
 * pd_in_t.h:24-49, state_out_t.h:24-78.  Field names and order are the ABI: user code,
 * the Agility blocks (libagilitycassie.a) and the ctypes wrapper all depend on them.
 * tests/test_abi.py checks sizes and offsets.
 */
#ifndef CASSIE_IO_TYPES_H
#define CASSIE_IO_TYPES_H

#include <stdbool.h>

/* packed (float32 wire) lengths, reference include/\*_t.h:20 */
#define CASSIE_OUT_T_PACKED_LEN 697
#define CASSIE_IN_T_PACKED_LEN 91
#define CASSIE_USER_IN_T_PACKED_LEN 58
#define PD_IN_T_PACKED_LEN 476
#define STATE_OUT_T_PACKED_LEN 493

typedef short DiagnosticCodes;

/* ------------------------------------------------------------ cassie_out_t --- */
typedef struct { bool dataGood; double stateOfCharge; double voltage[12]; double current; double temperature[4]; } battery_out_t;
typedef struct { double position; double velocity; } cassie_joint_out_t;
typedef struct {
    unsigned short statusWord;
    double position, velocity, torque, driveTemperature, dcLinkVoltage, torqueLimit, gearRatio;
} elmo_out_t;
typedef struct {
    elmo_out_t hipRollDrive, hipYawDrive, hipPitchDrive, kneeDrive, footDrive;
    cassie_joint_out_t shinJoint, tarsusJoint, footJoint;
    unsigned char medullaCounter;
    unsigned short medullaCpuLoad;
    bool reedSwitchState;
} cassie_leg_out_t;
typedef struct { bool radioReceiverSignalGood; bool receiverMedullaSignalGood; double channel[16]; } radio_out_t;
typedef struct {
    int etherCatStatus[6];
    int etherCatNotifications[21];
    double taskExecutionTime;
    unsigned int overloadCounter;
    double cpuTemperature;
} target_pc_out_t;
typedef struct {
    bool dataGood;
    unsigned short vpeStatus;
    double pressure, temperature;
    double magneticField[3], angularVelocity[3], linearAcceleration[3], orientation[4];
} vectornav_out_t;
typedef struct {
    target_pc_out_t targetPc;
    battery_out_t battery;
    radio_out_t radio;
    vectornav_out_t vectorNav;
    unsigned char medullaCounter;
    unsigned short medullaCpuLoad;
    bool bleederState, leftReedSwitchState, rightReedSwitchState;
    double vtmTemperature;
} cassie_pelvis_out_t;
typedef struct {
    cassie_pelvis_out_t pelvis;
    cassie_leg_out_t leftLeg, rightLeg;
    bool isCalibrated;
    DiagnosticCodes messages[4];
} cassie_out_t;

/* ------------------------------------------------------------- cassie_in_t --- */
typedef struct { unsigned short controlWord; double torque; } elmo_in_t;
typedef struct { elmo_in_t hipRollDrive, hipYawDrive, hipPitchDrive, kneeDrive, footDrive; } cassie_leg_in_t;
typedef struct { short channel[14]; } radio_in_t;
typedef struct { radio_in_t radio; bool sto; bool piezoState; unsigned char piezoTone; } cassie_pelvis_in_t;
typedef struct { cassie_pelvis_in_t pelvis; cassie_leg_in_t leftLeg, rightLeg; } cassie_in_t;

/* -------------------------------------------------------- cassie_user_in_t --- */
typedef struct { double torque[10]; short telemetry[9]; } cassie_user_in_t;

/* ----------------------------------------------------------------- pd_in_t --- */
typedef struct { double torque[5], pTarget[5], dTarget[5], pGain[5], dGain[5]; } pd_motor_in_t;
typedef struct { double torque[6], pTarget[6], dTarget[6], pGain[6], dGain[6]; } pd_task_in_t;
typedef struct { pd_task_in_t taskPd; pd_motor_in_t motorPd; } pd_leg_in_t;
typedef struct { pd_leg_in_t leftLeg, rightLeg; double telemetry[9]; } pd_in_t;

/* ------------------------------------------------------------- state_out_t --- */
typedef struct { double stateOfCharge; double current; } state_battery_out_t;
typedef struct {
    double position[3], orientation[4], footRotationalVelocity[3], footTranslationalVelocity[3];
    double toeForce[3], heelForce[3];
} state_foot_out_t;
typedef struct { double position[6], velocity[6]; } state_joint_out_t;
typedef struct { double position[10], velocity[10], torque[10]; } state_motor_out_t;
typedef struct {
    double position[3], orientation[4], rotationalVelocity[3], translationalVelocity[3];
    double translationalAcceleration[3], externalMoment[3], externalForce[3];
} state_pelvis_out_t;
typedef struct { double channel[16]; bool signalGood; } state_radio_out_t;
typedef struct { double height; double slope[2]; } state_terrain_out_t;
typedef struct {
    state_pelvis_out_t pelvis;
    state_foot_out_t leftFoot, rightFoot;
    state_terrain_out_t terrain;
    state_motor_out_t motor;
    state_joint_out_t joint;
    state_radio_out_t radio;
    state_battery_out_t battery;
} state_out_t;

/* sensor filter state exposed through cassie_sim_{drive,joint}_filter (reference src/cassiemujoco.c:210-217) */
typedef struct drive_filter { int x[9]; } drive_filter_t;
typedef struct joint_filter { double x[4]; double y[3]; } joint_filter_t;

#ifdef __cplusplus
extern "C" {
#endif
/* float32 wire (de)serialisers -- implemented by the Agility static library that is linked into the
 * product (reference src/libagilitycassie.a: cassie_out_t.o, cassie_in_t.o, ...) */
void pack_cassie_out_t(const cassie_out_t *bus, unsigned char *bytes);
void unpack_cassie_out_t(const unsigned char *bytes, cassie_out_t *bus);
void pack_cassie_in_t(const cassie_in_t *bus, unsigned char *bytes);
void unpack_cassie_in_t(const unsigned char *bytes, cassie_in_t *bus);
void pack_cassie_user_in_t(const cassie_user_in_t *bus, unsigned char *bytes);
void unpack_cassie_user_in_t(const unsigned char *bytes, cassie_user_in_t *bus);
void pack_pd_in_t(const pd_in_t *bus, unsigned char *bytes);
void unpack_pd_in_t(const unsigned char *bytes, pd_in_t *bus);
void pack_state_out_t(const state_out_t *bus, unsigned char *bytes);
void unpack_state_out_t(const unsigned char *bytes, state_out_t *bus);

/* the three closed Agility blocks (reference include/pd_input.h:30-35, cassie_core_sim.h:30-35,
 * state_output.h:29-34): Simulink-style alloc / copy / free / setup / step on opaque state */
typedef struct PdInput pd_input_t;
typedef struct CassieCoreSim cassie_core_sim_t;
typedef struct StateOutput state_output_t;
pd_input_t *pd_input_alloc(void);
void pd_input_copy(pd_input_t *dst, const pd_input_t *src);
void pd_input_free(pd_input_t *sys);
void pd_input_setup(pd_input_t *sys);
void pd_input_step(pd_input_t *sys, const pd_in_t *in1, const cassie_out_t *in2, cassie_user_in_t *out1);
cassie_core_sim_t *cassie_core_sim_alloc(void);
void cassie_core_sim_copy(cassie_core_sim_t *dst, const cassie_core_sim_t *src);
void cassie_core_sim_free(cassie_core_sim_t *sys);
void cassie_core_sim_setup(cassie_core_sim_t *sys);
void cassie_core_sim_step(cassie_core_sim_t *sys, const cassie_user_in_t *in1, const cassie_out_t *in2, cassie_in_t *out1);
state_output_t *state_output_alloc(void);
void state_output_copy(state_output_t *dst, const state_output_t *src);
void state_output_free(state_output_t *sys);
void state_output_setup(state_output_t *sys);
void state_output_step(state_output_t *sys, const cassie_out_t *in1, state_out_t *out1);
#ifdef __cplusplus
}
#endif
#endif
