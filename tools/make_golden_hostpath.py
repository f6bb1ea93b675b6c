#!/usr/bin/env python3
"""Generates tests/golden/hostpath_v1.npz from the REFERENCE's own code (oracle/_ref/libref_hostpath.so:
the reference's drive_encoder / joint_encoder / motor functions compiled where they lie, plus the Agility
blocks of src/libagilitycassie.a).  Needs /root/reference; the committed .npz travels to the GPU box.

Per step the chain of reference src/cassiemujoco.c:1147-1157 is driven on synthetic physics outputs:
pd_input_step -> cassie_core_sim_step -> motor x10 -> drive_encoder x10 -> joint_encoder x6 -> IMU copy ->
state_output_step, and the inputs plus every output (ctrl, cassie_out_t bytes, packed bytes, state_out_t
bytes) are stored.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cassie-mujoco-sim_amd"))
from cassie_amd import iotypes as T  # noqa: E402

subprocess.check_call(["bash", os.path.join(REPO, "oracle", "build_ref.sh")])
R = ctypes.CDLL(os.path.join(REPO, "oracle", "_ref", "libref_hostpath.so"))
assert R.ref_sizes(0) == ctypes.sizeof(T.cassie_out_t)
for f in ("pd_input_alloc", "cassie_core_sim_alloc", "state_output_alloc"):
    getattr(R, f).restype = ctypes.c_void_p
R.ref_motor.restype = ctypes.c_double
VP = ctypes.c_void_p
R.ref_set_model.argtypes = [VP] * 5
R.ref_drive_encoder.argtypes = [VP, VP, VP, ctypes.c_int]
R.ref_joint_encoder.argtypes = [VP, VP, VP, ctypes.c_int]
for f in ("pd_input_setup", "cassie_core_sim_setup", "state_output_setup"):
    getattr(R, f).argtypes = [VP]
R.pd_input_step.argtypes = [VP] * 4
R.cassie_core_sim_step.argtypes = [VP] * 4
R.state_output_step.argtypes = [VP] * 3
R.pack_cassie_out_t.argtypes = [VP, VP]
R.ref_motor.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_bool, ctypes.c_void_p]

# model constants of cassie.xml (model/cassie.xml:258-287)
bits = np.array([13, 13, 13, 13, 18, 18, 18, 13, 13, 13, 13, 13, 18, 18, 18, 13, 0, 0, 0, 0], dtype=np.float64)
objid = np.array([0, 1, 2, 3, 4, 9, 10, 14, 5, 6, 7, 8, 9, 20, 21, 25, 0, 0, 0, 0], dtype=np.int32)
gear = np.array([25, 25, 16, 16, 50] * 2, dtype=np.float64)
tmax = np.array([4.5, 4.5, 12.2, 12.2, 0.9] * 2, dtype=np.float64)
rpm = np.array([2900, 2900, 1300, 1300, 5500] * 2, dtype=np.float64)
R.ref_set_model(bits.ctypes.data, objid.ctypes.data, gear.ctypes.data, tmax.ctypes.data, rpm.ctypes.data)

DRIVE_SENSOR = [0, 1, 2, 3, 4, 8, 9, 10, 11, 12]
JOINT_SENSOR = [5, 6, 7, 13, 14, 15]


def drives(o):
    return [o.leftLeg.hipRollDrive, o.leftLeg.hipYawDrive, o.leftLeg.hipPitchDrive, o.leftLeg.kneeDrive, o.leftLeg.footDrive,
            o.rightLeg.hipRollDrive, o.rightLeg.hipYawDrive, o.rightLeg.hipPitchDrive, o.rightLeg.kneeDrive, o.rightLeg.footDrive]


def in_torques(i):
    return [i.leftLeg.hipRollDrive.torque, i.leftLeg.hipYawDrive.torque, i.leftLeg.hipPitchDrive.torque, i.leftLeg.kneeDrive.torque,
            i.leftLeg.footDrive.torque, i.rightLeg.hipRollDrive.torque, i.rightLeg.hipYawDrive.torque, i.rightLeg.hipPitchDrive.torque,
            i.rightLeg.kneeDrive.torque, i.rightLeg.footDrive.torque]


def cassie_out_initial():
    """cassie_out_init of the reference (src/cassiemujoco.c:695-734), restated for the generator."""
    o = T.cassie_out_t()
    o.isCalibrated = True
    p = o.pelvis
    p.medullaCounter, p.medullaCpuLoad, p.vtmTemperature = 1, 159, 40
    p.targetPc.etherCatStatus[1], p.targetPc.etherCatStatus[4] = 8, 1
    p.targetPc.taskExecutionTime, p.targetPc.cpuTemperature = 2e-4, 60
    p.battery.dataGood, p.battery.stateOfCharge = True, 1
    for i in range(4):
        p.battery.temperature[i] = 30
    for i in range(12):
        p.battery.voltage[i] = 4.2
    p.radio.radioReceiverSignalGood = p.radio.receiverMedullaSignalGood = True
    p.radio.channel[8] = 1
    p.vectorNav.dataGood, p.vectorNav.pressure, p.vectorNav.temperature = True, 101.325, 25
    for leg in (o.leftLeg, o.rightLeg):
        leg.medullaCounter, leg.medullaCpuLoad = 1, 94
        for d, tl, gr in ((leg.hipRollDrive, 140.63, 25), (leg.hipYawDrive, 140.63, 25), (leg.hipPitchDrive, 216.16, 16),
                          (leg.kneeDrive, 216.16, 16), (leg.footDrive, 45.14, 50)):
            d.statusWord, d.dcLinkVoltage, d.driveTemperature, d.torqueLimit, d.gearRatio = 0x0637, 48, 30, tl, gr
    return o


def main(nsteps=240, seed=20260925):
    rng = np.random.default_rng(seed)
    pd, core, est = R.pd_input_alloc(), R.cassie_core_sim_alloc(), R.state_output_alloc()
    for f, h in (("pd_input_setup", pd), ("cassie_core_sim_setup", core), ("state_output_setup", est)):
        getattr(R, f)(ctypes.c_void_p(h))
    out = cassie_out_initial()
    dfilt = np.zeros((10, 9), dtype=np.int32)
    jfilt = np.zeros((6, 7), dtype=np.float64)
    delay = np.zeros((10, 6), dtype=np.float64)
    # synthetic physics outputs: smooth joint trajectories around the nominal pose + IMU signals
    q_nom = np.array([0.0045, 0, 0.4973, -1.1997, -1.5968] * 2)
    g = {k: [] for k in ("pd_in", "sensordata", "actvel", "ctrl", "cassie_out", "packed", "state_out")}
    phase = rng.uniform(0, 6.28, 29)
    for t in range(nsteps):
        u = T.pd_in_t()
        for leg, o in ((u.leftLeg, 0), (u.rightLeg, 5)):
            for i in range(5):
                leg.motorPd.pTarget[i] = q_nom[o + i] + 0.3 * np.sin(0.05 * t + i)
                leg.motorPd.pGain[i] = [70, 70, 100, 100, 50][i]
                leg.motorPd.dGain[i] = [7, 7, 8, 8, 5][i]
                leg.motorPd.torque[i] = rng.uniform(-5, 5) if t % 7 == 0 else 0.0
            if t > 120:  # exercise the task-space PD inputs too
                for i in range(6):
                    leg.taskPd.pTarget[i] = rng.uniform(-0.1, 0.1)
                    leg.taskPd.pGain[i] = 10.0
        sd = np.zeros(29)
        motors = q_nom + 0.4 * np.sin(0.031 * t + phase[:10]) + 1e-3 * rng.standard_normal(10)
        sd[DRIVE_SENSOR] = motors * gear                      # actuatorpos sensors are motor-side (gear * q)
        sd[JOINT_SENSOR] = np.array([0.0, 1.4267, -1.5968, 0.0, 1.4267, -1.5968]) + 0.2 * np.sin(0.017 * t + phase[10:16])
        qd = np.array([np.cos(0.5 * a), np.sin(0.5 * a), 0, 0]) if (a := 0.3 * np.sin(0.01 * t)) is not None else None
        sd[16:20] = qd
        sd[20:23] = 0.5 * np.sin(0.02 * t + phase[16:19])
        sd[23:26] = np.array([0, 0, 9.81]) + 2.0 * np.sin(0.023 * t + phase[19:22])
        sd[26:29] = [0, -0.5, 0]
        av = gear * 0.4 * 0.031 / 5e-4 * np.cos(0.031 * t + phase[:10]) * (1.0 if t < 200 else 6.0)  # last part saturates the speed limit
        if t == 150:
            out.pelvis.radio.channel[8] = 0                   # STO asserted through the radio for 20 steps
        if t == 170:
            out.pelvis.radio.channel[8] = 1
        user_in, cin = T.cassie_user_in_t(), T.cassie_in_t()
        R.pd_input_step(ctypes.c_void_p(pd), ctypes.byref(u), ctypes.byref(out), ctypes.byref(user_in))
        R.cassie_core_sim_step(ctypes.c_void_p(core), ctypes.byref(user_in), ctypes.byref(out), ctypes.byref(cin))
        sto = out.pelvis.radio.channel[8] < 1
        ctrl = np.zeros(10)
        dl = drives(out)
        for i, cmd in enumerate(in_torques(cin)):
            c = ctypes.c_double(0)
            dl[i].torque = R.ref_motor(i, cmd, av[i], delay[i].ctypes.data, sto, ctypes.byref(c))
            ctrl[i] = c.value
        for i in range(10):
            R.ref_drive_encoder(ctypes.byref(dl[i]), sd.ctypes.data, dfilt[i].ctypes.data, DRIVE_SENSOR[i])
        jl = [out.leftLeg.shinJoint, out.leftLeg.tarsusJoint, out.leftLeg.footJoint, out.rightLeg.shinJoint, out.rightLeg.tarsusJoint,
              out.rightLeg.footJoint]
        for i in range(6):
            R.ref_joint_encoder(ctypes.byref(jl[i]), sd.ctypes.data, jfilt[i].ctypes.data, JOINT_SENSOR[i])
        vn = out.pelvis.vectorNav
        for i in range(4):
            vn.orientation[i] = sd[16 + i]
        for i in range(3):
            vn.angularVelocity[i], vn.linearAcceleration[i], vn.magneticField[i] = sd[20 + i], sd[23 + i], sd[26 + i]
        y = T.cassie_out_t.from_buffer_copy(out)
        so = T.state_out_t()
        R.state_output_step(ctypes.c_void_p(est), ctypes.byref(y), ctypes.byref(so))
        packed = (ctypes.c_ubyte * 697)()
        R.pack_cassie_out_t(ctypes.byref(y), packed)
        g["pd_in"].append(np.frombuffer(bytes(u), dtype=np.uint8))
        g["sensordata"].append(sd.copy()); g["actvel"].append(av.copy()); g["ctrl"].append(ctrl)
        g["cassie_out"].append(np.frombuffer(bytes(y), dtype=np.uint8))
        g["packed"].append(np.frombuffer(bytes(packed), dtype=np.uint8))
        g["state_out"].append(np.frombuffer(bytes(so), dtype=np.uint8))
    path = os.path.join(REPO, "tests", "golden", "hostpath_v1.npz")
    np.savez_compressed(path, **{k: np.array(v) for k, v in g.items()})
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
