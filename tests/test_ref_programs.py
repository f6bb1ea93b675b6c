"""Source-level drop-in for C users: the reference's own example programs (example/*.c), compiled UNMODIFIED from where
they lie in the reference tree against this repository's include/ and product library (oracle/build_ref.sh puts the
binaries into the git-ignored oracle/_ref/bin, which travels to the GPU box).  The CPU part checks that all of them build
and link; the GPU part runs the reference's UDP simulator and controller against each other and against the library."""
import ctypes
import os
import socket
import subprocess
import time

import numpy as np
import pytest

from cassie_amd import iotypes as T
from cassie_amd._lib import REPO_DIR, lib

VP = ctypes.c_void_p
REF_STAGE = os.path.join(REPO_DIR, "oracle", "_ref")
BIN = os.path.join(REF_STAGE, "bin")
PROGRAMS = ("cassiesim", "cassiectrl", "cassietest", "cassievideo", "test_doublevis", "test_heelforce", "test_hfield", "test_terrain")
staged = pytest.mark.skipif(not os.path.exists(os.path.join(BIN, "cassiesim")),
                            reason="reference programs not built (oracle/build_ref.sh runs where /root/reference exists)")


@pytest.mark.skipif(not os.path.exists("/root/reference/example/cassiesim.c"), reason="no reference tree on this box")
def test_reference_example_programs_compile_and_link_unmodified(built):
    """Every C program the reference ships in example/ compiles against include/cassiemujoco.h (+ udp.h and the per-struct
    headers it includes) and links against the product library with no source change."""
    subprocess.check_call(["bash", os.path.join(REPO_DIR, "oracle", "build_ref.sh")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    prod = os.path.join(REPO_DIR, "cassie-mujoco-sim_amd", "lib", "libcassiemujoco.so")
    exported = set(l.split()[-1] for l in subprocess.check_output(["nm", "-D", "--defined-only", prod], text=True).splitlines() if l.strip())
    for p in PROGRAMS:
        exe = os.path.join(BIN, p)
        assert os.path.exists(exe), p
        undefined = [l.split()[-1] for l in subprocess.check_output(["nm", "-D", "--undefined-only", exe], text=True).splitlines()]
        wanted = [s for s in undefined if s.startswith(("cassie_", "pack_", "unpack_", "udp_", "send_packet", "get_newest", "wait_for", "process_packet"))]
        assert wanted and all(s in exported for s in wanted), (p, [s for s in wanted if s not in exported])


def _qlog(path):
    raw = np.fromfile(path, dtype=np.float64)
    return raw[: raw.size // 68 * 68].reshape(-1, 68)    # time, qpos[35], qvel[32] per step (reference example/cassiesim.c:262-266)


@staged
@pytest.mark.gpu
def test_reference_cassiesim_and_cassiectrl_run_against_each_other(tmp_path):
    """The reference's UDP simulator and its example controller, both unmodified and both on this library, exchange
    packets in lock step in PD mode (null commands: the robot sinks to the floor); the simulator's qpos log shows one
    physics step of 0.5 ms per exchange and a finite state throughout."""
    port, cport = 27000 + os.getpid() % 2000, 29500 + os.getpid() % 2000
    qlog = str(tmp_path / "q.bin")
    cwd = os.path.join(REF_STAGE, "example")                  # the program opens ../model/cassie.xml
    simlog = open(tmp_path / "sim.log", "w")
    sim = subprocess.Popen([os.path.join(BIN, "cassiesim"), "-a", "127.0.0.1", "-p", str(port), "-x", "-q", qlog], cwd=cwd,
                           stdout=simlog, stderr=subprocess.STDOUT)
    ctrl = subprocess.Popen([os.path.join(BIN, "cassiectrl"), "-x", "-a", "127.0.0.1", "-p", str(port), "-b", "127.0.0.1", "-c", str(cport)],  # -a/-p: the simulator, -b/-c: local (cassiectrl.c:48-66; its usage text has them swapped)
                            stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    try:
        deadline = time.time() + 60
        while time.time() < deadline and (not os.path.exists(qlog) or os.path.getsize(qlog) < 68 * 8 * 600):
            assert sim.poll() is None and ctrl.poll() is None, (sim.poll(), ctrl.poll(), open(tmp_path / "sim.log").read()[-2000:])
            time.sleep(0.1)
    finally:
        ctrl.kill(); sim.kill(); ctrl.wait(); sim.wait()
    log = _qlog(qlog)
    assert len(log) >= 600
    assert np.all(np.isfinite(log))
    assert np.allclose(np.diff(log[:, 0]), 5e-4, atol=1e-12) and abs(log[0, 0] - 5e-4) < 1e-12
    assert abs(log[0, 3] - 1.01) < 1e-3 and log[:, 3].min() > 0.0 and log[-1, 3] < log[0, 3]     # pelvis height: starts at 1.01, sinks
    assert np.allclose(np.linalg.norm(log[:, 4:8], axis=1), 1.0, atol=1e-6)


@staged
@pytest.mark.gpu
def test_reference_cassiesim_replies_what_the_library_computes(built, tmp_path):
    """A controller (this test) drives the reference's unmodified cassiesim over UDP in PD mode: every reply is byte for
    byte pack_state_out_t of cassie_sim_step_pd on the same inputs, and the qpos it logs is the library's."""
    L = lib()
    L.cassie_sim_init.restype = VP
    L.cassie_sim_init.argtypes = [ctypes.c_char_p, ctypes.c_bool]
    L.cassie_sim_free.argtypes = [VP]
    L.cassie_sim_step_pd.argtypes = [VP, VP, VP]
    L.cassie_sim_qpos.restype = ctypes.POINTER(ctypes.c_double)
    L.cassie_sim_qpos.argtypes = [VP]
    L.pack_pd_in_t.argtypes = [VP, VP]
    L.unpack_pd_in_t.argtypes = [VP, VP]
    L.pack_state_out_t.argtypes = [VP, VP]
    port = 24000 + os.getpid() % 2000
    qlog = str(tmp_path / "q.bin")
    cwd = os.path.join(REF_STAGE, "example")
    sim = subprocess.Popen([os.path.join(BIN, "cassiesim"), "-a", "127.0.0.1", "-p", str(port), "-x", "-q", qlog], cwd=cwd,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ref = L.cassie_sim_init(os.path.join(REF_STAGE, "model", "cassie.xml").encode(), False)
    assert ref
    try:
        time.sleep(1.0)                                        # until the server process has bound its port
        sock = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        rng = np.random.default_rng(3)
        y = T.state_out_t()
        qs = []
        off = [0.0045, 0, 0.4973, -1.1997, -1.5968]
        for k in range(120):
            if k % 30 == 0:
                u = T.pd_in_t()
                for leg in (u.leftLeg, u.rightLeg):
                    for i in range(5):
                        leg.motorPd.pTarget[i] = off[i] + 0.2 * rng.uniform(-1, 1)
                        leg.motorPd.pGain[i] = [100, 100, 88, 96, 50][i]
                        leg.motorPd.dGain[i] = [10, 10, 8, 9.6, 5][i]
                packed = (ctypes.c_ubyte * 476)()
                L.pack_pd_in_t(ctypes.byref(u), packed)
                uw = T.pd_in_t()                               # what survives the float32 wire format
                L.unpack_pd_in_t(packed, ctypes.byref(uw))
            # the program binds its socket before it loads the model, so the first datagram waits in the socket buffer
            # while the simulator starts (model load, HIP initialisation); one send per step -- a resend would be a step
            sock.settimeout(180.0 if k == 0 else 10.0)
            sock.sendto(bytes([k & 0xff, 0]) + bytes(packed), ("127.0.0.1", port))
            reply, _ = sock.recvfrom(4096)
            assert len(reply) == 2 + 493
            L.cassie_sim_step_pd(ref, ctypes.byref(y), ctypes.byref(uw))
            want = (ctypes.c_ubyte * 493)()
            L.pack_state_out_t(ctypes.byref(y), want)
            assert reply[2:] == bytes(want), k
            qs.append(np.ctypeslib.as_array(L.cassie_sim_qpos(ref), (35,)).copy())
    finally:
        sim.kill(); sim.wait()
        L.cassie_sim_free(ref)
    log = _qlog(qlog)
    n = min(len(log), len(qs))
    assert n >= 50                                             # the tail of the log may sit in the killed process's stdio buffer
    assert np.array_equal(log[:n, 1:36], np.array(qs[:n]))
