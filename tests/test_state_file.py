"""Checkpoint files: cassie_state_save / cassie_state_load.  The CPU part checks the format and the
flat images of the Agility block states; the GPU part saves a running simulator and resumes it in another one."""
import ctypes
import os

import numpy as np
import pytest

from cassie_amd import iotypes as T
from cassie_amd._lib import REPO_DIR, lib

VP = ctypes.c_void_p


@pytest.fixture(scope="module")
def L(built):
    L = lib()
    L.cassie_state_alloc.restype = VP
    L.cassie_state_free.argtypes = [VP]
    L.cassie_state_save.argtypes = [VP, ctypes.c_char_p]
    L.cassie_state_load.argtypes = [VP, ctypes.c_char_p]
    for f in ("cassie_state_qpos", "cassie_state_qvel", "cassie_state_time"):
        getattr(L, f).restype = ctypes.POINTER(ctypes.c_double)
        getattr(L, f).argtypes = [VP]
    L.cassie_hostenv_image_size.restype = ctypes.c_size_t
    L.cassie_hostenv_alloc.restype = VP
    L.cassie_hostenv_free.argtypes = [VP]
    L.cassie_hostenv_to_image.argtypes = [VP, VP]
    L.cassie_hostenv_from_image.argtypes = [VP, VP]
    L.cassie_hostenv_step_pd_pre.argtypes = [VP] * 7
    L.cassie_hostenv_step_pd_post.argtypes = [VP] * 3
    L.cassie_hostmodel_from_model.argtypes = [VP, VP]
    return L


def test_agility_block_sizes_match_the_allocations(L):
    """The block states are serialised as 96 / 1240 / 4208 flat bytes: the sizes their allocators ask
    malloc for."""
    libc = ctypes.CDLL(None)
    libc.malloc_usable_size.restype = ctypes.c_size_t
    libc.malloc_usable_size.argtypes = [VP]
    for alloc, free, size in (("cassie_core_sim_alloc", "cassie_core_sim_free", 96), ("pd_input_alloc", "pd_input_free", 1240),
                              ("state_output_alloc", "state_output_free", 4208)):
        getattr(L, alloc).restype = VP
        getattr(L, free).argtypes = [VP]
        p = getattr(L, alloc)()
        usable = libc.malloc_usable_size(p)
        getattr(L, free)(p)
        assert size <= usable < size + 32, (alloc, usable)


def test_host_image_carries_the_whole_host_state(L, cassie):
    """Run the host chain for a while, image it into a fresh env, and both continue bit-identically (so nothing the
    blocks remember lives outside the image)."""
    from test_hostpath import GOLDEN, HostModel
    g = np.load(GOLDEN)
    hm = HostModel()
    assert L.cassie_hostmodel_from_model(cassie._h, ctypes.byref(hm)) == 0

    def step(env, t):
        u = T.pd_in_t.from_buffer_copy(g["pd_in"][t].tobytes())
        sd, av = np.ascontiguousarray(g["sensordata"][t]), np.ascontiguousarray(g["actvel"][t])
        ctrl = np.zeros(10)
        y, so = T.cassie_out_t(), T.state_out_t()
        L.cassie_hostenv_step_pd_pre(env, ctypes.byref(hm), ctypes.byref(u), sd.ctypes.data, av.ctypes.data, ctrl.ctypes.data, ctypes.byref(y))
        L.cassie_hostenv_step_pd_post(env, ctypes.byref(y), ctypes.byref(so))
        return ctrl.tobytes() + bytes(y) + bytes(so)
    a = L.cassie_hostenv_alloc()
    for t in range(100):
        step(a, t)
    n = L.cassie_hostenv_image_size()
    img = (ctypes.c_ubyte * n)()
    L.cassie_hostenv_to_image(a, img)
    b = L.cassie_hostenv_alloc()
    L.cassie_hostenv_from_image(b, img)
    for t in range(100, 160):
        assert step(a, t) == step(b, t), t
    L.cassie_hostenv_free(a)
    L.cassie_hostenv_free(b)


def test_a_copied_env_is_independent_of_its_source(L, cassie):
    """The Agility pd_input / state_output states hold pointers into themselves; the library's *_copy are plain copies, so
    in the reference a duplicated simulator's estimator keeps working on its parent's buffers.  Here copies are rebased:
    stepping a copy must not disturb the source, and the copy must continue exactly like the source would have."""
    from test_hostpath import GOLDEN, HostModel
    L.cassie_hostenv_copy.argtypes = [VP, VP]
    g = np.load(GOLDEN)
    hm = HostModel()
    assert L.cassie_hostmodel_from_model(cassie._h, ctypes.byref(hm)) == 0

    def step(env, t):
        u = T.pd_in_t.from_buffer_copy(g["pd_in"][t].tobytes())
        sd, av = np.ascontiguousarray(g["sensordata"][t]), np.ascontiguousarray(g["actvel"][t])
        ctrl = np.zeros(10)
        y, so = T.cassie_out_t(), T.state_out_t()
        L.cassie_hostenv_step_pd_pre(env, ctypes.byref(hm), ctypes.byref(u), sd.ctypes.data, av.ctypes.data, ctrl.ctypes.data, ctypes.byref(y))
        L.cassie_hostenv_step_pd_post(env, ctypes.byref(y), ctypes.byref(so))
        return ctrl.tobytes() + bytes(y) + bytes(so)
    a, twin, c = L.cassie_hostenv_alloc(), L.cassie_hostenv_alloc(), L.cassie_hostenv_alloc()
    for t in range(60):
        assert step(a, t) == step(twin, t)
    L.cassie_hostenv_copy(c, a)
    for t in range(120, 150):
        step(c, t)                                    # the copy wanders off on other inputs
    L.cassie_hostenv_copy(c, a)
    for t in range(60, 110):
        ra = step(a, t)
        assert ra == step(twin, t), t                 # the source never noticed
        assert ra == step(c, t), t                    # and a fresh copy continues like the source
    for e in (a, twin, c):
        L.cassie_hostenv_free(e)


def test_state_file_round_trip_and_rejection(L, tmp_path):
    s = L.cassie_state_alloc()
    rng = np.random.default_rng(0)
    np.ctypeslib.as_array(L.cassie_state_qpos(s), (35,))[:] = rng.standard_normal(35)
    np.ctypeslib.as_array(L.cassie_state_qvel(s), (32,))[:] = rng.standard_normal(32)
    L.cassie_state_time(s)[0] = 1.2345
    path = str(tmp_path / "state.bin").encode()
    assert L.cassie_state_save(s, path) == 0
    t = L.cassie_state_alloc()
    assert L.cassie_state_load(t, path) == 0
    assert np.array_equal(np.ctypeslib.as_array(L.cassie_state_qpos(t), (35,)), np.ctypeslib.as_array(L.cassie_state_qpos(s), (35,)))
    assert np.array_equal(np.ctypeslib.as_array(L.cassie_state_qvel(t), (32,)), np.ctypeslib.as_array(L.cassie_state_qvel(s), (32,)))
    assert L.cassie_state_time(t)[0] == 1.2345
    raw = open(path, "rb").read()
    assert raw[:8] == b"CASSIEST"
    open(path, "wb").write(b"NOTASTATE" + raw[9:])
    assert L.cassie_state_load(t, path) == -1                       # foreign magic
    open(path, "wb").write(raw[:-100])
    assert L.cassie_state_load(t, path) == -1                       # truncated
    open(path, "wb").write(raw + b"\0")
    assert L.cassie_state_load(t, path) == -1                       # trailing bytes
    flipped = bytearray(raw)
    flipped[len(raw) // 2] ^= 0x10
    open(path, "wb").write(bytes(flipped))
    assert L.cassie_state_load(t, path) == -1                       # a flipped bit in the body: checksum mismatch
    open(path, "wb").write(raw)
    assert L.cassie_state_load(t, path) == 0                        # the intact file still loads
    L.cassie_hostenv_blocks_verified.restype = ctypes.c_bool
    assert L.cassie_hostenv_blocks_verified()                       # the closed library's block sizes are what the image assumes
    assert L.cassie_state_load(t, str(tmp_path / "missing.bin").encode()) == -1
    assert L.cassie_state_time(t)[0] == 1.2345                      # failed loads leave the state alone
    L.cassie_state_free(s)
    L.cassie_state_free(t)


MODEL = os.path.join(REPO_DIR, "models", "cassie.cmodel").encode()


def _pd(rng):
    u = T.pd_in_t()
    off = [0.0045, 0, 0.4973, -1.1997, -1.5968]
    for leg in (u.leftLeg, u.rightLeg):
        for i in range(5):
            leg.motorPd.pTarget[i] = off[i] + 0.2 * rng.uniform(-1, 1)
            leg.motorPd.pGain[i] = [70, 70, 100, 100, 50][i]
            leg.motorPd.dGain[i] = [7, 7, 8, 8, 5][i]
    return u


@pytest.mark.gpu
def test_resume_from_a_checkpoint_file_on_the_gpu(L, tmp_path):
    """200 steps, save; a second simulator loads the file and both run 200 more steps: identical state_out_t bytes."""
    L.cassie_sim_init.restype = VP
    L.cassie_sim_init.argtypes = [ctypes.c_char_p, ctypes.c_bool]
    L.cassie_sim_free.argtypes = [VP]
    L.cassie_sim_step_pd.argtypes = [VP, VP, VP]
    L.cassie_get_state.argtypes = [VP, VP]
    L.cassie_set_state.argtypes = [VP, VP]
    rng = np.random.default_rng(5)
    a = L.cassie_sim_init(MODEL, False)
    u, y = _pd(rng), T.state_out_t()
    for _ in range(200):
        L.cassie_sim_step_pd(a, ctypes.byref(y), ctypes.byref(u))
    s = L.cassie_state_alloc()
    L.cassie_get_state(a, s)
    path = str(tmp_path / "ckpt.bin").encode()
    assert L.cassie_state_save(s, path) == 0
    b = L.cassie_sim_init(MODEL, False)
    t = L.cassie_state_alloc()
    assert L.cassie_state_load(t, path) == 0
    L.cassie_set_state(b, t)
    ya, yb = T.state_out_t(), T.state_out_t()
    for k in range(200):
        if k % 50 == 0:
            u = _pd(rng)
        L.cassie_sim_step_pd(a, ctypes.byref(ya), ctypes.byref(u))
        L.cassie_sim_step_pd(b, ctypes.byref(yb), ctypes.byref(u))
        assert bytes(ya) == bytes(yb), k
    for h in (s, t):
        L.cassie_state_free(h)
    L.cassie_sim_free(a)
    L.cassie_sim_free(b)


@pytest.mark.gpu
def test_udp_lockstep_server(L, tmp_path):
    """The cassiesim server (reference example/cassiesim.c role) in PD mode: 80 packed pd_in_t datagrams in, 80 packed
    state_out_t datagrams back, byte for byte what cassie_sim_step_pd + pack_state_out_t give directly; the state file it
    writes when it stops resumes the run."""
    import socket
    import subprocess
    exe = os.path.join(REPO_DIR, "cassie-mujoco-sim_amd", "bin", "cassiesim")
    if not os.path.exists(exe):
        pytest.skip("cassiesim not built")
    L.cassie_sim_init.restype = VP
    L.cassie_sim_init.argtypes = [ctypes.c_char_p, ctypes.c_bool]
    L.cassie_sim_free.argtypes = [VP]
    L.cassie_sim_step_pd.argtypes = [VP, VP, VP]
    L.pack_pd_in_t.argtypes = [VP, VP]
    L.pack_state_out_t.argtypes = [VP, VP]
    port = 26000 + os.getpid() % 3000
    statefile = str(tmp_path / "server_state.bin")
    srv = subprocess.Popen([exe, "-a", "127.0.0.1", "-p", str(port), "-x", "-m", MODEL.decode(), "-n", "80", "-s", statefile],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        line = srv.stdout.readline()
        assert "Waiting for input" in line, line
        sock = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        sock.settimeout(20)
        rng = np.random.default_rng(9)
        ref = L.cassie_sim_init(MODEL, False)
        y = T.state_out_t()
        for k in range(80):
            if k % 20 == 0:
                u = _pd(rng)
                packed = (ctypes.c_ubyte * 476)()
                L.pack_pd_in_t(ctypes.byref(u), packed)
                # the controller's inputs are what survives the float32 wire format
                uw = T.pd_in_t()
                L.unpack_pd_in_t.argtypes = [VP, VP]
                L.unpack_pd_in_t(packed, ctypes.byref(uw))
            sock.sendto(bytes([k & 0xff, 0]) + bytes(packed), ("127.0.0.1", port))
            reply, _ = sock.recvfrom(4096)
            assert len(reply) == 2 + 493
            L.cassie_sim_step_pd(ref, ctypes.byref(y), ctypes.byref(uw))
            want = (ctypes.c_ubyte * 493)()
            L.pack_state_out_t(ctypes.byref(y), want)
            assert reply[2:] == bytes(want), k
        srv.wait(timeout=30)
        assert srv.returncode == 0
        assert "80 steps" in srv.stdout.read()
        assert os.path.getsize(statefile) > 8000
        L.cassie_sim_free(ref)
    finally:
        if srv.poll() is None:
            srv.kill()
