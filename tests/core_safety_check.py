"""Shared by tests/test_core_safety.py and tools/make_golden_core_safety.py: sample generators for cassie_core_sim's safety
layer, the live binary (oracle/_ref/libref_hostpath.so: the reference's libagilitycassie.a behind a batch harness) and the
restatement the step kernel uses (csrc/pk_safety.h through the wave emulator's library).  Test infrastructure only."""
import ctypes
import os

import numpy as np

from cassie_amd._lib import REPO_DIR

REF_SO = os.path.join(REPO_DIR, "oracle", "_ref", "libref_hostpath.so")
NOMINAL = np.array([0.0045, 0, 0.4973, -1.1997, -1.5968, -0.0045, 0, 0.4973, -1.1997, -1.5968])
LIMITS = np.array([140.63, 140.63, 216.16, 216.16, 45.14] * 2)
LOWER = -np.array([0.11179938779914941, 0.23397243543875249, 0.7226646259971647, 2.572713633111154, 2.2934609527920613,
                   0.1990658503988659, 0.23397243543875249, 0.7226646259971647, 2.572713633111154, 2.2934609527920613])
UPPER = np.array([0.1990658503988659, 0.23397243543875249, 1.2462634015954637, -0.8830382858376185, -0.7608652381980153,
                  0.11179938779914941, 0.23397243543875249, 1.2462634015954637, -0.8830382858376185, -0.7608652381980153])


def have_live_binary():
    return os.path.exists(REF_SO)


def live(u, q, w, L, ch8, telemetry=None, fresh=True):
    """cassie_core_sim_step of the real binary on n samples -> (tau [n][10], radio shorts [n][14], (sto, piezoState, piezoTone) [n][3], controlWords [n][10])."""
    lib = ctypes.CDLL(REF_SO)
    n = len(ch8)
    a = lambda x, dt=np.float64: np.ascontiguousarray(x, dtype=dt)
    u, q, w, L, ch8 = a(u), a(q), a(w), a(L), a(ch8)
    tel = None if telemetry is None else a(telemetry, np.int16)
    tau, radio, flags, cw = np.zeros((n, 10)), np.zeros((n, 14), dtype=np.int16), np.zeros((n, 3), dtype=np.uint8), np.zeros((n, 10), dtype=np.uint16)
    lib.ref_core_sim_batch.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 6 + [ctypes.c_int] + [ctypes.c_void_p] * 4
    lib.ref_core_sim_batch(n, u.ctypes.data, q.ctypes.data, w.ctypes.data, L.ctypes.data, ch8.ctypes.data,
                           None if tel is None else tel.ctypes.data, 1 if fresh else 0, tau.ctypes.data, radio.ctypes.data, flags.ctypes.data, cw.ctypes.data)
    return tau, radio, flags, cw


def restated(u, q, w, L, sto):
    """csrc/pk_safety.h (what the kernel's drive-level pass calls) -> (tau [n][10], message bits [n])."""
    import emu_py
    lib = emu_py.lib()
    n = len(sto)
    a = lambda x, dt=np.float64: np.ascontiguousarray(x, dtype=dt)
    u, q, w, L, sto = a(u), a(q), a(w), a(L), a(sto, np.uint8)
    tau, msg = np.zeros((n, 10)), np.zeros(n, dtype=np.int32)
    lib.emu_core_safety.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 7
    lib.emu_core_safety.restype = None
    lib.emu_core_safety(n, u.ctypes.data, q.ctypes.data, w.ctypes.data, L.ctypes.data, sto.ctypes.data, tau.ctypes.data, msg.ctypes.data)
    return tau, msg


def queue_of(msg_bits):
    """radio.channel[1..4] of a FRESH block for the message bits of one step: the codes raised, highest first."""
    out = np.zeros((len(msg_bits), 4), dtype=np.int16)
    for i, m in enumerate(msg_bits):
        codes = ([635] if m & 1 else []) + ([630] if m & 2 else [])
        out[i, : len(codes)] = codes
    return out


def samples(n, seed, kind="mixed"):
    """(u, q, w, L, ch8).  `mixed`: a third well inside the limits, a third within +-0.2 rad of a bound of one or several joints
    (violations from 0 to past the 0.15 rad blend, the coupled hip pitch + knee rows), a third anywhere in +-3.3 rad; torques from
    nothing to ten times the limits; velocities to +-30 rad/s; STO in 5 %."""
    rng = np.random.default_rng(seed)
    q = np.tile(NOMINAL, (n, 1)) + rng.uniform(-0.05, 0.05, (n, 10))
    third = n // 3
    # near the bounds: pick joints, put them at a bound +- up to 0.2
    sel = rng.random((third, 10)) < rng.uniform(0.05, 0.6, (third, 1))
    side = rng.random((third, 10)) < 0.5
    near = np.where(side, LOWER, UPPER) + rng.uniform(-0.2, 0.2, (third, 10)) * rng.choice([1.0, 0.1, 0.001], (third, 1))
    q[third:2 * third] = np.where(sel, near, q[third:2 * third])
    # the coupled rows: hip pitch + knee around -3 pi / 4
    k = slice(third, third + third // 4)
    hp = rng.uniform(-0.7, 1.2, third // 4)
    q[k, 2] = hp; q[k, 3] = -2.356194490192345 - hp + rng.uniform(-0.2, 0.2, third // 4)
    q[2 * third:] = rng.uniform(-3.3, 3.3, (n - 2 * third, 10))
    u = rng.uniform(-1, 1, (n, 10)) * LIMITS * rng.choice([0.0, 0.1, 1.0, 1.0, 10.0], (n, 1))
    w = rng.uniform(-1, 1, (n, 10)) * rng.choice([0.0, 1.0, 30.0], (n, 1))
    L = np.tile(LIMITS, (n, 1))
    ch8 = np.where(rng.random(n) < 0.05, rng.choice([0.0, 0.5, 0.999, 1.5, -1.0], n), 1.0)
    return u, q, w, L, ch8


def adversarial():
    """Hand-made corners: every joint exactly at, one ulp inside and one ulp beyond each bound; violations of exactly the blend
    width; torques exactly at the limits (the >= of message 630); signed zeros; a zero and a negative limit; STO with
    negative torques (the sign of the zero)."""
    rows = []
    base = NOMINAL.copy()
    def add(q, u=None, w=None, L=None, ch8=1.0):
        rows.append((np.zeros(10) if u is None else np.asarray(u, float), np.asarray(q, float), np.zeros(10) if w is None else np.asarray(w, float),
                     LIMITS.copy() if L is None else np.asarray(L, float), ch8))
    for k in range(10):
        for b in (LOWER[k], UPPER[k]):
            for d in (0.0, np.spacing(b), -np.spacing(b), 0.15, -0.15, 0.15 + 1e-12, -0.15 - 1e-12, 0.149999999, -0.149999999, 1e-300, -1e-300):
                q = base.copy(); q[k] = b + d
                add(q, u=np.full(10, 3.0), w=np.linspace(-2, 2, 10))
                add(q, u=-LIMITS, w=np.full(10, -25.0))
    for u in (LIMITS, -LIMITS, LIMITS * (1 - 1e-16), np.nextafter(LIMITS, 0), np.nextafter(LIMITS, 1e9), np.zeros(10), -np.zeros(10)):
        add(base, u=u)
        add(base, u=u, ch8=0.0)
    add(base, u=np.full(10, -5.0), ch8=0.0)
    add(base, u=np.full(10, 5.0), L=np.zeros(10))
    add(base, u=np.full(10, 5.0), L=-LIMITS)
    add(base, u=np.full(10, -5.0), L=-LIMITS)
    q = base.copy(); q[2], q[3] = 0.3, -2.356194490192345 - 0.3 - 0.05
    add(q, u=np.full(10, 50.0), w=np.full(10, 3.0))
    q = base.copy(); q[7], q[8] = -0.7, -2.2
    add(q, u=np.full(10, -50.0), w=np.full(10, -3.0))
    return tuple(np.array([r[i] for r in rows]) for i in range(5))
