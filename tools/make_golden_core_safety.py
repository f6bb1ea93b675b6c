"""Generates tests/golden/core_safety_v1.npz: inputs and outputs of the REAL cassie_core_sim_step (reference
src/libagilitycassie.a, run through oracle/_ref/libref_hostpath.so: oracle/build_ref.sh) for the safety-layer restatement
csrc/pk_safety.h -- the hand-made corner cases and 3 000 mixed random samples of tests/core_safety_check.py, plus one sequence
through ONE block instance (the message queue is sticky).  Run in the build container (needs /root/reference)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cassie-mujoco-sim_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import core_safety_check as C  # noqa: E402

assert C.have_live_binary(), "build oracle/_ref first: bash oracle/build_ref.sh"
out = {}
for name, s in (("adv", C.adversarial()), ("mix", C.samples(3000, 2026))):
    u, q, w, L, ch8 = s
    tau, radio, flags, cw = C.live(u, q, w, L, ch8)
    assert not flags.any() and not cw.any()
    out.update({name + "_u": u, name + "_q": q, name + "_w": w, name + "_L": L, name + "_ch8": ch8, name + "_tau": tau, name + "_radio": radio})
# a sequence through one instance: clean steps, a torque-limit hit, clean, a joint-limit violation, clean, both
u, q, w, L, ch8 = C.samples(400, 7)
tel = np.random.default_rng(3).integers(-30000, 30000, (400, 9)).astype(np.int16)
tau, radio, flags, cw = C.live(u, q, w, L, ch8, telemetry=tel, fresh=False)
out.update({"seq_u": u, "seq_q": q, "seq_w": w, "seq_L": L, "seq_ch8": ch8, "seq_tel": tel, "seq_tau": tau, "seq_radio": radio})
path = os.path.join(REPO, "tests", "golden", "core_safety_v1.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes;", {k: v.shape for k, v in out.items() if k.endswith("_tau")})
