"""10^7 fresh samples of cassie_core_sim's safety layer: the restatement csrc/pk_safety.h (through the wave emulator's library)
against the live binary (oracle/_ref/libref_hostpath.so = the reference's libagilitycassie.a), bit for bit.  Prints a summary for
profiles/roundN/core_safety_soak.txt.  Needs /root/reference (build container)."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cassie-mujoco-sim_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))
import core_safety_check as C  # noqa: E402

total = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
chunk, done, bad, viol, lim, sto = 500_000, 0, 0, 0, 0, 0
t0 = time.time()
seed = 50_000
while done < total:
    u, q, w, L, ch8 = C.samples(chunk, seed)
    seed += 1
    tau, radio, flags, cw = C.live(u, q, w, L, ch8)
    mine, msg = C.restated(u, q, w, L, ch8 != 1.0)
    neq = np.any(tau.view(np.uint64) != mine.view(np.uint64), axis=1) | np.any(C.queue_of(msg) != radio[:, 1:5], axis=1)
    if neq.any():
        i = int(np.nonzero(neq)[0][0])
        print("FIRST DIFFERENCE seed %d sample %d\n q %r\n u %r\n w %r\n live %r\n mine %r" % (seed - 1, i, q[i], u[i], w[i], tau[i], mine[i]))
    bad += int(neq.sum()); viol += int((msg & 1 > 0).sum()); lim += int((msg & 2 > 0).sum()); sto += int((ch8 != 1.0).sum())
    done += chunk
print("cassie_core_sim safety layer: %d samples, %d differ from the live binary in any torque bit or message (%.0f s); "
      "%d with a violated joint-limit constraint, %d with a torque at its limit, %d under STO" % (done, bad, time.time() - t0, viol, lim, sto))
