#!/usr/bin/env python3
"""Generates tests/golden/physics_vN.npz: what the CPU oracle (oracle/cassie_oracle.c, + the host chain in drive-pd mode)
produces for the frozen workload of tests/golden_physics.py.  Run it ONLY for a deliberate change of a physics
definition, bump golden_physics.VERSION first, and say in the commit what changed: the CPU suite requires the oracle to
reproduce the committed file bit for bit and the GPU suite holds the HIP kernel to it within 1e-7."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "cassie-mujoco-sim_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    import golden_physics as G
    from cassie_amd import Model
    if os.path.exists(G.PATH) and "--force" not in sys.argv:
        raise SystemExit("%s exists: a changed definition needs a new VERSION in tests/golden_physics.py (or --force to overwrite)" % G.PATH)
    out = {"meta/version": np.array(G.VERSION), "meta/checkpoints": np.array(G.CHECKPOINTS), "meta/nenv": np.array(G.NENV)}
    for name in G.MODELS:
        model = Model(name)
        for mode in G.MODES:
            rec = G.oracle_rollout(model, name, mode)
            for field, v in rec.items():
                out[G.key(name, mode, field)] = v
            print("%-16s %-9s rows at step 1000: %s  sweeps: %s" % (name, mode, rec["counts"][-1][:, 1], rec["counts"][-1][:, 2]), flush=True)
    os.makedirs(os.path.dirname(G.PATH), exist_ok=True)
    np.savez_compressed(G.PATH, **out)
    print("wrote", G.PATH, os.path.getsize(G.PATH), "bytes")


if __name__ == "__main__":
    main()
