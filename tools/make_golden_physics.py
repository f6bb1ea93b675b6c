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
    # `--v2`: the file of the round-5 option (CM_FLAG_HFPRISM); default: the v1 file of the default definitions
    v2 = "--v2" in sys.argv
    path, version, models = (G.PATH2, G.VERSION2, G.MODELS2) if v2 else (G.PATH, G.VERSION, G.MODELS)
    if os.path.exists(path) and "--force" not in sys.argv:
        raise SystemExit("%s exists: a changed definition needs a new VERSION in tests/golden_physics.py (or --force to overwrite)" % path)
    out = {"meta/version": np.array(version), "meta/checkpoints": np.array(G.CHECKPOINTS), "meta/nenv": np.array(G.NENV)}
    for name in models:
        model = G.model_of(name)
        for mode in G.MODES:
            rec = G.oracle_rollout(model, name, mode)
            for field, v in rec.items():
                out[G.key(name, mode, field)] = v
            print("%-20s %-9s rows at step 1000: %s  sweeps: %s  most rows at a checkpoint: %d" % (name, mode, rec["counts"][-1][:, 1], rec["counts"][-1][:, 2], rec["counts"][:, :, 1].max()), flush=True)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
