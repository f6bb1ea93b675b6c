"""Batched derived getters (phys_batch_derive) on the CPU wave emulator against values computed
independently from the oracle's state with numpy: centre of mass / momentum from first principles, Jacobians by the
oracle's co_jac, the mass matrix, and the contact-force sums."""
import ctypes

import numpy as np

from cassie_amd import phys as P
from emu_py import EmuBatch
from oracle_py import Oracle, arr, lib as olib


from derive_check import check_derived_block, foot_ids as _ids


import pytest


@pytest.mark.parametrize("two_waves", [0, 1])
def test_derived_block_against_first_principles(cassie, two_waves):
    """(two_waves: the forward pass with the read-out enabled through the full instantiation's two-wave form -- what a
    cassie_sim_t, a batch of one env with the read-out on, runs since round 4)"""
    import emu_py
    emu_py.lib().emu_two_waves(two_waves)
    try:
        _derived_block(cassie)
    finally:
        emu_py.lib().emu_two_waves(0)


def _derived_block(cassie):
    pod = cassie.pod
    rng = np.random.default_rng(5)
    n = 3
    emu = EmuBatch(pod, n)
    q = np.tile(cassie.qpos_init(), (n, 1))
    q[:, 2] -= [0.0, 0.012, 0.02]                    # in the air, touching, pressed into the floor
    q[:, 7:] += 0.05 * rng.standard_normal((n, pod.nq - 7))
    v = rng.uniform(-0.5, 0.5, (n, pod.nv))
    emu.qpos[:], emu.qvel[:] = q, v
    ids = _ids(cassie)
    assert min(ids) >= 0
    D, QM = emu.derive(ids)
    check_derived_block(cassie, q, v, D, QM, ids)
    assert D[0, P.DRV_FOOT_FORCE + 2] == 0 and D[2, P.DRV_FOOT_FORCE + 2] > 50      # in the air vs pressed into the floor
    assert np.array_equal(emu.qpos, q)                                               # derive is a forward pass: state untouched
