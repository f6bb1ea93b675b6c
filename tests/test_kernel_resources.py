"""Regression guard for the step kernel's register allocation and LDS footprint (no GPU needed: hipcc cross-compiles).

The kernel runs one wave per SIMD on all 512 registers and 40 KB of LDS per env, and the allocator sits on a cliff at
this size: small source changes have sent 600 values to scratch all over the kernel (the `wv::touch` that
keeps the lane's column of Y live; the machine-LICM flag).  The headline instantiation -- the row-capped fast one of
cassie.xml -- must keep: no scratch instruction in its ISA, no VGPR spill, four workgroups per CU (LDS <= 40 KB), and
its Gram matrix and composite-inertia sums on the matrix core."""
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_fast_cassie_kernel_keeps_its_registers_and_its_lds(tmp_path):
    isa = tmp_path / "fast.s"
    env = dict(os.environ, MAXRS="31", KEEP=str(isa))
    out = subprocess.run(["bash", os.path.join(REPO, "tools", "kernel_resources.sh")], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    text = out.stdout
    val = lambda key: int(re.search(key + r"[^:]*: *(\d+)", text).group(1))
    assert val("VGPRs Spill") == 0, text
    assert val("LDS Size") <= 40960, text          # four single-wave workgroups per CU (160 KB)
    assert val("Occupancy") == 1, text
    asm = isa.read_text()
    body = [l.split(";")[0].strip() for l in asm.split("\n")]
    assert not any(l.startswith(("scratch_load", "scratch_store")) for l in body), "scratch traffic in the fast kernel"
    # A = Y Y^T: 3 tiles x 8 dof blocks; composite inertias: 2 body blocks x 8 summand blocks
    assert sum(l.startswith("v_mfma_f64_16x16x4_f64") for l in body) == 24 + 16
