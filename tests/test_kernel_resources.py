"""Regression guard for the step kernel's register allocation and LDS footprint (no GPU needed: hipcc cross-compiles).

The headline instantiation -- the row-capped fast one of cassie.xml in its two-wave form (round 4) -- runs two waves per SIMD on
256 registers each and 40 KB of LDS per env, and the allocator sits on a cliff at this size: small source changes have sent
hundreds of values to scratch all over the kernel (the `wv::touch` that keeps the lane's column of Y live; the
machine-LICM flag; a second call site of the inlined env step).  It must keep: 256 registers (two waves per SIMD), next to
no scratch traffic (a handful of launch-long values parked once), four two-wave workgroups per CU (LDS <= 40 KB), its Gram
matrix and composite-inertia sums on the matrix core, and no workgroup barrier beyond the four of a substep (F, J, P, E; X is a
pair of LDS flags since round 4) + the exits."""
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_fast_cassie_kernel_keeps_its_registers_and_its_lds(tmp_path):
    isa = tmp_path / "fast.s"
    env = dict(os.environ, MAXRS="31", NW="2", KEEP=str(isa))
    out = subprocess.run(["bash", os.path.join(REPO, "tools", "kernel_resources.sh")], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    text = out.stdout
    val = lambda key: int(re.search(key + r"[^:]*: *(\d+)", text).group(1))
    assert val("VGPRs Spill") <= 8, text
    assert val("ScratchSize") <= 192, text         # (the frame: spill slots of launch-long scalars; the traffic is counted below)
    assert val("LDS Size") <= 40960, text          # four two-wave workgroups per CU (160 KB)
    assert val("Occupancy") == 2, text             # two waves per SIMD: 256 registers each
    asm = isa.read_text()
    body = [l.split(";")[0].strip() for l in asm.split("\n")]
    assert sum(l.startswith(("scratch_load", "scratch_store")) for l in body) <= 8, "scratch traffic in the fast kernel"
    assert sum(l.startswith("s_barrier") for l in body) <= 16, "workgroup barriers: F, J, P, E per substep in each wave's program + the exits"
    # A = Y Y^T: 3 tiles x 8 dof blocks; composite inertias: 2 body blocks x 8 summand blocks
    assert sum(l.startswith("v_mfma_f64_16x16x4_f64") for l in body) == 24 + 16


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_tray_fast_kernel_keeps_its_gram_matrix_on_the_matrix_core(tmp_path):
    """The 40-dof model's row-capped instantiation (47 rows) in the form that ships since round 5 -- TWO wavefronts per env, 256
    registers each: the 48 x 48 Gram matrix of the staged tile as six tiles x ten dof blocks on the matrix core (+ the 16 instructions
    of the composite-inertia sums), through the staged tile's own LDS; two waves per SIMD; four workgroups per CU; and the spills that
    round 5 removed stay removed (round 4: 1 076 B of scratch and 551 spilled values, which made this form 7.5 % slower than one wave per
    env; now 204 B / 150, and 13.6 % faster: profiles/round5/tray_two_waves_ab.txt)."""
    isa = tmp_path / "tray_fast.s"
    env = dict(os.environ, MAXRS="47", NW="2", NVP="40", TOPO="TopoCassieTray38", FEAT="2", KEEP=str(isa))
    out = subprocess.run(["bash", os.path.join(REPO, "tools", "kernel_resources.sh")], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    text = out.stdout
    val = lambda key: int(re.search(key + r"[^:]*: *(\d+)", text).group(1))
    assert val("ScratchSize") <= 320, text
    assert val("VGPRs Spill") <= 200, text
    assert val("LDS Size") <= 40960, text
    assert val("Occupancy") == 2, text
    body = [l.split(";")[0].strip() for l in isa.read_text().split("\n")]
    assert sum(l.startswith(("scratch_load", "scratch_store")) for l in body) <= 160, "scratch traffic in the tray model's fast kernel"
    assert sum(l.startswith("v_mfma_f64_16x16x4_f64") for l in body) == 60 + 16


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_127_row_kernel_fits_a_cu(tmp_path):
    """The 127-row instantiation (two wavefronts of 512 registers, the staged matrix of 128 rows beside the body tiles): one workgroup
    must fit a CU's 160 KB of LDS with room for a fast workgroup beside it, and its spills stay a frame of launch-long values."""
    env = dict(os.environ, MAXRS="127", NW="2", WPS="1", WALK="true", FEAT="1")
    out = subprocess.run(["bash", os.path.join(REPO, "tools", "kernel_resources.sh")], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    text = out.stdout
    val = lambda key: int(re.search(key + r"[^:]*: *(\d+)", text).group(1))
    assert val("LDS Size") <= 86016, text
    assert val("ScratchSize") <= 512 and val("VGPRs Spill") <= 128, text


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_fast_hfield_kernel_keeps_its_registers_and_sets_its_wave_priorities(tmp_path):
    """The height-field model's row-capped instantiation (config 4's dominant kernel): two waves per SIMD, four workgroups per CU, a
    frame of launch-long values and next to no spilled vector values (round 4: 196 B / 29; today 156 B / 8), and the wave priorities
    of round 5 (wave.h CK_PRIO_*: one s_setprio at each phase boundary of the two wave programs -- the step from none to these was
    +3 % on config 4, profiles/round5/wave_priority_ab.txt) are still in the code."""
    isa = tmp_path / "hf_fast.s"
    env = dict(os.environ, MAXRS="31", NW="2", FEAT="1", KEEP=str(isa))
    out = subprocess.run(["bash", os.path.join(REPO, "tools", "kernel_resources.sh")], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    text = out.stdout
    val = lambda key: int(re.search(key + r"[^:]*: *(\d+)", text).group(1))
    assert val("VGPRs Spill") <= 24, text
    assert val("ScratchSize") <= 256, text
    assert val("LDS Size") <= 40960, text
    assert val("Occupancy") == 2, text
    body = [l.split(";")[0].strip() for l in isa.read_text().split("\n")]
    assert sum(l.startswith(("scratch_load", "scratch_store")) for l in body) <= 48, "scratch traffic in the height-field model's fast kernel"
    prios = sorted(set(int(l.split()[1]) for l in body if l.startswith("s_setprio")))
    assert prios == [0, 1, 3], prios


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="needs hipcc")
def test_in_place_form_of_the_fast_kernel_is_placed_like_the_plain_form(tmp_path):
    """Round 6: the fast kernel with the 63-row code behind its own 31-row code (INROWS) must run where the plain form runs -- two
    waves per SIMD, 40 KB of LDS per env, i.e. four envs per CU -- or switching a range to it would halve the chip.  The 63-row code in
    256 registers spills (it runs for the one substep that needs it); the frame stays under 1 KB per lane."""
    isa = tmp_path / "inplace.s"
    env = dict(os.environ, MAXRS="31", NW="2", INROWS="63", KEEP=str(isa))
    out = subprocess.run(["bash", os.path.join(REPO, "tools", "kernel_resources.sh")], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    text = out.stdout
    val = lambda key: int(re.search(key + r"[^:]*: *(\d+)", text).group(1))
    assert val("Occupancy") == 2, text
    assert val("LDS Size") <= 40960, text
    assert val("ScratchSize") <= 1024, text
    body = [l.split(";")[0].strip() for l in isa.read_text().split("\n")]
    # both codes are there: the 31-row Gram matrix (3 tiles x 8 dof blocks) and the 63-row one (10 tiles x 8), + the composite inertias twice
    assert sum(l.startswith("v_mfma_f64_16x16x4_f64") for l in body) > 24 + 16
