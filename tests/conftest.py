import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cassie-mujoco-sim_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _hip_device_count():
    """Number of HIP devices visible to this process (0 when the runtime cannot be loaded)."""
    import ctypes
    for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
        try:
            hip = ctypes.CDLL(name)
        except OSError:
            continue
        n = ctypes.c_int(0)
        try:
            return n.value if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0
        except Exception:
            return 0
    return 0


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without a GPU: the gpu-marked tests are skipped (the product has no CPU fallback, so they
    could only fail); with `-m gpu` on such a box they are skipped too, visibly."""
    if not any("gpu" in it.keywords for it in items) or _hip_device_count() > 0:
        return
    skip = pytest.mark.skip(reason="no HIP device visible: the physics library has no CPU fallback")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def built():
    """Makes sure the product library, the oracle and the emulator are built (no-op when up to date)."""
    import subprocess
    need = [os.path.join(REPO, "cassie-mujoco-sim_amd", "lib", "libcassiemujoco.so"),
            os.path.join(REPO, "oracle", "libcassie_oracle.so"),
            os.path.join(REPO, "tests", "emu", "libcassie_emu.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.check_call(["make", "-C", REPO, "-j8"], stdout=subprocess.DEVNULL)
    return True


@pytest.fixture(scope="session")
def cassie(built):
    from cassie_amd import Model
    return Model("cassie")
