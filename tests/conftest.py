import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "cassie-mujoco-sim_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
