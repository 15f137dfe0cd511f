import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def built():
    """Everything compiled once per session (HIP lib, oracle, emulator)."""
    import __graft_entry__
    __graft_entry__.build()
    return True


@pytest.fixture(scope='session')
def emu_library(built):
    from pybullet_multigoal_gym_amd._lib import PmgLibrary
    return PmgLibrary(os.path.join(ROOT, 'tests', 'emu', 'libpmg_emu.so'))


@pytest.fixture(scope='session')
def hip_library(built):
    from pybullet_multigoal_gym_amd._lib import default_library
    return default_library()
