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
def emu_library_small_rowspace(built, tmp_path_factory):
    """The emulator build of the product sources with the row-space solve of the one-object kernel limited to 4 contacts:
    an ordinary scene (object + fingers on the table) then takes the path of an env with MORE contacts than that solve
    holds (object x table run in row space + LDS rows for the rest)."""
    import subprocess
    from pybullet_multigoal_gym_amd._lib import PmgLibrary
    out = str(tmp_path_factory.mktemp('emu') / 'libpmg_emu_small.so')
    emu = os.path.join(ROOT, 'tests', 'emu')
    src = os.path.join(ROOT, 'pybullet_multigoal_gym_amd', 'csrc')
    subprocess.check_call(['g++', '-O2', '-fPIC', '-std=c++17', '-I' + emu, '-I' + src, '-Wno-unknown-pragmas', '-DPMG_OBJ_ROWSPACE_MAXC=4',
                           '-shared', '-o', out, os.path.join(emu, 'hip_emu.cpp'), os.path.join(emu, 'pmg_probe.cpp'),
                           os.path.join(src, 'pmg_api.cpp'), '-x', 'c++', os.path.join(src, 'pmg_kernels.hip'), '-lrt'])
    return PmgLibrary(out)


@pytest.fixture(scope='session')
def emu_library_serial_repeat(built, tmp_path_factory):
    """The emulator build WITHOUT the speculative third wavefront (-DPMG_CYL_SPEC=0): slide's list-0 kernel repeats the
    finger x puck pairs in double serially on the helper wavefront, round 5's layout."""
    import subprocess
    from pybullet_multigoal_gym_amd._lib import PmgLibrary
    out = str(tmp_path_factory.mktemp('emu') / 'libpmg_emu_serial.so')
    emu = os.path.join(ROOT, 'tests', 'emu')
    src = os.path.join(ROOT, 'pybullet_multigoal_gym_amd', 'csrc')
    subprocess.check_call(['g++', '-O2', '-fPIC', '-std=c++17', '-I' + emu, '-I' + src, '-Wno-unknown-pragmas', '-DPMG_CYL_SPEC=0',
                           '-shared', '-o', out, os.path.join(emu, 'hip_emu.cpp'), os.path.join(emu, 'pmg_probe.cpp'),
                           os.path.join(src, 'pmg_api.cpp'), '-x', 'c++', os.path.join(src, 'pmg_kernels.hip'), '-lrt'])
    return PmgLibrary(out)


@pytest.fixture(scope='session')
def hip_library(built):
    from pybullet_multigoal_gym_amd._lib import default_library
    return default_library()
