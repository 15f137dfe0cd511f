"""CPU tier: sanitizer builds (SURVEY.md section 5 lists sanitizers among the auxiliary subsystems a build should have).

* the oracle (oracle/pmg_oracle.c) under AddressSanitizer + UndefinedBehaviorSanitizer, driven through every task and
  option by oracle/sanitize_main.c;
* the PRODUCT's device and host sources (csrc/pmg_kernels.hip, pmg_api.cpp) compiled for the CPU emulator (tests/emu)
  under UndefinedBehaviorSanitizer -- out-of-range shifts, signed overflow, misaligned or null accesses, float -> int
  overflows in the kernels' index arithmetic -- stepping reach / push / slide / block_stack / chest_push through the
  C ABI.  (AddressSanitizer is not used there: the emulator switches fiber stacks under ASan's feet.)
Any report aborts the child process (-fno-sanitize-recover=all) and fails the test."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / 'oracle_san')
    subprocess.check_call(['gcc', '-O1', '-g', '-fopenmp', '-std=gnu11', '-fsanitize=address,undefined', '-fno-sanitize-recover=all',
                           '-I.', '-o', exe, 'pmg_oracle.c', 'sanitize_main.c', '-lm'], cwd=os.path.join(ROOT, 'oracle'))
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS='detect_leaks=1'))
    assert out.returncode == 0, out.stderr[-3000:]
    assert 'rc 0' in out.stdout and 'runtime error' not in out.stderr and 'AddressSanitizer' not in out.stderr


DRIVER = r'''
import sys, warnings
sys.path.insert(0, %r)
import numpy as np
import pybullet_multigoal_gym_amd as pmg
from pybullet_multigoal_gym_amd._lib import PmgLibrary
lib = PmgLibrary(sys.argv[1])
for task, kw, n in (('reach', {}, 2), ('push', {}, 1), ('slide', {}, 1), ('block_stack', {'num_block': 2}, 1), ('chest_push', {'num_block': 1}, 1)):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        env = pmg.make_env(task=task, num_envs=n, seed=1, seed_stride=1, _library=lib, **kw)
    env.reset()
    a = np.full((n, env.dims.action_dim), -1.0, np.float32)     # down to the table: contact paths
    for _ in range(2):
        env.step(a)
    env.reset(mask=np.ones(n, bool))
    env._compute_reward(np.zeros((5, env.dims.goal_dim), np.float32), np.ones((5, env.dims.goal_dim), np.float32))
    env.close()
    print(task, 'ok', flush=True)
'''


def test_product_sources_are_clean_under_ubsan_on_the_emulator(tmp_path):
    emu, src = os.path.join(ROOT, 'tests', 'emu'), os.path.join(ROOT, 'pybullet_multigoal_gym_amd', 'csrc')
    lib = str(tmp_path / 'libpmg_emu_ubsan.so')
    subprocess.check_call(['g++', '-O1', '-fPIC', '-std=c++17', '-I' + emu, '-I' + src, '-Wno-unknown-pragmas', '-w',
                           '-fsanitize=undefined', '-fno-sanitize-recover=all', '-shared', '-o', lib,
                           os.path.join(emu, 'hip_emu.cpp'), os.path.join(emu, 'pmg_probe.cpp'), os.path.join(src, 'pmg_api.cpp'),
                           '-x', 'c++', os.path.join(src, 'pmg_kernels.hip'), '-lrt'])
    out = subprocess.run([sys.executable, '-c', DRIVER % ROOT, lib], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-500:], out.stderr[-3000:])
    assert out.stdout.count(' ok') == 5 and 'runtime error' not in out.stderr
