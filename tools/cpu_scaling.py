"""Exploration helper: oracle throughput vs OpenMP thread count on this box."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, oracle_lib as O
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
try:
    print('cgroup cpu.max', open('/sys/fs/cgroup/cpu.max').read().strip())
except Exception as e:
    print('no cgroup cpu.max', e)
for th in (1, 8, 32, 64, 128, 256):
    n = 8 * th
    e = O.OracleEnv('reach', n, seed_stride=1, threads=th); e.reset()
    a = np.zeros((n, 3), np.float32)
    e.step(a)
    t = time.perf_counter()
    for _ in range(3): e.step(a)
    el = time.perf_counter() - t
    print('threads %3d: %8.0f env-steps/s (%.0f per thread)' % (th, 3 * n / el, 3 * n / el / th), flush=True)
