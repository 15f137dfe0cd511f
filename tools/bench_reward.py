#!/usr/bin/env python3
"""HBM roofline of the HER-relabelling kernels: B goal pairs resident in HBM,
[B,G] float32 x 2 in, float32 + uint8 out = 8 G + 5 B per item.  `bench_reward.py [B] [G]`, G = 3 (single-object tasks)
or 3 * num_block (block_stack / block_rearrange).  Prints one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pybullet_multigoal_gym_amd as pmg
import ctypes as C

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 26
G = int(sys.argv[2]) if len(sys.argv) > 2 else 3
_lib = None
if os.environ.get('PMG_BENCH_LIB'):   # kernel A/B: an alternative build of the library
    from pybullet_multigoal_gym_amd._lib import PmgLibrary
    _lib = PmgLibrary(os.environ['PMG_BENCH_LIB'])
def env_for_goal_dim(G):
    """An env whose goal_dim is G: 3 = single-object tasks; 3 nb = block_stack; 3 nb + 4 = block_stack with
    grip_informed_goal; 1 + 3 nb (+3) = chest_push (with grip_informed_goal)."""
    if G == 3:
        return pmg.make_env(task='reach', num_envs=64, _library=_lib)
    for nb in range(1, 6):
        for task, kw, g in (('block_stack', {}, 3 * nb), ('block_stack', {'grip_informed_goal': True}, 3 * nb + 4),
                            ('chest_push', {}, 1 + 3 * nb), ('chest_push', {'grip_informed_goal': True}, 4 + 3 * nb)):
            if g == G:
                return pmg.make_env(task=task, num_block=nb, num_envs=64, _library=_lib, **kw)
    raise SystemExit('no task has goal_dim %d' % G)


env = env_for_goal_dim(G)
assert env.dims.goal_dim == G
h = env.handle
ag, dg = h.device_alloc(B * 4 * G), h.device_alloc(B * 4 * G)
r, ok = h.device_alloc(B * 4), h.device_alloc(B)
chunk = 1 << 22
rs = np.random.RandomState(0)
for o in range(0, B, chunk):
    n = min(chunk, B - o)
    a = rs.uniform(-0.1, 0.1, (n, G)).astype(np.float32)
    h.upload(ag + o * 4 * G, a)
    h.upload(dg + o * 4 * G, a + (rs.uniform(-0.06, 0.06, (n, G)) / np.sqrt(G / 3)).astype(np.float32))
lib = h.L.lib
def launch():
    rc = lib.pmg_compute_reward_device(h.h, C.c_void_p(ag), C.c_void_p(dg), C.c_int64(B), C.c_void_p(r), C.c_void_p(ok))
    assert rc == 0
for _ in range(3):
    launch()
h.sync()
K = int(os.environ.get('PMG_REWARD_LAUNCHES', '20'))
t0 = time.perf_counter()
for _ in range(K):
    launch()
h.sync()
ms = (time.perf_counter() - t0) / K * 1e3
# spot check against numpy on the first chunk
n = 1 << 16
ha, hd, hr, hk = (np.empty((n, G), np.float32), np.empty((n, G), np.float32), np.empty(n, np.float32), np.empty(n, np.uint8))
h.download(ha, ag); h.download(hd, dg); h.download(hr, r); h.download(hk, ok)
d = np.linalg.norm(ha.astype(np.float64) - hd, axis=1)
clear = np.abs(d - 0.05) > 1e-6
assert np.array_equal(hr[clear], -(d > 0.05).astype(np.float32)[clear]) and np.array_equal(hk[clear] != 0, ~(d > 0.05)[clear])
gbs = B * (8 * G + 5) / (ms * 1e-3) / 1e9
print(json.dumps({'kernel': 'pmg_k_reward3' if G == 3 else 'pmg_k_reward_flat (G=%d)' % G, 'items': B, 'bytes_per_item': 8 * G + 5, 'read_bytes': B * 8 * G, 'write_bytes': B * 5, 'launches_timed': K, 'timer': 'host perf_counter around K launches + sync (rocprofv3 --kernel-trace figures: profiles/*_reward_kernel_trace.csv)', 'ms': ms, 'items_per_s': B / (ms * 1e-3),
                  'roofline': {'bound': 'hbm', 'achieved': gbs, 'peak': 8000.0, 'unit': 'GB/s', 'frac': gbs / 8000.0}}))
for p_ in (ag, dg, r, ok):
    h.device_free(p_)
env.close()
