"""Which envs are the long pole of a batched step?  (GPU box; PMG_ENV_CYCLES=1 is set here.)
   python tools/env_cycles.py <task> [num_envs] [steps]
Runs the benchmark's staggered random-policy loop through the host API and, for the last `steps` steps, reads the per-env
cycle counters (PMG_BUF_ENV_CYCLES) and the launch schedule: cycles per env-step by launch list, by contact count, and the
slowest envs with where their gripper is."""
import os, sys, json
import numpy as np
os.environ['PMG_ENV_CYCLES'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pybullet_multigoal_gym_amd as pmg

task = sys.argv[1] if len(sys.argv) > 1 else 'chest_push'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
kw = {'num_block': 4} if task in ('block_stack', 'block_rearrange', 'chest_push', 'chest_pick_and_place') else {}
if os.environ.get('PMG_TF_LIB'):                       # (kernel A/B: another build of the library, as tools/teacher_forced.py)
    from pybullet_multigoal_gym_amd._lib import PmgLibrary
    kw['_library'] = PmgLibrary(os.environ['PMG_TF_LIB'])
env = pmg.make_env(task=task, gripper='parallel_jaw', num_envs=N, seed=0, **kw)
h = env.handle
T = 50
env.reset()
rng = np.random.default_rng(0)
A = h.dims.action_dim
phase = np.arange(N) % T
for t in range(T):                                   # pre-roll: env i reaches phase i mod T
    env.step(rng.uniform(-1, 1, (N, A)).astype(np.float32))
    m = (phase == t % T)
    if m.any(): env.reset(mask=m)
rows = []
for t in range(steps):
    obs = env.step(rng.uniform(-1, 1, (N, A)).astype(np.float32))[0]
    cyc = h.env_cycles().astype(np.int64)
    sc = h.schedule()
    lst = np.full(N, 1); lst[sc['prone']] = 0; lst[sc['redo']] = 2
    st = h.get_state()
    for i in range(N):
        rows.append((t, i, lst[i], int(cyc[i, 0]) * 64, int(cyc[i, 1]), float(st[i, 18]), float(st[i, 19]), float(st[i, 20])))
    m = (phase == (T + t) % T)
    if m.any(): env.reset(mask=m)
r = np.array(rows, dtype=np.float64)
ms = r[:, 3] / 2.3e6                                  # ~2.3 GHz shader clock
print('task %s, %d envs, %d steps: per env-step wavefront time (ms at 2.3 GHz)' % (task, N, steps))
for l, name in ((0, 'list 0 (full store)'), (1, 'list 1 (fast path)'), (2, 'redo')):
    sel = r[:, 2] == l
    if sel.any():
        x = ms[sel]
        print('  %-20s %6.2f %% of env-steps  p50 %.2f  p90 %.2f  p99 %.2f  max %.2f ms' % (name, 100 * sel.mean(), *np.percentile(x, [50, 90, 99]), x.max()))
print('  by largest contact count of a substep (all lists):')
for lo, hi in ((0, 0), (1, 8), (9, 16), (17, 24), (25, 32), (33, 40), (41, 48), (49, 99)):
    sel = (r[:, 4] >= lo) & (r[:, 4] <= hi)
    if sel.any(): print('    nc %2d..%2d: %6.2f %%  p50 %.2f  max %.2f ms' % (lo, hi, 100 * sel.mean(), np.median(ms[sel]), ms[sel].max()))
worst = np.argsort(-ms)[:12]
print('  slowest env-steps: (step, env, list, ms, max contacts, tip target x y z)')
for k in worst: print('   ', int(r[k, 0]), int(r[k, 1]), int(r[k, 2]), '%.2f' % ms[k], int(r[k, 4]), '%.3f %.3f %.3f' % tuple(r[k, 5:8]))
per_step_max = [ms[r[:, 0] == t].max() for t in range(steps)]
print('  slowest env per step: mean %.2f ms (min %.2f max %.2f)' % (np.mean(per_step_max), min(per_step_max), max(per_step_max)))
