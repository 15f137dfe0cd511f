"""-m gpu: the TAIL of contact parity, through the C ABI.

(a) Teacher-forced: every step the device is re-synchronised to the float64 oracle's state (pmg_set_state), both take the
    same action and the single-step deviation (100 substeps, contacts included) is measured over 1024 envs x 50 steps --
    chaotic contact code tested without the chaos.  Bars are on the MAXIMUM over all 51 200 env-steps where the dynamics
    has no bifurcations (reach, push, pick_and_place), on the maximum plus an outlier count for the multi-block tasks, and
    -- for slide (a cylinder on its rim) and the chest tasks (gripper against walls / door), whose single steps DO
    bifurcate in any float32 arithmetic -- on the number of outliers relative to the float32 build of the oracle itself,
    the precision floor of the same algorithm.  Success flags must agree wherever the distance is 1e-4 off the threshold.
(b) Whole 50-step episodes without re-synchronisation: p99 of the object-position error and the number of envs beyond
    1e-3, again against the float32 oracle's own numbers.
Round-2 measurements (tools/teacher_forced.py, tools/stat_parity.py on an MI355X) are quoted next to each bar.
"""
import os
import sys

import numpy as np
import pytest

import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

pytestmark = pytest.mark.gpu

MB = {'block_stack': 4, 'block_rearrange': 3, 'chest_push': 2, 'chest_pick_and_place': 2}

# task -> {quantity: (max bar, allowed count above 1e-4[, allowed count above the max bar])}; measured maxima in the comments.
# The third entry exists for the multi-block tasks only: among 51 200 single steps of 100 substeps each, one or two contain
# a contact bifurcation in float32 (a block edge catching or missing a finger corner one substep apart; the float32 build
# of the ORACLE has 30 such steps beyond 1e-3 on the same states, tests/test_gpu_scripted.py) -- every other step is held
# to BASELINE.json's 1e-3 on its maximum.
ABSOLUTE = {
    'reach': {'tip_pos': (2e-5, 0), 'q_arm': (1e-4, 0), 'q_finger': (1e-4, 0)},                          # 5.9e-6, 3.0e-5, 2.0e-5
    'push': {'tip_pos': (1e-4, 0), 'block_pos': (5e-5, 0), 'q_arm': (1e-3, 10), 'q_finger': (2e-4, 0)},   # 1.5e-5, 6.4e-6, 2.0e-4 (3), 5.2e-5
    'pick_and_place': {'tip_pos': (1e-4, 0), 'block_pos': (1e-4, 0), 'q_arm': (3e-4, 0)},               # 2.2e-5, 2.7e-5, 5.9e-5
    'block_stack': {'tip_pos': (1e-3, 10, 2), 'block_pos': (1e-3, 12, 2), 'q_arm': (1e-3, 10, 2)},      # 3.4e-4 (3), 6.1e-4 (4), 1.9e-3 (4; 1 beyond 1e-3)
    'block_rearrange': {'tip_pos': (1e-3, 10, 2), 'block_pos': (1e-3, 20, 2), 'q_arm': (1e-3, 20, 2)},  # 2.6e-4 (2), 2.5e-4 (6), 7.5e-4 (6)
}
RELATIVE = ['slide', 'chest_push', 'chest_pick_and_place']


def _kw(task):
    return {'num_block': MB[task]} if task in MB else {}


@pytest.mark.parametrize('task', sorted(ABSOLUTE))
def test_teacher_forced_single_step_maximum(built, task):
    import teacher_forced as TF
    r = TF.run(task, 1024, 50, _kw(task), device=True, threads=oracle_lib.usable_threads())
    assert r['flag_mismatches'] == 0, r['flag_mismatches']          # reward / goal_achieved identical off the threshold
    for name, spec in ABSOLUTE[task].items():
        bar, count, beyond = spec if len(spec) == 3 else spec + (0,)
        s = r['stats'][name]
        print(task, name, s)
        assert bar <= 1e-3
        assert (s['max'] <= bar) if beyond == 0 else (s['n_gt_1e-3'] <= beyond and s['max'] <= 1e-2), (task, name, s)
        assert s['n_gt_1e-4'] <= count, (task, name, s)
        assert s['p99'] <= 2e-5, (task, name, s)                    # 99 % of all env-steps: float32 rounding


@pytest.mark.parametrize('task', RELATIVE)
def test_teacher_forced_outliers_no_worse_than_float32_oracle(built, task):
    """slide / chest: a single step can bifurcate (puck tipping over its rim, gripper wedged at a wall), so a handful of
    the 51 200 env-steps differ grossly in ANY float32 arithmetic.  The device may not have more of them than the
    float32 build of the oracle (+50 % and 5), and away from them must be float32-exact."""
    import teacher_forced as TF
    dev = TF.run(task, 1024, 50, _kw(task), device=True, threads=oracle_lib.usable_threads())
    f32 = TF.run(task, 1024, 50, _kw(task), device=False, threads=oracle_lib.usable_threads())
    assert dev['flag_mismatches'] <= f32['flag_mismatches'] + 2
    for name in ('block_pos', 'tip_pos', 'q_arm') + (('door_q',) if task.startswith('chest') else ()):
        d, f = dev['stats'][name], f32['stats'][name]
        assert d['n_gt_1e-3'] <= 1.5 * f['n_gt_1e-3'] + 5, (task, name, d, f)     # measured: slide 19 vs 29, chest_push q_arm 68 vs 73
        assert d['p99'] <= 2e-5, (task, name, d)                                  # measured <= 7.3e-6 (the oracle's own f32: 3.7e-4)
        assert d['p50'] <= 2e-6, (task, name, d)


@pytest.mark.parametrize('task', ['push', 'pick_and_place', 'block_stack', 'block_rearrange', 'chest_push'])
def test_whole_episode_tail_vs_float32_oracle(built, task):
    """50 random-policy steps without re-synchronisation, 1024 envs: the device's error against the float64 oracle --
    p99 and the count beyond 1e-3 of the object positions, and the fraction of differing success flags -- held to the
    float32 oracle's own error against the float64 oracle (same algorithm, same seeds, same actions)."""
    import warnings
    import oracle_lib
    import pybullet_multigoal_gym_amd as pmg
    N, T = 1024, 50
    kw = _kw(task)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        env = pmg.make_env(task=task, num_envs=N, seed=0, seed_stride=1, **kw)
    th = oracle_lib.usable_threads()
    o64 = oracle_lib.OracleEnv(task, N, seed_base=0, seed_stride=1, threads=th, **kw)
    o32 = oracle_lib.OracleEnv(task, N, seed_base=0, seed_stride=1, threads=th, f32=True, **kw)
    o64.reset(), o32.reset()
    env.reset(), o64.reset(), o32.reset()
    rs = np.random.RandomState(12345)
    A = env.dims.action_dim
    flag_dev = flag_f32 = 0
    for t in range(T):
        a = rs.uniform(-1, 1, (N, A)).astype(np.float32)
        o, r, d, info = env.step(a)
        a64, r64, d64, ok64 = o64.step(a)
        a32, r32, d32, ok32 = o32.step(a)
        flag_dev += int((info['goal_achieved'] != ok64).sum())
        flag_f32 += int((ok32 != ok64).sum())
    err = np.abs(o['achieved_goal'] - a64['achieved_goal']).max(1)
    spr = np.abs(a32['achieved_goal'] - a64['achieved_goal']).max(1)
    p99d, p99f = np.percentile(err, 99), np.percentile(spr, 99)
    nd, nf = int((err > 1e-3).sum()), int((spr > 1e-3).sum())
    print('whole-episode', task, 'p99 dev %.2e f32 %.2e | >1e-3 dev %d f32 %d | flags dev %d f32 %d' % (p99d, p99f, nd, nf, flag_dev, flag_f32))
    assert p99d <= 1.5 * p99f + 2e-4, (p99d, p99f)
    assert nd <= 1.25 * nf + 0.005 * N, (nd, nf)
    assert flag_dev <= 1.5 * flag_f32 + 0.002 * N * T, (flag_dev, flag_f32)
    env.close()
