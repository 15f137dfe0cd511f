"""-m gpu: the TAIL of contact parity, through the C ABI.

(a) Teacher-forced: every step the device is re-synchronised to the float64 oracle's state (pmg_set_state), both take the
    same action and the single-step deviation (100 substeps, contacts included) is measured over 1024 envs x 50 steps --
    chaotic contact code tested without the chaos.  Bars are on the MAXIMUM over all 51 200 env-steps where the dynamics
    has no bifurcations (reach, push, pick_and_place), on the maximum plus an outlier count for the multi-block tasks, and
    -- for slide (a cylinder on its rim) and the chest tasks (gripper against walls / door), whose single steps DO
    bifurcate -- on the number of gross steps relative to the CHAOS FLOOR (the float64 oracle against itself under
    float32-sized state noise; the tasks that exceed twice that floor are listed with their caps).  Success flags must
    agree wherever the distance is off the threshold by more than the step's own position error + 1e-4.
(b) Whole 50-step episodes without re-synchronisation: p99 of the object-position error and the number of envs beyond
    1e-3, against the chaos floor's own numbers.
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


# Gross single steps (beyond 1e-3) are held to twice the chaos floor + 3 for EVERY (task, quantity): the list of exemptions
# (ABOVE_FLOOR: round 4 twelve caps of 35 ... 1500, round 5 five of 6 ... 22) is gone.  Round 6: slide repeats
# every finger x puck pair in contact in double -- speculatively, on a third wavefront of its list-0 workgroups --, the chest tasks
# every chest cylinder pair in contact (gripper base x walls / door, handle x fingers), in every chest kernel.  Measured at four
# times this sample (204 800 steps, profiles/r06_chaos_floor_large_sample_204800_steps.txt), device / floor: slide 0 / 3 / 1 against
# 0 / 0 / 1; chest_push 0 / 1 / 5 / 0 against 0.5 / 2.5 / 3.5 / 0; chest_pick_and_place 7 / 0 / 21 / 12 against 5.5 / 0 / 14 / 8.
P99 = {}


@pytest.mark.parametrize('task', RELATIVE)
def test_teacher_forced_outliers_against_the_chaos_floor(built, task):
    """slide / chest: a single step can bifurcate (puck tipping over its rim, gripper wedged at a wall).  The yardstick is
    the CHAOS FLOOR: the float64 oracle against itself with the state moved by one float32 ulp and rounded to float32
    after every substep (tools/teacher_forced.py, perturb=2) -- how often a step bifurcates under float32-sized state
    noise whatever the arithmetic.  Gross steps (beyond 1e-3): twice the floor + 3, no exemptions; away from them the device must
    be float32-exact."""
    import teacher_forced as TF
    dev = TF.run(task, 1024, 50, _kw(task), device=True, threads=oracle_lib.usable_threads(), perturb=2)
    assert dev['flag_mismatches'] <= 2
    for name in ('block_pos', 'tip_pos', 'q_arm') + (('door_q',) if task.startswith('chest') else ()):
        d, c = dev['stats'][name], dev['chaos'][name]
        print(task, name, 'device', d, 'chaos', c)
        assert d['n_gt_1e-3'] <= 2 * c['floor'] + 3, (task, name, d, c)
        assert d['p99'] <= P99.get((task, name), 2e-5), (task, name, d)
        assert d['p50'] <= 2e-6, (task, name, d)


@pytest.mark.parametrize('task', RELATIVE)
def test_single_steps_at_the_chaos_floor_large_sample(built, task):
    """The same comparison at four times the sample -- 4096 envs x 50 random-policy steps = 204 800 teacher-forced single steps:
    rounds 1-5 passed the 51 200-step bar with slide's 15-25 gross puck steps hidden inside `2 x floor + 3` plus a cap; at this sample
    slide's floor is 0 and a build without the double repeat of the finger x puck contacts measures 25 (chest_push joints 19 against a
    floor of 3.5, chest_pick_and_place joints 56 against 14).  Bar: <= 2 x floor + 3 on every quantity (measured, round 6: slide puck
    3 / tip 0 / joints 1; chest_push 1 / 0 / 5, door 0; chest_pick_and_place 0 / 6 / 18, door 9)."""
    import teacher_forced as TF
    dev = TF.run(task, 4096, 50, _kw(task), device=True, threads=oracle_lib.usable_threads(), perturb=2)
    assert dev['flag_mismatches'] <= 8
    for name in ('block_pos', 'tip_pos', 'q_arm') + (('door_q',) if task.startswith('chest') else ()):
        d, c = dev['stats'][name], dev['chaos'][name]
        print(task, 'x 204 800', name, 'device', d['n_gt_1e-3'], 'floor', c['floor_per_perturbed_oracle'])
        assert d['n'] == 204800
        assert d['n_gt_1e-3'] <= 2 * c['floor'] + 3, (name, d, c)
        assert d['p99'] <= 2e-5 and d['p50'] <= 2e-6, (name, d)


# whole episodes: (p99 of the object-position error, envs beyond 1e-3, differing flags) where twice the chaos floor does not hold.
# Round 4 listed chest_push-2 (p99 1.3e-2 against the floor's 1.7e-3, 42 envs of 1024 beyond 1e-3 against 15: caps 2e-2 / 60 /
# 205); round 5 measures 3.1e-3 / 25 / 0 against 1.7e-3 / 14 / 0 -- inside the bar: the list is empty.
EPISODE_ABOVE_FLOOR = {}


@pytest.mark.parametrize('task', ['push', 'pick_and_place', 'block_stack', 'block_rearrange', 'chest_push'])
def test_whole_episode_tail_against_the_chaos_floor(built, task):
    """50 random-policy steps without re-synchronisation, 1024 envs: the device's error against the float64 oracle --
    p99 and the count beyond 1e-3 of the object positions, and the number of differing success flags -- held to twice
    the chaos floor's own (oracle_lib.FloorOracle: float64 arithmetic, the state rounded to float32 after every substep;
    same seeds, same actions), or to the listed cap where that does not hold."""
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
    ofl = oracle_lib.FloorOracle(task, N, seed_base=0, seed_stride=1, threads=th, **kw)
    o64.reset(), ofl.reset()
    env.reset(), o64.reset(), ofl.reset()
    rs = np.random.RandomState(12345)
    A = env.dims.action_dim
    flag_dev = flag_fl = 0
    for t in range(T):
        a = rs.uniform(-1, 1, (N, A)).astype(np.float32)
        o, r, d, info = env.step(a)
        a64, r64, d64, ok64 = o64.step(a)
        afl, rfl, dfl, okfl = ofl.step(a)
        flag_dev += int((info['goal_achieved'] != ok64).sum())
        flag_fl += int((okfl != ok64).sum())
    err = np.abs(o['achieved_goal'] - a64['achieved_goal']).max(1)
    spr = np.abs(afl['achieved_goal'] - a64['achieved_goal']).max(1)
    p99d, p99f = np.percentile(err, 99), np.percentile(spr, 99)
    nd, nf = int((err > 1e-3).sum()), int((spr > 1e-3).sum())
    print('whole-episode', task, 'p99 dev %.2e floor %.2e | >1e-3 dev %d floor %d | flags dev %d floor %d' % (p99d, p99f, nd, nf, flag_dev, flag_fl))
    cap = EPISODE_ABOVE_FLOOR.get(task)
    if cap is None:
        assert p99d <= 2 * p99f + 2e-4, (p99d, p99f)
        assert nd <= 2 * nf + 0.005 * N, (nd, nf)
        assert flag_dev <= 2 * flag_fl + 0.002 * N * T, (flag_dev, flag_fl)
    else:
        assert p99d <= cap[0] and nd <= cap[1] and flag_dev <= cap[2], (p99d, nd, flag_dev, cap)
    env.close()
