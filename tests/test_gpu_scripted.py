"""-m gpu: the tasks are solved ON THE DEVICE, and the device agrees with the oracle while they are being solved.

tools/scripted_policies.py: hand-written controllers (open -> descend -> close -> lift -> carry; push-to-goal in
axis-aligned legs; stack 2 and 4; open the chest's door / lid by its handle and push / drop a block in) that read only
the observations.  Two families, through the C ABI:

(a) solvability, free-running, 256 envs per task: the success rate on the device meets the same bars as the oracle's
    (tests/test_scripted_tasks.py) and lies within 4 % of the float64 oracle's rate on the same seeds; the launch
    schedule read back every step proves that the one-env-per-wavefront lists (the gripper-on-object kernels:
    `pmg_k_step_list<1,24,0>` with the row-space solve and its LDS fallback beyond 16 contacts, `pmg_k_step_list<5,48,0>`,
    the chest kernel) carried those episodes.
(b) teacher-forced along the scripted trajectory (tools/teacher_forced.py with the policy as the action source): every
    step the device is re-synchronised to the float64 oracle's state, both take the controller's action, and the
    single-step deviation (100 substeps with the gripper ON the object) is held to bars on p99 / p99.9 and to a count of
    gross outliers no larger than the float32 build of the oracle itself produces (a block slipping in the fingers or a
    tower starting to topple one substep earlier or later is a bifurcation in ANY float32 arithmetic).  Success flags
    must be identical wherever the goal distance is 1e-4 off the threshold.
Round-3 measurements on an MI355X are quoted next to the bars.
"""
import os
import sys

import numpy as np
import pytest

import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

pytestmark = pytest.mark.gpu

# "ever succeeded" bars over ALL 256 envs / over the envs whose object starts where the arm gets behind it at once
# (scripted_suite.feasible_mask) -- the review's 95 % for every task but slide; oracle rates measured on the same seeds in
# the comments (the device's are within 1-2 envs of them, printed by the test)
BARS = {
    'reach': (1.0, 1.0),                     # 1.000
    'pick_and_place': (0.95, 0.95),          # 1.000
    'push': (0.95, 0.97),                    # 0.992; feasible 1.000
    'block_stack_2': (0.95, 0.95),           # 0.980
    'block_rearrange_2': (0.84, 0.84),       # device 0.883 / oracle 0.895 (the other block is an obstacle nobody plans around)
    'block_stack_4': (0.95, 0.95),           # 0.977
    'chest_push': (0.93, 0.97),              # 0.957; feasible 1.000
    'chest_pick_and_place': (0.95, 0.95),    # 1.000
}


@pytest.mark.parametrize('name', sorted(BARS))
def test_device_solves_the_task_with_the_scripted_policy(built, name):
    import scripted_suite as SS
    dev, ever_dev = SS.run(name, 'device', 256)
    ora, ever_ora = SS.run(name, 'oracle', 256)
    bar_all, bar_feasible = BARS[name]
    assert dev['success_ever'] >= bar_all, dev
    assert dev['success_ever_feasible'] >= bar_feasible, dev
    assert abs(dev['success_ever'] - ora['success_ever']) <= 0.04, (dev, ora)
    assert np.mean(ever_dev != ever_ora) <= 0.06, (dev, ora)       # mostly the SAME envs succeed
    if name != 'reach':   # the gripper-on-object kernels did the work, not the fast paths
        assert dev['env_steps_one_env_lists'] >= 0.3 * dev['N'] * dev['T'], dev
    print(name, 'device', dev, 'oracle', ora)


# task -> (kwargs, episode steps, {quantity: p99 bar}).  Beside the p99 bar every quantity carries a COUNT of gross single
# steps (beyond 1e-3: a contact made or missed, not rounding), held to twice the CHAOS FLOOR + 3 -- the count of the float64
# oracle against itself with its state moved by one float32 ulp and rounded to float32 after every substep
# (tools/teacher_forced.py, perturb=2: precision-independent).  Measured device / floor / float32 oracle counts of round 4
# in the comments (profiles/r04_chaos_floor.txt).
TEACHER = {
    'pick_and_place': ({}, 60, {'tip_pos': 1e-4, 'block_pos': 3e-4, 'q_arm': 2e-4}),                 # 0 / 0 / 0 everywhere
    'push': ({}, 300, {'tip_pos': 2e-5, 'block_pos': 1e-4, 'q_arm': 5e-5}),                         # tip 2 / 0.5 / 5, block 5 / 3 / 33, q_arm 2 / 2.5 / 10
    'slide': ({}, 60, {'tip_pos': 2e-5, 'block_pos': 1e-4, 'q_arm': 5e-5}),                         # round 5: tip 0 / 0, block 15 / 8, q_arm 3 / 0.5 (device / floor)
    'block_stack': ({'num_block': 4}, 340, {'tip_pos': 1e-4, 'block_pos': 2e-4, 'q_arm': 1e-4}),    # 1 / 0 / 2, 2 / 0 / 11, 2 / 0 / 8
    'block_rearrange': ({'num_block': 2}, 400, {'tip_pos': 5e-5, 'block_pos': 5e-4, 'q_arm': 1e-4}),  # 10 / 6.5 / 16, 62 / 52 / 147, 33 / 24 / 142
    'chest_push': ({'num_block': 1}, 360, {'tip_pos': 5e-5, 'block_pos': 5e-4, 'q_arm': 2e-4, 'door_q': 2e-5}),   # p99 measured 1.3e-5 / 2.4e-4 / 6.8e-5 / 2.7e-6 (the floor's: 1.2e-5 / 2.4e-4 / 6.3e-5 / 2.6e-6); counts: twice the floor + 3
    'chest_pick_and_place': ({'num_block': 1}, 100, {'tip_pos': 1e-4, 'block_pos': 2e-4, 'q_arm': 1e-4, 'door_q': 2e-5}),   # 0 / <= 1 / 0 everywhere
}
# Gross steps: twice the chaos floor + 3 on every (task, quantity) -- the exemption list (ABOVE_FLOOR: round 4 caps of 40 ... 1500,
# round 5 chest_push tip 9 / door 6) is gone: chest_push repeats every chest cylinder pair in contact in double (round 6).


@pytest.mark.parametrize('task', sorted(TEACHER))
def test_teacher_forced_along_the_scripted_trajectory(built, task):
    import scripted_policies as SP
    import teacher_forced as TF
    kw, T, bars = TEACHER[task]
    N = 256
    kw = dict(kw, max_episode_steps=T)
    pkw = {'num_block': kw['num_block']} if 'num_block' in kw else {}
    th = oracle_lib.usable_threads()
    dev = TF.run(task, N, T, kw, device=True, threads=th, policy=SP.make_policy(task, N, **pkw), keep_schedule=True, perturb=2)
    print(task, 'final success (oracle trajectory) %.3f' % dev['final_success'], 'lists', dev['schedule_env_steps'])
    for name in bars:
        print('   %-10s device %s' % (name, dev['stats'][name]))
        print('   %-10s chaos  %s' % (name, dev['chaos'][name]))
    assert dev['flag_mismatches'] == 0, dev['flag_mismatches']
    assert dev['schedule_env_steps']['prone'] >= 0.3 * N * T, dev['schedule_env_steps']     # one-env lists under load
    if task != 'slide':
        assert dev['final_success'] >= 0.6                         # the trajectory compared is one that solves the task
    for name, p99 in bars.items():
        d, c = dev['stats'][name], dev['chaos'][name]
        assert p99 <= 1e-3
        assert d['p99'] <= p99, (task, name, d)
        assert d['n_gt_1e-3'] <= 2 * c['floor'] + 3, (task, name, d, c)             # gross steps: twice the chaos floor, no exemptions
        assert d['p50'] <= 2e-6, (task, name, d)                   # the typical step: float32 rounding
