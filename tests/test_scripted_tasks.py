"""CPU tier: the tasks are SOLVABLE, and the kernels agree with the oracle while they are being solved.

The reference's tasks exist to be solved (R/README.md:13-26; kuka_single_step_envs.py:4-16; kuka_multi_step_envs.py:34-87,
256-342).  tools/scripted_policies.py holds hand-written controllers that do it from the observations alone -- open,
descend, close, lift, carry; push-to-goal; stack; open the chest and drop / push a block in.  Here:

(a) solvability on the oracle: success rates of 64 envs per task against bars a little under the measured rates;
(b) the emulated product kernels (tests/emu: the HIP sources compiled for the CPU), teacher-forced along those
    trajectories at the moments that matter -- fingers closing on the block, the lift, a block set down on another one,
    the gripper pushing a block, the fingers on the chest lid's handle -- i.e. the gripper-on-object solver paths
    (`obj_contact_pgs_rowspace`, the one-env-per-wavefront lists) that a random policy practically never reaches.
The -m gpu counterpart (tests/test_gpu_scripted.py) runs both families on the device at 256 envs.
"""
import os
import sys
import warnings

import numpy as np
import pytest

import oracle_lib as O
import pybullet_multigoal_gym_amd as pmg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import scripted_policies as SP  # noqa: E402
import scripted_suite as SS     # noqa: E402

# task -> (bar on "ever succeeded" over all envs, bar over the envs whose object starts where the arm gets behind it at
# once (scripted_suite.feasible_mask)); measured (64 / 256 envs, float64 oracle) in the comments.  Since the far-edge
# detour of the push legs, the off-centre grasp and the staging of far-edge blocks (tools/scripted_policies.py) every task
# but slide is solved for >= 95 % of ALL envs at 256 envs; the 64-env bars sit one or two envs under the measured rates.
BARS = {
    'reach': (1.0, 1.0),                     # 1.000 / 1.000
    'pick_and_place': (0.97, 0.97),          # 1.000 / 1.000
    'push': (0.95, 0.97),                    # 0.984 / 0.992 (feasible 1.000)
    'block_stack_2': (0.95, 0.95),           # 1.000 / 0.980
    'block_rearrange_2': (0.80, 0.80),       # 0.875 / 0.895: two blocks pushed to their slots one after the other, the other block an obstacle nobody plans around
    'block_stack_4': (0.92, 0.92),           # 0.969 / 0.977 (ever; 0.86-0.89 still standing at step 340: a four-high tower creeps)
    'chest_push': (0.90, 0.97),              # 0.938 / 0.957 (feasible 1.000)
    'chest_pick_and_place': (0.97, 0.97),    # 1.000 / 1.000
}


@pytest.mark.parametrize('name', sorted(BARS))
def test_oracle_solves_the_task_with_the_scripted_policy(built, name):
    out, ever = SS.run(name, 'oracle', 64)
    bar_all, bar_feasible = BARS[name]
    assert out['success_ever'] >= bar_all, out
    assert out['success_ever_feasible'] >= bar_feasible, out


def test_oracle_slide_is_pushed_to_the_edge_of_the_workspace(built):
    """Slide cannot be solved quasi-statically in this reference: the goal box lies 0.4 m beyond the object box
    (kuka_single_step_base_env.py:66-69), the tip target moves 1 cm per step (kuka.py:209, <= 0.15 m/s peak under the
    kp = 0.03 position motors) and the puck decelerates at mu*g = 0.49 m/s^2, i.e. coasts ~2 cm.  What the controller
    can do is push the puck to the lower x limit of the tip box -- which it does, for every puck it can get behind."""
    T = 60
    env = O.OracleEnv('slide', 32, seed_base=0, seed_stride=1, threads=O.usable_threads(), max_episode_steps=T)
    env.reset()
    obs = env.reset()
    x0 = obs['achieved_goal'][:, 0].copy()
    pol = SP.make_policy('slide', 32)
    obs, ok, ever = SP.rollout(env, pol, T, obs)
    x1 = obs['achieved_goal'][:, 0]
    reachable = x0 <= -0.46                             # the arm is slow to get behind a puck at the far edge of its reach
    assert reachable.sum() >= 20
    assert np.all(x1[reachable] < x0[reachable] - 0.05)   # every such puck travelled towards the goal box ...
    assert np.all(x1[reachable] < -0.70)                # ... up to where the tip box ends (-0.67 - finger - radius)
    assert np.all(np.abs(obs['achieved_goal'][:, 2] - 0.17) < 2e-3)   # flat on the long table


def _quiet_env(task, lib, n, **kw):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        return pmg.make_env(task=task, num_envs=n, seed=0, seed_stride=1, _library=lib, **kw)


def _fly(task, n, T, watch, **kw):
    """Run the scripted policy on the float64 oracle; `watch(t, policy, obs)` -> True marks step t for the comparison.
    Returns [(state before, action, state after, obs after, ok after)] of the marked steps."""
    ora = O.OracleEnv(task, n, seed_base=0, seed_stride=1, threads=O.usable_threads(), max_episode_steps=T, **kw)
    ora.reset()
    obs = ora.reset()
    pol = SP.make_policy(task, n, **({'num_block': kw['num_block']} if 'num_block' in kw else {}))
    marked = []
    for t in range(T):
        a = pol.act(obs)
        take = watch(t, pol, obs)
        s0 = ora.get_state() if take else None
        obs, r, d, ok = ora.step(a)
        if take:
            marked.append((s0, a, ora.get_state(), obs, ok))
    return marked


def _teacher_forced(env, marked, nb, tip_bar, q_bar, blk_bar):
    worst = {'tip': 0.0, 'q': 0.0, 'blk': 0.0}
    for s0, a, s1, obs1, ok1 in marked:
        env.set_state(s0)
        o, r, d, info = env.step(a)
        s = env.get_state()
        worst['tip'] = max(worst['tip'], float(np.abs(o['observation'][:, :3] - obs1['observation'][:, :3]).max()))
        worst['q'] = max(worst['q'], float(np.abs(s[:, :9] - s1[:, :9]).max()))
        for b in range(nb):
            worst['blk'] = max(worst['blk'], float(np.abs(s[:, 64 + 13 * b:67 + 13 * b] - s1[:, 64 + 13 * b:67 + 13 * b]).max()))
        dist = np.linalg.norm(obs1['achieved_goal'].astype(np.float64) - obs1['desired_goal'], axis=1)
        clear = np.abs(dist - 0.05) > 1e-4
        assert np.array_equal(info['goal_achieved'][clear], ok1[clear])
    assert worst['tip'] <= tip_bar and worst['q'] <= q_bar and worst['blk'] <= blk_bar, worst
    return worst


def test_emulated_grasp_and_lift_matches_oracle(emu_library):
    """pick_and_place: the three steps in which the fingers close on the block and the first two of the carry -- finger x
    block contacts on both sides, the block leaving the table (one free object in row space, one env per wavefront)."""
    carried = []

    def watch(t, pol, obs):
        if np.any(pol.phase == 2):
            return True
        if np.all(pol.phase == 5) and len(carried) < 2:
            carried.append(t)
            return True
        return False
    marked = _fly('pick_and_place', 2, 30, watch)   # (two envs: one goal on the table, one in the air)
    assert 4 <= len(marked) <= 8 and len(carried) == 2
    env = _quiet_env('pick_and_place', emu_library, 2, max_episode_steps=30)
    env.reset()
    w = _teacher_forced(env, marked[-5:], 1, tip_bar=1e-5, q_bar=2e-5, blk_bar=2e-5)
    held = marked[-1][3]['observation']
    assert np.any(held[:, 5] > 0.18) and np.all(np.abs(held[:, 3:6] - held[:, 0:3]).max(1) < 0.01), 'the block is in the gripper'
    env.close()
    print('grasp-and-lift, emulated kernels vs oracle:', w)


def test_emulated_push_contact_matches_oracle(emu_library):
    """push: steps of the closed gripper shoving the block along the table."""
    marked = _fly('push', 1, 60, lambda t, pol, obs: bool(np.all(pol.phase == 3)))
    shoving = [m for m in marked if np.all(np.abs(m[2][:, 64:66] - m[0][:, 64:66]).max(1) > 0.004)]   # the block moves
    assert len(shoving) >= 2
    env = _quiet_env('push', emu_library, 1, max_episode_steps=60)
    env.reset()
    w = _teacher_forced(env, shoving[:3], 1, tip_bar=1e-5, q_bar=2e-5, blk_bar=2e-5)
    env.close()
    print('push, emulated kernels vs oracle:', w)


def test_emulated_stacking_set_down_matches_oracle(emu_library):
    """block_stack-2: the step in which the carried block comes down on the other one and the step the fingers let go
    (block x block + finger x block contacts: the multi-block one-env list)."""
    marked = _fly('block_stack', 1, 130, lambda t, pol, obs: bool(pol.cur[0] == 1 and ((pol.phase[0] == 5 and pol.count[0] >= 1) or pol.phase[0] == 6)),
                  num_block=2)
    assert len(marked) >= 2
    env = _quiet_env('block_stack', emu_library, 1, max_episode_steps=130, num_block=2)
    env.reset()
    w = _teacher_forced(env, marked[:2], 2, tip_bar=1e-5, q_bar=2e-5, blk_bar=2e-5)
    env.close()
    print('set-down on a block, emulated kernels vs oracle:', w)
