"""CPU tier: the chest tasks (chest_push: front sliding door; chest_pick_and_place: up sliding lid) -- SURVEY.md 8(f)-4.
Oracle semantics against an independent restatement of the reference's goal code and against analytic door physics,
then the product's device code (fiber emulator build) against the oracle."""
import warnings

import numpy as np
import pytest

import oracle_lib as O
import pybullet_multigoal_gym_amd as pmg

CENTRE = np.array([-0.65, 0.0, 0.175])
TOP = np.array([-0.65, 0.0, 0.3])


def _ref_goal(task, nb, grip, ind, blocks, tip, closeness):
    """kuka_multi_step_envs.py:256-342 (pick and place) / 405-475 (push) for sub-goal `ind` (None = the final goal);
    the open-the-door sub-goal is cut to the goal length without grip_informed_goal (see oracle chest_goal())."""
    pnp = task == 'chest_pick_and_place'
    door = [0.10 if pnp else 0.12]
    blocks = [np.array(b, float) for b in blocks]
    if ind is None:
        g = [door] + [CENTRE] * nb
        if grip:
            g += [TOP, [0.06]] if pnp else [CENTRE + [0.03, 0, 0]]
        return np.concatenate(g)
    if ind == 0:
        g = [door] + blocks
        if grip:
            g += [tip] + ([[closeness]] if pnp else [])
        return np.concatenate(g)
    if not grip:
        return np.concatenate([door] + [CENTRE if i <= ind - 1 else blocks[i] for i in range(nb)])
    per = 3 if pnp else 2
    j, ph = (ind - 1) // per, (ind - 1) % per
    g = [CENTRE if i < j else blocks[i] for i in range(nb)]
    if pnp:
        if ph == 0:
            g += [blocks[j], [0.03]]
        elif ph == 1:
            g[j] = TOP
            g += [TOP, [0.03]]
        else:
            g[j] = CENTRE
            g += [TOP, [0.06]]
    else:
        if ph == 0:
            g += [blocks[j] + [0.03, 0, 0]]
        else:
            g[j] = CENTRE
            g += [CENTRE + [0.03, 0, 0]]
    return np.concatenate([door] + g)


@pytest.mark.parametrize('task', ['chest_push', 'chest_pick_and_place'])
@pytest.mark.parametrize('grip', [False, True])
def test_oracle_chest_layout_and_goals(built, task, grip):
    nb = 3
    pnp = task == 'chest_pick_and_place'
    ora = O.OracleEnv(task, 1, num_block=nb, seed_base=5, seed_stride=1, grip_informed_goal=grip, task_decomposition=True)
    d = ora.dims
    assert (d.action_dim, d.observation_dim, d.policy_state_dim) == (4 if pnp else 3, 8 + 16 * nb + 20, 4 + 3 * nb + 19)
    assert d.goal_dim == 1 + 3 * nb + ((4 if pnp else 3) if grip else 0)
    ora.reset()
    o = ora.reset()
    st = ora.get_state()[0]
    blocks = [st[64 + 13 * b:67 + 13 * b] for b in range(nb)]
    # blocks start outside the chest, inside the shifted object box (kuka_multi_step_base_env.py:102-105)
    for b in blocks:
        assert -0.54 <= b[0] <= -0.40 and -0.15 <= b[1] <= 0.15 and b[2] == np.float32(0.175)
    obs = o['observation'][0]
    tip, closeness = obs[:3], obs[3]
    ag = o['achieved_goal'][0]
    assert ag[0] == 0 and np.array_equal(ag[1:1 + 3 * nb], np.concatenate(blocks))
    if grip:
        assert np.array_equal(ag[1 + 3 * nb:4 + 3 * nb], tip) and (not pnp or ag[-1] == closeness)
    # door joint state + the three key points ride behind the blocks
    tail = obs[8 + 16 * nb:]
    assert tail[0] == 0 and tail[1] == 0
    kp = tail[2:].reshape(3, 6)
    if pnp:   # lid: left / right / handle key points (chest_up_sliding_door.urdf)
        assert np.allclose(kp[:, :3], [[-0.6, 0.07, 0.267], [-0.6, -0.07, 0.267], [-0.555, 0.065, 0.26702]], atol=1e-5)
    else:
        assert np.allclose(kp[:, :3], [[-0.597, -0.07, 0.21], [-0.597, 0.07, 0.21], [-0.577, 0.0, 0.25001]], atol=1e-5)
    assert np.array_equal(o['policy_state'][0][4 + 3 * nb:], np.r_[tail[0], tail[2:]])
    # after reset sub_goal_ind = -1: the final goal; then every sub-goal against the reference restatement
    steps = nb * ((3 if pnp else 2) if grip else 1) + 1
    assert np.allclose(o['desired_goal'][0], _ref_goal(task, nb, grip, None, blocks, tip, closeness), atol=1e-6)
    for ind in range(steps):
        ora.set_sub_goal(ind)
        g = ora.reset(mask=np.zeros(1, bool))['desired_goal'][0]
        assert np.allclose(g, _ref_goal(task, nb, grip, ind, blocks, tip, closeness)[:d.goal_dim], atol=1e-6), ind
    ora.set_sub_goal(-1)
    assert np.allclose(ora.reset(mask=np.zeros(1, bool))['desired_goal'][0], _ref_goal(task, nb, grip, steps - 1, blocks, tip, closeness), atol=1e-6)
    with pytest.raises(Exception):
        ora.set_sub_goal(steps)


@pytest.mark.parametrize('task,upper,mass', [('chest_push', 0.12, 2.0), ('chest_pick_and_place', 0.10, 4.0)])
def test_oracle_chest_door_physics(built, task, upper, mass):
    """The door is one prismatic DoF without friction, damping or gravity along its axis: it coasts at constant
    velocity, stops at its limits, and once found within 1 cm of the open state the position motor holds it there
    (kuka_multi_step_base_env.py:296-298)."""
    ora = O.OracleEnv(task, 1, num_block=1, seed_base=1, seed_stride=1)
    ora.reset()
    A = ora.dims.action_dim
    up = np.zeros((1, A), np.float32)
    up[0, 2] = 1.0
    for _ in range(6):
        ora.step(up)                                   # lift the gripper clear of everything
    st = ora.get_state().copy()
    st[0, 48:51] = [0.03, 0.1, 0.0]                    # door at 3 cm moving at 0.1 m/s, motor off
    ora.set_state(st)
    o = ora.step(np.zeros((1, A), np.float32))[0]
    q, qd = o['achieved_goal'][0, 0], o['observation'][0, -20 + 1]
    assert abs(qd - 0.1) < 1e-6 and abs(q - (0.03 + 0.1 * 0.2)) < 1e-6        # 100 substeps x 2 ms, nothing acts on it
    for _ in range(5):
        o = ora.step(np.zeros((1, A), np.float32))[0]
    q, qd = o['achieved_goal'][0, 0], o['observation'][0, -20 + 1]
    assert abs(q - upper) < 2e-3 and abs(qd) < 1e-3                            # stopped by the upper limit
    assert ora.get_state()[0, 50] == 1.0                                       # found open: the motor is latched
    st = ora.get_state().copy()
    st[0, 49] = -0.2                                                           # knock it back towards closed
    ora.set_state(st)
    for _ in range(3):
        o = ora.step(np.zeros((1, A), np.float32))[0]
    assert abs(o['achieved_goal'][0, 0] - upper) < 0.011                       # 500 N on 2-4 kg: it does not get away
    # closed door, motor off, pushed against the lower limit: stays at 0
    ora.reset()
    st = ora.get_state().copy()
    st[0, 48:51] = [0.0, -0.3, 0.0]
    ora.set_state(st)
    o = ora.step(np.zeros((1, A), np.float32))[0]
    assert abs(o['achieved_goal'][0, 0]) < 1e-3 and ora.get_state()[0, 50] == 0.0


def test_oracle_chest_walls_hold_blocks(built):
    """A block dropped inside the chest rests on the table between the walls; one thrown at the back wall from inside
    stays inside; the closed front door keeps a block that slides towards it out."""
    ora = O.OracleEnv('chest_push', 1, num_block=2, seed_base=1, seed_stride=1)
    ora.reset()
    st = ora.get_state().copy()
    st[0, 64:67] = [-0.65, 0.0, 0.20]; st[0, 71:74] = [-0.4, 0.0, 0.0]        # inside, moving at the back wall
    st[0, 77:80] = [-0.56, 0.0, 0.175]; st[0, 84:87] = [-0.3, 0.0, 0.0]       # outside, sliding at the closed door
    ora.set_state(st)
    for _ in range(4):
        o = ora.step(np.zeros((1, 3), np.float32))[0]
    b0, b1 = o['achieved_goal'][0, 1:4], o['achieved_goal'][0, 4:7]
    assert -0.695 + 0.015 - 2e-3 <= b0[0] <= -0.6 and abs(b0[2] - 0.175) < 1e-3  # stopped by the back wall, on the table
    assert b1[0] >= -0.592 + 0.015 - 2e-3 and abs(b1[2] - 0.175) < 1e-3           # stopped by the door's front face
    assert abs(o['achieved_goal'][0, 0]) < 1e-4                                    # a push along x does not move the door


def _quiet_env(task, lib, **kw):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        return pmg.make_env(task=task, num_envs=1, seed_stride=1, _library=lib, **kw)


def _compare(o, oo, tol_obs, tol_pos=2e-4):
    assert np.abs(o['observation'] - oo['observation']).max() < tol_obs
    assert np.abs(o['policy_state'] - oo['policy_state']).max() < tol_obs
    assert np.abs(o['achieved_goal'] - oo['achieved_goal']).max() < tol_pos
    assert np.abs(o['desired_goal'] - oo['desired_goal']).max() < tol_pos


def test_emulated_chest_push_drags_the_door_like_the_oracle(emu_library):
    """The gripper base pressed against the sliding door drags it open by friction while the arm slides along it: walls,
    door slot, door limit and the motor latch on the device against the oracle, step by step."""
    env = _quiet_env('chest_push', emu_library, num_block=2, seed=3)
    ora = O.OracleEnv('chest_push', 1, num_block=2, seed_base=3, seed_stride=1)
    ora.reset()
    o, oo = env.reset(), ora.reset()
    for k in o:
        assert np.abs(o[k] - oo[k]).max() < 1e-6
    assert np.abs(env.get_state() - ora.get_state()).max() < 1e-6
    a = np.zeros((1, 3), np.float32)
    for t in range(13):
        a[0] = [-1, 0, 1 if t < 8 else 0]
        if t < 7:                      # the approach is flown by the oracle alone (the emulator is slow) ...
            oo, ro, do, oko = ora.step(a)
            continue
        if t == 7:                     # ... the device takes over from its state as the gripper base reaches the door
            env.set_state(ora.get_state())
        o, r, d, info = env.step(a)
        oo, ro, do, oko = ora.step(a)
        _compare(o, oo, 2e-3)
        assert r[0] == ro[0]
    assert oo['achieved_goal'][0, 0] > 0.11                      # the door did open ...
    assert ora.get_state()[0, 50] == 1.0 and env.get_state()[0, 50] == 1.0   # ... and both latched the motor
    env.close()


@pytest.mark.parametrize('task', ['chest_push', 'chest_pick_and_place'])
def test_emulated_chest_blocks_walls_and_moving_door_match_oracle(emu_library, task):
    """Blocks thrown at the walls from inside the chest and at the door from outside, the door itself moving."""
    pnp = task == 'chest_pick_and_place'
    env = _quiet_env(task, emu_library, num_block=2, seed=4)
    # stick / slip of the block rubbing along the moving door amplifies rounding: the float32 build of the oracle is
    # the like-for-like reference here (the float64 one ends 0.5 mm away on the door after three steps)
    ora = O.OracleEnv(task, 1, num_block=2, seed_base=4, seed_stride=1, f32=True)
    ora.reset(), env.reset()
    st = ora.get_state().copy()
    st[0, 48:51] = [0.02, 0.15, 0.0]
    st[0, 64:67] = [-0.65, 0.03, 0.19]; st[0, 71:74] = [-0.3, 0.25, 0.0]
    st[0, 77:80] = [-0.56, -0.02, 0.175 if not pnp else 0.30]; st[0, 84:87] = [-0.3, 0.0, 0.0]
    if pnp:                                                      # the second block falls on the lid
        st[0, 77:79] = [-0.65, 0.0]; st[0, 84:87] = 0
    env.set_state(st), ora.set_state(st)
    a = np.zeros((1, env.dims.action_dim), np.float32)
    for t in range(3):
        o, r, d, info = env.step(a)
        oo, ro, do, oko = ora.step(a)
        _compare(o, oo, 1e-2, 1e-4)   # obs carries the (jittery) relative velocities of resting blocks; poses are tight
    env.close()


@pytest.mark.parametrize('task', ['chest_push', 'chest_pick_and_place'])
def test_emulated_chest_sub_goals_match_oracle(emu_library, task):
    nb = 2
    kw = dict(num_block=nb, grip_informed_goal=True, task_decomposition=True)
    env = _quiet_env(task, emu_library, seed=6, **kw)
    ora = O.OracleEnv(task, 1, seed_base=6, seed_stride=1, **kw)
    ora.reset()
    o, oo = env.reset(), ora.reset()
    assert np.abs(o['desired_goal'] - oo['desired_goal']).max() < 1e-6
    steps = env.num_steps
    assert steps == nb * (3 if task == 'chest_pick_and_place' else 2) + 1
    subs = env.sub_goals
    for k in range(steps):
        g = env.set_sub_goal(k)
        ora.set_sub_goal(k)
        ref = ora.reset(mask=np.zeros(1, bool))['desired_goal']
        assert np.abs(g - ref).max() < 1e-6 and np.abs(subs[k] - ref).max() < 1e-6
    # stepping under sub-goal 0: the gripper goal follows the gripper, reward is about the door alone
    env.set_sub_goal(0), ora.set_sub_goal(0)
    a = np.zeros((1, env.dims.action_dim), np.float32)
    a[0, 2] = 1
    o, r, d, info = env.step(a)
    oo, ro, do, oko = ora.step(a)
    _compare(o, oo, 2e-3)
    assert np.array_equal(o['desired_goal'][0, 1 + 3 * nb:], o['achieved_goal'][0, 1 + 3 * nb:])
    assert r[0] == ro[0] == -1.0                                   # door still closed: 0.10 / 0.12 away
    with pytest.raises(Exception):
        env.set_sub_goal(steps)
    with pytest.raises(Exception):
        env.set_goal(np.zeros((1, env.dims.goal_dim), np.float32))
    env.close()




@pytest.mark.parametrize('task', ['chest_push', 'chest_pick_and_place'])
def test_emulated_chest_curriculum_matches_oracle(emu_library, task):
    """The device's reset kernel (MT19937, level / moved-block draws over num_block + 1 levels, schedule) against the oracle
    -- which tests/test_reference_golden.py holds to the reference's own _generate_curriculum -- plus the gripper-informed
    goal of level 0 (the gripper stays where it is)."""
    nb, seed = 3, 3
    budget = 8 * (nb + 1)
    env = _quiet_env(task, emu_library, num_block=nb, seed=seed, use_curriculum=True, num_goals_to_generate=budget)
    env.activate_curriculum_update()
    # the constructor consumed one reset (base_env.py:84) before the update was switched on: it drew episode 0's
    # blocks and level without counting it, so the schedule is compared from the second episode of a fresh stream
    ora = O.OracleEnv(task, 1, num_block=nb, seed_base=seed, seed_stride=1, use_curriculum=True, num_goals_to_generate=budget)
    ora.reset()
    ora.curriculum_update(True)
    for _ in range(12):
        o, oo = env.reset(), ora.reset()
        c = ora.curriculum()
        assert np.array_equal(o['desired_goal'], oo['desired_goal'])
        assert np.array_equal(env.last_curriculum_level, c['level']) and np.array_equal(env.curriculum_goal_step, c['goal_step'])
        assert np.array_equal(env.curriculum_prob, c['prob']) and np.array_equal(env.num_generated_goals_per_curriculum, c['generated'])
        assert np.array_equal(env.get_state()[:, 63], ora.get_state()[:, 63])
    assert c['generated'].sum() == 12
    env.close()
    # level 0 with gripper-informed goals: desired gripper pose = achieved gripper pose
    env = _quiet_env(task, emu_library, num_block=2, seed=1, use_curriculum=True, grip_informed_goal=True)
    ora = O.OracleEnv(task, 1, num_block=2, seed_base=1, seed_stride=1, use_curriculum=True, grip_informed_goal=True)
    ora.reset()
    o, oo = env.reset(), ora.reset()
    assert env.last_curriculum_level[0] == 0
    assert np.abs(o['desired_goal'] - oo['desired_goal']).max() < 1e-6
    assert np.array_equal(o['desired_goal'][0, 7:], o['achieved_goal'][0, 7:])
    env.close()


def test_emulated_finger_opens_the_door_by_its_handle(emu_library):
    """What the task is about: the closed fingers come in over the door, next to the handle cylinder, and push it
    sideways -- finger x handle (cylinder x box) contacts driving the door DoF.  The oracle flies the approach, the device
    takes over from its state for the push."""
    env = _quiet_env('chest_push', emu_library, num_block=1, seed=3)
    ora = O.OracleEnv('chest_push', 1, num_block=1, seed_base=3, seed_stride=1, f32=True)
    ora.reset(), env.reset(), ora.reset()
    for a, n in (([0, 0, 1], 6), ([0, -1, 0], 3), ([-1, 0, 0], 7)):
        for _ in range(n):
            oo = ora.step(np.float32([a]))[0]
    assert abs(oo['achieved_goal'][0, 0]) < 1e-3                  # about to touch: the door is still closed
    env.set_state(ora.get_state())
    for _ in range(3):
        o = env.step(np.float32([[0, 1, 0]]))[0]
        oo = ora.step(np.float32([[0, 1, 0]]))[0]
        _compare(o, oo, 5e-2, 1e-4)   # poses tight; the velocities of a finger scraping along the handle are noisy
    assert oo['achieved_goal'][0, 0] > 0.015                      # pushed open by almost 2 cm in three steps
    env.close()


def test_emulated_chest_two_list_split_and_redo(emu_library):
    """The chest tasks' launch plan (round 3): an env whose gripper can touch the chest or a block keeps the full layout
    (list 0: 48 contacts, a stage slot per pair), everybody else runs ContactLds<6, 30> with stage slots handed out by
    rank among the surviving pairs (list 1, 20 KB of LDS instead of 32), and a list-1 env whose contacts do not fit is
    recomputed by pmg_k_redo_chest.  Three envs, one of each kind, against the oracle."""
    nb = 5
    env = _quiet_env('chest_push', emu_library, num_block=nb, seed=5)
    env.close()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        env = pmg.make_env(task='chest_push', num_envs=3, seed=5, seed_stride=1, num_block=nb, _library=emu_library)
    ora = O.OracleEnv('chest_push', 3, num_block=nb, seed_base=5, seed_stride=1)
    ora.reset()
    env.reset(), ora.reset()
    # env 1: the oracle flies the gripper to the chest's front door (as the door-dragging test above)
    a = np.zeros((3, 3), np.float32)
    for t in range(7):
        a[:] = 0
        a[1] = [-1, 0, 1]
        ora.step(a)
    st = ora.get_state()
    # env 2: gripper far from everything, the five blocks in a tight row: 20 table contacts + 4 x 4 block x block = 36 > 30
    for b in range(nb):
        st[2, 64 + 13 * b:77 + 13 * b] = [-0.45, -0.09 + 0.0302 * b, 0.175, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0]
    # env 0: gripper where it starts, the blocks spread out away from it
    for b, y in enumerate([-0.15, -0.09, 0.09, 0.15, 0.21]):
        st[0, 64 + 13 * b:77 + 13 * b] = [-0.45, y, 0.175, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0]
    ora.set_state(st)
    env.set_state(ora.get_state())
    a[:] = 0
    a[1] = [-1, 0, 0]
    for t in range(2):
        o, r, d, info = env.step(a)
        oo, ro, do, oko = ora.step(a)
        sch = env.handle.schedule()
        assert list(sch['prone']) == [1] and sorted(sch['free']) == [0, 2] and list(sch['redo']) == [2], sch
        _compare(o, oo, 2e-3)
        assert np.array_equal(r, ro)
    env.close()
