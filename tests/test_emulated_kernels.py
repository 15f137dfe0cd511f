"""CPU tier: the product's HIP device code, compiled for the fiber emulator (tests/emu), against
the oracle.  Checks the kernel LOGIC lane by lane; speed and the real gfx950 build are -m gpu."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
import pybullet_multigoal_gym_amd as pmg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fp(a):
    return a.ctypes.data_as(C.c_void_p)


def test_device_dynamics_functions_match_oracle(emu_library):
    lib = C.CDLL(emu_library.path)
    rs = np.random.RandomState(0)
    for _ in range(3):
        q = np.float32(np.r_[rs.uniform(-1, 1, 7) + [0, -0.5, 0, 1.7, 0, -0.8, 0], rs.uniform(0, 0.035, 2)])
        qd = np.float32(np.r_[rs.uniform(-2, 2, 7), rs.uniform(-0.1, 0.1, 2)])
        tau = np.float32(rs.uniform(-1, 1, 9))
        qdd, mi, tip = np.zeros(9, np.float32), np.zeros(81, np.float32), np.zeros(12, np.float32)
        lib.pmge_probe_dynamics(_fp(q), _fp(qd), _fp(tau), _fp(qdd), _fp(mi), _fp(tip))
        ref = O.fdyn(q.astype(float), qd.astype(float), tau.astype(float))
        assert np.abs(qdd - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())
        mref = O.minv(q.astype(float))
        assert np.abs(mi.reshape(9, 9) - mref).max() < 1e-5 * np.abs(mref).max()    # float32 Gauss-Jordan, cond(M) ~ 1e3
        p, R = O.fk_tip(q.astype(float))
        assert np.abs(tip[:3] - p).max() < 1e-6 and np.abs(tip[3:].reshape(3, 3) - R).max() < 1e-6


def test_device_ik_matches_oracle(emu_library):
    lib = C.CDLL(emu_library.path)
    q0 = np.float32([0, -0.5592432, 0, 1.733180, 0, -0.8501557, 0, 0.035, 0.035])
    for tgt in ([-0.52, 0.0, 0.25], [-0.45, 0.1, 0.30]):
        out = np.zeros(9, np.float32)
        lib.pmge_probe_ik(_fp(q0), _fp(np.float32(tgt)), _fp(out))
        ref, it = O.ik(q0.astype(float), tgt)
        assert np.abs(out - ref).max() < 5e-5


@pytest.mark.parametrize('task,kw,steps,tol', [
    ('reach', {}, 2, 2e-5),
    ('reach', {'joint_control': True}, 2, 2e-5),
    ('push', {}, 1, 2e-3),
    ('block_stack', {'num_block': 2}, 1, 2e-3),
])
def test_emulated_step_kernel_matches_oracle(emu_library, task, kw, steps, tol):
    N = 1
    env = pmg.make_env(task=task, num_envs=N, seed=3, seed_stride=1, _library=emu_library, **kw)
    ora = O.OracleEnv(task, N, seed_base=3, seed_stride=1, **kw)
    ora.reset()
    o, oo = env.reset(), ora.reset()
    assert np.array_equal(o['desired_goal'], oo['desired_goal'])          # device MT19937 == numpy stream
    assert np.abs(env.get_state() - ora.get_state()).max() < 1e-6
    rs = np.random.RandomState(5)
    for _ in range(steps):
        a = rs.uniform(-1, 1, (N, env.dims.action_dim)).astype(np.float32)
        o, r, d, info = env.step(a)
        oo, ro, do, oko = ora.step(a)
    assert np.abs(env.get_state()[:, :9] - ora.get_state()[:, :9]).max() < tol   # joint angles
    assert np.abs(o['achieved_goal'] - oo['achieved_goal']).max() < tol
    assert np.array_equal(r, ro) and np.array_equal(d, do)
    env.close()


def _pair(emu_library, task, **kw):
    env = pmg.make_env(task=task, num_envs=1, seed=3, seed_stride=1, _library=emu_library, **kw)
    ora = O.OracleEnv(task, 1, seed_base=3, seed_stride=1, **kw)
    ora.reset()
    env.reset(), ora.reset()
    assert np.abs(env.get_state() - ora.get_state()).max() < 1e-6
    return env, ora


def _drive(env, ora, actions):
    for a in actions:
        a = np.float32(a).reshape(1, -1)
        o, r, d, _ = env.step(a)
        oo, ro, do, _ = ora.step(a)
    se, so = env.get_state()[0], ora.get_state()[0]
    assert np.abs(se[:9] - so[:9]).max() < 2e-5                 # joint angles
    assert np.abs(se[64:71] - so[64:71]).max() < 5e-5           # object pose
    assert np.abs(se[71:77] - so[71:77]).max() < 2e-3           # object twist (solver early-exit floor)
    assert np.array_equal(r, ro) and np.array_equal(d, do)
    return so


def test_emulated_slide_puck_matches_oracle(emu_library):
    """cylinder x box narrowphase on the device: the puck resting on the long table (mu 0.05), then pushed by
    the closed fingers (kuka_single_step_base_env.py:53-56,66-69; cylinder_bulk.urdf)."""
    env, ora = _pair(emu_library, 'slide')
    assert np.array_equal(env.reset()['desired_goal'], ora.reset()['desired_goal'])
    st = ora.get_state().copy()
    st[0, 64:67] = [-0.52, 0.045, 0.170]
    env.set_state(st), ora.set_state(st)
    so = _drive(env, ora, [[0, 1, 0], [0, 1, 0]])
    assert so[65] > 0.06 and abs(so[66] - 0.170) < 1e-4        # it moved along +y and stayed on the table
    env.close()


@pytest.mark.parametrize('task,kw,rest_z', [('push', {}, 0.016), ('slide', {}, 0.011), ('block_stack', {'num_block': 2}, 0.016)])
def test_emulated_object_off_the_table_lands_on_the_floor(emu_library, task, kw, rest_z):
    """The robot URDF's base plane (iiwa14_parallel_jaw.urdf:37-58: 5 x 5 x 0.002 m box at the origin, friction 1): an object
    that has left the table lands on it at z = 0.001 + its half height and stays there -- on the emulated kernels as in the
    oracle -- instead of falling for the rest of the episode (rounds 1-5)."""
    env, ora = _pair(emu_library, task, **kw)
    st = ora.get_state().copy()
    edge = (-0.70 if task == 'slide' else -0.52) + (0.5 if task == 'slide' else 0.25)       # the table's +x side wall
    st[0, 64:67] = [edge + 0.04, 0.02, 0.17]
    st[0, 67:71] = [0, 0, 0, 1]; st[0, 71:77] = 0
    env.set_state(st), ora.set_state(st)
    A = env.dims.action_dim
    for t in range(4):
        a = np.zeros((1, A), np.float32)
        o, r, d, _ = env.step(a)
        oo, ro, do, _ = ora.step(a)
    se, so = env.get_state()[0], ora.get_state()[0]
    assert abs(so[66] - rest_z) < 2e-4 and abs(se[66] - rest_z) < 2e-4, (se[64:67], so[64:67])   # at rest on the floor, both
    assert np.abs(se[64:71] - so[64:71]).max() < 5e-5 and np.abs(se[71:77] - so[71:77]).max() < 2e-3
    assert np.abs(se[:9] - so[:9]).max() < 2e-5
    env.close()


@pytest.mark.parametrize('task', ['slide'])
def test_speculative_double_repeat_is_bit_identical_to_the_serial_repeat(emu_library, emu_library_serial_repeat, task):
    """List 0 of slide: three wavefronts per workgroup -- env, float narrowphase, and the finger x puck pairs, which are repeated in
    double whenever they are in contact, computed beside it; the fingers pushing the puck.  States and outputs EQUAL to the build
    that repeats those pairs serially behind the float pass (-DPMG_CYL_SPEC=0, round 5's layout), and the double results are
    really taken (the emulator counts them)."""
    import ctypes as C
    N = 3
    kw = {} if task == 'slide' else {'num_block': 2}
    envs = [pmg.make_env(task=task, num_envs=N, seed=3, seed_stride=1, _library=lib, **kw) for lib in (emu_library, emu_library_serial_repeat)]
    for e in envs:
        e.reset()
    st = envs[0].get_state().copy()
    if task == 'slide':
        st[:, 64] = -0.52 + np.float32([0.0, 0.003, -0.002]); st[:, 65] = 0.045 + np.float32([0.0, -0.002, 0.003]); st[:, 66] = 0.170
        st[:, 67:71] = [0, 0, 0, 1]; st[:, 71:77] = 0
        acts = [[0, 1, 0], [0, 1, 0], [0.3, 1, 0]]
    else:
        acts = [[-1, 0.2 * k, 0.5] for k in (-1, 0, 1, 0, 1)]
    for e in envs:
        e.set_state(st)
    counter = C.CDLL(emu_library.path).pmge_cyl_spec_taken
    counter.restype = C.c_longlong
    before = counter()
    for t, a in enumerate(acts):
        a = np.tile(np.float32(a), (N, 1))
        if task != 'slide':
            a[:, 1] *= np.float32([1.0, 0.5, -1.0])
        outs = [e.step(a) for e in envs]
        assert all(np.array_equal(outs[0][0][k], outs[1][0][k]) for k in outs[0][0])
        assert np.array_equal(envs[0].get_state(), envs[1].get_state())
        if task == 'slide' or t >= 1:
            assert envs[0].handle.schedule()['prone'].size == N         # every env ran on list 0: the three-wavefront kernel
    assert counter() - before > 50                                        # fingers on the puck / gripper base on the door: double results taken
    if task == 'slide':
        assert (envs[0].get_state()[:, 65] > 0.06).all()
    for e in envs:
        e.close()


def test_emulated_gripper_base_contact_matches_oracle(emu_library):
    """gripper-base cylinder (link 7) x block pairs: a block wedged between the open fingers against the palm."""
    env, ora = _pair(emu_library, 'pick_and_place')
    st = ora.get_state().copy()
    st[0, 64:67] = [-0.52, 0.0, 0.25 + 0.0295]
    env.set_state(st), ora.set_state(st)
    so = _drive(env, ora, [[0, 0, 0, 1], [0, 0, -1, 1]])
    assert so[66] > 0.25                                         # held up by the contacts, not in free fall
    env.close()


def test_emulated_stacked_blocks_behind_the_table_run_match_oracle(emu_library):
    """The block x table run is solved in row space (table_run_*), the rows behind it (block x block, finger x block) as LDS
    rows that move the same blocks: a block dropped onto another one, then the closed gripper pressed onto the stack --
    the run's unclamped impulses must be re-derived from the velocity changes at every phase change."""
    env = _make_quiet('block_stack', emu_library, num_block=2, seed=3)
    ora = O.OracleEnv('block_stack', 1, num_block=2, seed_base=3, seed_stride=1)
    ora.reset()
    env.reset(), ora.reset()
    st = ora.get_state().copy()
    st[0, 64:77] = [-0.52, 0.0, 0.175, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0]                     # block 0 on the table under the tip
    st[0, 77:90] = [-0.515, 0.004, 0.175 + 0.0305, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0]         # block 1 half a millimetre above it, offset
    st[0, 18:21] = [-0.52, 0.0, 0.26]
    env.set_state(st), ora.set_state(st)
    for a in ([0, 0, -1, 1], [0, 0, -1, 1], [0, 0.3, -1, 1]):
        a = np.float32(a).reshape(1, -1)
        o, r, d, _ = env.step(a)
        oo, ro, do, _ = ora.step(a)
    se, so = env.get_state()[0], ora.get_state()[0]
    assert np.abs(se[:9] - so[:9]).max() < 5e-5
    assert np.abs(se[64:71] - so[64:71]).max() < 2e-4 and np.abs(se[77:84] - so[77:84]).max() < 2e-4
    assert so[79] > 0.19                                          # block 1 rests on block 0 (not fallen through, not thrown off)
    env.close()


def test_emulated_object_with_more_contacts_than_the_row_space_solve_holds(emu_library_small_rowspace):
    """One free object, one env per wavefront: up to 16 contacts are solved in row space; beyond that the object x table run
    goes to row space and the rest stays LDS rows built one per lane (object_run_*, build_rows_by_lane).  A build with the
    limit lowered to 4 contacts takes that path as soon as the fingers touch the table next to the object they squeeze."""
    env, ora = _pair(emu_library_small_rowspace, 'pick_and_place')
    st = ora.get_state().copy()
    st[0, 64:67] = [-0.52, 0.0, 0.175]
    st[0, 18:21] = [-0.52, 0.0, 0.176]
    env.set_state(st), ora.set_state(st)
    so = _drive(env, ora, [[0, 0, -1, 1], [0, 0, -1, -1]])             # (a third, sliding step decorrelates in float32 on the LDS-row path, old or new)
    assert abs(so[66] - 0.175) < 5e-3                                  # the object stays on the table between the fingers
    env.close()


def _golden(name):
    import json
    return json.load(open(os.path.join(ROOT, 'tests', 'golden', name)))


@pytest.mark.parametrize('name', ['sampling_block_rearrange3_curriculum', 'sampling_block_stack4_curriculum', 'sampling_push',
                                  'sampling_chest_push2_curriculum'])
def test_emulated_reset_kernel_replays_reference_sampling(emu_library, name):
    """The device reset kernel (MT19937, numpy choice(p=) / choice(replace=False), shuffles, rejection loops, curriculum
    schedule) against what the REFERENCE's own _task_reset / _generate_goal / _generate_curriculum drew for several
    seeds (tests/golden/ref_sampling_*.json, tools/gen_reference_fixtures.py) -- no oracle in between."""
    import ref_replay as R
    fx = R.load(os.path.join(ROOT, 'tests', 'golden', 'ref_%s.json' % name))
    env = R.ProductAdapter(fx, library=emu_library)
    R.replay(fx, env, tol_static=2e-5)
    env.close()


def _make_quiet(task, lib, **kw):
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        return pmg.make_env(task=task, num_envs=1, seed_stride=1, _library=lib, **kw)


def test_emulated_sub_goals_and_dynamic_goal_match_oracle(emu_library):
    """task_decomposition: set_sub_goal(k) keeps the first k+1 blocks of the order at their targets and every other
    block 'at its goal' wherever it currently is (kuka_multi_step_base_env.py:154-177,309-312), re-derived after
    every step; reward follows the active sub-goal."""
    nb = 3
    env = _make_quiet('block_stack', emu_library, num_block=nb, seed=3, task_decomposition=True)
    ora = O.OracleEnv('block_stack', 1, num_block=nb, seed_base=3, seed_stride=1, task_decomposition=True)
    ora.reset()
    o, oo = env.reset(), ora.reset()
    assert np.array_equal(o['desired_goal'], oo['desired_goal'])          # sub_goal_ind = -1: the full stack
    st = ora.get_state()[0]
    order = st[40:40 + nb].astype(int)
    for k in range(nb):
        g = env.set_sub_goal(k)
        ora.set_sub_goal(k)
        ref = ora.reset(mask=np.zeros(1, bool))['desired_goal']          # re-observe without resetting
        assert np.array_equal(g, ref)
        assert np.array_equal(env.sub_goals[k], ref)
        for i in range(nb):
            b = order[i]
            want = st[48 + 3 * b:51 + 3 * b] if i <= k else st[64 + 13 * b:67 + 13 * b]
            assert np.array_equal(g[0, 3 * b:3 * b + 3], want)
    # with sub-goal 0 active and block order[0] teleported onto its target, the step reports success
    env.set_sub_goal(0), ora.set_sub_goal(0)
    b0 = order[0]
    st2 = ora.get_state().copy()
    st2[0, 64 + 13 * b0:67 + 13 * b0] = st2[0, 48 + 3 * b0:51 + 3 * b0]
    env.set_state(st2), ora.set_state(st2)
    a = np.zeros((1, 4), np.float32)
    o, r, d, info = env.step(a)
    oo, ro, do, oko = ora.step(a)
    assert np.abs(o['desired_goal'] - oo['desired_goal']).max() < 1e-5     # the other blocks' goals track their poses
    assert info['goal_achieved'][0] and oko[0] and r[0] == 0.0 and ro[0] == 0.0
    assert env.set_sub_goal(-1) is not None and not np.array_equal(env.set_sub_goal(-1), o['desired_goal'])
    with pytest.raises(Exception):
        env.set_sub_goal(nb)
    env.close()


def test_emulated_rearrange_step_matches_oracle(emu_library):
    env = _make_quiet('block_rearrange', emu_library, num_block=2, seed=3)
    ora = O.OracleEnv('block_rearrange', 1, num_block=2, seed_base=3, seed_stride=1)
    ora.reset()
    o, oo = env.reset(), ora.reset()
    assert np.array_equal(o['desired_goal'], oo['desired_goal'])
    assert np.abs(env.get_state() - ora.get_state()).max() < 1e-6
    a = np.float32([[0.5, -1.0, 0.3]])
    o, r, d, _ = env.step(a)
    oo, ro, do, _ = ora.step(a)
    assert o['observation'].shape == (1, 8 + 16 * 2) and env.dims.action_dim == 3
    assert np.abs(o['observation'] - oo['observation']).max() < 2e-3
    assert np.abs(o['observation'][0, 3]) == 0 and np.abs(o['observation'][0, 7]) == 0     # no gripper terms (kuka.py:245-246)
    assert np.array_equal(r, ro) and np.array_equal(d, do)
    env.close()


@pytest.mark.parametrize('kw', [dict(task_decomposition=True), dict(use_curriculum=True), dict()])
def test_emulated_grip_informed_goals_match_oracle(emu_library, kw):
    """grip_informed_goal (kuka_multi_step_envs.py:75-77,91-111,143-145): goals carry the gripper-tip target and the
    0.03 finger width; sub-goals come in (pick, place) pairs; the achieved goal appends tip xyz + finger closeness."""
    nb = 3
    env = _make_quiet('block_stack', emu_library, num_block=nb, seed=3, grip_informed_goal=True, **kw)
    ora = O.OracleEnv('block_stack', 1, num_block=nb, seed_base=3, seed_stride=1, grip_informed_goal=True, **kw)
    ora.reset()
    o, oo = env.reset(), ora.reset()
    assert env.dims.goal_dim == 3 * nb + 4 and o['desired_goal'].shape == (1, 13)
    assert np.array_equal(o['desired_goal'], oo['desired_goal'])
    assert np.abs(o['achieved_goal'] - oo['achieved_goal']).max() < 1e-6
    st = ora.get_state()[0]
    order = st[40:40 + nb].astype(int)
    tgt = lambda b: st[48 + 3 * b:51 + 3 * b]
    pos = lambda b: st[64 + 13 * b:67 + 13 * b]
    top = order[nb - 1]
    if kw.get('use_curriculum'):
        lvl = env.last_curriculum_level[0]
        assert np.array_equal(o['desired_goal'][0, 9:12], tgt(order[lvl]))                 # :143
    else:
        assert np.array_equal(o['desired_goal'][0, 9:12], tgt(top))                        # :76 (and sub_goals[-1])
    assert o['desired_goal'][0, 12] == np.float32(0.03)
    if kw.get('task_decomposition'):
        assert env.num_steps == 2 * nb
        subs = env.sub_goals
        for s in range(2 * nb):
            g = env.set_sub_goal(s)
            ora.set_sub_goal(s)
            assert np.array_equal(g, ora.reset(mask=np.zeros(1, bool))['desired_goal']) and np.array_equal(g, subs[s])
            j, pick = s // 2, s % 2 == 0
            for i in range(nb):
                b = order[i]
                assert np.array_equal(g[0, 3 * b:3 * b + 3], tgt(b) if (i < j if pick else i <= j) else pos(b))
            assert np.array_equal(g[0, 9:12], pos(order[j]) if pick else tgt(order[j]))
        with pytest.raises(Exception):
            env.set_sub_goal(2 * nb)
        env.set_sub_goal(0), ora.set_sub_goal(0)
    a = np.float32([[0.2, -0.4, -1.0, 1.0]])
    o, r, d, info = env.step(a)
    oo, ro, do, oko = ora.step(a)
    assert np.abs(o['desired_goal'] - oo['desired_goal']).max() < 1e-4
    assert np.abs(o['achieved_goal'] - oo['achieved_goal']).max() < 1e-4
    assert np.array_equal(r, ro) and np.array_equal(info['goal_achieved'], oko)
    r2, ok2 = env._compute_reward(o['achieved_goal'], o['desired_goal'])                   # HER path with goal_dim 13
    assert np.array_equal(r2, r) and np.array_equal(ok2, info['goal_achieved'])
    env.close()


def test_emulated_row_packed_reach_and_redo(emu_library):
    """Four envs per wavefront for the contact-free reach envs (pmg_packed.h): 6 envs = one full wave + one with
    two live rows; then a misprediction -- an env whose plan entry says 'away from the table' while its fingers
    are 7 mm inside it -- must be caught by the per-substep predicate and recomputed by pmg_k_redo."""
    N = 6
    env = pmg.make_env(task='reach', num_envs=N, seed=3, seed_stride=1, _library=emu_library)
    ora = O.OracleEnv('reach', N, seed_base=3, seed_stride=1)
    ora.reset()
    env.reset(), ora.reset()
    rs = np.random.RandomState(5)
    a = rs.uniform(-1, 1, (N, 3)).astype(np.float32)
    o, r, d, _ = env.step(a)
    oo, ro, do, _ = ora.step(a)
    sch = env.handle.schedule()
    assert len(sch['prone']) == 0 and sorted(sch['free']) == list(range(N)) and len(sch['redo']) == 0
    assert np.abs(o['observation'] - oo['observation']).max() < 2e-5 and np.array_equal(r, ro)
    assert np.abs(env.get_state() - ora.get_state()).max() < 2e-5
    st = ora.get_state().copy()
    q_low, _ = O.ik(st[1, :9].astype(float), [-0.52, 0.0, 0.168])
    st[1, :7] = q_low[:7]; st[1, 9:18] = 0; st[1, 18:21] = [-0.52, 0, 0.30]; st[1, 21:28] = q_low[:7]
    env.set_state(st), ora.set_state(st)
    o, r, d, _ = env.step(np.zeros((N, 3), np.float32))
    oo, ro, do, _ = ora.step(np.zeros((N, 3), np.float32))
    sch = env.handle.schedule()
    assert list(sch['redo']) == [1] and len(sch['prone']) == 0
    assert np.abs(o['observation'] - oo['observation']).max() < 2e-5
    assert np.abs(env.get_state() - ora.get_state()).max() < 5e-5
    assert oo['observation'][1, 2] > 0.17                      # the table pushed the fingers back out
    env.close()


def test_emulated_row_packed_object_tasks_and_overflow_redo(emu_library):
    """push with four envs per wavefront (each row its own 12-contact store): parity with the oracle, then an env the
    plan routes to the packed list (tip TARGET 10 cm from the block) although its closed fingers are touching the
    block on the table -- more than 12 contacts -- must be given up and recomputed by pmg_k_redo_obj."""
    N = 5
    env = pmg.make_env(task='push', num_envs=N, seed=3, seed_stride=1, _library=emu_library)
    ora = O.OracleEnv('push', N, seed_base=3, seed_stride=1)
    ora.reset()
    env.reset(), ora.reset()
    a = np.random.RandomState(5).uniform(-1, 1, (N, 3)).astype(np.float32)
    env.step(a), ora.step(a)
    sch = env.handle.schedule()
    assert len(sch['free']) + len(sch['prone']) == N and len(sch['free']) >= 4 and len(sch['redo']) == 0
    se, so = env.get_state(), ora.get_state()
    assert np.abs(se[:, :9] - so[:, :9]).max() < 2e-5 and np.abs(se[:, 64:71] - so[:, 64:71]).max() < 5e-5
    st = so.copy()
    tipxy = ora.reset(mask=np.zeros(N, bool))['observation'][2, :2]
    st[2, 64:67] = [tipxy[0], tipxy[1] + 0.027, 0.175]          # block against the side of the closed fingers
    st[2, 67:71] = [0, 0, 0, 1]; st[2, 71:77] = 0
    st[2, 18:21] = [tipxy[0], tipxy[1] - 0.10, 0.176]           # ... while the tip target is 10+ cm away
    env.set_state(st), ora.set_state(st)
    z = np.zeros((N, 3), np.float32)
    o, r, d, _ = env.step(z)
    oo, ro, do, _ = ora.step(z)
    sch = env.handle.schedule()
    assert 2 in sch['free'] and list(sch['redo']) == [2]
    se, so = env.get_state(), ora.get_state()
    assert np.abs(se[:, :9] - so[:, :9]).max() < 5e-5 and np.abs(se[:, 64:67] - so[:, 64:67]).max() < 2e-4
    env.close()


@pytest.mark.parametrize('kind', ['box', 'cyl'])
def test_device_narrowphase_matches_oracle_on_random_pairs(emu_library, kind):
    """The HIP narrowphase functions (box_box_fast -> box_box, cyl_box), called directly in the emulator build, against
    the oracle's on randomly oriented pairs in the regime the simulation lives in (brought together until first touch,
    then a little deeper): same number of points, normals, depths and witness points."""
    lib = C.CDLL(emu_library.path)
    lib.pmge_probe_narrowphase.restype = C.c_int
    rs = np.random.RandomState(4)

    def rot():
        q = rs.normal(size=4); q /= np.linalg.norm(q); x, y, z, w = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    hb = np.float32([0.015, 0.015, 0.015])
    ha = np.float32([0.0125, 0.005, 0.04]) if kind == 'box' else np.float32([0.03, 0.03, 0.01])
    checked = tilted = loose = general_loose = 0
    for trial in range(100):
        Ra, Rb = (np.eye(3), np.eye(3)) if trial % 4 == 0 else (rot(), rot())
        if trial % 4 == 1:
            # bodies tilted by <= 3 degrees out of the table plane, any yaw: rim x edge crossings, the extrapolated closest
            # pair and the edge refinement of cyl_box (round 4) are taken here
            from test_oracle_physics import _rot_axis
            Ra = _rot_axis(rs.normal(size=3), rs.uniform(0, 0.05))
            Rb = _rot_axis(rs.normal(size=3), rs.uniform(0, 0.05)) @ _rot_axis(np.array([0.0, 0.0, 1.0]), rs.uniform(0, 2 * np.pi))
        cb = rs.uniform(-0.1, 0.1, 3)
        u = rs.normal(size=3); u /= np.linalg.norm(u)
        for step in range(0, 200):
            ca = cb + u * (0.09 - 0.0005 * step)
            ref = (O.box_box(ca, Ra.ravel(), ha, cb, Rb.ravel(), hb) if kind == 'box'
                   else O.cyl_box(ca, Ra.ravel(), 0.03, 0.01, cb, Rb.ravel(), hb))
            if len(ref):
                break
        for extra in (0.0, 0.0007):                       # at first touch and slightly deeper
            ca2 = np.float32(ca - u * extra)
            a32 = [np.float32(x) for x in (ca2, Ra.ravel(), ha, cb, Rb.ravel(), hb)]
            ref = (O.box_box(*[x.astype(float) for x in a32]) if kind == 'box'
                   else O.cyl_box(a32[0].astype(float), a32[1].astype(float), 0.03, 0.01, a32[3].astype(float), a32[4].astype(float), hb))
            out = np.zeros(40, np.float32)
            n = lib.pmge_probe_narrowphase(0 if kind == 'box' else 1, *[_fp(x) for x in a32], C.c_float(0.002), _fp(out))
            got = out.reshape(4, 10)[:n]
            if n != len(ref):
                # a point sitting exactly on the margin / a tie between axes may flip between float32 and float64
                assert abs(n - len(ref)) <= 1 and (len(ref) == 0 or np.abs(ref[:, 9]).max() < 0.0021)
                continue
            if n == 0:
                continue
            checked += 1
            order_g, order_r = np.lexsort(got[:, :3].round(4).T), np.lexsort(ref[:, :3].round(4).T)
            en = np.abs(got[order_g][:, 6:9] - ref[order_r][:, 6:9]).max()
            ed = np.abs(got[order_g][:, 9] - ref[order_r][:, 9]).max()
            ep = np.abs(got[order_g][:, 0:6] - ref[order_r][:, 0:6]).max()
            if trial % 4 == 1:
                # two almost parallel features: the closest pair's position along them, and with it the last third of a degree
                # of the normal, is ill-conditioned -- float32 and float64 settle on different points of a flat minimum.
                # The depth is not: strict bar on it, loose bars on the rest, and a count of the strict misses
                tilted += 1
                assert ed < 1e-4 and en < 2e-2 and ep < 1e-2, (en, ed, ep)
                loose += int(en >= 2e-4 or ed >= 2e-5 or ep >= 5e-5)
                continue
            if not (en < 2e-4 and ed < 2e-5 and ep < 5e-5):
                # (cyl, any orientation) the closest-feature direction of two features that are nearly parallel by chance
                general_loose += 1
                assert kind == 'cyl' and ed < 1e-4 and en < 2e-2 and ep < 1e-2, (en, ed, ep)
    print('tilted pairs %d, beyond the strict bars %d; other pairs %d, beyond the strict bars %d' % (tilted, loose, checked - tilted, general_loose))
    assert loose <= 0.25 * max(tilted, 1) and general_loose <= 0.03 * checked
    assert checked > 130


def test_device_cyl_box_corner_in_the_side_matches_oracle(emu_library):
    """The product's cyl_box on a box corner touching / inside the cylinder's side (a finger's edge against the puck): the
    planar corner-to-circle distance, as the oracle -- in float32 the coinciding closest points of penetrating shapes are
    1e-8 apart, not 0, so the switch to the radial axis sits at 1 um in both."""
    lib = C.CDLL(emu_library.path)
    lib.pmge_probe_narrowphase.restype = C.c_int
    I = np.eye(3, dtype=np.float32).ravel()
    cc, ha, hb = np.float32([-0.495, 0.0979, 0.17]), np.float32([0.03, 0.03, 0.01]), np.float32([0.0125, 0.005, 0.04])
    checked = 0
    for yoff in (0.0157, 0.0257):
        for gap in np.arange(0.042, 0.028, -0.001):
            cb = np.float32([cc[0] + gap, cc[1] + yoff, 0.207])
            out = np.zeros(40, np.float32)
            n = lib.pmge_probe_narrowphase(1, _fp(cc), _fp(I), _fp(ha), _fp(cb), _fp(I), _fp(hb), C.c_float(0.002), _fp(out))
            ref = O.cyl_box(cc.astype(float), I.astype(float), 0.03, 0.01, cb.astype(float), I.astype(float), hb.astype(float))
            assert n == len(ref), (yoff, gap, n, len(ref))
            if n:
                got = out.reshape(4, 10)[:n]
                assert abs(got[:, 9].min() - ref[:, 9].min()) < 2e-6 and np.abs(got[0, 6:9] - ref[0, 6:9]).max() < 1e-4, (yoff, gap, got[:, 9], ref[:, 9])
                checked += 1
    assert checked > 10


def test_axis_aligned_partner_front_end_is_bit_identical(emu_library):
    """The reach kernel's narrowphase instantiation knows that box B (the table) is axis-aligned and leaves the sums with
    exact zeros out of the separating-axis front end: same contacts, same bits as the general routine on fingers in
    random poses over, in and beside a table-sized box."""
    lib = C.CDLL(emu_library.path)
    lib.pmge_probe_narrowphase.restype = C.c_int
    rs = np.random.RandomState(3)
    ha, hb = np.float32([0.0125, 0.005, 0.04]), np.float32([0.5, 0.4, 0.1])
    Rb = np.eye(3, dtype=np.float32)
    hit = 0
    for trial in range(600):
        q = rs.normal(size=4) * ([1, 1, 1, 1] if trial % 3 == 0 else [0.02, 0.02, 1, 1]); q /= np.linalg.norm(q); x, y, z, w = q
        Ra = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                       [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                       [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        ext = np.abs(Ra[2]) @ ha                                     # the finger's half extent along z
        cb = np.float32([0, 0, 0])
        ca = np.float32([rs.uniform(-0.52, 0.52), rs.uniform(-0.42, 0.42), 0.1 + ext + rs.uniform(-0.002, 0.003)])
        a32 = [np.float32(v) for v in (ca, Ra.ravel(), ha, cb, Rb.ravel(), hb)]
        o1, o2 = np.zeros(40, np.float32), np.zeros(40, np.float32)
        n1 = lib.pmge_probe_narrowphase(3, *[_fp(v) for v in a32], C.c_float(0.002), _fp(o1))
        n2 = lib.pmge_probe_narrowphase(2, *[_fp(v) for v in a32], C.c_float(0.002), _fp(o2))
        assert n1 == n2, (trial, n1, n2)
        assert np.array_equal(o1[:10 * n1].view(np.uint32), o2[:10 * n2].view(np.uint32)), trial
        hit += n1 > 0
    assert hit > 250, hit


def test_face_clip_is_bit_identical_to_the_general_box_box_routine(emu_library):
    """box_box_fast serves a face contact whose incident face is NOT inside the reference face (a finger on a cube, cubes
    stacked off-centre) with box_face_clip: clip passes over all vertices at once, batched LDS traffic.  It must return
    exactly what the general routine (box_box: vertex-by-vertex Sutherland-Hodgman through the workspace) returns -- same
    points, same order, same bits -- on fingers against cubes, cubes on cubes and cubes on a table-sized box, in random
    orientations, touching, deeper and with more than four clipped vertices inside the margin."""
    lib = C.CDLL(emu_library.path)
    lib.pmge_probe_narrowphase.restype = C.c_int
    rs = np.random.RandomState(11)

    def rot(small):
        q = rs.normal(size=4) * ([small, small, 1.0, 1.0] if small else 1.0); q /= np.linalg.norm(q); x, y, z, w = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    shapes = [(np.float32([0.0125, 0.005, 0.04]), np.float32([0.015] * 3)),      # finger x cube
              (np.float32([0.015] * 3), np.float32([0.015] * 3)),                # cube x cube
              (np.float32([0.015] * 3), np.float32([0.5, 0.5, 0.1]))]            # cube x table-sized box
    lib.pmge_face_clip_count.restype = C.c_longlong
    calls0 = lib.pmge_face_clip_count()
    clipped = many = 0
    for trial in range(900):
        ha, hb = shapes[trial % 3]
        small = 0.0 if trial % 2 else rs.choice([0.003, 0.03, 0.3])               # nearly face-parallel poses clip most
        Ra, Rb = rot(small), (np.eye(3) if trial % 5 else rot(small))
        n_ax = Rb[:, rs.randint(3)] * rs.choice([-1, 1])
        reach = np.abs(Ra.T @ n_ax) @ ha + np.abs(Rb.T @ n_ax) @ hb
        lateral = rs.normal(size=3); lateral -= n_ax * (lateral @ n_ax)
        lateral *= rs.uniform(0, 1) * float(min(hb.min(), 0.03)) / max(np.linalg.norm(lateral), 1e-9)
        cb = rs.uniform(-0.1, 0.1, 3)
        ca = cb + n_ax * (reach - rs.uniform(-0.001, 0.003)) + lateral
        a32 = [np.float32(x) for x in (ca, Ra.ravel(), ha, cb, Rb.ravel(), hb)]
        o_new, o_old = np.zeros(40, np.float32), np.zeros(40, np.float32)
        n_new = lib.pmge_probe_narrowphase(0, *[_fp(x) for x in a32], C.c_float(0.002), _fp(o_new))
        n_old = lib.pmge_probe_narrowphase(2, *[_fp(x) for x in a32], C.c_float(0.002), _fp(o_old))
        assert n_new == n_old, (trial, n_new, n_old)
        assert np.array_equal(o_new[:10 * n_new].view(np.uint32), o_old[:10 * n_old].view(np.uint32)), (trial, o_new[:10 * n_new], o_old[:10 * n_old])
        clipped += n_new > 0
        many += n_new == 4
    assert clipped > 300 and many > 100, (clipped, many)
    assert lib.pmge_face_clip_count() - calls0 > 250          # the new path is what answered


@pytest.mark.parametrize('nb,frac_down', [(0, 0.3), (1, 0.05), (1, 0.4)])
def test_two_pass_plan_writes_the_same_lists_as_the_single_workgroup_plan(built, nb, frac_down):
    """Batches beyond 65 536 envs are planned by ceil(N / 1024) workgroups in two passes (pmg_k_plan_count /
    pmg_k_plan_scatter).  On a batch both plans can take (3 000 envs = three workgroups, the last one ragged) they must
    write identical launch lists and counts -- for reach, and for one free object with few / many envs down at the
    table (the promotion rule decided from the grand totals, on either side of its threshold)."""
    import ctypes as C
    lib = C.CDLL(os.path.join(ROOT, 'tests', 'emu', 'libpmg_emu.so'))
    N, adim = 3000, 3
    rs = np.random.RandomState(nb * 7 + int(frac_down * 100))
    hot = np.zeros((N, 32), np.float32)
    hot[:, 18] = rs.uniform(-0.67, -0.37, N); hot[:, 19] = rs.uniform(-0.2, 0.2, N)
    hot[:, 20] = np.where(rs.uniform(0, 1, N) < frac_down, 0.175 + rs.uniform(0, 0.01, N), rs.uniform(0.19, 0.5, N))
    blocks = np.zeros((N, 13 * max(nb, 1)), np.float32)
    blocks[:, 0] = rs.uniform(-0.64, -0.40, N); blocks[:, 1] = rs.uniform(-0.15, 0.15, N); blocks[:, 2] = 0.175; blocks[:, 6] = 1
    near = rs.uniform(0, 1, N) < 0.1                       # some grippers right at their object: class 0
    hot[near, 18:21] = blocks[near, 0:3] + np.float32([0.0, 0.0, 0.02])
    actions = rs.uniform(-1, 1, (N, adim)).astype(np.float32)
    out, flags = [], []
    for two_pass in (0, 1):
        sc = np.full(3 + 3 * N, -7, np.int32)
        rc = lib.pmge_probe_plan(C.c_int(N), C.c_int(nb), hot.ctypes.data_as(C.c_void_p), blocks.ctypes.data_as(C.c_void_p),
                                 actions.ctypes.data_as(C.c_void_p), C.c_int(adim), C.c_int(1536), C.c_int(two_pass),
                                 sc.ctypes.data_as(C.c_void_p))
        assert rc in (0, 1)
        out.append(sc)
        flags.append(rc)
    a, b = out
    # the promotion flag (list 0 takes issue priority when the fingers-down class was moved there): same in both plans,
    # set exactly for one object with few envs down at the table
    assert flags[0] == flags[1] == (1 if (nb == 1 and frac_down < 0.125) else 0)
    n0, n1 = int(a[0]), int(a[1])
    assert n0 + n1 == N and (n0, n1) == (int(b[0]), int(b[1])) and 0 < n0 < N
    assert np.array_equal(a[2:2 + n0], b[2:2 + n0]) and np.array_equal(a[2 + N:2 + N + n1], b[2 + N:2 + N + n1])
    assert a[2 + 2 * N] == 0 and b[2 + 2 * N] == 0
    assert sorted(np.concatenate([b[2:2 + n0], b[2 + N:2 + N + n1]]).tolist()) == list(range(N))    # a partition of the batch


@pytest.mark.parametrize('task', ['push', 'pick_and_place', 'slide'])
def test_emulated_sharded_batch_is_bit_identical_to_the_unsharded_batch(emu_library, task):
    """The product's kernels on the emulator: 24 envs in one batch against 3 shards of 8, half of the batch with its fingers
    driven onto the table (the class whose kernel depended on batch-wide counts until round 5) -- outputs and state rows EQUAL
    (tests/test_gpu_parity.py runs the same at 512 / 2 and 4096 / 8 on the device)."""
    from test_gpu_parity import _sharded_equals_unsharded
    _sharded_equals_unsharded(emu_library, task, 24, 3, 4)


def _quat_R64(q):
    x, y, z, w = [float(v) for v in q]
    s = 2.0 / (x * x + y * y + z * z + w * w)
    xs, ys, zs = x * s, y * s, z * s
    wx, wy, wz, xx, xy, xz, yy, yz, zz = w * xs, w * ys, w * zs, x * xs, x * ys, x * zs, y * ys, y * zs, z * zs
    return np.array([[1 - (yy + zz), xy - wz, xz + wy], [xy + wz, 1 - (xx + zz), yz - wx], [xz - wy, yz + wx, 1 - (xx + yy)]])


def test_double_forward_kinematics_of_the_contact_links_matches_the_oracle(emu_library):
    """fk64_link (the double poses cyl_redo64 takes for the gripper base and the fingers, from float32 joint angles) against
    the float64 oracle's kinematics() through its Bullet-call-level world: link frames of Bullet links 12 / 13 / 15."""
    lib = C.CDLL(emu_library.path)
    ora = O.OracleEnv('reach', 1, seed_base=0)
    ora.reset()
    ol = ora.lib
    rs = np.random.RandomState(3)
    for trial in range(20):
        q = np.float32(np.concatenate([rs.uniform(-2, 2, 7), rs.uniform(0, 0.035, 2)]))
        for d in range(9):
            ol.pmgo_bw_reset_joint(ora.h, 0, d, C.c_double(float(q[d])), C.c_double(0.0))
        for body, link in ((7, 12), (5, 13), (6, 15)):                 # BODY_GBASE, BODY_FINGER1, BODY_FINGER2
            p, R, ref = np.zeros(3), np.zeros(9), np.zeros(13)
            lib.pmge_probe_fk64(_fp(q), body, _fp(p), _fp(R))
            ol.pmgo_bw_link_state(ora.h, link, _fp(ref))
            assert np.abs(p - ref[:3]).max() < 1e-12, (trial, body, p, ref[:3])
            assert np.abs(R.reshape(3, 3) - _quat_R64(ref[3:7])).max() < 1e-12
    ora.close()


def test_double_repeat_of_a_cylinder_pair_is_the_float64_oracle(emu_library):
    """cyl_redo64 (cyl_box<double> on poses re-derived in double from the float32 state) against the float64 oracle's cyl_box on
    the same poses: the slide puck -- any small tilt, any yaw -- against the table and near its edge.  Same count, points and
    normals to 1e-6 (the outputs are float32), depths to 1e-8.  Beside it the float32 pass on float32 poses: its gross
    disagreements with the oracle are counted (the resting puck has none: the repeat is for vertex / edge contacts)."""
    lib = C.CDLL(emu_library.path)
    lib.pmge_probe_cyl_redo64.restype = C.c_int
    lib.pmge_probe_cyl_amb.restype = C.c_int
    rs = np.random.RandomState(5)
    I3 = np.eye(3)
    checked = flagged = gross_unflagged = 0
    q9 = np.zeros(9, np.float32)
    door = np.zeros(4, np.float32)
    for trial in range(300):
        tilt = 10.0 ** rs.uniform(-8, -1.5) * rs.normal(size=2)
        yaw = rs.uniform(0, 2 * np.pi)
        quat = np.array([tilt[0] / 2, tilt[1] / 2, np.sin(yaw / 2), np.cos(yaw / 2)])
        quat /= np.linalg.norm(quat)
        edge = trial % 3 == 0
        blk = np.zeros(13, np.float32)
        blk[0:3] = [-0.70 + (0.5 - rs.uniform(0, 0.04) if edge else rs.uniform(-0.3, 0.3)), rs.uniform(-0.3, 0.3), 0.16 + 0.01 + rs.uniform(-2e-4, 1.5e-3)]
        blk[3:7] = quat
        tc, th = np.float32([-0.70, 0.0, 0.08]), np.float32([0.5, 0.45, 0.08])
        out = np.zeros(40, np.float32)
        kc = np.zeros(24, np.float32); kc[0:3] = tc; kc[3:6] = th
        n = lib.pmge_probe_cyl_redo64(-1, 0, -1, -1, _fp(q9), _fp(blk), _fp(door), _fp(kc), C.c_float(0.03), C.c_float(0.01), _fp(out))
        R = _quat_R64(blk[3:7])
        ref = O.cyl_box(blk[0:3].astype(float), R.ravel(), 0.03, 0.01, tc.astype(float), I3.ravel(), th.astype(float))
        assert n == len(ref), (trial, n, len(ref))
        if n == 0:
            continue
        got = out.reshape(4, 10)[:n]
        assert np.abs(got[:, 6:9] - ref[:, 6:9]).max() < 1e-6 and np.abs(got[:, 9] - ref[:, 9]).max() < 1e-8 and np.abs(got[:, 0:6] - ref[:, 0:6]).max() < 1e-6, (trial, got, ref)
        checked += 1
        Rf = np.float32(R)                                     # the float pass: float32 poses
        outf, amb = np.zeros(40, np.float32), C.c_float(0)
        nf = lib.pmge_probe_cyl_amb(_fp(blk), _fp(Rf), C.c_float(0.03), C.c_float(0.01), _fp(tc), _fp(np.float32(I3)), _fp(th), C.c_float(0.002), _fp(outf), C.byref(amb))
        gf = outf.reshape(4, 10)[:nf]
        # gross: another number of points, another normal, a point 3 mm from the oracle's or 20 um deeper (two candidates of the
        # same depth 1 mm apart on the rim may swap in the reduction to four points: not a different contact)
        gross = nf != n or np.abs(np.sort(gf[:, 0:3], axis=0) - np.sort(ref[:, 0:3], axis=0)).max() > 3e-3 or np.abs(gf[0, 6:9] - ref[0, 6:9]).max() > 1e-2 \
            or np.abs(np.sort(gf[:, 9]) - np.sort(ref[:, 9])).max() > 2e-5
        flagged += int(amb.value < 1.0)
        gross_unflagged += int(gross and not amb.value < 1.0)
    print('cylinder pairs in contact %d, float pass flagged ambiguous %d, gross float32 answers not flagged %d' % (checked, flagged, gross_unflagged))
    assert checked > 150 and gross_unflagged <= 0.02 * checked and flagged < 0.2 * checked
