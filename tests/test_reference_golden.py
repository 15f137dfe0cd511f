"""tests/golden/ref_*.json were produced by the REFERENCE's own Python (its real make_env / reset / step / _get_obs /
_compute_reward / curricula / sub-goals, run by tools/gen_reference_fixtures.py on a scripted Bullet client whose physics
calls land in the oracle).  Here the same sessions are demanded

  * from the oracle's own env entry points: every orchestration row of SURVEY.md section 8(a) restated in
    oracle/pmg_oracle.c must give what the reference's code gives, to float32 output rounding (the physics underneath is
    shared, so any larger difference is an orchestration difference);
  * from the product's kernels (CPU emulator here, the HIP library under -m gpu) through the un-batched host API
    (num_envs=None: BASELINE.json configs[0]'s shapes), at the float32 bars of DESIGN.md section 5.

The physics itself (row a21) is NOT pinned by these files: PyBullet is absent, the oracle restates it ([BULLET-PRIOR]).
"""
import os
import re

import pytest

import ref_replay as R

PATHS = R.fixture_paths()
NAMES = [os.path.basename(p)[4:-5] for p in PATHS]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fixtures_exist():
    assert len(PATHS) >= 24, 'run tools/gen_reference_fixtures.py (build container only)'
    tasks = {R.load(p)['task'] for p in PATHS}
    assert tasks == {'reach', 'push', 'pick_and_place', 'slide', 'block_stack', 'block_rearrange', 'chest_push', 'chest_pick_and_place'}


@pytest.mark.parametrize('path', PATHS, ids=NAMES)
def test_oracle_reproduces_reference_session(built, path):
    fx = R.load(path)
    env = R.OracleAdapter(fx)
    R.replay(fx, env, tol_static=1.5e-7, tol_traj=1.5e-7)    # float32 rounding of O(1) values: 6e-8
    env.close()
    assert fx['urdf_fk_max_err'] < 1e-10      # the oracle's link kinematics vs the URDF text, checked at every getLinkState


def test_world_parameters_the_reference_sets(built):
    """base_env.py:203-220 as executed: gravity, contact ERP, 0.04 s / 20 substeps / 5 solver iterations -- against the
    constants compiled into the kernels."""
    src = open(os.path.join(ROOT, 'pybullet_multigoal_gym_amd', 'csrc', 'pmg_device_body.inc')).read()

    def const(name):
        return float(re.search(r'constexpr \w+ %s = ([-0-9.e]+)f?;' % name, src).group(1))
    for p in PATHS:
        wp = R.load(p)['world_params']
        assert wp['gravity'] == [0, 0, -const('GRAVITY')]
        assert wp['contact_erp'] == pytest.approx(const('CONTACT_ERP'), abs=1e-7)
        assert wp['fixedTimeStep'] == pytest.approx(const('PHYSICS_DT'), abs=1e-9)
        assert wp['numSubSteps'] == const('SUBSTEPS') and wp['numSolverIterations'] == const('SOLVER_ITERS')
        assert wp['fixedTimeStep'] / wp['numSubSteps'] == pytest.approx(const('DT'), abs=1e-9)


@pytest.mark.parametrize('path', PATHS, ids=NAMES)
def test_spaces_and_bullet_call_counts(built, emu_library, path):
    """Shapes of the reference's spaces, TimeLimit, and how often the reference really steps the simulation."""
    import warnings
    import pybullet_multigoal_gym_amd as pmg
    fx = R.load(path)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        env = pmg.make_env(task=fx['task'], num_envs=None, _library=emu_library, **fx['make_kwargs'])
    assert env.action_space.shape == (fx['action_dim'],)
    assert list(env.action_space.low) == fx['action_low'] and list(env.action_space.high) == fx['action_high']
    assert env._max_episode_steps == fx['max_episode_steps']
    for k, shape in fx['observation_space'].items():      # the reference's keys: state, policy_state, achieved_goal, desired_goal
        assert list(env.observation_space.spaces[k].shape) == shape, k
    env.close()
    steps = sum(1 for e in fx['events'] if e['op'] == 'step')
    assert fx['bullet_calls'].get('stepSimulation', 0) == 5 * steps            # kuka.py:223-225
    assert fx['bullet_calls']['calculateInverseKinematics'] >= 2               # constructor: robot.reset() + env.reset()


# float32 device vs the float64 physics under the fixtures, over WHOLE episodes: resets / goals / curricula / sub-goals to
# 2e-5 (positions of a reset are IK solutions); trajectories 2e-4 (reach: 2e-5) and the velocity columns at BASELINE.json's
# 1e-3 (Bullet's solver stops iterating at 3e-4 m/s of residual; rounds 3-4: 1e-3 / 5e-3).  FIXED bars for every session.  Rounds 3-4 relaxed seven
# sessions (`RELAXED`: up to 5e-3 / 2e-2) wherever the float32 build of the ORACLE strayed from the fixture -- the device
# measures 3e-5 (positions) and 4.5e-4 (velocity columns) on those very sessions (round 5, gpurun_out/r05h), so the list and
# the float32-oracle yardstick are gone.
GPU_BARS = {'reach': dict(tol_traj=2e-5, tol_vel=1e-3)}     # 2.5e-7; velocities 2.2e-4 with joint control (fingers on the table)
GPU_DEFAULT = dict(tol_traj=2e-4, tol_vel=1e-3)   # measured over all 36 sessions, whole episodes (round 5): positions <= 3.0e-5, velocity columns <= 5.4e-4


def _replay_whole_episodes(path, library):
    fx = R.load(path)
    bars = dict(GPU_BARS.get(fx['task'], GPU_DEFAULT))
    env = R.ProductAdapter(fx, library=library)
    worst = R.replay(fx, env, tol_static=2e-5, threshold_guard=2e-3, traj_steps=None, **bars)
    env.close()
    print('worst', os.path.basename(path), worst, 'bars', bars)


@pytest.mark.parametrize('path', PATHS, ids=NAMES)
def test_emulated_kernels_reproduce_reference_session(built, emu_library, path):
    """The product's kernel sources on the CPU emulator: every reference session, whole episodes, at the device's bars."""
    _replay_whole_episodes(path, emu_library)


@pytest.mark.gpu
@pytest.mark.parametrize('path', PATHS, ids=NAMES)
def test_hip_reproduces_reference_session(built, hip_library, path):
    _replay_whole_episodes(path, hip_library)


# ---------------------------------------------------------------------------------------------------------------------
# The day a machine with pybullet~=3.0.6 + gym~=0.17.3 runs `tools/gen_reference_fixtures.py --real`, the same sessions on
# the REAL reference land in tests/golden/real_*.json and these tests pin the physics at BASELINE.json's
# 1e-3.  Until then: skipped, and the physics stays [BULLET-PRIOR] -- "parity unpinned" for that row.
import glob  # noqa: E402

REAL = sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', 'real_*.json')))


@pytest.mark.skipif(not REAL, reason='no capture from real PyBullet exists (pybullet / gym are absent here): physics parity unpinned')
@pytest.mark.parametrize('path', REAL or [None])
def test_oracle_matches_real_pybullet_capture(built, path):
    fx = R.load(path)
    env = R.OracleAdapter(fx)
    R.replay(fx, env, tol_static=1e-3, tol_traj=1e-3, tol_vel=5e-2, threshold_guard=2e-3, check_internal=False)
    env.close()


ALTERNATIVES = {'motor_impulse_dt': [0.04, 0.002], 'warm_start': [0.0, 0.85], 'link_damping': [0.04, 0.0],
                'damping_per_substep': [0.0, 1.0], 'residual_threshold': [1e-7, 0.0], 'friction_dirs': [2.0, 1.0],
                'contact_warm_start': [0.0, 0.85]}


def rank_priors(fixtures, alternatives=None):
    """Replay `fixtures` under every combination of the oracle's switchable [BULLET-PRIOR] choices and rank the
    combinations by the mean trajectory error -- the verdict a first real PyBullet capture gets on day one."""
    import itertools
    import oracle_lib
    alternatives = alternatives or ALTERNATIVES
    names = sorted(alternatives)
    table = []
    try:
        for combo in itertools.product(*[alternatives[n] for n in names]):
            for n, v in zip(names, combo):
                oracle_lib.set_prior(n, v)
            errs = []
            for fx in fixtures:
                env = R.OracleAdapter(fx)
                try:
                    w = R.replay(fx, env, tol_static=1e9, tol_traj=1e9, tol_vel=1e9, threshold_guard=1e9, check_internal=False)
                    errs.append(w['traj'])
                finally:
                    env.close()
            table.append((float(sum(errs) / len(errs)), dict(zip(names, combo))))
    finally:
        oracle_lib.reset_priors()
    return sorted(table, key=lambda t: t[0])


def test_prior_switches_change_the_physics_and_reset(built):
    """The switchable [BULLET-PRIOR] choices exist, move the trajectories, and the defaults (what the product compiles in
    and what tests/golden/ref_*.json were recorded with) come back after pmgo_reset_priors()."""
    import oracle_lib
    assert set(ALTERNATIVES) <= set(oracle_lib.prior_names())
    fx = R.load(os.path.join(ROOT, 'tests', 'golden', 'ref_push_joint.json'))
    ranked = rank_priors([fx], {'motor_impulse_dt': [0.04, 0.002], 'warm_start': [0.0, 0.85]})
    assert ranked[0][1] == {'motor_impulse_dt': 0.04, 'warm_start': 0.0} and ranked[0][0] < 1e-6      # the recorded combination
    assert ranked[1][0] > 1e-5                                                                          # any other one moves it
    assert oracle_lib.get_prior('motor_impulse_dt') == 0.04 and oracle_lib.get_prior('warm_start') == 0.0
    env = R.OracleAdapter(fx)
    R.replay(fx, env, tol_static=1.5e-7, tol_traj=1.5e-7)
    env.close()


@pytest.mark.skipif(not REAL, reason='no capture from real PyBullet exists: nothing to rank the [BULLET-PRIOR] alternatives against')
def test_rank_prior_alternatives_against_real_capture(built):
    ranked = rank_priors([R.load(p) for p in REAL])
    for err, combo in ranked[:8]:
        print('%.3e %s' % (err, combo))
    assert ranked[0][0] < 1e-3, 'no combination of the switchable priors reaches BASELINE.json\'s 1e-3: look at the structural ones (manifold, GJK)'
