"""pybullet_multigoal_gym_amd -- MI355X-native vectorised Kuka multigoal environments.

``make_env`` keeps the reference's signature (P/__init__.py:4-11) and adds the
batch geometry (``num_envs``, ``device``, ``seed``, ``seed_stride``).  Options
of the reference that lie outside the accelerated hot path raise
``NotImplementedError`` instead of being silently ignored.
"""
from .envs import KukaVecEnv  # noqa: F401

__all__ = ['make_env', 'KukaVecEnv']

_TASKS = ['push', 'reach', 'slide', 'pick_and_place',
          'block_stack', 'block_rearrange', 'chest_pick_and_place', 'chest_push',
          'primitive_push_assemble', 'primitive_push_reach', 'insertion']
_GRIPPERS = ['robotiq85', 'parallel_jaw']
_ACCELERATED = ['reach', 'push', 'slide', 'pick_and_place', 'block_stack', 'block_rearrange', 'chest_push',
                'chest_pick_and_place']


def make_env(task='reach', gripper='parallel_jaw', num_block=5, render=False, binary_reward=True,
             grip_informed_goal=False, task_decomposition=False,
             joint_control=False, max_episode_steps=50, distance_threshold=0.05,
             primitive=None,
             image_observation=False, depth_image=False, goal_image=False, point_cloud=False, state_noise=False,
             visualize_target=True,
             camera_setup=None, observation_cam_id=None, goal_cam_id=0,
             use_curriculum=False, num_goals_to_generate=1e6,
             num_envs=None, device=0, seed=0, seed_stride=1, env_index_offset=0, dtype='float32', _library=None):
    assert gripper in _GRIPPERS, 'invalid gripper: {}, only support: {}'.format(gripper, _GRIPPERS)
    if task not in _TASKS:
        raise ValueError('invalid task name: {}, only support: {}'.format(task, _TASKS))

    def unsupported(what):
        raise NotImplementedError('%s is outside the MI355X hot path of this build (accelerated: tasks %s, '
                                  "gripper 'parallel_jaw', state observations)" % (what, _ACCELERATED))
    if task not in _ACCELERATED:
        unsupported("task '%s'" % task)
    if gripper != 'parallel_jaw':
        unsupported("gripper '%s'" % gripper)
    if render:
        unsupported('render=True (GUI)')
    if image_observation or depth_image or goal_image or point_cloud:
        unsupported('image / depth / point-cloud observations')
    if camera_setup is not None:
        unsupported('camera_setup')
    if primitive is not None:
        unsupported('push primitives')
    if state_noise:
        unsupported('state_noise')
    if task in ('block_stack', 'block_rearrange'):
        assert num_block <= 5, "only support up to 5 blocks"
        if task == 'block_rearrange':   # kuka_multi_step_envs.py:158-159
            assert not task_decomposition, 'Block rearranging task does not support task decomposition.'
            assert not grip_informed_goal, 'Block rearranging task does not support gripper informed goal representation.'
    elif task in ('chest_push', 'chest_pick_and_place'):
        assert num_block <= 5, "only support up to 5 blocks"
    elif task_decomposition or use_curriculum or grip_informed_goal:
        unsupported('task_decomposition / use_curriculum / grip_informed_goal outside the multi-block tasks')
    return KukaVecEnv(task=task, num_envs=num_envs, binary_reward=binary_reward, joint_control=joint_control,
                      max_episode_steps=max_episode_steps, distance_threshold=distance_threshold, num_block=num_block,
                      seed=seed, seed_stride=seed_stride, device=device, env_index_offset=env_index_offset,
                      dtype=dtype, task_decomposition=task_decomposition, use_curriculum=use_curriculum,
                      num_goals_to_generate=num_goals_to_generate, grip_informed_goal=grip_informed_goal,
                      _library=_library)
