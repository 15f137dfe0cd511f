"""tools/refharness -- run the REFERENCE's own Python without PyBullet or gym (build container only).

The reference (/root/reference, pure Python) cannot be imported as is: `gym`, `pybullet`, `pybullet_utils` and
`quaternion` are absent from this image.  This package supplies

  * `stubs.install()`: minimal stand-in modules for what the reference imports from gym 0.17.3 (Env, Wrapper,
    spaces.Box / Dict / MultiDiscrete, utils.seeding.np_random, envs.registration register / make / registry with
    the TimeLimit wrapper) -- third-party code, restated from its published behaviour -- plus `pybullet`
    (constants only) and `pybullet_utils.bullet_client`;
  * `fake_bullet.FakeBulletClient`: a scripted Bullet client.  It parses the reference's URDF files for joint /
    link names, indices, limits and kinematics, and forwards the physics calls (resetJointState,
    setJointMotorControl*, calculateInverseKinematics, stepSimulation, getLinkState, base poses) to ONE world of
    oracle/libpmg_oracle.so through its pmgo_bw_* entry points.

With these in place `pybullet_multigoal_gym.make_env(...)` builds the reference's real env objects, and every line
of their orchestration (sampling, goal generation, observation assembly, reward, TimeLimit, curricula, sub-goals,
motor commands, the float32 action product, call order) executes as written by the reference's authors; only the
physics underneath is the oracle's.  tools/gen_reference_fixtures.py records what that pair produces into
tests/golden/ref_*.json; the tests then demand the same numbers from the oracle's own env entry points and from the
HIP library.  Nothing here (and nothing of the reference) travels to the GPU box -- only the JSON does.
"""
