"""CPU tier: analytic invariants of the oracle's physics."""
import numpy as np
import pytest

import oracle_lib as O

REST = np.array([0, -0.5592432, 0, 1.733180, 0, -0.8501557, 0, 0.02, 0.01])


def test_inverse_mass_matrix_is_spd(built):
    M = O.minv(REST)
    assert np.abs(M - M.T).max() < 1e-12
    assert np.linalg.eigvalsh(M).min() > 0


def test_velocity_product_terms_match_christoffel_symbols(built):
    """C(q, qd) qd of the articulated-body algorithm equals the Christoffel form built from
    finite differences of M(q) = (M^-1)^-1; link damping removed by differencing in qd."""
    rs = np.random.RandomState(0)
    q = np.array([0.3, -0.5, 0.2, 1.7, -0.4, -0.8, 0.5, 0.02, 0.01])
    qd = rs.uniform(-1, 1, 9) * 1e-2      # small: the quadratic damping term is O(|qd|^2 * 0.04)
    M = lambda x: np.linalg.inv(O.minv(x))
    z = np.zeros(9)
    g = -M(q) @ O.fdyn(q, z, z)
    # bias(qd) = C qd qd + D(qd);  take the even part in qd to drop the (odd) linear damping
    bias = lambda v: -M(q) @ O.fdyn(q, v, z) - g
    c_even = 0.5 * (bias(qd) + bias(-qd))
    h = 1e-5
    dM = [(M(q + h * np.eye(9)[k]) - M(q - h * np.eye(9)[k])) / (2 * h) for k in range(9)]
    c = np.array([sum((dM[k][i, j] - 0.5 * dM[i][j, k]) * qd[j] * qd[k] for j in range(9) for k in range(9)) for i in range(9)])
    assert np.abs(c_even - c).max() < 2e-7 + 0.05 * np.abs(c).max()


def test_ik_reaches_the_start_pose(built):
    q, it = O.ik(np.r_[REST[:7], 0.035, 0.035], [-0.52, 0.0, 0.25])
    p, R = O.fk_tip(q)
    assert it <= 40 and np.abs(p - [-0.52, 0, 0.25]).max() < 1e-4
    assert np.abs(R - np.diag([-1, 1, -1])).max() < 2e-3


def test_arm_holds_pose_and_tracks_the_target(built):
    env = O.OracleEnv('reach', 2, seed_stride=1)
    env.reset()
    o0 = env.reset()
    for _ in range(5):
        o, r, d, ok = env.step(np.zeros((2, 3), np.float32))
    assert np.abs(o['observation'] - o0['observation']).max() < 5e-4      # gravity sag under the PD motor
    a = np.tile(np.float32([1, 0, 0]), (2, 1))
    for _ in range(5):
        o, r, d, ok = env.step(a)
    assert np.allclose(o['observation'][:, 0] - o0['observation'][:, 0], 0.05, atol=3e-3)
    s = env.get_state()
    assert np.allclose(s[:, 18] - o0['observation'][:, 0], 0.05, atol=1e-6)   # ee target += 0.01 per unit action


def test_tip_target_is_clipped_to_the_workspace(built):
    env = O.OracleEnv('reach', 1)
    env.reset()
    for _ in range(30):
        env.step(np.float32([[1, 1, -1]]))
    s = env.get_state()[0]
    assert np.allclose(s[18:21], [-0.37, 0.20, 0.175], atol=1e-7)          # kuka.py:40-41,210-212


def test_block_rests_on_the_table_and_closeness_formula(built):
    env = O.OracleEnv('pick_and_place', 4, seed_stride=1)
    env.reset()
    a = np.zeros((4, 4), np.float32)
    a[:, 3] = -1   # open the gripper (kuka.py:171: -1 -> 0.0 = open)
    for _ in range(6):
        o, r, d, ok = env.step(a)
    s = env.get_state()
    assert np.abs(s[:, 66] - 0.175).max() < 2e-4                            # block z
    assert np.abs(s[:, 64 + 7:64 + 13]).max() < 2e-3                        # at rest (solver residual floor)
    closeness = o['observation'][:, 6]
    assert np.allclose(closeness, 0.07 - s[:, 7] - s[:, 8], atol=1e-6)      # SURVEY.md finding 0-6
    assert (closeness > 0.06).all()


def test_time_limit_and_reward_semantics(built):
    env = O.OracleEnv('reach', 1, max_episode_steps=3)
    env.reset()
    dones = [env.step(np.zeros((1, 3), np.float32))[2][0] for _ in range(4)]
    assert dones == [False, False, True, True]
    r, ok = env.compute_reward(np.float32([[0, 0, 0], [0, 0, 0]]), np.float32([[0.04, 0, 0], [0.06, 0, 0]]))
    assert r.tolist() == [-0.0, -1.0] and np.signbit(r[0]) and ok.tolist() == [True, False]
    dense = O.OracleEnv('reach', 1, binary_reward=False)
    r, ok = dense.compute_reward(np.float32([[0, 0, 0]]), np.float32([[0.3, 0.4, 0]]))
    assert abs(r[0] + 0.5) < 1e-6 and not ok[0]


def test_box_box_face_contact(built):
    I = np.eye(3).ravel()
    c = O.box_box([0, 0, 0.0149], I, [0.015] * 3, [0, 0, -0.08], I, [0.25, 0.35, 0.08])
    assert len(c) == 4
    assert np.allclose(c[:, 6:9], [0, 0, 1])                      # normal from B (table) to A (block)
    assert np.allclose(c[:, 9], -1e-4, atol=1e-9)                 # 0.1 mm penetration
    assert sorted(map(tuple, np.round(np.abs(c[:, 0:2]), 6))) == [(0.015, 0.015)] * 4
    assert len(O.box_box([0, 0, 0.018], I, [0.015] * 3, [0, 0, -0.08], I, [0.25, 0.35, 0.08])) == 0   # beyond the 2 mm margin
