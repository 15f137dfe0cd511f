#!/usr/bin/env python3
"""Generate tests/golden/*.json -- the vectors that CAN be pinned without PyBullet.

Runs in the build container only.  Sources of truth:
  * numpy's legacy RandomState (bit-stable MT19937 stream) seeded the way gym 0.17.3's
    seeding.np_random does (sha512(str(seed))[:8] -> little-endian uint32 words);
  * the reference's sampling rules, restated here in numpy with file:line citations
    (P/ = /root/reference/pybullet_multigoal_gym/), so that the oracle's and the HIP
    kernel's draws can be checked draw-by-draw:
      P/robots/kuka.py:35-51                      workspace boxes
      P/envs/base_envs/kuka_single_step_base_env.py:104-143   object + goal sampling
      P/envs/base_envs/kuka_multi_step_base_env.py:221-235    block placement
      P/envs/task_envs/kuka_multi_step_envs.py:34-74          stack order + base target
  * the analytic FK known answer of SURVEY.md section 7-1.
The reference package itself cannot be imported (gym / pybullet are absent).
"""
import hashlib
import json
import os

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')


def gym_np_random(seed):
    h = hashlib.sha512(str(seed).encode('utf8')).digest()[:8]
    big = int.from_bytes(h, 'little')
    ints = []
    while big > 0:
        big, mod = divmod(big, 2 ** 32)
        ints.append(mod)
    rs = np.random.RandomState()
    rs.seed(ints or [0])
    return rs, ints or [0]


def boxes(task):
    tip = np.array([-0.52, 0.0, 0.25])
    if task in ('push', 'slide'):
        tip[-1] = 0.175 + 0.001
    lower = np.array([-0.67, -0.20, 0.175])
    # kuka_single_step_envs.py:14,29,44,57: slide is built with obj_range=0.1, target_range=0.2 (the others 0.15 / 0.15)
    obj_range, tgt_range = (0.1, 0.2) if task == 'slide' else (0.15, 0.15)
    obj_lo, obj_hi = tip.copy() - obj_range, tip.copy() + obj_range
    obj_lo[0] += 0.03; obj_hi[0] -= 0.03
    tgt_lo, tgt_hi = tip.copy() - tgt_range, tip.copy() + tgt_range
    tgt_lo[0] += 0.03; tgt_lo[-1] = lower[-1]; tgt_hi[0] -= 0.03
    if task == 'slide':  # kuka_single_step_base_env.py:66-69
        tgt_lo[0] -= 0.4; tgt_hi[0] -= 0.4
    return tip, obj_lo, obj_hi, tgt_lo, tgt_hi


def single_reset(rs, task):
    tip, obj_lo, obj_hi, tgt_lo, tgt_hi = boxes(task)
    has_obj, grasping, in_air = task != 'reach', task == 'pick_and_place', task in ('reach', 'pick_and_place')
    obj = None
    center = tip.copy()
    obj_z = 0.170 if task == 'slide' else 0.175   # kuka_single_step_base_env.py:49,56
    if has_obj:
        xy = tip[:2]
        while np.linalg.norm(xy - tip[:2]) < 0.1:
            xy = rs.uniform(obj_lo[:-1], obj_hi[:-1])
        obj = np.append(xy, obj_z)
        center = obj
    while True:
        g = rs.uniform(tgt_lo, tgt_hi)
        if np.linalg.norm(g - center) > 0.1:
            break
    if not in_air:
        g[2] = obj_z
    elif grasping:
        if rs.uniform(0, 1) >= 0.5:
            g[2] = obj_z
    return obj, g


def stack_reset(rs, nb):
    tip, obj_lo, obj_hi, tgt_lo, tgt_hi = boxes('block_stack')
    poses = []
    for _ in range(nb):
        while True:
            xy = rs.uniform(obj_lo[:-1], obj_hi[:-1])
            if all(np.linalg.norm(xy - p[:-1]) > 0.06 for p in poses + [tip]):
                poses.append(np.concatenate((xy, [0.175])))
                break
    order = np.arange(nb, dtype=int)
    rs.shuffle(order)
    while True:
        b = rs.uniform(tgt_lo[:-1], tgt_hi[:-1])
        if all(np.linalg.norm(b - p[:-1]) > 0.08 for p in poses):
            break
    goal = [None] * nb
    for k in range(nb):
        goal[order[k]] = [b[0], b[1], 0.175 + 0.03 * k]
    return poses, order.tolist(), b.tolist(), np.concatenate(goal)


class Curriculum:
    """kuka_multi_step_base_env.py:121-140 (state) and :350-379 (_update_curriculum_prob), restated."""

    def __init__(self, n, num_goals_to_generate):
        self.n = n
        self.prob = np.concatenate([[1.0], np.zeros(n - 1)])
        self.per = num_goals_to_generate // n
        self.count = np.zeros(n)
        self.update = True

    def draw(self, rs):
        level = rs.choice(self.n, p=self.prob)            # the REAL numpy choice: pins the oracle's restatement
        return int(level)

    def account(self, level):
        if not self.update:
            return
        self.count[level] += 1
        fin = self.count >= self.per
        half = self.count >= (self.per / 2)
        self.prob[fin] = 0.0
        if half[0] and not fin[0]:
            self.prob[0] = 0.5; self.prob[1] = 0.5
        for i in range(1, self.n - 1):
            if fin[i - 1] and not fin[i]:
                if half[i]:
                    self.prob[i] = 0.5; self.prob[i + 1] = 0.5
                else:
                    self.prob[i] = 1.0
        if fin[-2]:
            self.prob[-1] = 1.0


def multi_blocks(rs, nb, tip, obj_lo, obj_hi):
    poses = []
    for _ in range(nb):
        while True:
            xy = rs.uniform(obj_lo[:-1], obj_hi[:-1])
            if all(np.linalg.norm(xy - p[:-1]) > 0.06 for p in poses + [tip]):
                poses.append(np.concatenate((xy, [0.175])))
                break
    return poses


def stack_curriculum_reset(rs, nb, cur):
    """kuka_multi_step_envs.py:34-87 + :124-148 with use_curriculum=True."""
    poses, order, base, goal = stack_reset(rs, nb)
    level = cur.draw(rs)
    cur.account(level)
    targets = [[base[0], base[1], 0.175 + 0.03 * k] for k in range(nb)]
    dg = [None] * nb
    for i in range(nb):
        dg[order[i]] = targets[i] if i <= level else poses[order[i]].tolist()
    return {'blocks': [p.tolist() for p in poses], 'order': order, 'base': base, 'level': level,
            'goal_step': level * 25 + 50, 'desired_goal': np.concatenate(dg).tolist(),
            'prob': cur.prob.tolist(), 'generated': cur.count.tolist()}


def rearrange_reset(rs, nb, cur=None):
    """kuka_multi_step_envs.py:174-227 (tip starts on the table: kuka_multi_step_envs.py:169)."""
    tip, obj_lo, obj_hi, tgt_lo, tgt_hi = boxes('push')
    poses = multi_blocks(rs, nb, tip, obj_lo, obj_hi)
    targets = []
    for _ in range(nb):
        while True:
            xy = rs.uniform(tgt_lo[:-1], tgt_hi[:-1])
            if all(np.linalg.norm(xy - p[:-1]) > 0.06 for p in targets + poses):
                targets.append(np.concatenate((xy, [0.175])))
                break
    out = {'blocks': [p.tolist() for p in poses], 'targets': [t.tolist() for t in targets]}
    if cur is None:
        out['desired_goal'] = np.concatenate(targets).tolist()
        return out
    level = cur.draw(rs)
    moved = np.sort(rs.choice(np.arange(nb), size=level + 1, replace=False), kind='stable').tolist()
    cur.account(level)
    tq = [t.copy() for t in targets]
    dg = []
    for i in range(nb):
        if i in moved:
            dg.append(tq[0].copy()); del tq[0]
        else:
            dg.append(poses[i].copy())
    out.update({'level': level, 'goal_step': level * 25 + 50, 'moved': [int(m) for m in moved],
                'desired_goal': np.concatenate(dg).tolist(), 'prob': cur.prob.tolist(), 'generated': cur.count.tolist()})
    return out


def chest_reset(rs, nb, pnp, cur=None):
    """kuka_multi_step_base_env.py:97-110, 221-250 (chest=True) + kuka_multi_step_envs.py:256-283, 344-383 (pick and
    place) / 405-431, 477-517 (push): blocks in the shifted object box, no random target, num_block + 1 curriculum levels."""
    tip = np.array([-0.52, 0.0, 0.25 if pnp else 0.175 + 0.001])
    obj_lo, obj_hi = tip.copy() - 0.1, tip.copy() + 0.1
    obj_lo[0] += 0.03; obj_hi[0] -= 0.03
    obj_lo[0] += 0.05; obj_hi[0] += 0.05; obj_lo[1] -= 0.05; obj_hi[1] += 0.05
    poses = multi_blocks(rs, nb, tip, obj_lo, obj_hi)
    centre = np.array([-0.7, 0.0, 0.21]); centre[0] += 0.05; centre[2] = 0.175
    out = {'blocks': [p.tolist() for p in poses]}
    door = [0.10 if pnp else 0.12]
    if cur is None:
        out['desired_goal'] = np.concatenate([door] + [centre] * nb).tolist()
        return out
    level = cur.draw(rs)
    moved = np.sort(rs.choice(np.arange(nb), size=level, replace=False), kind='stable').tolist()
    cur.account(level)
    dg = [door] + [centre if i in moved else poses[i] for i in range(nb)]
    out.update({'level': level, 'goal_step': level * 25 + 50, 'moved': [int(m) for m in moved],
                'desired_goal': np.concatenate(dg).tolist(), 'prob': cur.prob.tolist(), 'generated': cur.count.tolist()})
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = {}
    for seed in [0, 1, 2, 7, 12345, 2 ** 32 + 5, 2 ** 63 - 1]:
        rs, ints = gym_np_random(seed)
        d = rs.random_sample(64).tolist()
        perm = np.arange(5)
        rs.shuffle(perm)
        rng[str(seed)] = {'init_key': ints, 'random_sample_64': d, 'then_shuffle_arange5': perm.tolist()}
    json.dump(rng, open(os.path.join(OUT, 'rng.json'), 'w'), indent=0)

    samp = {}
    for task in ['reach', 'push', 'pick_and_place', 'slide']:
        for seed in [0, 3]:
            rs, _ = gym_np_random(seed)
            eps = []
            for ep in range(6):  # episode 0 is the reset consumed by the reference's constructor (base_env.py:84)
                obj, g = single_reset(rs, task)
                eps.append({'object': None if obj is None else obj.tolist(), 'goal': g.tolist()})
            samp['%s/%d' % (task, seed)] = eps
    for nb in [2, 4, 5]:
        for seed in [0, 3]:
            rs, _ = gym_np_random(seed)
            eps = []
            for ep in range(4):
                poses, order, base, goal = stack_reset(rs, nb)
                eps.append({'blocks': [p.tolist() for p in poses], 'order': order, 'base': base, 'goal': goal.tolist()})
            samp['block_stack%d/%d' % (nb, seed)] = eps
    json.dump(samp, open(os.path.join(OUT, 'sampling.json'), 'w'), indent=0)

    # multi-step bookkeeping: block_rearrange sampling, and the curriculum draw / probability schedule with a
    # small goal budget (num_goals_to_generate = 8*nb) so that every probability transition is walked through
    def episodes(fn, count):
        # the reference's schedule can leave prob summing to 0.5 when level i+1 uses up its budget before level i
        # (numpy then raises "probabilities do not sum to 1" inside the reference): the sequence stops there
        out = []
        for _ in range(count):
            try:
                out.append(fn())
            except ValueError:
                break
        return out

    multi = {}
    for nb in [2, 3, 5]:
        for seed in [0, 3]:
            rs, _ = gym_np_random(seed)
            multi['rearrange%d/%d' % (nb, seed)] = [rearrange_reset(rs, nb) for _ in range(4)]
            rs, _ = gym_np_random(seed)
            cur = Curriculum(nb, 8 * nb)
            multi['rearrange%d_curriculum/%d' % (nb, seed)] = episodes(lambda: rearrange_reset(rs, nb, cur), 10 * nb)
            rs, _ = gym_np_random(seed)
            cur = Curriculum(nb, 8 * nb)
            multi['block_stack%d_curriculum/%d' % (nb, seed)] = episodes(lambda: stack_curriculum_reset(rs, nb, cur), 10 * nb)
    json.dump({'num_goals_to_generate_per_block': 8, 'episodes': multi}, open(os.path.join(OUT, 'multistep.json'), 'w'), indent=0)

    chest = {}
    for task, pnp in [('chest_push', False), ('chest_pick_and_place', True)]:
        for nb in [1, 3, 5]:
            for seed in [0, 3]:
                rs, _ = gym_np_random(seed)
                chest['%s%d/%d' % (task, nb, seed)] = [chest_reset(rs, nb, pnp) for _ in range(3)]
                rs, _ = gym_np_random(seed)
                cur = Curriculum(nb + 1, 8 * (nb + 1))
                chest['%s%d_curriculum/%d' % (task, nb, seed)] = episodes(lambda: chest_reset(rs, nb, pnp, cur), 10 * (nb + 1))
    json.dump({'num_goals_to_generate_per_level': 8, 'episodes': chest}, open(os.path.join(OUT, 'chest.json'), 'w'), indent=0)

    fk = {'rest_pose': [0, -0.5592432, 0, 1.733180, 0, -0.8501557, 0, 0.035, 0.035],
          'tip_position': [-0.522923, 0.0, 0.250773], 'tip_position_tol': 1e-5,
          'tip_rotation': [[-1, 0, 0], [0, 1, 0], [0, 0, -1]], 'tip_rotation_tol': 1.1e-3,
          'source': 'SURVEY.md section 7-1 (computed independently in the survey session)'}
    json.dump(fk, open(os.path.join(OUT, 'fk.json'), 'w'), indent=0)
    print('wrote', sorted(os.listdir(OUT)))


if __name__ == '__main__':
    main()
