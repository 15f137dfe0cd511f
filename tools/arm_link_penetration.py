#!/usr/bin/env python3
"""How often do the iiwa ARM links (which neither the kernels nor the oracle collide: SURVEY section 3.6, DESIGN deviations)
reach into the table or into an object?  Measured on the float64 oracle (CPU), random policy, for joint control
(kuka.py:204-206: +-0.05 rad per joint per step -- the arm may sweep anywhere) and, for scale, tip control (tool clipped to the
box above the table: expected 0).

After every env-step (= 100 substeps) the bounding BOX of every link's collision mesh (link frame, from the STL files:
tests/golden/model.json `col_aabb`) is placed by a numpy forward kinematics over the PyBullet link list and tested against
  * the table (solid box, top at z = 0.16): penetration = depth of the box's deepest corner under the table top, for corners
    whose (x, y) lies inside the table's rectangle;
  * every free object: penetration = inscribed-sphere radius (0.015 cube, 0.01 puck) minus the distance of the object's centre
    from the link's box.
A mesh fills its bounding box only partly, so the counts are UPPER bounds.  Links: iiwa_link_3..7 (link_0..2 cannot reach the
table: the robot stands beside it) and, for comparison, the gripper base cylinder and the fingers, which ARE collided.

    python tools/arm_link_penetration.py [N=2048] [T=300]  ->  one JSON line per (task, control mode)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle_lib as O   # noqa: E402

MODEL = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'model.json')))
BL = MODEL['bullet_links']
ARM = ['iiwa_link_3', 'iiwa_link_4', 'iiwa_link_5', 'iiwa_link_6', 'iiwa_link_7']
COLLIDED = ['iiwa_gripper_base_link', 'iiwa_gripper_finger1', 'iiwa_gripper_finger2']


def link_frames(q):
    """q [N, 9] -> {name: (p [N, 3], R [N, 3, 3])}: oracle kinematics() / fk64_link, vectorised."""
    N = q.shape[0]
    P, Rm = [], []
    out = {}
    for b in BL:
        if b['parent'] < 0:
            pp, Rp = np.zeros((N, 3)), np.tile(np.eye(3), (N, 1, 1))
        else:
            pp, Rp = P[b['parent']], Rm[b['parent']]
        R0 = Rp @ np.array(b['R'])
        p = pp + Rp @ np.array(b['xyz'])
        ax = np.array(b['axis'], float)
        if b['type'] == 0:
            a = q[:, b['dof']]
            c, s = np.cos(a), np.sin(a)
            K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
            Rq = np.eye(3)[None] * c[:, None, None] + s[:, None, None] * K[None] + (1 - c)[:, None, None] * np.outer(ax, ax)[None]
            R0 = R0 @ Rq
        elif b['type'] == 1:
            p = p + (R0 @ ax) * q[:, b['dof']][:, None]
        P.append(p); Rm.append(R0)
        out[b['name']] = (p, R0)
    return out


def box_corners(p, R, lo, hi):
    lo, hi = np.array(lo), np.array(hi)
    c = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])   # [8, 3]
    return p[:, None, :] + np.einsum('nij,kj->nki', R, c)                                                  # [N, 8, 3]


def table_penetration(p, R, lo, hi, tc, th):
    w = box_corners(p, R, lo, hi)
    inside = (np.abs(w[..., 0] - tc[0]) <= th[0]) & (np.abs(w[..., 1] - tc[1]) <= th[1])
    depth = np.where(inside, (tc[2] + th[2]) - w[..., 2], -1.0)
    return depth.max(1)


def object_penetration(p, R, lo, hi, centre, r_in):
    lo, hi = np.array(lo), np.array(hi)
    loc = np.einsum('nji,nj->ni', R, centre - p)                       # the object's centre in the link frame
    d = np.maximum(np.maximum(lo - loc, loc - hi), 0.0)
    return r_in - np.sqrt((d * d).sum(1))


def run(task, kw, N, T, threads):
    ora = O.OracleEnv(task, N, seed_base=1, seed_stride=1, threads=threads, **kw)
    ora.reset()
    rs = np.random.RandomState(0)
    A = ora.dims.action_dim
    slide = task == 'slide'
    tc = np.array([-0.70 if slide else -0.52, 0.0, 0.08])
    th = np.array(MODEL['long_table' if slide else 'table']['ext']) / 2
    nb = 0 if task == 'reach' else (kw.get('num_block', 1) if task.startswith('block') or task.startswith('chest') else 1)
    r_in = 0.01 if slide else 0.015
    aabb = {b['name']: b['col_aabb'] for b in BL if b['col_aabb']}
    cnt = {k: dict(table=0, obj=0, table_max=0.0, obj_max=0.0) for k in ARM + COLLIDED}
    any_arm = 0
    for t in range(T):
        if t % 50 == 0:
            ora.reset()
        ora.step(rs.uniform(-1, 1, (N, A)).astype(np.float32))
        st = ora.get_state().astype(np.float64)
        fr = link_frames(st[:, :9])
        step_any = np.zeros(N, bool)
        for name in ARM + COLLIDED:
            p, R = fr[name]
            lo, hi = aabb[name]
            dt = table_penetration(p, R, lo, hi, tc, th)
            do = np.full(N, -1.0)
            for b in range(nb):
                do = np.maximum(do, object_penetration(p, R, lo, hi, st[:, 64 + 13 * b:67 + 13 * b], r_in))
            c = cnt[name]
            c['table'] += int((dt > 0.005).sum()); c['obj'] += int((do > 0.005).sum())
            c['table_max'] = max(c['table_max'], float(dt.max())); c['obj_max'] = max(c['obj_max'], float(do.max()))
            if name in ARM:
                step_any |= (dt > 0.005) | (do > 0.005)
        any_arm += int(step_any.sum())
    ora.close()
    tot = N * T
    return dict(task=task, **kw, env_steps=tot, arm_link_penetration_gt_5mm_fraction=any_arm / tot,
                per_link={k: dict(table_frac=v['table'] / tot, object_frac=v['obj'] / tot, table_max_m=round(v['table_max'], 4),
                                  object_max_m=round(v['obj_max'], 4)) for k, v in cnt.items()})


if __name__ == '__main__':
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    th = O.usable_threads()
    for task, kw in [('reach', {'joint_control': True}), ('push', {'joint_control': True}), ('pick_and_place', {'joint_control': True}),
                     ('block_stack', {'num_block': 4, 'joint_control': True}), ('reach', {}), ('push', {}), ('pick_and_place', {}),
                     ('block_stack', {'num_block': 4})]:
        print(json.dumps(run(task, kw, N, T, th)), flush=True)
