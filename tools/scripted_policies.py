#!/usr/bin/env python3
"""Scripted (hand-written, observation-feedback) policies that SOLVE the manipulation tasks, vectorised over envs.

Why: the reference's tasks exist to be solved (R/README.md:13-26; kuka_single_step_envs.py:4-16 PickAndPlace with
`grasping=True, target_in_the_air=True`; kuka_multi_step_envs.py:34-87 stacking, :256-342 chest).  A random policy
almost never puts the gripper on an object, so these controllers are what drives the gripper-on-object solver paths --
grasp, lift, carry, push-to-goal, stack, open-the-door-and-drop -- in the solvability and teacher-forced parity tests
(tests/test_scripted_tasks.py, tests/test_gpu_scripted.py).

Every policy reads ONLY the observation dict an agent would get (`observation`, `achieved_goal`, `desired_goal`; layouts
of SURVEY.md appendix B / kuka_single_step_base_env.py:193-235 / kuka_multi_step_base_env.py:255-336) and returns
actions in [-1, 1] (kuka.py:167-225: 1 cm of tip-target travel per unit, last entry = finger command, +1 closed).
They work unchanged on the oracle (tests/oracle_lib.OracleEnv) and on the HIP library (KukaVecEnv).

    tools/scripted_policies.py <task> [N] [T]      # runs the policy on the CPU oracle and prints the success rate
"""
import os
import sys

import numpy as np

STEP = 0.01            # tip-target travel per unit action (kuka.py:209)
TABLE_Z = 0.175        # block centre resting on the table = lower tip clip (kuka.py:40)
CHEST_CENTRE = np.array([-0.65, 0.0, 0.175])
GRASP_X = -0.412       # grasp points stay behind this x (see _PickPlaceCore)
SAFE_X = -0.415        # carry / place targets stay behind this x: beyond it the arm is nearly stretched and the DLS IK creeps
TIP_LOW, TIP_HIGH = np.array([-0.67, -0.20, 0.175]), np.array([-0.37, 0.20, 0.55])   # kuka.py:40-41


def _toward(tip, target, gain=1.0):
    return np.clip(gain * (target - tip) / STEP, -1.0, 1.0)


class Policy:
    """Base: per-env integer phase + counters; subclasses fill `_act`."""
    action_dim = 3

    def __init__(self, n):
        self.n = n
        self.phase = np.zeros(n, np.int32)
        self.count = np.zeros(n, np.int32)

    def reset(self, mask=None):
        m = slice(None) if mask is None else np.asarray(mask, bool)
        self.phase[m] = 0
        self.count[m] = 0

    def act(self, obs):
        a = np.zeros((self.n, self.action_dim), np.float32)
        self._act(obs, a)
        return np.clip(a, -1.0, 1.0).astype(np.float32)

    def _set_phase(self, mask, p):
        self.phase[mask] = p
        self.count[mask] = 0


class ReachPolicy(Policy):
    def _act(self, obs, a):
        a[:, :3] = _toward(obs['observation'][:, -3:] if obs['observation'].shape[1] == 3 else obs['achieved_goal'],
                           obs['desired_goal'])


class _PickPlaceCore:
    """The grasp-and-carry state machine shared by pick_and_place, stacking, rearranging and the chest drop.
    phases: 0 fly above the block (open) - 1 descend (open) - 2 close - 3 lift - 4 carry above the target -
            5 lower to the target - 6 open - 7 retreat upwards - 8 done."""

    def __init__(self, n, clearance=0.06, grasp_steps=3, release_steps=2):
        self.n = n
        self.clearance = clearance
        self.grasp_steps = grasp_steps
        self.release_steps = release_steps

    def step(self, pol, idx, tip, blk, target, travel_z, a, hold=False, drop_from=None, direct=False, clamp_x=True, fly_z=None):
        """Advance the envs `idx` (bool mask): block at `blk`, to be put at `target`.  `hold`: keep the block in the
        closed gripper at the target (single pick_and_place with an in-air goal).  Returns the mask of finished envs."""
        ph = pol.phase
        e = np.zeros_like(tip)
        grip = np.full(len(tip), -1.0)
        target = target.copy()
        if clamp_x:
            target[:, 0] = np.minimum(target[:, 0], SAFE_X)    # the arm's IK misbehaves at the far edge of the tip box
        lost = idx & (ph >= 3) & (ph <= 5) & (np.abs(blk - tip).max(1) > 0.035)
        pol._set_phase(lost, 0)                                # dropped the block: grasp it again
        # the grasp point: the block's centre, but not beyond where the arm gets quickly (GRASP_X: the fingers are 2.5 cm
        # wide in x, a block at the far edge of the object box is gripped up to 1.2 cm off-centre)
        grasp = blk.copy()
        grasp[:, 0] = np.minimum(grasp[:, 0], GRASP_X)
        above = grasp + np.array([0, 0, self.clearance])
        if fly_z is not None:                                  # fly over everything that stands on the table
            above[:, 2] = np.maximum(above[:, 2], fly_z)
        m = idx & (ph == 0)
        e[m] = (above - tip)[m]
        arrived = m & (np.abs(grasp[:, :2] - tip[:, :2]).max(1) < 0.006)
        pol._set_phase(arrived, 1)
        m = idx & (ph == 1)
        e[m] = (grasp - tip)[m]
        arrived = m & (np.abs(grasp - tip).max(1) < 0.006)
        pol._set_phase(arrived, 2)
        m = idx & (ph == 2)
        grip[m] = 1.0
        pol.count[m] += 1
        pol._set_phase(m & (pol.count >= self.grasp_steps), 5 if direct else 3)
        m = idx & (ph == 3)
        grip[m] = 1.0
        up = tip.copy()
        up[:, 2] = travel_z
        need_lift = np.abs(target[:, :2] - blk[:, :2]).max(1) > 0.01   # straight up only if there is xy travel to do
        e[m] = (up - tip)[m]
        pol._set_phase(m & ((tip[:, 2] > travel_z - 0.01) | ~need_lift), 4)
        m = idx & (ph == 4)
        grip[m] = 1.0
        over = target.copy()
        over[:, 2] = np.where(need_lift, np.maximum(travel_z, target[:, 2]), target[:, 2])
        e[m] = (over - blk)[m]
        pol._set_phase(m & (np.abs(target[:, :2] - blk[:, :2]).max(1) < 0.004), 5)
        m = idx & (ph == 5)
        grip[m] = 1.0
        goal = target if drop_from is None else drop_from
        e[m] = (goal - blk)[m]
        over_it = m & (np.abs(goal[:, :2] - blk[:, :2]).max(1) < 0.004)
        pol.count[over_it] += 1                                # a block set down on another one may rest a little high
        at = over_it & ((np.abs(goal - blk).max(1) < 0.004) | ((pol.count > 5) & (np.abs(goal - blk).max(1) < 0.012)))
        if not hold:
            pol._set_phase(at, 6)
        m = idx & (ph == 6)
        grip[m] = -1.0
        e[m, :2] = (blk - tip)[m, :2]
        pol.count[m] += 1
        pol._set_phase(m & (pol.count >= self.release_steps), 7)
        m = idx & (ph == 7)
        grip[m] = -1.0
        up = tip.copy()
        up[:, 2] = target[:, 2] + 0.05
        e[m] = (up - tip)[m]
        done = m & (tip[:, 2] > target[:, 2] + 0.038)
        a[idx, :3] = np.clip(e[idx] / STEP, -1, 1)
        a[idx, 3] = grip[idx]
        return done


class PickAndPlacePolicy(Policy):
    """open -> above the block -> descend -> close -> carry to the goal (in the air or on the table) and hold it there."""
    action_dim = 4

    def __init__(self, n):
        super().__init__(n)
        self.core = _PickPlaceCore(n)

    def _act(self, obs, a):
        ob = obs['observation']
        tip, blk = ob[:, 0:3].astype(np.float64), ob[:, 3:6].astype(np.float64)
        goal = obs['desired_goal'].astype(np.float64)
        # one block, nothing to fly over: carry along the straight (L-infinity) line -- 50 steps are tight
        self.core.step(self, np.ones(self.n, bool), tip, blk, goal, tip[:, 2], a, hold=True, direct=True)


class PushPolicy(Policy):
    """Closed gripper, axis-aligned legs.  The finger faces are axis-aligned (fixed tool orientation, kuka.py:42) and
    Bullet's friction is a pyramid (two independent btPlaneSpace1 directions per contact), so a block pushed obliquely
    slides along the face instead of following the gripper: push along y until the y error is gone, then along x (larger
    error first; a leg whose error is under `skip` is left out).  Per leg: rise, fly behind the block, come down, push.
    phases: 0 pick the leg - 1 rise / fly - 2 descend - 3 push."""

    def __init__(self, n, standoff=0.042, fly_z=0.222, ground_z=0.177, skip=0.02, tol=0.006, side_y=0.15):
        super().__init__(n)
        self.standoff, self.fly_z, self.ground_z, self.skip, self.tol = standoff, fly_z, ground_z, skip, tol
        self.axis = np.zeros(n, np.int32)
        # far edge of the tip box: the arm cannot get behind a block at x > -0.43 near the centre line (the IK creeps; it
        # reaches x = -0.378 only at |y| >= 0.15).  A fly phase that stalls there triggers a detour: push the block
        # sideways to |y| = side_y first (stage 1), then towards the robot with a short stand-off (stage 2)
        self.side_y = side_y
        self.detour = np.zeros(n, np.int32)
        self.side = np.ones(n)

    def reset(self, mask=None):
        super().reset(mask)
        self.detour[slice(None) if mask is None else np.asarray(mask, bool)] = 0

    def _act(self, obs, a):
        ob = obs['observation']
        tip, blk = ob[:, 0:3].astype(np.float64), ob[:, 3:6].astype(np.float64)
        goal = obs['desired_goal'].astype(np.float64)
        self.push_step(np.ones(self.n, bool), tip, blk, goal, a)

    def push_step(self, idx, tip, blk, goal, a):
        """Advance the envs `idx`; returns the mask of envs whose block is at its goal (both errors under `skip`)."""
        n = self.n
        rows = np.arange(n)
        err = goal[:, :2] - blk[:, :2]
        finished = idx & (np.abs(err).max(1) < self.skip)
        ph = self.phase
        # the detour's own targets: stage 1 = sideways to side * side_y, stage 2 = the x error alone
        d1, d2 = self.detour == 1, self.detour == 2
        err[d1, 0] = 0.0
        err[d1, 1] = (self.side * self.side_y - blk[:, 1])[d1]
        self.detour[d1 & (np.abs(err[:, 1]) < 0.012)] = 2
        d1, d2 = self.detour == 1, self.detour == 2
        err[d2, 0] = (goal[:, 0] - blk[:, 0])[d2]
        err[d2, 1] = 0.0
        self.detour[d2 & ((np.abs(err[:, 0]) < self.tol + 0.004) | (blk[:, 0] < -0.47))] = 0   # back where the arm gets behind it anywhere
        finished &= self.detour == 0
        m = idx & (ph == 0) & ~finished
        self.axis[m] = np.argmax(np.abs(err), axis=1)[m]
        self._set_phase(m, 1)
        k = self.axis
        sgn = np.sign(err[rows, k])
        sgn[sgn == 0] = 1.0
        behind = blk[:, :2].copy()
        behind[rows, k] -= sgn * np.where(self.detour == 2, 0.034, self.standoff)
        behind = np.clip(behind, TIP_LOW[:2] + 0.004, TIP_HIGH[:2] - 0.004)   # the tip target is clipped to this box (kuka.py:40-41)
        e = np.zeros((n, 3))
        m = idx & (ph == 1) & ~finished              # rise, then fly
        tgt = np.concatenate([behind, np.full((n, 1), self.fly_z)], 1)
        low = tip[:, 2] < self.fly_z - 0.012
        near_block = np.abs(tip[:, :2] - blk[:, :2]).max(1) < 0.045
        crossing = low & near_block                  # straight up while next to the block
        tgt[crossing, :2] = tip[crossing, :2]
        already = (np.abs(behind - tip[:, :2]).max(1) < 0.012) & low    # e.g. the second leg starts where it stands
        e[m] = (tgt - tip)[m]
        self.count[m] += 1
        stalled = m & (self.count > 35) & (self.detour == 0) & (k == 0) & (sgn < 0) & (blk[:, 0] > -0.45)
        self.detour[stalled] = 1
        self.side[stalled] = np.where(blk[stalled, 1] >= 0, 1.0, -1.0)
        self._set_phase(stalled, 0)
        self._set_phase(m & ~stalled & ((np.abs(behind - tip[:, :2]).max(1) < 0.005) | already), 2)
        m = idx & (ph == 2) & ~finished              # descend
        tgt = np.concatenate([behind, np.full((n, 1), self.ground_z)], 1)
        e[m] = (tgt - tip)[m]
        self._set_phase(m & (tip[:, 2] < self.ground_z + 0.004), 3)
        m = idx & (ph == 3) & ~finished              # push along the leg's axis, hold the other coordinate on the block
        other = 1 - k
        v = np.zeros((n, 2))
        v[rows, k] = sgn * np.minimum(1.0, (np.abs(err[rows, k]) + 0.004) / STEP) * STEP
        v[rows, other] = blk[rows, other] - tip[rows, other]
        e[m, :2] = v[m]
        e[m, 2] = (self.ground_z - tip[:, 2])[m]
        leg_done = m & (np.abs(err[rows, k]) < self.tol)
        slipped = m & ((np.abs(blk[rows, other] - tip[rows, other]) > 0.02) | ((tip[rows, k] - blk[rows, k]) * sgn > 0.0))   # beside / past the block
        self._set_phase(leg_done | slipped, 0)
        a[idx, :3] = np.clip(e[idx] / STEP, -1, 1)
        a[finished, :3] = 0.0
        return finished


def multi_block_views(obs, nb):
    """tip xyz, finger closeness and the [n, nb, 3] block positions of a multi-block observation
    (kuka_multi_step_base_env.py:264-283; joint_control=False)."""
    ob = obs['observation'].astype(np.float64)
    blocks = np.stack([ob[:, 8 + 16 * b:11 + 16 * b] for b in range(nb)], axis=1)
    return ob[:, 0:3], ob[:, 3], blocks


class StackPolicy(Policy):
    """block_stack (kuka_multi_step_envs.py:34-87): the desired goal holds one target per block, the stack's levels
    0.03 apart; pick the blocks in the order of their target heights and set each one down on the previous one.
    A block already within `tol` of its target is left alone."""
    action_dim = 4

    def __init__(self, n, num_block=4, tol=0.012):
        super().__init__(n)
        self.nb, self.tol = num_block, tol
        self.core = _PickPlaceCore(n)
        self.cur = np.zeros(n, np.int32)           # position in the stacking order
        self.staging = np.zeros(n, bool)           # pulling a far-edge block in before it is stacked

    def reset(self, mask=None):
        super().reset(mask)
        m = slice(None) if mask is None else np.asarray(mask, bool)
        self.cur[m] = 0
        self.staging[m] = False

    def _act(self, obs, a):
        n, nb = self.n, self.nb
        rows = np.arange(n)
        tip, closeness, blocks = multi_block_views(obs, nb)
        targets = obs['desired_goal'].astype(np.float64)[:, :3 * nb].reshape(n, nb, 3)
        order = np.argsort(targets[:, :, 2], axis=1, kind='stable')

        def place_of(k):
            """Where the k-th block of the order goes: the base to its target on the table (kept off the far edge of
            the tip box), every other block ON the block below it, wherever that one came to rest."""
            b = order[rows, k]
            place = targets[rows, b].copy()
            base = k == 0
            place[base, 0] = np.minimum(place[base, 0], SAFE_X)
            below = order[rows, np.maximum(k - 1, 0)]
            place[~base, :2] = blocks[rows, below][~base, :2]
            return b, place

        for _ in range(nb):                        # between grasps: step over the blocks that are where they belong
            k = np.minimum(self.cur, nb - 1)
            b, place = place_of(k)
            ok = np.abs(blocks[rows, b] - place).max(1) < self.tol
            skip = ok & (self.phase == 0) & (self.cur < nb)
            self.cur[skip] += 1
        active = self.cur < nb
        k = np.minimum(self.cur, nb - 1)
        b, place = place_of(k)
        # a block at the far edge of the object box can only be gripped off-centre (GRASP_X), and set down on another block
        # like that its overhanging fingers shove the tower: pull it in first -- put it on the table 6 cm closer, at a free
        # spot, and pick it up again, centred
        mine = blocks[rows, b]
        far = active & (mine[:, 0] > GRASP_X + 0.002) & (self.phase == 0)
        self.staging[far] = True
        self.staging[self.staging & (mine[:, 0] <= GRASP_X + 0.002) & (self.phase == 0)] = False
        if self.staging.any():
            spot = np.stack([np.full(n, -0.47), mine[:, 1], np.full(n, TABLE_Z)], 1)
            for _ in range(3):                     # slide the spot along y away from the other blocks
                d = blocks[:, :, :2] - spot[:, None, :2]
                d[rows, b] = 9.0
                near = np.linalg.norm(d, axis=2).min(1) < 0.055
                spot[near, 1] += np.where(spot[near, 1] > 0, -0.06, 0.06)
            place[self.staging] = spot[self.staging]
        # carry height: the carried block's underside clears whatever stands at the target (and the blocks on the table)
        travel = np.maximum(place[:, 2], TABLE_Z) + 0.045
        fly = blocks[:, :, 2].max(1) + 0.065
        done = self.core.step(self, active, tip, blocks[rows, b], place, travel, a, clamp_x=False, fly_z=fly)
        self._set_phase(done, 0)                   # the check above decides whether it worked or is tried again
        idle = ~active                             # everything stacked: hover open-handed above the stack
        a[idle, :3] = 0.0
        a[idle, 3] = -1.0


class ChestPushPolicy(Policy):
    """chest_push (kuka_multi_step_envs.py:388-517, chest_front_sliding_door.urdf): the closed fingers rise over the
    door, come down beside the handle and push it along +y until the door latches open (chest.py:59-68); then every
    block is pushed into the chest -- first along y to the doorway's centre line, then along -x through it.
    phases: 10 rise - 11 fly beside the handle - 12 push the handle - 13 back off; then the push legs per block."""

    def __init__(self, n, num_block=1):
        super().__init__(n)
        self.nb = num_block
        self.push = PushPolicy(n, skip=0.012)
        self.phase[:] = 10
        self.cur = np.zeros(n, np.int32)

    def reset(self, mask=None):
        m = slice(None) if mask is None else np.asarray(mask, bool)
        self.phase[m] = 10
        self.count[m] = 0
        self.cur[m] = 0
        self.push.reset(mask)

    def _act(self, obs, a):
        n, nb = self.n, self.nb
        rows = np.arange(n)
        tip, _, blocks = multi_block_views(obs, nb)
        door = obs['achieved_goal'][:, 0].astype(np.float64)
        ph = self.phase
        e = np.zeros((n, 3))
        m = ph == 10
        e[m, 2] = 0.236 - tip[m, 2]
        self._set_phase(m & (tip[:, 2] > 0.23), 11)
        m = ph == 11
        beside = np.array([-0.577, -0.035, 0.236])   # the finger backs just clear the door face (x = -0.592)
        e[m] = (beside - tip)[m]
        self._set_phase(m & (np.abs(beside - tip).max(1) < 0.006), 12)
        m = ph == 12
        e[m] = np.array([0.0, 0.01, 0.0])
        e[m, 0] = -0.577 - tip[m, 0]
        e[m, 2] = 0.236 - tip[m, 2]
        self._set_phase(m & (door > 0.117), 13)
        m = ph == 13                                   # away from the handle before the arm comes down again
        e[m] = np.array([0.01, 0.0, 0.0])
        self._set_phase(m & (tip[:, 0] > -0.56), 14)
        a[:, :3] = np.clip(e / STEP, -1, 1)
        pushing = ph == 14
        if pushing.any():
            self.push.phase[~pushing] = 0
            k = np.minimum(self.cur, nb - 1)
            blk = blocks[rows, k]
            goal = np.tile(CHEST_CENTRE, (n, 1))
            goal[:, 0] += 0.03 * k                     # later blocks stop behind the earlier ones
            off_line = np.abs(blk[:, 1] - goal[:, 1]) > 0.012
            outside = blk[:, 0] > -0.585               # still in front of the doorway
            hold = off_line & outside & (self.push.detour == 0)       # (a far-edge detour of the push legs has its own targets)
            goal[hold, 0] = blk[hold, 0]                              # first to the centre line, only then in
            act = pushing & (self.cur < nb)
            pa = np.zeros((n, 3), np.float32)
            fin = self.push.push_step(act, tip, blk, goal, pa)
            fin &= ~(off_line & outside) & (self.push.detour == 0)
            a[act] = pa[act]
            self.cur[fin] += 1
            self.push.phase[fin] = 0
            a[pushing & (self.cur >= nb)] = 0.0


class ChestPickAndPlacePolicy(Policy):
    """chest_pick_and_place (kuka_multi_step_envs.py:230-383, chest_up_sliding_door.urdf): the closed fingers push the
    lid's handle along -x until the lid latches open, then every block is picked, carried over the chest and dropped in.
    phases: 10 fly in front of the handle - 11 push - 12 rise; then the grasp-and-carry machine per block."""
    action_dim = 4

    def __init__(self, n, num_block=1):
        super().__init__(n)
        self.nb = num_block
        self.core = _PickPlaceCore(n)
        self.phase[:] = 10
        self.cur = np.zeros(n, np.int32)

    def reset(self, mask=None):
        m = slice(None) if mask is None else np.asarray(mask, bool)
        self.phase[m] = 10
        self.count[m] = 0
        self.cur[m] = 0

    def _act(self, obs, a):
        n, nb = self.n, self.nb
        rows = np.arange(n)
        tip, _, blocks = multi_block_views(obs, nb)
        door = obs['achieved_goal'][:, 0].astype(np.float64)
        ph = self.phase
        e = np.zeros((n, 3))
        grip = np.ones(n)
        m = ph == 10
        front = np.array([-0.53, 0.065, 0.24])
        e[m] = (front - tip)[m]
        self._set_phase(m & (np.abs(front - tip).max(1) < 0.006), 11)
        m = ph == 11
        e[m] = np.array([-0.01, 0.0, 0.0])
        e[m, 1] = 0.065 - tip[m, 1]
        e[m, 2] = 0.24 - tip[m, 2]
        self._set_phase(m & (door > 0.097), 12)
        m = ph == 12
        e[m] = np.array([0.005, 0.0, 0.01])
        self._set_phase(m & (tip[:, 2] > 0.30), 0)
        lid = ph >= 10
        a[lid, :3] = np.clip(e[lid] / STEP, -1, 1)
        a[lid, 3] = grip[lid]
        work = ~lid & (self.cur < nb)
        if work.any():
            k = np.minimum(self.cur, nb - 1)
            blk = blocks[rows, k]
            inside = (np.abs(blk[:, 0] - CHEST_CENTRE[0]) < 0.045) & (np.abs(blk[:, 1]) < 0.055) & (blk[:, 2] < 0.2)
            nxt = work & inside & (ph == 0)
            self.cur[nxt] += 1
            work = work & (self.cur < nb)
            k = np.minimum(self.cur, nb - 1)
            blk = blocks[rows, k]
            target = np.tile(CHEST_CENTRE, (n, 1))
            over = target.copy()
            over[:, 2] = 0.30
            travel = np.full(n, 0.31)
            done = self.core.step(self, work, tip, blk, target, travel, a, drop_from=over, clamp_x=False,
                                  fly_z=np.full(n, 0.24))
            self._set_phase(done, 0)
        idle = ~lid & (self.cur >= nb)
        a[idle, :3] = 0.0
        a[idle, 3] = -1.0


class RearrangePolicy(Policy):
    """block_rearrange (kuka_multi_step_envs.py:151-227): closed gripper, every block is pushed to its target slot on the
    table with the axis-aligned legs of PushPolicy, one block after the other (blocks already within `skip` of their slot
    are left alone; the others are obstacles nobody plans around -- with two blocks that rarely matters)."""

    def __init__(self, n, num_block=2):
        super().__init__(n)
        self.nb = num_block
        self.push = PushPolicy(n, skip=0.015)
        self.cur = np.zeros(n, np.int32)

    def reset(self, mask=None):
        super().reset(mask)
        self.cur[slice(None) if mask is None else np.asarray(mask, bool)] = 0
        self.push.reset(mask)

    def _act(self, obs, a):
        n, nb = self.n, self.nb
        rows = np.arange(n)
        tip, _, blocks = multi_block_views(obs, nb)
        targets = obs['desired_goal'].astype(np.float64)[:, :3 * nb].reshape(n, nb, 3)
        err = np.abs(targets[:, :, :2] - blocks[:, :, :2]).max(2)
        todo = err >= self.push.skip
        # keep working on the current block until it is there, then the next one that is not
        k = np.minimum(self.cur, nb - 1)
        switch = (~todo[rows, k]) & (self.push.phase == 0) | (self.cur >= nb)
        nxt = np.where(todo.any(1), np.argmax(todo, axis=1), nb)
        self.cur[switch] = nxt[switch]
        act = self.cur < nb
        k = np.minimum(self.cur, nb - 1)
        pa = np.zeros((n, 3), np.float32)
        fin = self.push.push_step(act, tip, blocks[rows, k], targets[rows, k], pa)
        a[act] = pa[act]
        self.push.phase[fin] = 0
        a[~act] = 0.0


def make_policy(task, n, **kw):
    if task == 'slide':      # the puck (radius 0.03, kuka_single_step_envs.py:49-59) needs a longer stand-off than the cube
        kw.setdefault('standoff', 0.06)
    return {'reach': ReachPolicy, 'push': PushPolicy, 'pick_and_place': PickAndPlacePolicy, 'slide': PushPolicy,
            'block_stack': StackPolicy, 'block_rearrange': RearrangePolicy, 'chest_push': ChestPushPolicy,
            'chest_pick_and_place': ChestPickAndPlacePolicy}[task](n, **kw)


# ---- running a policy on either backend -------------------------------------------------------------------------------
def step_env(env, a):
    """(obs, reward, done, goal_achieved) from the oracle wrapper or the HIP env."""
    out = env.step(a)
    ok = out[3]['goal_achieved'] if isinstance(out[3], dict) else out[3]
    return out[0], out[1], out[2], np.asarray(ok, bool)


def rollout(env, policy, T, obs, on_step=None):
    """Run `policy` for T steps from observation `obs`; returns (final obs, success at the last step, ever succeeded)."""
    ever = np.zeros(policy.n, bool)
    ok = ever
    for t in range(T):
        a = policy.act(obs)
        obs, r, d, ok = step_env(env, a)
        ever |= ok
        if on_step is not None:
            on_step(t, a, obs, r, ok)
    return obs, ok, ever


if __name__ == '__main__':
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import oracle_lib
    task = sys.argv[1] if len(sys.argv) > 1 else 'pick_and_place'
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    kw = {'num_block': int(sys.argv[4]) if len(sys.argv) > 4 else 2} if task.startswith(('block', 'chest')) else {}
    env = oracle_lib.OracleEnv(task, N, seed_base=0, seed_stride=1, threads=oracle_lib.usable_threads(), max_episode_steps=T, **kw)
    env.reset()
    obs = env.reset()
    pol = make_policy(task, N, **kw)

    def show(t, a, o, r, ok):
        if t % 5 == 4 or t == T - 1:
            print(t, 'phase hist', np.bincount(pol.phase, minlength=9), 'success %.3f' % ok.mean())
    obs, ok, ever = rollout(env, pol, T, obs, show)
    print('success at the end %.3f, ever %.3f; failed envs: %s' % (ok.mean(), ever.mean(), np.nonzero(~ok)[0][:20]))
