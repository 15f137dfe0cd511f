#!/usr/bin/env python3
"""Teacher-forced parity statistics (GPU box): every step the device is re-synchronised to the float64 oracle's state
(pmg_set_state), both take the same action, and the SINGLE-STEP deviation (100 substeps) is recorded -- the way to test
chaotic contact code without the chaos.  Prints the max / p99.9 / p99 per quantity; tests/test_gpu_tail_parity.py holds the
bars derived from these numbers.   tools/teacher_forced.py <task> [N] [T] [f32] [scripted]   (f32: the float32 ORACLE
instead of the device, for the precision floor of the same algorithm; scripted: the actions come from the task-solving
controllers of tools/scripted_policies.py, fed with the float64 oracle's observations -- grasp, lift, carry, push, stack,
open-the-door-and-drop -- instead of the random policy, so the gripper-on-object solver paths are the ones compared)"""
import json
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle_lib  # noqa: E402


def fields(task, nb):
    chest = task.startswith('chest')
    f = {'q_arm': slice(0, 7), 'q_finger': slice(7, 9), 'qd': slice(9, 18)}
    if chest:
        f['door_q'] = slice(48, 49)
        f['door_qd'] = slice(49, 50)
    return f


def block_views(state, nb):
    b = state[:, 64:64 + 13 * nb].reshape(len(state), nb, 13)
    return b[..., 0:3], b[..., 3:7], b[..., 7:10], b[..., 10:13]


def physical_columns(task, nb):
    """Columns of a state row that hold continuous physical state (joint angles / velocities, block poses / velocities, the
    chest door): what a float32 representation rounds -- not the targets, counters, flags and levels beside them."""
    cols = list(range(0, 18))
    if task.startswith('chest'):
        cols += [48, 49]
    cols += list(range(64, 64 + 13 * nb))
    return np.array(cols)


def ulp_perturbed(state, cols, rs):
    """The state with every physical entry moved to a NEIGHBOURING float32 (one ulp up or down at random): the size of the
    noise any float32 code has in its state, in a form that can be fed to the float64 oracle."""
    out = np.array(state, np.float64)
    v = out[:, cols].astype(np.float32)
    up = rs.random_sample(v.shape) < 0.5
    v = np.where(up, np.nextafter(v, np.float32(np.inf)), np.nextafter(v, np.float32(-np.inf)))
    out[:, cols] = v.astype(np.float64)
    return out


GROSS = 1e-3      # a single step beyond this is a bifurcation (a contact made or missed), not rounding


def run(task, N=1024, T=50, kw=None, device=True, threads=16, seed=12345, lib=None, policy=None, keep_schedule=False, perturb=0):
    """policy: None = uniform random actions, else an object with act(obs) (tools/scripted_policies.py) driven by the
    float64 oracle's observations.  keep_schedule: also count, per step, the envs on the device's launch lists.
    perturb = K > 0: the CHAOS FLOOR -- K more float64 oracles take every step from the same state with its physical
    entries moved by one float32 ulp (ulp_perturbed) and round their state to float32 after every one of the 100 substeps
    (the oracle's `state_f32_per_substep` switch; all arithmetic stays float64); their deviation from the plain float64
    step is what ANY implementation that keeps its state in float32 must expect, whatever its arithmetic.  out['chaos'] then holds, per quantity, the counts of
    steps beyond 1e-3 of each perturbed oracle, and for the compared implementation the number of its gross steps that
    are NOT gross in any perturbed oracle either ('off_floor')."""
    kw = dict(kw or {})
    nb = 0 if task == 'reach' else (kw.get('num_block', 4) if task.startswith(('block', 'chest')) else 1)
    o64 = oracle_lib.OracleEnv(task, N, seed_base=0, seed_stride=1, threads=threads, **kw)
    o64.reset()
    obs64 = o64.reset()
    sched = {}
    pert = []
    for k in range(perturb):
        e = oracle_lib.OracleEnv(task, N, seed_base=0, seed_stride=1, threads=threads, **kw)
        e.reset(); e.reset()
        pert.append(e)
    prs = np.random.RandomState(777)
    pcols = physical_columns(task, nb)
    pworst = [dict() for _ in pert]
    gross_dev = {}     # quantity -> list of [N] bool per step
    gross_any = {}
    if device:
        import pybullet_multigoal_gym_amd as pmg
        if lib is None and os.environ.get('PMG_TF_LIB'):           # (kernel A/B: another build of the library, tools/ab.sh style)
            from pybullet_multigoal_gym_amd._lib import PmgLibrary
            lib = PmgLibrary(os.environ['PMG_TF_LIB'])
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            dev = pmg.make_env(task=task, num_envs=N, seed=0, seed_stride=1, _library=lib, **kw)
        dev.reset()
    else:
        o32 = oracle_lib.OracleEnv(task, N, seed_base=0, seed_stride=1, threads=threads, f32=True, **kw)
        o32.reset()
        o32.reset()
    rs = np.random.RandomState(seed)
    A = o64.dims.action_dim
    thr = kw.get('distance_threshold', 0.05)
    worst = {}
    flag_mismatch = 0
    flag_total = 0

    def note(name, err, into=None):
        err = np.asarray(err, np.float64).reshape(len(err), -1).max(1)
        w = (worst if into is None else into).setdefault(name, [])
        w.append(err)
        return err

    for t in range(T):
        a = rs.uniform(-1, 1, (N, A)).astype(np.float32) if policy is None else policy.act(obs64)
        s0 = o64.get_state()
        if device:
            dev.set_state(s0)
            od, rd, dd, info = dev.step(a)
            okd = info['goal_achieved']
            sd = dev.get_state()
            if keep_schedule:
                for key, v in dev.handle.schedule().items():
                    sched[key] = sched.get(key, 0) + len(v)
        else:
            o32.set_state(s0)
            od, rd, dd, okd = o32.step(a)
            sd = o32.get_state()
        oo, ro, do, oko = o64.step(a)
        obs64 = oo
        so = o64.get_state()
        tsl = slice(0, 3) if not kw.get('joint_control') else slice(7, 10)

        def compare(sx, ox, into):
            e = {}
            for name, sl in fields(task, nb).items():
                e[name] = note(name, np.abs(sx[:, sl] - so[:, sl]), into)
            if nb:
                pd, qd_, vd, wd = block_views(sx, nb)
                po, qo, vo, wo = block_views(so, nb)
                e['block_pos'] = note('block_pos', np.abs(pd - po), into)
                note('block_quat', np.minimum(np.abs(qd_ - qo), np.abs(qd_ + qo)), into)
                note('block_vel', np.abs(vd - vo), into)
                note('block_omega', np.abs(wd - wo), into)
            e['tip_pos'] = note('tip_pos', np.abs(ox['observation'][:, tsl] - oo['observation'][:, tsl]), into)
            return e
        e_dev = compare(sd, od, None)
        if pert:
            e_any = {}
            for k, pe in enumerate(pert):
                pe.set_state(ulp_perturbed(s0, pcols, prs))
                oracle_lib.set_prior('state_f32_per_substep', 1.0)      # (process-wide switch: on for this call only)
                try:
                    op, _, _, _ = pe.step(a)
                finally:
                    oracle_lib.set_prior('state_f32_per_substep', 0.0)
                for name, v in compare(pe.get_state(), op, pworst[k]).items():
                    e_any[name] = np.maximum(e_any.get(name, 0.0), v)
            for name in e_dev:
                gross_dev.setdefault(name, []).append(e_dev[name] > GROSS)
                gross_any.setdefault(name, []).append(e_any[name] > GROSS)
        dist = np.linalg.norm(oo['achieved_goal'].astype(np.float64) - oo['desired_goal'], axis=1)
        # a flag is a function of the achieved goal: it may differ where this step's own position error (barred and counted
        # separately) reaches the threshold -- "clear" = further from the threshold than that error + 1e-4
        moved = np.linalg.norm(od['achieved_goal'].astype(np.float64) - oo['achieved_goal'], axis=1)
        clear = np.abs(dist - thr) > 1e-4 + moved
        flag_total += int(clear.sum())
        flag_mismatch += int((np.asarray(okd)[clear] != np.asarray(oko)[clear]).sum())
    out = {'task': task, 'kw': kw, 'N': N, 'T': T, 'who': 'device' if device else 'float32 oracle', 'flags_off_threshold': flag_total,
           'flag_mismatches': flag_mismatch, 'stats': {}, 'policy': 'random' if policy is None else type(policy).__name__,
           'final_success': float(np.mean(oko)), 'schedule_env_steps': sched}
    for name, w in worst.items():
        e = np.concatenate(w)
        out['stats'][name] = {'max': float(e.max()), 'p99.9': float(np.percentile(e, 99.9)), 'p99': float(np.percentile(e, 99)),
                              'p50': float(np.percentile(e, 50)), 'n_gt_1e-4': int((e > 1e-4).sum()), 'n_gt_1e-3': int((e > 1e-3).sum()), 'n': int(e.size)}
    if pert:
        ch = {}
        for name in gross_dev:
            gd, ga = np.array(gross_dev[name]), np.array(gross_any[name])
            per_k = [int((np.concatenate(pw[name]) > GROSS).sum()) for pw in pworst]
            ch[name] = {'floor_per_perturbed_oracle': per_k, 'floor': float(np.mean(per_k)), 'gross': int(gd.sum()),
                        'off_floor': int((gd & ~ga).sum()),
                        'perturbed_p99': float(np.mean([np.percentile(np.concatenate(pw[name]), 99) for pw in pworst]))}
        out['chaos'] = ch
        out['perturbed_oracles'] = perturb
    for e in pert:
        e.close()
    return out


if __name__ == '__main__':
    task = sys.argv[1] if len(sys.argv) > 1 else 'push'
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    dev = 'f32' not in sys.argv[4:]
    kw = {'num_block': {'block_stack': 4, 'block_rearrange': 3, 'chest_push': 2, 'chest_pick_and_place': 2}.get(task, 4)} if task.startswith(('block', 'chest')) else {}
    pol = None
    if 'scripted' in sys.argv[4:]:
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import scripted_policies
        if task.startswith('chest'):
            kw['num_block'] = 1
        if task == 'block_rearrange':
            kw['num_block'] = 2
        kw['max_episode_steps'] = T
        pol = scripted_policies.make_policy(task, N, **({'num_block': kw['num_block']} if 'num_block' in kw else {}))
    r = run(task, N, T, kw, device=dev, policy=pol, keep_schedule=dev, perturb=2 if 'chaos' in sys.argv[4:] else 0)
    print(json.dumps(r))
