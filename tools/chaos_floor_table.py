#!/usr/bin/env python3
"""GPU box: the single-step outlier counts of the DEVICE next to the chaos floor (float64 oracle with its state moved by one
float32 ulp and rounded to float32 after every substep, tools/teacher_forced.py) and next to the float32 build of the oracle,
per task, under the scripted task-solving policies and under the random policy.  One JSON line per (task, policy) and a
table; profiles/r04_chaos_floor.{jsonl,txt} are this script's output.
    tools/chaos_floor_table.py [scripted|random|all] [nof32] > gpurun_out/chaos_floor.jsonl   (nof32: without the float32-oracle column)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import oracle_lib  # noqa: E402
import scripted_policies as SP  # noqa: E402
import teacher_forced as TF  # noqa: E402

SCRIPTED = {'pick_and_place': ({}, 60), 'push': ({}, 300), 'slide': ({}, 60), 'block_stack': ({'num_block': 4}, 340),
            'block_rearrange': ({'num_block': 2}, 400), 'chest_push': ({'num_block': 1}, 360), 'chest_pick_and_place': ({'num_block': 1}, 100)}
RANDOM = {'reach': {}, 'push': {}, 'pick_and_place': {}, 'slide': {}, 'block_stack': {'num_block': 4}, 'block_rearrange': {'num_block': 3},
          'chest_push': {'num_block': 2}, 'chest_pick_and_place': {'num_block': 2}}
QUANT = ('tip_pos', 'block_pos', 'q_arm', 'door_q')


def one(task, kw, N, T, policy_name):
    th = oracle_lib.usable_threads()
    pkw = {'num_block': kw['num_block']} if 'num_block' in kw else {}
    mk = (lambda: SP.make_policy(task, N, **pkw)) if policy_name == 'scripted' else (lambda: None)
    kw = dict(kw, max_episode_steps=T) if policy_name == 'scripted' else dict(kw)
    dev = TF.run(task, N, T, kw, device=True, threads=th, policy=mk(), perturb=2)
    f32 = TF.run(task, N, T, kw, device=False, threads=th, policy=mk()) if 'nof32' not in sys.argv[2:] else None
    row = {'task': task, 'policy': policy_name, 'N': N, 'T': T, 'kw': kw, 'quantities': {}}
    for q in QUANT:
        if q not in dev['stats']:
            continue
        d, c = dev['stats'][q], dev['chaos'][q]
        f = f32['stats'][q] if f32 is not None else {'n_gt_1e-3': -1, 'p99': float('nan')}
        row['quantities'][q] = {'env_steps': d['n'], 'device_gt_1e-3': d['n_gt_1e-3'], 'device_off_floor': c['off_floor'], 'floor_gt_1e-3': c['floor_per_perturbed_oracle'],
                                'f32_oracle_gt_1e-3': f['n_gt_1e-3'], 'device_p99': d['p99'], 'floor_p99': c['perturbed_p99'], 'f32_oracle_p99': f['p99'],
                                'device_max': d['max'], 'device_p50': d['p50']}
    print(json.dumps(row), flush=True)
    return row


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    rows = []
    if which in ('scripted', 'all'):
        for task, (kw, T) in SCRIPTED.items():
            rows.append(one(task, kw, 256, T, 'scripted'))
    if which in ('random', 'all'):
        for task, kw in RANDOM.items():
            rows.append(one(task, kw, 1024, 50, 'random'))
    out = sys.stderr
    print('%-22s %-9s %-10s %9s | steps beyond 1e-3: %8s %14s %8s | p99: %9s %9s %9s' % ('task', 'policy', 'quantity', 'env-steps', 'device', 'floor (2 runs)', 'f32 orc', 'device', 'floor', 'f32 orc'), file=out)
    for r in rows:
        for q, v in r['quantities'].items():
            print('%-22s %-9s %-10s %9d | %27d %14s %8d | %14.1e %9.1e %9.1e' % (r['task'], r['policy'], q, v['env_steps'], v['device_gt_1e-3'], v['floor_gt_1e-3'], v['f32_oracle_gt_1e-3'],
                                                                              v['device_p99'], v['floor_p99'], v['f32_oracle_p99']), file=out)


if __name__ == '__main__':
    main()
