#!/usr/bin/env python3
"""GPU box: the chaos-floor comparison of tools/chaos_floor_table.py at four times the sample -- 4096 envs x 50 random-policy steps
per task (204 800 single steps) for the tasks that were above the floor until round 5 -- so that the ratios device / floor rest on
more than a handful of events.   tools/chaos_floor_large.py > gpurun_out/chaos_floor_large.jsonl"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import oracle_lib  # noqa: E402
import teacher_forced as TF  # noqa: E402

TASKS = {'slide': {}, 'chest_push': {'num_block': 2}, 'chest_pick_and_place': {'num_block': 2}, 'push': {}, 'block_stack': {'num_block': 4}}
only = sys.argv[1:]
for task, kw in TASKS.items():
    if only and task not in only:
        continue
    r = TF.run(task, 4096, 50, kw, device=True, threads=oracle_lib.usable_threads(), perturb=2)
    row = {'task': task, 'N': 4096, 'T': 50, 'kw': kw, 'quantities': {}}
    for q in ('tip_pos', 'block_pos', 'q_arm', 'door_q'):
        if q in r['stats']:
            d, c = r['stats'][q], r['chaos'][q]
            row['quantities'][q] = {'env_steps': d['n'], 'device_gt_1e-3': d['n_gt_1e-3'], 'floor_gt_1e-3': c['floor_per_perturbed_oracle'], 'device_off_floor': c['off_floor'],
                                    'device_p99': d['p99'], 'floor_p99': c['perturbed_p99'], 'device_p99.9': d['p99.9'], 'device_max': d['max']}
            print('%-22s %-10s %7d steps | beyond 1e-3: device %4d floor %s | p99 %.1e / %.1e | p99.9 %.1e' % (task, q, d['n'], d['n_gt_1e-3'], c['floor_per_perturbed_oracle'], d['p99'], c['perturbed_p99'], d['p99.9']), file=sys.stderr, flush=True)
    print(json.dumps(row), flush=True)
