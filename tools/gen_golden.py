#!/usr/bin/env python3
"""Generate tests/golden/rng.json and fk.json.

Build container only.  Two vectors that do not come out of the reference's code:
  * rng.json: numpy's legacy RandomState (bit-stable MT19937 stream) seeded the way gym 0.17.3's seeding.np_random does
    (sha512(str(seed))[:8] -> little-endian uint32 words): raw float64 draws and a shuffle, for the oracle's and the
    device's MT19937 / init_by_array / shuffle restatements;
  * fk.json: the analytic FK known answer of SURVEY.md section 7-1.
Everything about WHAT the tasks draw (object / goal sampling, stack orders, curricula, sub-goals) is recorded from the
reference's own code by tools/gen_reference_fixtures.py (tests/golden/ref_*.json); no sampling rule is re-typed here.
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
    fk = {'rest_pose': [0, -0.5592432, 0, 1.733180, 0, -0.8501557, 0, 0.035, 0.035],
          'tip_position': [-0.522923, 0.0, 0.250773], 'tip_position_tol': 1e-5,
          'tip_rotation': [[-1, 0, 0], [0, 1, 0], [0, 0, -1]], 'tip_rotation_tol': 1.1e-3,
          'source': 'SURVEY.md section 7-1 (computed independently in the survey session)'}
    json.dump(fk, open(os.path.join(OUT, 'fk.json'), 'w'), indent=0)
    print('wrote rng.json, fk.json')


if __name__ == '__main__':
    main()
