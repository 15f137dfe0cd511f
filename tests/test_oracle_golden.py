"""CPU tier: the oracle against the vectors that do not come from the reference's code -- numpy's RandomState stream, the
FK known answer, the model constants (tools/gen_golden.py, tools/gen_model.py).  What the tasks DRAW and return is checked
against the reference's own code in tests/test_reference_golden.py (tests/golden/ref_*.json)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return json.load(open(os.path.join(G, name)))


def test_rng_stream_matches_numpy_randomstate(built):
    for seed, v in load('rng.json').items():
        d, perm = O.rng_probe(int(seed), 64, 5)
        assert d.tolist() == v['random_sample_64'], seed        # bit-exact float64 draws
        assert perm.tolist() == v['then_shuffle_arange5'], seed


def test_fk_known_answer(built):
    fk = load('fk.json')
    p, R = O.fk_tip(fk['rest_pose'])
    assert np.abs(p - fk['tip_position']).max() < fk['tip_position_tol']
    assert np.abs(R - np.array(fk['tip_rotation'])).max() < fk['tip_rotation_tol']


def test_model_header_matches_fixture_and_reference_assets(built):
    """include/pmg_model.h is generated data: it must agree with tests/golden/model.json, and --
    when the reference tree is mounted (this container only) -- with a fresh parse of the URDFs."""
    import re
    root = os.path.dirname(G)
    hdr = open(os.path.join(root, '..', 'include', 'pmg_model.h')).read()
    model = load('model.json')
    m = re.search(r'#define PMG_MB_MASS \{([^}]*)\}', hdr)
    masses = [float(x) for x in m.group(1).split(',')]
    want = [sum(s['mass'] for s in model['sub'][link]) for link in model['mov_link']]
    assert np.allclose(masses, want, atol=1e-12)
    assert abs(sum(masses) + 5.0 - 24.41) < 0.3          # ~24.4 kg arm + gripper (link_0 = 5 kg is static)
    assert model['row_order'][:9] == [11, 12, 9, 10, 13, 16, 17, 14, 15]
    if os.path.exists('/root/reference/pybullet_multigoal_gym/assets'):
        import subprocess, sys, tempfile, shutil
        keep = open(os.path.join(root, '..', 'include', 'pmg_model.h')).read()
        subprocess.check_call([sys.executable, os.path.join(root, '..', 'tools', 'gen_model.py')], stdout=subprocess.DEVNULL)
        assert open(os.path.join(root, '..', 'include', 'pmg_model.h')).read() == keep, 'pmg_model.h is stale'
