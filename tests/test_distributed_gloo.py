"""CPU tier: the N>1 path with world_size 2 over gloo.  Each rank steps its shard (on the emulator
build of the C ABI), the packed shards are all-gathered, and rank 0 checks the stacked result
against ONE un-sharded oracle env of the full size: same seeds per global env index, same order."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, total, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch.distributed as dist
    import pybullet_multigoal_gym_amd as pmg
    from pybullet_multigoal_gym_amd import distributed as D
    from pybullet_multigoal_gym_amd._lib import PmgLibrary
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    emu = PmgLibrary(os.path.join(ROOT, 'tests', 'emu', 'libpmg_emu.so'))
    env = D.make_sharded_env(pmg.make_env, total, world, rank, task='reach', seed=11, seed_stride=1, _library=emu)
    start, stop = D.shard_bounds(total, world, rank)
    actions = np.random.RandomState(7).uniform(-1, 1, (total, 3)).astype(np.float32)
    env.reset()
    obs, r, d, info = env.step(actions[start:stop])
    packed = D.allgather_host(D.pack_outputs(env, obs, r, d, info['goal_achieved']))
    if rank == 0:
        np.save(ret, packed)
    env.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_equal_one_unsharded_env(built, tmp_path):
    import torch.multiprocessing as mp
    import oracle_lib as O
    from pybullet_multigoal_gym_amd import distributed as D
    total, world = 4, 2
    assert [D.shard_bounds(5, 2, r) for r in (0, 1)] == [(0, 3), (3, 5)]
    ret = str(tmp_path / 'packed.npy')
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, total, ret), nprocs=world, join=True)
    packed = np.load(ret)
    ora = O.OracleEnv('reach', total, seed_base=11, seed_stride=1)
    ora.reset()
    ora.reset()
    actions = np.random.RandomState(7).uniform(-1, 1, (total, 3)).astype(np.float32)
    oo, ro, do, oko = ora.step(actions)
    obs, r, done, ok = D.unpack_outputs(ora.dims, packed)
    assert packed.shape == (total, ora.dims.packed_dim)
    assert np.array_equal(obs['desired_goal'], oo['desired_goal'])   # per-env seeds follow the GLOBAL index
    assert np.abs(obs['observation'] - oo['observation']).max() < 2e-5
    assert np.array_equal(r, ro) and np.array_equal(done, do)
