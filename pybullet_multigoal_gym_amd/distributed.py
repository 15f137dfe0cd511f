"""Multi-GPU sharding of the batched env: one process per GPU, envs split in
contiguous shards, and ONE collective per batched step -- an RCCL all-gather
(inside the C-ABI library) of each rank's packed observation shard.

Envs never interact (one Bullet world per env in the reference,
P/envs/base_envs/base_env.py:203-220), so nothing else is exchanged.
torch.distributed is used only to rendezvous (broadcast the RCCL unique id)
and, on CPU-only test rigs, as the gloo stand-in for the gather.
"""
import numpy as np


def shard_bounds(total_envs, world_size, rank):
    """Contiguous shard [start, stop) of `rank`: env i lives on rank i // ceil(total / world)."""
    per = -(-total_envs // world_size)
    start = min(rank * per, total_envs)
    return start, min(start + per, total_envs)


def make_sharded_env(make_env, total_envs, world_size, rank, **kw):
    """This rank's shard of a `total_envs`-env job; env seeds are those of the global index."""
    start, stop = shard_bounds(total_envs, world_size, rank)
    if stop <= start:
        raise ValueError('rank %d of %d has no envs (total %d)' % (rank, world_size, total_envs))
    return make_env(num_envs=stop - start, env_index_offset=start, **kw)


def init_rccl(env, rank, world_size, dist=None):
    """Create the RCCL communicator of `env`'s handle; the unique id travels over torch.distributed."""
    if dist is None:
        import torch.distributed as dist
    uid = [env.handle.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    env.handle.comm_init(rank, world_size, uid[0])


def pack_outputs(env, obs, reward, done, goal_achieved):
    """Host-side row layout of PMG_BUF_PACKED: obs | policy | ag | dg | reward | goal_achieved | done."""
    n = env.num_envs
    cols = [np.asarray(obs[k], np.float32).reshape(n, -1) for k in ('observation', 'policy_state', 'achieved_goal', 'desired_goal')]
    cols += [np.asarray(reward, np.float32).reshape(n, 1), np.asarray(goal_achieved, np.float32).reshape(n, 1),
             np.asarray(done, np.float32).reshape(n, 1)]
    return np.ascontiguousarray(np.concatenate(cols, axis=1))


def unpack_outputs(dims, packed):
    """Inverse of pack_outputs for a stacked [N_total, packed_dim] array."""
    o, p, g = dims.observation_dim, dims.policy_state_dim, dims.goal_dim
    c = np.cumsum([0, o, p, g, g, 1, 1, 1])
    obs = {'observation': packed[:, c[0]:c[1]], 'policy_state': packed[:, c[1]:c[2]],
           'achieved_goal': packed[:, c[2]:c[3]], 'desired_goal': packed[:, c[3]:c[4]]}
    return obs, packed[:, c[4]], packed[:, c[6]] != 0, packed[:, c[5]] != 0


def allgather_host(packed_local, dist=None):
    """Gather equally-sized packed shards through torch.distributed (gloo on CPU rigs)."""
    import torch
    if dist is None:
        import torch.distributed as dist
    world = dist.get_world_size()
    t = torch.from_numpy(packed_local)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return np.concatenate([x.numpy() for x in out], axis=0)
