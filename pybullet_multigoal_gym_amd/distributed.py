"""Multi-GPU sharding of the batched env: one process per GPU, envs split in equal contiguous shards, and ONE collective
per batched step -- an RCCL all-gather (inside the C-ABI library) of each rank's packed observation shard.

Envs never interact (one Bullet world per env in the reference, P/envs/base_envs/base_env.py:203-220), so nothing else is
exchanged.  The host side needs three small things from its peers -- the 128-byte RCCL unique id of rank 0, a barrier
and a max over ranks of a wall time -- and gets them from `Rendezvous`, a few dozen lines of stdlib TCP: no PyTorch, no
MPI.  (`torch.distributed.run` may still LAUNCH the ranks: only RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT are read.)
"""
import os
import pickle
import socket
import struct
import time

import numpy as np

RDV_PORT_OFFSET = 17   # torch.distributed.run keeps its own store on MASTER_PORT: ours listens this far above it


def shard_bounds(total_envs, world_size, rank):
    """Contiguous shard [start, stop) of `rank`.  Shards are EQUAL: the all-gather of the packed rows (ncclAllGather,
    and its host stand-in) sends the same count from every rank, so a job whose env count does not divide by the world
    size is refused instead of hanging or corrupting the gather."""
    if total_envs % world_size != 0:
        raise ValueError('total_envs=%d is not a multiple of world_size=%d: the packed all-gather needs equal shards '
                         '(pad the job to %d envs)' % (total_envs, world_size, -(-total_envs // world_size) * world_size))
    per = total_envs // world_size
    return rank * per, (rank + 1) * per


def make_sharded_env(make_env, total_envs, world_size, rank, **kw):
    """This rank's shard of a `total_envs`-env job; env seeds are those of the global index."""
    start, stop = shard_bounds(total_envs, world_size, rank)
    return make_env(num_envs=stop - start, env_index_offset=start, **kw)


def _send(sock, obj):
    blob = pickle.dumps(obj, protocol=4)
    sock.sendall(struct.pack('<Q', len(blob)) + blob)


def _recv(sock):
    hdr = b''
    while len(hdr) < 8:
        chunk = sock.recv(8 - len(hdr))
        if not chunk:
            raise ConnectionError('rendezvous peer closed the connection')
        hdr += chunk
    n, = struct.unpack('<Q', hdr)
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(1 << 20, n - len(buf)))
        if not chunk:
            raise ConnectionError('rendezvous peer closed the connection')
        buf += chunk
    return pickle.loads(bytes(buf))


class Rendezvous:
    """Rank 0 listens, everybody else connects; every collective is "all ranks send one object to rank 0, rank 0 sends
    the list back" -- the ranks call them in the same order, so no threads and no tags are needed.  Control plane only:
    the data path of the job is the RCCL all-gather inside the library."""

    def __init__(self, rank, world_size, addr=None, port=None, timeout=180.0):
        self.rank, self.world = int(rank), int(world_size)
        addr = addr or os.environ.get('MASTER_ADDR', '127.0.0.1')
        if port is None:
            port = int(os.environ['PMG_RDV_PORT']) if 'PMG_RDV_PORT' in os.environ else int(os.environ.get('MASTER_PORT', '29400')) + RDV_PORT_OFFSET
        self.peers = []
        self.sock = None
        if self.world == 1:
            return
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr if addr not in ('localhost',) else '127.0.0.1', port))
            srv.listen(self.world)
            srv.settimeout(timeout)
            by_rank = {}
            while len(by_rank) < self.world - 1:
                conn, _ = srv.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(timeout)
                by_rank[_recv(conn)] = conn
            srv.close()
            self.peers = [by_rank[r] for r in range(1, self.world)]
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    s = socket.create_connection((addr, port), timeout=5.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(timeout)
            _send(s, self.rank)
            self.sock = s

    @classmethod
    def from_env(cls, **kw):
        return cls(int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), **kw)

    def allgather(self, obj):
        """[obj of rank 0, obj of rank 1, ...] on every rank."""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            out = [obj] + [_recv(c) for c in self.peers]
            for c in self.peers:
                _send(c, out)
            return out
        _send(self.sock, obj)
        return _recv(self.sock)

    def broadcast(self, obj, src=0):
        return self.allgather(obj if self.rank == src else None)[src]

    def barrier(self):
        self.allgather(None)

    def max(self, x):
        return max(self.allgather(x))

    def min(self, x):
        return min(self.allgather(x))

    def close(self):
        for c in self.peers:
            c.close()
        if self.sock is not None:
            self.sock.close()
        self.peers, self.sock = [], None


def init_rccl(env, rdv, _force_fail=False):
    """Create the RCCL communicator of `env`'s handle; rank 0's unique id travels over the rendezvous.  Returns True when
    EVERY rank has its communicator (all ranks get the same answer, so they can take a fallback together); a failure is
    reported on stderr by the rank it happened on, never swallowed."""
    import sys
    uid = None
    if rdv.rank == 0 and not _force_fail:
        try:
            uid = env.handle.comm_unique_id()
        except Exception as ex:   # noqa: BLE001
            print('rank 0: no RCCL unique id (%s)' % ex, file=sys.stderr, flush=True)
    uid = rdv.broadcast(uid)
    ok = 0
    if uid is not None:
        try:
            env.handle.comm_init(rdv.rank, rdv.world, uid)
            ok = 1
        except Exception as ex:   # noqa: BLE001
            print('rank %d: RCCL communicator failed (%s)' % (rdv.rank, ex), file=sys.stderr, flush=True)
    elif _force_fail and rdv.rank == 0:
        print('rank 0: RCCL communicator failure forced by the caller', file=sys.stderr, flush=True)
    return rdv.min(ok) == 1


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


def allgather_host(packed_local, rdv):
    """Host fallback of the all-gather (PCIe + TCP, equal shards): used when no RCCL communicator could be created."""
    parts = rdv.allgather(np.ascontiguousarray(packed_local))
    if any(p.shape != parts[0].shape for p in parts):
        raise ValueError('unequal shards in the packed all-gather: %s' % [p.shape for p in parts])
    return np.concatenate(parts, axis=0)
