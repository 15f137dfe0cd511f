"""Multi-GPU sharding of the batched env: one process per GPU, envs split in equal contiguous shards, and ONE collective
per batched step -- an RCCL all-gather (inside the C-ABI library) of each rank's packed observation shard.

Envs never interact (one Bullet world per env in the reference, P/envs/base_envs/base_env.py:203-220), so nothing else is
exchanged.  The host side needs three small things from its peers -- the 128-byte RCCL unique id of rank 0, a barrier
and a max over ranks of a wall time -- and gets them from `Rendezvous`, stdlib TCP with a fixed typed framing (no pickle: rank 0 never
deserialises objects from the network), a job token in the hello and rank validation: no PyTorch, no MPI.  (`torch.distributed.run` may still LAUNCH the ranks: only RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT are read.)
"""
import hmac
import os
import sys
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


# ---- wire format -----------------------------------------------------------------------------------------------------
# The payloads are None, ints, floats, the 128-byte RCCL id and float arrays -- nothing needs pickle, and unpickling what
# an unauthenticated TCP peer sends would hand it code execution on rank 0.  A message is
#   magic 'PMG1' | type tag u8 | payload length u32 (capped) | payload
# with the payload itself typed: N none, I int64, F float64, B bytes, S utf-8 string, A ndarray (dtype code u8, ndim u8, dims u32..., raw
# data), L list (count u32, then that many messages).
MAGIC = b'PMG1'
MAX_PAYLOAD = 1 << 30          # 1 GiB: far above any packed shard, far below "allocate whatever the peer says"
MAX_LIST = 1 << 16
_DTYPES = {0: np.dtype('<f4'), 1: np.dtype('<f8'), 2: np.dtype('<i4'), 3: np.dtype('<i8'), 4: np.dtype('u1')}
_DTYPE_CODE = {v: k for k, v in _DTYPES.items()}


class ProtocolError(ConnectionError):
    pass


def _encode(obj):
    if obj is None:
        return b'N', b''
    if isinstance(obj, (bool, int, np.integer)):
        return b'I', struct.pack('<q', int(obj))
    if isinstance(obj, (float, np.floating)):
        return b'F', struct.pack('<d', float(obj))
    if isinstance(obj, (bytes, bytearray)):
        return b'B', bytes(obj)
    if isinstance(obj, str):
        return b'S', obj.encode('utf-8')
    if isinstance(obj, np.ndarray):
        a = np.ascontiguousarray(obj)
        dt = a.dtype.newbyteorder('<') if a.dtype.byteorder == '>' else a.dtype
        if np.dtype(dt) not in _DTYPE_CODE:
            raise TypeError('rendezvous: unsupported array dtype %s' % a.dtype)
        head = struct.pack('<BB', _DTYPE_CODE[np.dtype(dt)], a.ndim) + struct.pack('<%dI' % a.ndim, *a.shape)
        return b'A', head + a.astype(dt, copy=False).tobytes()
    if isinstance(obj, (list, tuple)):
        if len(obj) > MAX_LIST:
            raise ValueError('rendezvous: list too long')
        parts = [struct.pack('<I', len(obj))]
        for x in obj:
            t, p = _encode(x)
            parts.append(t + struct.pack('<I', len(p)) + p)
        return b'L', b''.join(parts)
    raise TypeError('rendezvous: cannot send a %s (only None, int, float, bytes, str, ndarray and lists of those)' % type(obj).__name__)


def _decode(tag, payload):
    if tag == b'N':
        if payload:
            raise ProtocolError('malformed none')
        return None
    if tag == b'I':
        if len(payload) != 8:
            raise ProtocolError('malformed int')
        return struct.unpack('<q', payload)[0]
    if tag == b'F':
        if len(payload) != 8:
            raise ProtocolError('malformed float')
        return struct.unpack('<d', payload)[0]
    if tag == b'B':
        return bytes(payload)
    if tag == b'S':
        try:
            return bytes(payload).decode('utf-8')
        except UnicodeDecodeError:
            raise ProtocolError('malformed string')
    if tag == b'A':
        if len(payload) < 2:
            raise ProtocolError('malformed array header')
        code, ndim = struct.unpack_from('<BB', payload, 0)
        if code not in _DTYPES or ndim > 8 or len(payload) < 2 + 4 * ndim:
            raise ProtocolError('malformed array header')
        shape = struct.unpack_from('<%dI' % ndim, payload, 2)
        dt = _DTYPES[code]
        count = int(np.prod(shape, dtype=np.int64)) if ndim else 1
        data = memoryview(payload)[2 + 4 * ndim:]
        if count * dt.itemsize != len(data):
            raise ProtocolError('array size does not match its shape')
        return np.frombuffer(data, dtype=dt, count=count).reshape(shape).copy()
    if tag == b'L':
        if len(payload) < 4:
            raise ProtocolError('malformed list')
        n, = struct.unpack_from('<I', payload, 0)
        if n > MAX_LIST:
            raise ProtocolError('list too long')
        off, out = 4, []
        for _ in range(n):
            if len(payload) < off + 5:
                raise ProtocolError('truncated list')
            t = bytes(payload[off:off + 1])
            ln, = struct.unpack_from('<I', payload, off + 1)
            off += 5
            if len(payload) < off + ln:
                raise ProtocolError('truncated list item')
            out.append(_decode(t, payload[off:off + ln]))
            off += ln
        if off != len(payload):
            raise ProtocolError('trailing bytes in list')
        return out
    raise ProtocolError('unknown type tag %r' % tag)


def _send(sock, obj):
    tag, payload = _encode(obj)
    if len(payload) > MAX_PAYLOAD:
        raise ValueError('rendezvous: message of %d bytes exceeds the %d-byte cap' % (len(payload), MAX_PAYLOAD))
    sock.sendall(MAGIC + tag + struct.pack('<I', len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(1 << 20, n - len(buf)))
        if not chunk:
            raise ConnectionError('rendezvous peer closed the connection')
        buf += chunk
    return bytes(buf)


def _recv(sock, max_payload=MAX_PAYLOAD):
    hdr = _recv_exact(sock, 9)
    if hdr[:4] != MAGIC:
        raise ProtocolError('bad magic: not a rendezvous peer')
    tag = hdr[4:5]
    n, = struct.unpack('<I', hdr[5:9])
    if n > max_payload:
        raise ProtocolError('message of %d bytes exceeds the cap of %d' % (n, max_payload))
    return _decode(tag, _recv_exact(sock, n))


def _is_loopback(addr):
    return addr in ('localhost', '::1') or addr.startswith('127.')


def _is_local_address(addr):
    """True when `addr` names THIS machine: a loopback name / address, or a host name / address that resolves to one of this
    host's own interfaces (a launcher that exports the node's hostname or FQDN as MASTER_ADDR for a single-node job:
    `torchrun --standalone`, SLURM wrappers).  An address is taken for one of ours when a socket can be bound to it."""
    if _is_loopback(addr):
        return True
    try:
        infos = socket.getaddrinfo(addr, None, proto=socket.IPPROTO_TCP)
    except OSError:
        return False
    for fam, _, _, _, sa in infos:
        ip = sa[0]
        if ip.startswith('127.') or ip == '::1':
            return True
        try:
            with socket.socket(fam, socket.SOCK_DGRAM) as probe:
                probe.bind((ip, 0))
            return True
        except OSError:
            continue
    return False


def job_token(addr='127.0.0.1'):
    """Token the hello of every rank carries.  With PMG_RDV_TOKEN set by the launcher it is a shared secret.  Without it the
    token is derived from the launcher's run id, master address / port and world size -- values anyone who can reach the
    port can guess -- so it only rejects STALE or ACCIDENTAL peers (another job, a port scanner), it does not authenticate.
    That is acceptable for a single-node job -- the rendezvous address is this machine: loopback, or the node's own host
    name / interface address (_is_local_address) -- which is every case bench.py launches; when the master address is
    ANOTHER machine (multi-node) PMG_RDV_TOKEN is REQUIRED and its absence is an error."""
    import hashlib
    tok = os.environ.get('PMG_RDV_TOKEN')
    if tok is None:
        # rank 0 is by definition local to MASTER_ADDR, so the address test below cannot see a multi-node job from there: the
        # launcher's own count can (torchrun exports LOCAL_WORLD_SIZE).  Without a shared secret a multi-node job is refused on
        # EVERY rank, and a single-node one listens on loopback only (Rendezvous: _derived_token_endpoint)
        lws, ws = os.environ.get('LOCAL_WORLD_SIZE'), os.environ.get('WORLD_SIZE')
        if lws and ws and int(ws) > int(lws):
            raise RuntimeError('rendezvous of a multi-node job (WORLD_SIZE %s > LOCAL_WORLD_SIZE %s): set PMG_RDV_TOKEN to a secret shared by '
                               'the ranks -- the derived token is guessable and only protects a single-node rendezvous on loopback' % (ws, lws))
        if not _is_local_address(addr) and os.environ.get('PMG_RDV_SINGLE_NODE') != '1':
            raise RuntimeError('rendezvous on %s, which is not an address of this machine: set PMG_RDV_TOKEN to a secret shared by '
                               'the ranks (the derived token only protects a single-node rendezvous against stale peers; a '
                               'single-node launcher whose master address does not resolve here can say PMG_RDV_SINGLE_NODE=1)' % addr)
        tok = '|'.join(os.environ.get(k, '') for k in ('TORCHELASTIC_RUN_ID', 'MASTER_ADDR', 'MASTER_PORT', 'PMG_RDV_PORT', 'WORLD_SIZE'))
    return hashlib.sha256(('pmg-rdv:' + tok).encode()).digest()


def _derived_token_endpoint(addr):
    """Where the rendezvous really listens / connects.  With the DERIVED (guessable) token the job is single-node by job_token()'s
    checks, so the listener stays on loopback whatever name the launcher exported as MASTER_ADDR: nobody off this machine can
    reach it.  With PMG_RDV_TOKEN (a shared secret) the address is used as given."""
    if 'PMG_RDV_TOKEN' in os.environ:
        return '127.0.0.1' if addr == 'localhost' else addr
    return '127.0.0.1'


class Rendezvous:
    """Rank 0 listens, everybody else connects; every collective is "all ranks send one object to rank 0, rank 0 sends
    the list back" -- the ranks call them in the same order, so no threads and no tags are needed.  Control plane only:
    the data path of the job is the RCCL all-gather inside the library."""

    def __init__(self, rank, world_size, addr=None, port=None, timeout=None):
        self.rank, self.world = int(rank), int(world_size)
        if timeout is None:      # generous: on a fresh box the ranks' first import of the GPU libraries can take minutes
            timeout = float(os.environ.get('PMG_RDV_TIMEOUT', '600'))
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
            token = job_token(addr)                 # (raises for a multi-node job without PMG_RDV_TOKEN before anything listens)
            srv.bind((_derived_token_endpoint(addr), port))
            srv.listen(self.world)
            srv.settimeout(timeout)
            by_rank = {}
            deadline = time.time() + timeout
            rejected = []

            def give_up(why):
                srv.close()
                for c in by_rank.values():          # do not leave the ranks that did arrive hanging on an open socket
                    c.close()
                raise TimeoutError('rendezvous: %d of %d ranks arrived within %.0f s (%s)%s' % (
                    len(by_rank) + 1, self.world, timeout, why,
                    '; rejected peers: ' + '; '.join(rejected[-8:]) if rejected else ''))
            while len(by_rank) < self.world - 1:
                left = deadline - time.time()
                if left <= 0:
                    give_up('deadline')
                srv.settimeout(min(left, 5.0))
                try:
                    conn, peer = srv.accept()
                except socket.timeout:
                    continue
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(2.0)                # an idle stranger holds the accept loop for 2 s, not 10
                why = None
                try:      # hello = [token bytes, rank]; anything else (a port scanner, a stale job) is dropped, not fatal
                    hello = _recv(conn, max_payload=256)
                    if not (isinstance(hello, list) and len(hello) == 2 and isinstance(hello[0], bytes) and isinstance(hello[1], int)):
                        why = 'malformed hello'
                    elif not hmac.compare_digest(hello[0], token):
                        why = 'wrong job token (do the ranks share PMG_RDV_TOKEN / MASTER_ADDR / MASTER_PORT / WORLD_SIZE?)'
                    elif not 1 <= hello[1] < self.world:
                        why = 'rank %d out of range' % hello[1]
                    elif hello[1] in by_rank:
                        why = 'duplicate rank %d' % hello[1]
                except (ConnectionError, OSError, struct.error, ValueError) as ex:
                    why = 'unreadable hello (%s)' % type(ex).__name__
                if why is not None:
                    msg = '%s:%s %s' % (peer[0], peer[1], why)
                    rejected.append(msg)
                    print('rendezvous (rank 0): dropped a connection from ' + msg, file=sys.stderr, flush=True)
                    try:
                        # the reply to the hello: a legitimate rank fails fast instead of timing out.  The REASON goes to rank 0's
                        # stderr only -- the peer is unauthenticated
                        _send(conn, 'NAK')
                    except (ConnectionError, OSError):
                        pass
                    conn.close()
                    continue
                try:
                    _send(conn, 'ACK')              # the handshake: the client reads this reply in its constructor
                except (ConnectionError, OSError):
                    conn.close()
                    continue
                conn.settimeout(timeout)
                by_rank[hello[1]] = conn
            srv.close()
            self.peers = [by_rank[r] for r in range(1, self.world)]
        else:
            token = job_token(addr)                 # (first: a multi-node job without PMG_RDV_TOKEN fails here, not in a connect timeout)
            deadline = time.time() + timeout
            while True:
                try:
                    s = socket.create_connection((_derived_token_endpoint(addr), port), timeout=5.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(timeout)
            _send(s, [token, self.rank])
            # the handshake: rank 0 answers the hello with ACK or NAK before any collective -- no sentinel values inside
            # collective payloads, and a refused rank learns it here, not from a broken pipe in its first all-gather
            try:
                reply = _recv(s, max_payload=16)
            except (ConnectionError, OSError) as ex:
                s.close()
                raise ConnectionError('rendezvous: rank 0 closed the connection during the hello (%s)' % type(ex).__name__)
            if reply != 'ACK':
                s.close()
                raise ConnectionError('rendezvous: rank 0 refused this rank (wrong job token, duplicate or out-of-range rank: see rank 0\'s stderr)')
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
