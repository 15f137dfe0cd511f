"""ctypes binding of the C ABI in include/pmg.h (libpmg_hip.so, built by hipcc for gfx950).

There is no Python/NumPy fallback: if the shared library is missing this
module raises at load time with the build command.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIBRARY = os.path.join(_HERE, 'csrc', 'libpmg_hip.so')

TASK_IDS = {'reach': 0, 'push': 1, 'pick_and_place': 2, 'slide': 3, 'block_stack': 4, 'block_rearrange': 5,
            'chest_push': 6, 'chest_pick_and_place': 7}
PMG_BUF_PACKED = 7
PMG_BUF_STATE = 8
PMG_BUF_SCHED = 9
PMG_BUF_ENV_CYCLES = 10


class PmgConfig(C.Structure):
    _fields_ = [('struct_size', C.c_int32), ('task', C.c_int32), ('num_envs', C.c_int32), ('num_block', C.c_int32),
                ('binary_reward', C.c_int32), ('joint_control', C.c_int32), ('max_episode_steps', C.c_int32),
                ('device', C.c_int32), ('distance_threshold', C.c_float), ('random_order', C.c_int32),
                ('seed_base', C.c_uint64), ('seed_stride', C.c_uint64), ('env_index_offset', C.c_int32),
                ('task_decomposition', C.c_int32), ('use_curriculum', C.c_int32), ('num_goals_to_generate', C.c_int32),
                ('grip_informed_goal', C.c_int32), ('reserved', C.c_int32 * 3)]


class PmgDims(C.Structure):
    _fields_ = [('num_envs', C.c_int32), ('action_dim', C.c_int32), ('observation_dim', C.c_int32),
                ('policy_state_dim', C.c_int32), ('goal_dim', C.c_int32), ('state_dim', C.c_int32),
                ('packed_dim', C.c_int32), ('reserved', C.c_int32)]


class PmgError(RuntimeError):
    pass


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class PmgLibrary:
    """A loaded libpmg_hip.so with typed entry points."""

    SYMBOLS = ['pmg_create', 'pmg_destroy', 'pmg_device_count', 'pmg_get_dims', 'pmg_last_error', 'pmg_seed', 'pmg_reset', 'pmg_step',
               'pmg_reset_device', 'pmg_reset_done_device', 'pmg_step_device', 'pmg_device_ptr', 'pmg_stream', 'pmg_sync', 'pmg_read_outputs',
               'pmg_compute_reward', 'pmg_compute_reward_device', 'pmg_get_state', 'pmg_set_state', 'pmg_set_goal',
               'pmg_comm_unique_id', 'pmg_comm_init', 'pmg_allgather_packed', 'pmg_comm_overlap', 'pmg_allgather_packed_async', 'pmg_allgather_wait', 'pmg_timing_reset', 'pmg_timing_every', 'pmg_timing_read',
               'pmg_device_alloc', 'pmg_device_free', 'pmg_upload', 'pmg_download',
               'pmg_set_sub_goal', 'pmg_curriculum_update', 'pmg_curriculum_read', 'pmg_timing_stats', 'pmg_get_rng', 'pmg_set_rng', 'pmg_comm_timing']

    def device_count(self):
        return int(self.lib.pmg_device_count())

    def __init__(self, path=None):
        self.path = path or DEFAULT_LIBRARY
        if not os.path.exists(self.path):
            raise PmgError('%s not found: the HIP extension is not built (run `python -c "import __graft_entry__ as g; '
                           'g.build()"` or `make -C pybullet_multigoal_gym_amd/csrc`). There is no CPU fallback.' % self.path)
        self.lib = C.CDLL(self.path)
        L = self.lib
        for s in self.SYMBOLS:
            getattr(L, s)  # raises AttributeError if the ABI is incomplete
        L.pmg_last_error.restype = C.c_char_p
        L.pmg_last_error.argtypes = [C.c_void_p]
        L.pmg_destroy.restype = None
        L.pmg_destroy.argtypes = [C.c_void_p]
        for name in self.SYMBOLS:
            if name not in ('pmg_last_error', 'pmg_destroy'):
                getattr(L, name).restype = C.c_int

    def error(self, handle=None):
        msg = self.lib.pmg_last_error(handle)
        return msg.decode() if msg else ''


_default = None


def default_library():
    global _default
    if _default is None:
        _default = PmgLibrary()
    return _default


class PmgHandle:
    """One pmg_env handle (N envs on one GPU)."""

    def __init__(self, library, **cfg_kw):
        self.L = library
        cfg = PmgConfig()
        cfg.struct_size = C.sizeof(PmgConfig)
        for k, v in cfg_kw.items():
            setattr(cfg, k, v)
        self.cfg = cfg
        self.h = C.c_void_p()
        rc = library.lib.pmg_create(C.byref(cfg), C.byref(self.h))
        if rc != 0:
            self.h = None
            raise PmgError('pmg_create failed (%d): %s' % (rc, library.error(None)))
        self.dims = PmgDims()
        self._check(library.lib.pmg_get_dims(self.h, C.byref(self.dims)))
        self.N = self.dims.num_envs

    def _check(self, rc):
        if rc != 0:
            raise PmgError('pmg call failed (%d): %s' % (rc, self.L.error(self.h)))

    def close(self):
        if getattr(self, 'h', None):
            self.L.lib.pmg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- host-buffer calls -------------------------------------------------
    def _obs_bufs(self):
        d, N = self.dims, self.N
        return (np.empty((N, d.observation_dim), np.float32), np.empty((N, d.policy_state_dim), np.float32),
                np.empty((N, d.goal_dim), np.float32), np.empty((N, d.goal_dim), np.float32))

    def seed(self, base, stride):
        self._check(self.L.lib.pmg_seed(self.h, C.c_uint64(base), C.c_uint64(stride)))

    def reset(self, mask=None):
        o, p, a, g = self._obs_bufs()
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8).reshape(self.N)
        self._check(self.L.lib.pmg_reset(self.h, _p(m), _p(o), _p(p), _p(a), _p(g)))
        return o, p, a, g

    def step(self, actions):
        o, p, a, g = self._obs_bufs()
        r = np.empty(self.N, np.float32)
        ok = np.empty(self.N, np.uint8)
        dn = np.empty(self.N, np.uint8)
        self._check(self.L.lib.pmg_step(self.h, _p(actions), _p(o), _p(p), _p(a), _p(g), _p(r), _p(ok), _p(dn)))
        return o, p, a, g, r, ok.astype(np.bool_), dn.astype(np.bool_)

    def read_outputs(self):
        o, p, a, g = self._obs_bufs()
        r = np.empty(self.N, np.float32)
        ok = np.empty(self.N, np.uint8)
        dn = np.empty(self.N, np.uint8)
        self._check(self.L.lib.pmg_read_outputs(self.h, _p(o), _p(p), _p(a), _p(g), _p(r), _p(ok), _p(dn)))
        return o, p, a, g, r, ok.astype(np.bool_), dn.astype(np.bool_)

    def compute_reward(self, ag, dg):
        G = self.dims.goal_dim
        ag = np.ascontiguousarray(ag, np.float32)
        dg = np.ascontiguousarray(dg, np.float32)
        if ag.shape != dg.shape or ag.shape[-1] != G:
            raise ValueError('achieved_goal %s / desired_goal %s must share a shape ending in %d' % (ag.shape, dg.shape, G))
        B = ag.size // G
        r = np.empty(B, np.float32)
        ok = np.empty(B, np.uint8)
        self._check(self.L.lib.pmg_compute_reward(self.h, _p(ag), _p(dg), C.c_int64(B), _p(r), _p(ok)))
        return r.reshape(ag.shape[:-1]), ok.astype(np.bool_).reshape(ag.shape[:-1])

    def get_state(self):
        s = np.empty((self.N, self.dims.state_dim), np.float32)
        self._check(self.L.lib.pmg_get_state(self.h, _p(s)))
        return s

    def set_state(self, s):
        s = np.ascontiguousarray(s, np.float32)
        if s.shape != (self.N, self.dims.state_dim):
            raise ValueError('state must have shape %s' % ((self.N, self.dims.state_dim),))
        self._check(self.L.lib.pmg_set_state(self.h, _p(s)))

    def set_goal(self, goals, mask=None):
        goals = np.ascontiguousarray(goals, np.float32).reshape(self.N, self.dims.goal_dim)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8).reshape(self.N)
        self._check(self.L.lib.pmg_set_goal(self.h, _p(m), _p(goals)))

    # -- multi-step task bookkeeping ----------------------------------------
    def set_sub_goal(self, sub_goal_ind, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8).reshape(self.N)
        self._check(self.L.lib.pmg_set_sub_goal(self.h, _p(m), C.c_int32(int(sub_goal_ind))))

    def curriculum_update(self, enabled):
        self._check(self.L.lib.pmg_curriculum_update(self.h, C.c_int32(int(bool(enabled)))))

    def curriculum_read(self):
        nb = self.cfg.num_block + (1 if self.cfg.task in (TASK_IDS['chest_push'], TASK_IDS['chest_pick_and_place']) else 0)
        level, goal_step = np.empty(self.N, np.int32), np.empty(self.N, np.int32)
        prob, generated = np.empty((self.N, nb), np.float32), np.empty((self.N, nb), np.float32)
        self._check(self.L.lib.pmg_curriculum_read(self.h, _p(level), _p(goal_step), _p(prob), _p(generated)))
        return level, goal_step, prob, generated

    # -- device-resident calls --------------------------------------------
    def step_device(self, d_actions_ptr):
        self._check(self.L.lib.pmg_step_device(self.h, C.c_void_p(d_actions_ptr)))

    def reset_device(self, d_mask_ptr=None):
        self._check(self.L.lib.pmg_reset_device(self.h, C.c_void_p(d_mask_ptr) if d_mask_ptr else None))

    def reset_done_device(self):
        """Reset, on the device, the envs whose episode has ended (TimeLimit); no host mask (include/pmg.h)."""
        self._check(self.L.lib.pmg_reset_done_device(self.h))

    def device_ptr(self, which=PMG_BUF_PACKED):
        p = C.c_void_p()
        self._check(self.L.lib.pmg_device_ptr(self.h, C.c_int(which), C.byref(p)))
        return p.value

    def schedule(self):
        """Launch schedule of the last step (diagnostics): dict(prone=[...], free=[...], redo=[...]) of env indices."""
        n = self.N
        buf = np.empty(3 + 3 * n, np.int32)
        self.sync()
        self.download(buf, self.device_ptr(PMG_BUF_SCHED))
        return {'prone': buf[2:2 + buf[0]].copy(), 'free': buf[2 + n:2 + n + buf[1]].copy(),
                'redo': buf[3 + 2 * n:3 + 2 * n + buf[2 + 2 * n]].copy()}

    def env_cycles(self):
        """Per-env cost of the last step (diagnostics; the handle must have been created with PMG_ENV_CYCLES=1 in the
        environment): [N, 2] int32 = shader cycles / 64 of the env's wavefront, largest contact count of a substep."""
        buf = np.empty((self.N, 2), np.int32)
        self.sync()
        self.download(buf, self.device_ptr(PMG_BUF_ENV_CYCLES))
        return buf

    def stream(self):
        p = C.c_void_p()
        self._check(self.L.lib.pmg_stream(self.h, C.byref(p)))
        return p.value

    def sync(self):
        self._check(self.L.lib.pmg_sync(self.h))

    def device_alloc(self, nbytes):
        p = C.c_void_p()
        self._check(self.L.lib.pmg_device_alloc(self.h, C.c_uint64(nbytes), C.byref(p)))
        return p.value

    def device_free(self, ptr):
        self._check(self.L.lib.pmg_device_free(self.h, C.c_void_p(ptr)))

    def upload(self, d_ptr, array):
        a = np.ascontiguousarray(array)
        self._check(self.L.lib.pmg_upload(self.h, C.c_void_p(d_ptr), _p(a), C.c_uint64(a.nbytes)))

    def download(self, array, d_ptr):
        assert array.flags['C_CONTIGUOUS']
        self._check(self.L.lib.pmg_download(self.h, _p(array), C.c_void_p(d_ptr), C.c_uint64(array.nbytes)))

    def timing_reset(self):
        self._check(self.L.lib.pmg_timing_reset(self.h))

    def timing_every(self, n):
        """Events around every n-th batched step only (an event costs ~6 us of idle queue on either side of the step)."""
        self._check(self.L.lib.pmg_timing_every(self.h, C.c_int(n)))

    def timing_read(self):
        ms = C.c_double()
        n = C.c_int64()
        self._check(self.L.lib.pmg_timing_read(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def timing_stats(self):
        """(min, avg, max) ms of the step-kernel launches since timing_reset(), and their count."""
        lo, avg, hi = C.c_double(), C.c_double(), C.c_double()
        n = C.c_int64()
        self._check(self.L.lib.pmg_timing_stats(self.h, C.byref(lo), C.byref(avg), C.byref(hi), C.byref(n)))
        return lo.value, avg.value, hi.value, n.value

    def get_rng(self):
        w = np.empty((self.N, 625), np.uint32)
        self._check(self.L.lib.pmg_get_rng(self.h, _p(w)))
        return w

    def set_rng(self, words):
        w = np.ascontiguousarray(words, np.uint32)
        if w.shape != (self.N, 625):
            raise ValueError('rng words must have shape (%d, 625)' % self.N)
        self._check(self.L.lib.pmg_set_rng(self.h, _p(w)))

    def comm_unique_id(self):
        buf = (C.c_uint8 * 128)()
        rc = self.L.lib.pmg_comm_unique_id(buf)
        if rc != 0:
            raise PmgError('pmg_comm_unique_id failed (%d)' % rc)
        return bytes(buf)

    def comm_init(self, rank, nranks, uid):
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        self._check(self.L.lib.pmg_comm_init(self.h, C.c_int(rank), C.c_int(nranks), buf))

    def comm_timing(self):
        """(avg, max) ms of the all-gathers since timing_reset() on this rank's stream, and their count."""
        avg, hi, n = C.c_double(), C.c_double(), C.c_int64()
        self._check(self.L.lib.pmg_comm_timing(self.h, C.byref(avg), C.byref(hi), C.byref(n)))
        return avg.value, hi.value, n.value

    def allgather_packed(self, d_out_ptr):
        self._check(self.L.lib.pmg_allgather_packed(self.h, C.c_void_p(d_out_ptr)))

    def comm_overlap(self, enabled=True):
        """Double-buffer the packed rows so that allgather_packed_async() of step t runs beside step t + 1 (include/pmg.h)."""
        self._check(self.L.lib.pmg_comm_overlap(self.h, C.c_int32(1 if enabled else 0)))

    def allgather_packed_async(self, d_out_ptr):
        self._check(self.L.lib.pmg_allgather_packed_async(self.h, C.c_void_p(d_out_ptr)))

    def allgather_wait(self, host=True):
        self._check(self.L.lib.pmg_allgather_wait(self.h, C.c_int32(1 if host else 0)))
