"""ctypes binding of oracle/libpmg_oracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement the HIP path is checked against.  Nothing
under pybullet_multigoal_gym_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, 'oracle')

TASKS = {'reach': 0, 'push': 1, 'pick_and_place': 2, 'slide': 3, 'block_stack': 4, 'block_rearrange': 5, 'chest_push': 6, 'chest_pick_and_place': 7}


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


def make_config(task, num_envs, num_block=4, binary_reward=True, joint_control=False, max_episode_steps=50,
                distance_threshold=0.05, seed_base=0, seed_stride=0, random_order=True, device=0, env_index_offset=0,
                task_decomposition=False, use_curriculum=False, num_goals_to_generate=0, grip_informed_goal=False):
    c = PmgConfig()
    c.struct_size = C.sizeof(PmgConfig)
    c.task = TASKS[task]
    c.num_envs = num_envs
    c.num_block = num_block
    c.binary_reward = int(binary_reward)
    c.joint_control = int(joint_control)
    c.max_episode_steps = max_episode_steps
    c.device = device
    c.distance_threshold = distance_threshold
    c.random_order = int(random_order)
    c.seed_base = seed_base
    c.seed_stride = seed_stride
    c.env_index_offset = env_index_offset
    c.task_decomposition = int(task_decomposition)
    c.use_curriculum = int(use_curriculum)
    c.num_goals_to_generate = int(num_goals_to_generate)
    c.grip_informed_goal = int(grip_informed_goal)
    return c


def usable_threads(cap=16):
    """OpenMP threads worth asking for: the affinity mask capped by the cgroup CPU quota (and by `cap`)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, min(n, cap))


def build():
    subprocess.check_call(['make', '-C', ODIR, '-s'])


_libs = {}


def load(f32=False):
    name = 'libpmg_oracle_f32.so' if f32 else 'libpmg_oracle.so'
    if name not in _libs:
        path = os.path.join(ODIR, name)
        if not os.path.exists(path):
            build()
        lib = C.CDLL(path)
        lib.pmgo_last_error.restype = C.c_char_p
        _libs[name] = lib
    return _libs[name]


def _fp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleEnv:
    """Batched env with the product's call shapes, backed by the CPU restatement."""

    def __init__(self, task, num_envs, f32=False, threads=1, **kw):
        self.lib = load(f32)
        self.cfg = make_config(task, num_envs, **kw)
        self.h = C.c_void_p()
        rc = self.lib.pmgo_create(C.byref(self.cfg), C.byref(self.h))
        if rc != 0:
            raise RuntimeError(self.lib.pmgo_last_error(None).decode())
        self.dims = PmgDims()
        self.lib.pmgo_get_dims(self.h, C.byref(self.dims))
        self.lib.pmgo_set_threads(self.h, threads)
        self.N = num_envs

    def close(self):
        if self.h:
            self.lib.pmgo_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _bufs(self):
        d, N = self.dims, self.N
        return (np.zeros((N, d.observation_dim), np.float32), np.zeros((N, d.policy_state_dim), np.float32),
                np.zeros((N, d.goal_dim), np.float32), np.zeros((N, d.goal_dim), np.float32))

    def seed(self, base, stride=0):
        self.lib.pmgo_seed(self.h, C.c_uint64(base), C.c_uint64(stride))

    def reset(self, mask=None):
        o, p, a, g = self._bufs()
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        rc = self.lib.pmgo_reset(self.h, _fp(m), _fp(o), _fp(p), _fp(a), _fp(g))
        assert rc == 0
        return dict(observation=o, policy_state=p, achieved_goal=a, desired_goal=g)

    def step(self, actions):
        actions = np.ascontiguousarray(actions, np.float32).reshape(self.N, self.dims.action_dim)
        o, p, a, g = self._bufs()
        r = np.zeros(self.N, np.float32)
        ok = np.zeros(self.N, np.uint8)
        dn = np.zeros(self.N, np.uint8)
        rc = self.lib.pmgo_step(self.h, _fp(actions), _fp(o), _fp(p), _fp(a), _fp(g), _fp(r), _fp(ok), _fp(dn))
        if rc != 0:
            raise RuntimeError(self.lib.pmgo_last_error(self.h).decode())
        return dict(observation=o, policy_state=p, achieved_goal=a, desired_goal=g), r, dn.astype(bool), ok.astype(bool)

    def compute_reward(self, ag, dg):
        ag = np.ascontiguousarray(ag, np.float32)
        dg = np.ascontiguousarray(dg, np.float32)
        B = ag.size // self.dims.goal_dim
        r = np.zeros(B, np.float32)
        ok = np.zeros(B, np.uint8)
        self.lib.pmgo_compute_reward(self.h, _fp(ag), _fp(dg), C.c_int64(B), _fp(r), _fp(ok))
        return r, ok.astype(bool)

    def set_sub_goal(self, ind, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        rc = self.lib.pmgo_set_sub_goal(self.h, _fp(m), C.c_int32(ind))
        if rc != 0:
            raise RuntimeError(self.lib.pmgo_last_error(self.h).decode())

    def curriculum_update(self, enabled):
        rc = self.lib.pmgo_curriculum_update(self.h, C.c_int32(int(enabled)))
        if rc != 0:
            raise RuntimeError(self.lib.pmgo_last_error(self.h).decode())

    def curriculum(self):
        nb = self.cfg.num_block + (1 if self.cfg.task in (TASKS['chest_push'], TASKS['chest_pick_and_place']) else 0)
        lv, gs = np.zeros(self.N, np.int32), np.zeros(self.N, np.int32)
        pr, gen = np.zeros((self.N, nb), np.float32), np.zeros((self.N, nb), np.float32)
        rc = self.lib.pmgo_curriculum_read(self.h, _fp(lv), _fp(gs), _fp(pr), _fp(gen))
        if rc != 0:
            raise RuntimeError(self.lib.pmgo_last_error(self.h).decode())
        return dict(level=lv, goal_step=gs, prob=pr, generated=gen)

    def get_state(self):
        s = np.zeros((self.N, self.dims.state_dim), np.float32)
        self.lib.pmgo_get_state(self.h, _fp(s))
        return s

    def set_state(self, s):
        s = np.ascontiguousarray(s, np.float32)
        assert s.shape == (self.N, self.dims.state_dim)
        self.lib.pmgo_set_state(self.h, _fp(s))

    def set_goal(self, goals, mask=None):
        goals = np.ascontiguousarray(goals, np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self.lib.pmgo_set_goal(self.h, _fp(m), _fp(goals))


def fk_tip(q, f32=False):
    lib = load(f32)
    q = np.ascontiguousarray(q, np.float64)
    pos = np.zeros(3)
    rot = np.zeros(9)
    lib.pmgo_fk_tip(_fp(q), _fp(pos), _fp(rot))
    return pos, rot.reshape(3, 3)


def ik(q0, pos, quat=(0, -1, 0, 0), max_iter=40, thr=1e-5, f32=False):
    lib = load(f32)
    q0 = np.ascontiguousarray(q0, np.float64)
    pos = np.ascontiguousarray(pos, np.float64)
    quat = np.ascontiguousarray(quat, np.float64)
    out = np.zeros(9)
    lib.pmgo_ik.restype = C.c_int
    it = lib.pmgo_ik(_fp(q0), _fp(pos), _fp(quat), C.c_int(max_iter), C.c_double(thr), _fp(out))
    return out, it


def minv(q, f32=False):
    lib = load(f32)
    q = np.ascontiguousarray(q, np.float64)
    out = np.zeros(81)
    lib.pmgo_minv(_fp(q), _fp(out))
    return out.reshape(9, 9)


def fdyn(q, qd, tau, f32=False):
    lib = load(f32)
    a = [np.ascontiguousarray(x, np.float64) for x in (q, qd, tau)]
    out = np.zeros(9)
    lib.pmgo_fdyn(_fp(a[0]), _fp(a[1]), _fp(a[2]), _fp(out))
    return out


def box_box(ca, Ra, ha, cb, Rb, hb, margin=0.002, f32=False):
    lib = load(f32)
    a = [np.ascontiguousarray(x, np.float64) for x in (ca, Ra, ha, cb, Rb, hb)]
    out = np.zeros(40)
    lib.pmgo_box_box.restype = C.c_int
    n = lib.pmgo_box_box(*[_fp(x) for x in a], C.c_double(margin), _fp(out))
    return out.reshape(4, 10)[:n]


class FloorOracle(OracleEnv):
    """The CHAOS FLOOR yardstick: the float64 oracle with its joint / block / door state rounded to float32 after every
    substep (`state_f32_per_substep`, oracle/pmg_oracle.c) -- float64 arithmetic on a float32 state, the best any
    implementation that keeps its state in float32 can do.  How far it strays from the plain float64 oracle, and how often
    by more than 1e-3 (a contact made or missed one substep apart), does not depend on anybody's arithmetic: the bars of
    the -m gpu contact tests are multiples of THAT, not of the float32 build of the oracle (a poor float32 code whose own
    outliers are 10-100 x as many, profiles/r04_chaos_floor.txt)."""

    def __init__(self, task, num_envs, **kw):
        kw.pop('f32', None)
        OracleEnv.__init__(self, task, num_envs, f32=False, **kw)

    def step(self, actions):
        set_prior('state_f32_per_substep', 1.0)        # process-wide switch: on for this call only
        try:
            return OracleEnv.step(self, actions)
        finally:
            set_prior('state_f32_per_substep', 0.0)


def cyl_box(cc, Rc, rad, hl, cb, Rb, hb, margin=0.002, f32=False):
    lib = load(f32)
    a = [np.ascontiguousarray(x, np.float64) for x in (cc, Rc, cb, Rb, hb)]
    out = np.zeros(40)
    lib.pmgo_cyl_box.restype = C.c_int
    n = lib.pmgo_cyl_box(_fp(a[0]), _fp(a[1]), C.c_double(rad), C.c_double(hl), _fp(a[2]), _fp(a[3]), _fp(a[4]), C.c_double(margin), _fp(out))
    return out.reshape(4, 10)[:n]


def rng_probe(seed, n_double, shuffle_n=0):
    lib = load()
    d = np.zeros(max(n_double, 1))
    p = np.zeros(max(shuffle_n, 1), np.int32)
    lib.pmgo_rng_probe(C.c_uint64(seed), C.c_int(n_double), _fp(d), C.c_int(shuffle_n), _fp(p))
    return d[:n_double], p[:shuffle_n]


# ---- [BULLET-PRIOR] switches of the oracle (process-wide; see oracle/pmg_oracle.c) ----
def prior_names():
    lib = load()
    lib.pmgo_prior_name.restype = C.c_char_p
    return [lib.pmgo_prior_name(i).decode() for i in range(lib.pmgo_prior_count())]


def set_prior(name, value, f32=False):
    lib = load(f32)
    rc = lib.pmgo_set_prior(name.encode(), C.c_double(float(value)))
    if rc != 0:
        raise KeyError(name)


def get_prior(name, f32=False):
    lib = load(f32)
    lib.pmgo_get_prior.restype = C.c_double
    return lib.pmgo_get_prior(name.encode())


def reset_priors():
    for f32 in (False, True):
        load(f32).pmgo_reset_priors()
