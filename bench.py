#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched Kuka env on N MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          # N > 1: spawns one rank per GPU itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   # or is launched per rank

One "step" = one batched env.step() over 4096 envs per GPU (BASELINE.json configs[1]: task='reach', 4096 vectorised
envs, random policy, state obs), actions pre-generated and resident in HBM, and -- for N > 1 -- one RCCL all-gather of
the packed observation shard per step.  Episodes are max_episode_steps=50 long and their phases are STAGGERED, as in an
RL loop that resets each env when its own episode ends: env i starts at phase i mod 50 (an untimed pre-roll of one
episode brings the batch there) and every batched step is followed by the masked reset of the 1/50 of the batch whose
episode just ended (pmg_reset_done_device: the device finds them by their elapsed-step counters, no host mask) -- inside
the timed region.  Any window of K steps therefore sees every episode phase in the same
proportion, and `value` no longer depends on where in an episode a short window falls (`--lockstep` restores the old
all-envs-in-phase loop).  W untimed warm-up steps, then EXACTLY K timed steps between barrier + stream sync; the slowest
rank's time counts; rank 0 prints ONE JSON line.  Secondary fields, never `value`: `full_episodes` (the same loop over
two whole episodes) and `host_api` (numpy in / numpy out through env.step(), PCIe inclusive).

The env is the C-ABI HIP library (ctypes).  No PyTorch anywhere: the ranks rendezvous over stdlib TCP
(pybullet_multigoal_gym_amd.distributed.Rendezvous) for the 128-byte RCCL id, the barriers and the max over ranks.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver only supports dmabuf IPC: RCCL's cross-process buffer sharing needs this before any HIP/HSA init
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

# algorithmic bytes per env-step (state r+w, action, outputs)
ALGO_BYTES = {'reach': 298, 'push': 486, 'slide': 486, 'pick_and_place': 490, 'block_stack': 1246, 'block_rearrange': 1242,
              # chest tasks with 4 blocks: state (24 + 4*13 + 3 door floats, order, counter + RNG cursor) r+w 704, action, outputs 618
              'chest_push': 1334, 'chest_pick_and_place': 1338}
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
# MI355X_MICROARCH.md "Wave scheduling": 256 CUs x 4 SIMD-32; a wave64 VALU instruction issues over 2 cycles
VALU_PEAK_INSTS = 1024 * 2.4e9 / 2          # 1.2288e12 wave-instructions / s
FP32_PEAK_TFLOPS = 157.3                    # 256 CU x 4 SIMD x 32 lanes x 2 flop x 2.4 GHz (vector fp32, = the fp32 MFMA rate)


def _committed(pattern):
    import glob
    return sorted(glob.glob(os.path.join(ROOT, 'profiles', pattern)), reverse=True)   # newest round tag first


def kernel_source_hash():
    """sha256 over the sources libpmg_hip.so is built from (csrc/*, include/*.h, the Makefile with its flags): what ties a
    committed counter pass to the kernels that are running (library_hash() ties it to the binary as well)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, 'pybullet_multigoal_gym_amd', 'csrc', '*.[hc]*')) +
                   glob.glob(os.path.join(ROOT, 'pybullet_multigoal_gym_amd', 'csrc', '*.inc')) +
                   glob.glob(os.path.join(ROOT, 'pybullet_multigoal_gym_amd', 'csrc', 'Makefile')) +
                   glob.glob(os.path.join(ROOT, 'include', '*.h')))
    for f in files:
        if f.endswith(('.so', '.o')):
            continue
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def library_hash(path=None):
    """sha256 of the libpmg_hip.so that is (or would be) loaded.  hipcc builds of the same sources are byte-identical (the
    round-3 review rebuilt the library and compared), so this ties a counter pass to the exact binary that ran."""
    import hashlib
    path = path or os.path.join(ROOT, 'pybullet_multigoal_gym_amd', 'csrc', 'libpmg_hip.so')
    try:
        return hashlib.sha256(open(path, 'rb').read()).hexdigest()[:16]
    except OSError:
        return None


def committed_traffic(task, n_envs):
    """HBM bytes per batched step from the committed rocprofv3 PMC passes (profiles/*_pmc_traffic.json: FETCH_SIZE and
    WRITE_SIZE collected in separate --pmc runs of this same command, KiB -> bytes, with the calibration factors of the
    same file applied -- MI355X_MICROARCH.md: calibrate on a known byte count in your own access pattern).  None when no
    pass is committed for this workload."""
    for f in _committed('*_%s%d_pmc_traffic.json' % (task, n_envs)):
        try:
            d = json.load(open(f))
            if d.get('task') == task and d.get('envs_per_gpu') == n_envs:
                return float(d['hbm_bytes_per_launch']), d.get('kernel_source_sha16'), os.path.basename(f), d.get('library_sha16')
        except Exception:
            continue
    return None, None, None, None


def committed_counters(task, n_envs):
    """Per-step instruction counters of the committed PMC pass (profiles/*_pmc_summary.json), or {}."""
    for f in _committed('*_%s%d_pmc_summary.json' % (task, n_envs)):
        try:
            d = json.load(open(f))
            stamp = d.pop('_stamp', {})
            return {k: float(v['mean_per_launch']) for k, v in d.items()}, stamp.get('kernel_source_sha16'), os.path.basename(f)
        except Exception:
            continue
    return {}, None, None


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


def cpu_baseline(task, budget_s=12.0):
    """The oracle (CPU restatement, kind 'port') timed on this box's host cores on a bounded sample of the same
    workload: 64 envs per core, random actions, until ~budget_s of wall time has passed (at least 3 batched steps)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import oracle_lib
    cores = usable_cores()
    n = 64 * cores
    ora = oracle_lib.OracleEnv(task, n, seed_base=0, seed_stride=1, threads=cores)
    ora.reset()
    rs = np.random.RandomState(12345)
    A = ora.dims.action_dim
    steps = 0
    t0 = time.perf_counter()
    while True:
        ora.step(rs.uniform(-1, 1, (n, A)).astype(np.float32))
        steps += 1
        el = time.perf_counter() - t0
        if steps >= 3 and el >= budget_s:
            break
    ora.close()
    return {'value': n * steps / el, 'unit': 'env-steps/s', 'cores': cores, 'kind': 'port',
            'sample': '%d envs x %d steps of task=%s on %d OpenMP threads (oracle/pmg_oracle.c, float64); '
                      'PyBullet itself is absent on this box, so this is NOT the reference and no speed-up over the '
                      'reference may be read off it' % (n, steps, task, cores)}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: spawn one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE in the env,
    a free TCP port for the rendezvous) and pass rank 0's line through."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1',
                   PMG_RDV_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--task', default='reach')
    ap.add_argument('--envs-per-gpu', type=int, default=4096)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the secondary full_episodes / host_api measurements')
    ap.add_argument('--episode-steps', type=int, default=50)
    ap.add_argument('--dense-reward', action='store_true', help='binary_reward=False (BASELINE.json configs[3] runs both)')
    ap.add_argument('--lib', default=None, help='alternative libpmg_hip.so build (kernel A/B experiments)')
    ap.add_argument('--lockstep', action='store_true', help='all envs in the same episode phase, one full reset every episode-steps steps (the round-1/2 loop)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))

    import pybullet_multigoal_gym_amd as pmg
    from pybullet_multigoal_gym_amd._lib import PmgLibrary
    from pybullet_multigoal_gym_amd.distributed import Rendezvous, init_rccl

    # RCCL prints a version banner on fd 1 when a communicator comes up: keep fd 1 for the ONE JSON line, send the rest of
    # this process's stdout to stderr
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.exit('bench.py: WORLD_SIZE=%d but --gpus %d' % (world, args.gpus))
    # PMG_BENCH_FORCE_DIST=1 takes the multi-rank code path (rendezvous, RCCL communicator, per-step all-gather) even
    # with one rank: lets a 1-GPU box validate what the N > 1 launch will execute
    multi = world > 1 or bool(os.environ.get('PMG_BENCH_FORCE_DIST'))
    rdv = Rendezvous(rank, world) if multi else None

    N, K, W, T = args.envs_per_gpu, args.steps, args.warmup, args.episode_steps
    # a launcher that restricts visibility per rank (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES = one GPU each) leaves every
    # rank with a single device 0: fall back to it instead of failing on device LOCAL_RANK
    device = local_rank
    try:
        ndev = PmgLibrary(args.lib).device_count()
        if ndev == 1 and local_rank >= 1:
            device = 0
            print('bench.py: rank %d sees one device (per-rank visibility); using device 0' % rank, file=sys.stderr)
        elif ndev > 1 and local_rank >= ndev:
            # more ranks than visible GPUs on a node that shows several: two ranks would share a device -- RCCL fails or hangs
            # on a duplicate GPU, and a weak-scaling number from shared devices would mislead
            sys.exit('bench.py: rank %d has LOCAL_RANK %d but only %d devices are visible: refusing to share a GPU between ranks' % (rank, local_rank, ndev))
    except Exception:
        pass
    env = pmg.make_env(task=args.task, num_envs=N, num_block=4, device=device, seed=0, seed_stride=1,
                       env_index_offset=rank * N, max_episode_steps=T, binary_reward=not args.dense_reward,
                       _library=PmgLibrary(args.lib) if args.lib else None)
    h = env.handle
    A = env.dims.action_dim
    gathered = None
    host_gather = False
    collective = 'none (one rank: nothing to gather)'
    if multi:
        # PMG_BENCH_FORCE_COMM_FAIL: test hook for the fallback below
        if init_rccl(env, rdv, _force_fail=bool(os.environ.get('PMG_BENCH_FORCE_COMM_FAIL'))):
            # the all-gather of step t runs on the library's communication stream beside step t + 1 (pmg_comm_overlap: packed rows
            # double-buffered; the gathered tables alternate too, as a consumer of table t would need while gather t + 1 is in
            # flight).  PMG_BENCH_NO_OVERLAP=1: the in-stream all-gather of rounds 1-4, for A/B
            overlap = not os.environ.get('PMG_BENCH_NO_OVERLAP')
            gathered = [h.device_alloc(world * N * env.dims.packed_dim * 4) for _ in range(2 if overlap else 1)]
            if overlap:
                h.comm_overlap(True)
            collective = 'one RCCL all-gather of the packed obs rows per step (%d B per rank), %s' % (
                N * env.dims.packed_dim * 4, 'on a communication stream, overlapped with the next step' if overlap else 'in the step stream')
        else:
            # every rank takes the SAME fallback, and the JSON line says so: packed rows to the host, TCP all-gather
            collective = 'FALLBACK: RCCL init failed, host all-gather of packed obs (PCIe + TCP inclusive)'
            host_gather = True
    # synthetic random policy: a table of batches of U(-1,1) float32 actions, resident in HBM
    E = 0 if args.no_extras else 2 * T                     # extra whole episodes for the secondary measurement
    stagger = not args.lockstep
    P = T if stagger else 0                                # untimed pre-roll: one episode, brings env i to phase i mod T
    table = np.random.RandomState(12345 + rank).uniform(-1, 1, (P + K + W + E, N, A)).astype(np.float32)
    actions = h.device_alloc(table.nbytes)
    h.upload(actions, table)
    stride = N * A * 4
    mine = np.empty((N, env.dims.packed_dim), np.float32) if host_gather else None
    masks = None
    if stagger:
        # masks[p][i] = 1 where (i + p) % T == 0: after batched step t the envs with (global index + t + 1) % T == 0 have
        # finished their episode (TimeLimit: done) and are reset before the next step, N/T of them every step
        gi = rank * N + np.arange(N)
        m = np.stack([((gi + p) % T == 0) for p in range(T)]).astype(np.uint8)
        masks = h.device_alloc(m.nbytes)
        h.upload(masks, m)

    host_gather_s = [0.0, 0]                               # FALLBACK path only: seconds in download + TCP all-gather, and how many

    def run(first, count, phase0=0):
        for t in range(first, first + count):
            if not stagger and (t - phase0) % T == 0:
                h.reset_device(None)
            h.step_device(actions + t * stride)
            if stagger and t < P:
                h.reset_device(masks + ((t + 1) % T) * N)  # untimed pre-roll: the mask table sets up the staggered phases
            elif stagger:
                h.reset_done_device()                      # from then on the envs whose TimeLimit ran out reset themselves on the device
            if gathered is not None and len(gathered) == 2:
                h.allgather_packed_async(gathered[t & 1])
            elif gathered is not None:
                h.allgather_packed(gathered[0])
            elif host_gather:
                h.sync()
                tg = time.perf_counter()
                h.download(mine, h.device_ptr())
                rdv.allgather(mine)
                host_gather_s[0] += time.perf_counter() - tg
                host_gather_s[1] += 1

    def fence():
        h.sync()                       # the library's stream: every kernel and the all-gather
        if multi:
            rdv.barrier()

    def timed(first, count, phase0=0):
        fence()
        h.timing_reset()
        t0 = time.perf_counter()
        run(first, count, phase0)
        fence()
        el = time.perf_counter() - t0
        return rdv.max(el) if multi else el

    # HIP events bracket every 4th batched step (every step in short runs): each event idles the queue for ~6 us, and all
    # steps run the same launch sequence -- staggered phases make every step statistically the same
    timing_every = 4 if K >= 16 else 1
    h.timing_every(timing_every)
    if stagger:
        h.reset_device(None)
    run(0, P + W)
    el_local = None

    def timed_keep(first, count, phase0=0):
        nonlocal el_local
        fence()
        h.timing_reset()
        t0 = time.perf_counter()
        run(first, count, phase0)
        h.sync()
        el_local = time.perf_counter() - t0                # this rank's own clock, before it waits for the others
        fence()
        el_all = time.perf_counter() - t0
        return rdv.max(el_all) if multi else el_all

    host_gather_s[0], host_gather_s[1] = 0.0, 0
    el = timed_keep(P + W, K)
    hg_ms = host_gather_s[0] / max(1, host_gather_s[1]) * 1e3
    kmin, kernel_ms, kmax, launches = h.timing_stats()
    per_rank = None
    if multi:
        # diagnosability of the N > 1 run: every rank's own wall clock, step-kernel average and all-gather events
        cavg, cmax, cn = h.comm_timing() if gathered is not None else (0.0, 0.0, 0)
        rows = rdv.allgather(np.array([el_local / K * 1e3, kernel_ms, kmax, cavg, cmax, float(cn), hg_ms], np.float64))
        if rank == 0:
            rows = np.asarray(rows).reshape(world, 7)
            per_rank = {'ms_per_step': [round(float(x), 4) for x in rows[:, 0]],
                        'kernel_ms': [round(float(x), 4) for x in rows[:, 1]],
                        'kernel_ms_max': [round(float(x), 4) for x in rows[:, 2]],
                        'allgather_ms': [round(float(x), 4) for x in rows[:, 3]],
                        'allgather_ms_max': [round(float(x), 4) for x in rows[:, 4]],
                        'allgather_launches': [int(x) for x in rows[:, 5]],
                        'host_gather_ms': [round(float(x), 4) for x in rows[:, 6]],   # FALLBACK only: D2H + TCP all-gather per step
                        'rank_spread_ms': round(float(rows[:, 0].max() - rows[:, 0].min()), 4),
                        'slowest_rank': int(rows[:, 0].argmax()),
                        'note': 'ms_per_step: each rank\'s own clock over its K steps (value uses the slowest rank incl. the '
                                'closing barrier); allgather_ms: HIP events around ncclAllGather on the rank\'s stream, i.e. '
                                'transfer + the wait for the slowest peer; expected transfer for %d B per rank over xGMI: '
                                '%.1f us at 50 GB/s' % (N * env.dims.packed_dim * 4, (world - 1) * N * env.dims.packed_dim * 4 / 50e9 * 1e6)}
    full = None
    if E:
        el_full = timed(P + K + W, E, phase0=P + K + W)    # lockstep: starts with a reset; exactly two whole episodes
        fmin, favg, fmax, fl = h.timing_stats()
        full = {'value': world * N * E / el_full, 'unit': 'env-steps/s', 'steps': E, 'ms_per_step': el_full / E * 1e3,
                'kernel_ms': {'min': fmin, 'avg': favg, 'max': fmax},
                'note': 'the same loop over %d more steps = %d whole %d-step episodes (every episode phase weighted equally)' % (E, E // T, T)}
    host = None
    if E and world == 1:
        # what a numpy-only RL loop gets: env.step(numpy actions) -> numpy observations (H2D actions, kernel, D2H packed
        # rows, unpack); one whole episode.  Never `value`.
        hs = T
        acts = table[P:P + hs]
        env.reset()
        t0 = time.perf_counter()
        for t in range(hs):
            env.step(acts[t])
        elh = time.perf_counter() - t0
        host = {'value': N * hs / elh, 'unit': 'env-steps/s', 'steps': hs, 'ms_per_step': elh / hs * 1e3,
                'bytes_h2d_per_step': N * A * 4, 'bytes_d2h_per_step': N * env.dims.packed_dim * 4,
                'path': 'env.step(numpy [N,A]) -> dict of numpy arrays, reward, done, info (PCIe inclusive)'}

    if rank == 0:
        value = world * N * K / el
        algo = ALGO_BYTES[args.task] * N
        achieved = algo / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        src_hash = kernel_source_hash()
        traffic, traffic_src, traffic_file, traffic_lib = committed_traffic(args.task, N)
        lib_hash = library_hash(args.lib)
        cnt, cnt_src, cnt_file = committed_counters(args.task, N)
        out = {
            'metric': 'env-steps/sec at N_envs=4096/GPU, KukaReach' if args.task == 'reach' and N == 4096
                      else 'env-steps/sec at N_envs=%d/GPU, %s' % (N, args.task),
            'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': el / K * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': "task='%s', %d vectorised envs/GPU, random policy U(-1,1), state obs, %s reward, "
                                   '%d-step episodes (%s), 100 substeps/env-step'
                                   % (args.task, N, 'dense' if args.dense_reward else 'binary', T,
                                      'phases staggered: env i at phase i mod %d, device-side reset (pmg_reset_done_device) of the 1/%d of the batch whose TimeLimit ran out after every step, inside the timed region' % (T, T)
                                      if stagger else 'lockstep: one reset of the whole batch every %d steps' % T),
                       'global_envs': world * N, 'parallelism': 'env-shard x%d; %s' % (world, collective),
                       'window': ('every episode phase is present in every batched step (staggered), so a %d-step window is '
                                  'representative; %d reset launches inside it' % (K, K)) if stagger else
                                 'timed steps = episode steps %d..%d of %d-step episodes%s' % (
                                     W % T, W % T + K - 1, T, '' if K >= T else
                                     ' (shorter than an episode: see full_episodes for the phase-weighted rate)')},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS,
                         # counters come from a committed rocprofv3 --pmc pass: reported only when that pass profiled
                         # THESE kernel sources (sha256 of csrc/ + include/ stamped into the pass), else null
                         'traffic': traffic if (traffic_src == src_hash) else None,
                         'traffic_source': {'file': traffic_file, 'profiled_kernel_sources': traffic_src, 'running_kernel_sources': src_hash,
                                            'match': traffic_src == src_hash, 'profiled_library': traffic_lib, 'running_library': lib_hash,
                                            'library_match': (traffic_lib == lib_hash) if traffic_lib else None},
                         'kernel': 'pmg_k_step_reach2 / pmg_k_step_reach (the device picks one per step, DESIGN.md 3.1f) + pmg_k_redo' if args.task == 'reach' else 'pmg_k_step_list / pmg_k_step_obj4 family (two concurrent launches + redo)',
                         'kernel_ms': kernel_ms, 'kernel_ms_min': kmin, 'kernel_ms_max': kmax, 'launches': launches,
                         'timed_every': timing_every,
                         'algorithmic_bytes_per_env_step': ALGO_BYTES[args.task],
                         'note': 'serial 100-substep rigid-body chain per env held in registers: HBM-light by construction '
                                 '(SURVEY.md 8d).  What binds it is the dependent-issue latency of that chain (one wavefront '
                                 'issues ~1 instruction / 4.3 cycles; a batched step lasts as long as its slowest wavefront = '
                                 'an env with finger-table contacts: kernel_ms_max vs kernel_ms_min); see roofline.valu'},
        }
        if cnt.get('SQ_INSTS_VALU') and kernel_ms > 0 and cnt_src == src_hash:
            vi = cnt['SQ_INSTS_VALU']
            v = {'insts_per_launch': vi, 'peak_insts_per_s': VALU_PEAK_INSTS, 'util': vi / (kernel_ms * 1e-3) / VALU_PEAK_INSTS,
                 'source': 'SQ_INSTS_VALU of the committed rocprofv3 --pmc pass over the live kernel time; peak = 1024 SIMD-32 '
                           'x 2.4 GHz / 2 cycles per wave64 instruction'}
            fl = [cnt.get('SQ_INSTS_VALU_ADD_F32'), cnt.get('SQ_INSTS_VALU_MUL_F32'), cnt.get('SQ_INSTS_VALU_FMA_F32'), cnt.get('SQ_INSTS_VALU_TRANS_F32')]
            if all(x is not None for x in fl):
                flops = 64.0 * (fl[0] + fl[1] + 2.0 * fl[2] + fl[3])      # per launch, 64 lanes per wave instruction
                v.update(fp32_flop_per_launch=flops, achieved_tflops=flops / (kernel_ms * 1e-3) / 1e12, peak_tflops=FP32_PEAK_TFLOPS,
                         flop_frac=flops / (kernel_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                         flop_source='64 x (ADD_F32 + MUL_F32 + 2 FMA_F32 + TRANS_F32) wave-instruction counters: an upper bound, idle lanes included')
            v['counter_file'] = cnt_file
            v['profiled_kernel_sources'] = cnt_src
            out['roofline']['valu'] = v
        if per_rank is not None:
            out['per_rank'] = per_rank
        if full is not None:
            out['full_episodes'] = full
        if host is not None:
            out['host_api'] = host
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.task)
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    if os.environ.get('PMG_ASSERT_NO_TORCH') and rank == 0:   # test hook: the ranks run without PyTorch
        assert 'torch' not in sys.modules, 'torch was imported by a bench rank'
        print('no torch in rank 0 (%d modules loaded)' % len(sys.modules), file=sys.stderr, flush=True)
    h.device_free(actions)
    if masks is not None:
        h.device_free(masks)
    if gathered is not None:
        for g in gathered:
            h.device_free(g)
    env.close()
    if rdv is not None:
        rdv.barrier()
        rdv.close()


if __name__ == '__main__':
    main()
