#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched Kuka env on N MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one batched env.step() over 4096 envs per GPU (BASELINE.json
configs[1]: task='reach', 4096 vectorised envs, random policy, state obs),
actions pre-generated and resident in HBM, episodes reset every
max_episode_steps=50 steps inside the timed region, and -- for N > 1 -- one RCCL
all-gather of the packed observation shard per step.  Rank 0 prints ONE JSON line.

The env is the C-ABI HIP library (ctypes); the synthetic action table is
uploaded once into a device buffer it owns.  torch is imported only for the
torch.distributed (gloo) rendezvous / barrier of multi-GPU runs -- it never
creates a HIP context here (its wheel bundles a different HIP runtime than
/opt/rocm).
"""
import argparse
import json
import os
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
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def committed_traffic(task, n_envs):
    """HBM bytes per pmg_k_step launch from the committed rocprofv3 PMC passes (profiles/*_pmc_traffic.json:
    FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this same command, KiB -> bytes; the
    guide's x2 FETCH correction applies to 16 B/lane streaming reads only -- these are dword accesses, so
    the raw counters are reported).  None when no pass is committed for this workload."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json'))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get('task') == task and d.get('envs_per_gpu') == n_envs:
            best = d
    return None if best is None else float(best['hbm_bytes_per_launch'])


def committed_valu(task, n_envs):
    """SQ_INSTS_VALU per pmg_k_step launch from the committed PMC pass (profiles/*_pmc_summary.json), or None."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_%s%d_pmc_summary.json' % (task, n_envs))), reverse=True):
        try:
            return float(json.load(open(f))['SQ_INSTS_VALU']['mean_per_launch'])
        except Exception:
            continue
    return None


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
    """The oracle (CPU restatement, kind 'port') timed on this box's host cores on a
    bounded sample of the same workload: 64 envs per core, random actions, until
    ~budget_s of wall time has passed (at least 3 batched steps)."""
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
                      'PyBullet itself is absent on this box' % (n, steps, task, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--task', default='reach')
    ap.add_argument('--envs-per-gpu', type=int, default=4096)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--episode-steps', type=int, default=50)
    ap.add_argument('--dense-reward', action='store_true', help='binary_reward=False (BASELINE.json configs[3] runs both)')
    ap.add_argument('--lib', default=None, help='alternative libpmg_hip.so build (kernel A/B experiments)')
    args = ap.parse_args()

    import pybullet_multigoal_gym_amd as pmg

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    # PMG_BENCH_FORCE_DIST=1 takes the multi-rank code path (torch.distributed rendezvous, RCCL communicator,
    # per-step all-gather) even with one rank: lets a 1-GPU box validate what the N > 1 launch will execute
    multi = world > 1 or bool(os.environ.get('PMG_BENCH_FORCE_DIST'))
    if multi:
        import torch
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo')  # rendezvous + barrier only; the data path is RCCL inside the library
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus

    N, K, W, T = args.envs_per_gpu, args.steps, args.warmup, args.episode_steps
    from pybullet_multigoal_gym_amd._lib import PmgLibrary
    env = pmg.make_env(task=args.task, num_envs=N, num_block=4, device=local_rank, seed=0, seed_stride=1,
                       env_index_offset=rank * N, max_episode_steps=T, binary_reward=not args.dense_reward,
                       _library=PmgLibrary(args.lib) if args.lib else None)
    h = env.handle
    A = env.dims.action_dim
    gathered = None
    host_gather = None
    collective = 'RCCL all-gather of packed obs'
    if multi:
        uid = [h.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        try:
            if os.environ.get('PMG_BENCH_FORCE_COMM_FAIL'):   # test hook for the fallback below
                raise RuntimeError('forced by PMG_BENCH_FORCE_COMM_FAIL')
            h.comm_init(rank, world, uid[0])
            ok = 1
        except Exception as ex:   # noqa: BLE001 -- reported below, never silent
            print('rank %d: RCCL communicator failed (%s)' % (rank, ex), file=sys.stderr, flush=True)
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag[0]) == 1:
            gathered = h.device_alloc(world * N * env.dims.packed_dim * 4)
        else:
            # every rank takes the SAME fallback, and the JSON line says so: packed rows to the host, gloo all-gather
            collective = 'FALLBACK: RCCL init failed, host gloo all-gather of packed obs (PCIe-inclusive)'
            mine = torch.empty((N, env.dims.packed_dim), dtype=torch.float32)
            host_gather = (mine, [torch.empty_like(mine) for _ in range(world)])
    # synthetic random policy: a table of K+W batches of U(-1,1) float32 actions, resident in HBM
    table = np.random.RandomState(12345 + rank).uniform(-1, 1, (K + W, N, A)).astype(np.float32)
    actions = h.device_alloc(table.nbytes)
    h.upload(actions, table)
    stride = N * A * 4

    def run(first, count):
        for t in range(first, first + count):
            if t % T == 0:
                h.reset_device(None)
            h.step_device(actions + t * stride)
            if gathered is not None:
                h.allgather_packed(gathered)
            elif host_gather is not None:
                h.sync()
                h.download(host_gather[0].numpy(), h.device_ptr())
                dist.all_gather(host_gather[1], host_gather[0])

    def fence():
        h.sync()                       # the library's stream: every kernel and the all-gather
        if 'torch' in sys.modules and sys.modules['torch'].cuda.is_initialized():
            sys.modules['torch'].cuda.synchronize()
        if multi:
            dist.barrier()

    run(0, W)
    fence()
    h.timing_reset()
    t0 = time.perf_counter()
    run(W, K)
    fence()
    el = time.perf_counter() - t0
    if multi:
        tt = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)   # slowest rank
        el = float(tt[0])
    kernel_ms, launches = h.timing_read()

    if rank == 0:
        value = world * N * K / el
        algo = ALGO_BYTES[args.task] * N
        achieved = algo / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        out = {
            'metric': 'env-steps/sec at N_envs=4096/GPU, KukaReach' if args.task == 'reach' and N == 4096
                      else 'env-steps/sec at N_envs=%d/GPU, %s' % (N, args.task),
            'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': el / K * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': "task='%s', %d vectorised envs/GPU, random policy U(-1,1), state obs, %s reward, "
                                   'reset every %d steps, 100 substeps/env-step'
                                   % (args.task, N, 'dense' if args.dense_reward else 'binary', T),
                       'global_envs': world * N, 'parallelism': 'env-shard x%d, %s' % (world, collective)},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': committed_traffic(args.task, N),
                         'kernel': 'pmg_k_step_reach (+ pmg_k_redo)' if args.task == 'reach' else 'pmg_k_step<NB,MAXC,CYL>', 'kernel_ms': kernel_ms, 'launches': launches,
                         'algorithmic_bytes_per_env_step': ALGO_BYTES[args.task],
                         'note': 'serial 100-substep rigid-body chain per env held in registers: HBM-light by '
                                 'construction (SURVEY.md 8d); the binding resource is the dependency latency of that chain '
                                 '(slowest wavefront = envs with finger-table contacts) and VALU issue, see roofline.valu'},
        }
        vi = committed_valu(args.task, N)
        if vi is not None and kernel_ms > 0:
            # the resource that actually binds this path: wave64 VALU issue, 1 instruction / 4 cycles / SIMD,
            # 1024 SIMDs at the 2.4 GHz peak engine clock (MI355X_MICROARCH.md)
            out['roofline']['valu'] = {'insts_per_launch': vi, 'peak_insts_per_s': 1024 * 2.4e9 / 4,
                                       'util': vi / (kernel_ms * 1e-3) / (1024 * 2.4e9 / 4),
                                       'source': 'SQ_INSTS_VALU of the committed rocprofv3 --pmc pass over the live kernel time'}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.task)
        print(json.dumps(out), flush=True)
    h.device_free(actions)
    if gathered is not None:
        h.device_free(gathered)
    env.close()
    if multi:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
