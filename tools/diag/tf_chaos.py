import sys, json
sys.path.insert(0, 'tests'); sys.path.insert(0, 'tools')
import oracle_lib, scripted_policies as SP, teacher_forced as TF
task = sys.argv[1]; N = int(sys.argv[2]); T = int(sys.argv[3]); nbk = int(sys.argv[4]) if len(sys.argv) > 4 else 0
kw = {'max_episode_steps': T}
if nbk: kw['num_block'] = nbk
pkw = {'num_block': nbk} if nbk else {}
r = TF.run(task, N, T, kw, device=False, threads=8, policy=SP.make_policy(task, N, **pkw), perturb=2)
for k in ('tip_pos', 'block_pos', 'q_arm'):
    print(k, 'f32 oracle:', {a: r['stats'][k][a] for a in ('p99', 'n_gt_1e-3', 'n')}, 'chaos:', r['chaos'][k])
