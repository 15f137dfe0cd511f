import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, 'tools')
import oracle_lib, scripted_policies as SP, teacher_forced as TF
task = sys.argv[1] if len(sys.argv) > 1 else 'chest_push'
N = 64
kw = {'num_block': 1, 'max_episode_steps': 360} if task == 'chest_push' else {'max_episode_steps': 60}
pkw = {'num_block': 1} if 'num_block' in kw else {}
f32 = TF.run(task, N, kw['max_episode_steps'], kw, device=False, threads=8, policy=SP.make_policy(task, N, **pkw))
for k in ('tip_pos', 'block_pos', 'q_arm'):
    print(k, {a: f32['stats'][k][a] for a in ('p99', 'p99.9', 'n_gt_1e-3', 'n')})
