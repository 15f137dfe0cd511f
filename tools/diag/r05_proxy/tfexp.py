import sys, os, json
sys.path.insert(0,'/root/repo/tools'); sys.path.insert(0,'/root/repo/tests')
import oracle_lib
oracle_lib.ODIR = os.environ.get('ORC_DIR', '/tmp/orc_exp')
import teacher_forced as TF, scripted_policies as SP
task=sys.argv[1]; N=int(sys.argv[2]); T=int(sys.argv[3]); mode=sys.argv[4]  # scripted|random
nbs={'block_stack':4,'block_rearrange':3,'chest_push':2,'chest_pick_and_place':2}
kw={'num_block':nbs[task]} if task in nbs else {}
pol=None
if mode=='scripted':
    if task.startswith('chest'): kw['num_block']=1
    if task=='block_rearrange': kw['num_block']=2
    kw['max_episode_steps']=T
    pol=SP.make_policy(task,N,**({'num_block':kw['num_block']} if 'num_block' in kw else {}))
r=TF.run(task,N,T,kw,device=False,threads=8,policy=pol,perturb=2)
out={q:(r['stats'][q]['n_gt_1e-3'], r['chaos'][q]['floor_per_perturbed_oracle'], '%.1e'%r['stats'][q]['p99']) for q in ('tip_pos','block_pos','q_arm','door_q') if q in r['stats']}
print(task, mode, N, T, 'EXP_SAT64=%s'%os.environ.get('EXP_SAT64'), out)
