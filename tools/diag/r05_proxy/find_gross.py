import sys, os, json, pickle
sys.path.insert(0,'/root/repo/tools'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import oracle_lib
oracle_lib.ODIR='/tmp/orc_exp'
import scripted_policies as SP
task=sys.argv[1]; N=int(sys.argv[2]); T=int(sys.argv[3]); mode=sys.argv[4]
nbs={'block_stack':4,'block_rearrange':3,'chest_push':2,'chest_pick_and_place':2}
kw={'num_block':nbs[task]} if task in nbs else {}
pol=None
if mode=='scripted':
    if task.startswith('chest'): kw['num_block']=1
    if task=='block_rearrange': kw['num_block']=2
    kw['max_episode_steps']=T
    pol=SP.make_policy(task,N,**({'num_block':kw['num_block']} if 'num_block' in kw else {}))
o64=oracle_lib.OracleEnv(task,N,seed_base=0,seed_stride=1,threads=8,**kw); o64.reset(); obs=o64.reset()
o32=oracle_lib.OracleEnv(task,N,seed_base=0,seed_stride=1,threads=8,f32=True,**kw); o32.reset(); o32.reset()
rs=np.random.RandomState(12345); A=o64.dims.action_dim
cases=[]
for t in range(T):
    a = rs.uniform(-1,1,(N,A)).astype(np.float32) if pol is None else pol.act(obs)
    s0=o64.get_state()
    o32.set_state(s0); o32.step(a); sd=o32.get_state()
    obs,_,_,_=o64.step(a); so=o64.get_state()
    err=np.abs(sd[:,0:7]-so[:,0:7]).max(1)
    for i in np.nonzero(err>1e-3)[0]:
        cases.append((t,int(i),s0[i].copy(),a[i].copy(),float(err[i])))
print(len(cases),'gross q_arm steps')
pickle.dump((task,kw,cases),open('/tmp/gross_%s_%s.pkl'%(task,mode),'wb'))
