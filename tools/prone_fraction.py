import sys, numpy as np
sys.path.insert(0,'/root/repo')
import pybullet_multigoal_gym_amd as pmg
N=4096
env=pmg.make_env(task='reach',num_envs=N,seed=0,seed_stride=1)
rs=np.random.RandomState(12345)
env.reset()
fr=[]
for t in range(50):
    a=rs.uniform(-1,1,(N,3)).astype(np.float32)
    z=env.get_state()[:,20]
    zn=np.clip(z+a[:,2]*0.01,0.175,0.55)
    fr.append(float((np.minimum(z,zn)<0.187).mean()))
    env.step(a)
print('prone fraction by step:', [round(f,3) for f in fr[::5]], 'mean', round(float(np.mean(fr)),3), 'max', round(max(fr),3))
# actual contact fraction (fingers within margin): tip z < 0.177
