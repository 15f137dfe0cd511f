import sys, ctypes as C, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import pybullet_multigoal_gym_amd as pmg
from pybullet_multigoal_gym_amd._lib import PmgLibrary
emu=PmgLibrary('/root/repo/tests/emu/libpmg_emu.so')
lib=C.CDLL(emu.path); lib.pmge_cyl_contact_count.restype=C.c_longlong; lib.pmge_cyl_redo_count.restype=C.c_longlong
task=sys.argv[1]; N=int(sys.argv[2]); T=int(sys.argv[3])
kw={'num_block':2} if task.startswith('chest') else {}
env=pmg.make_env(task=task,num_envs=N,seed=0,seed_stride=1,_library=emu,**kw)
env.reset()
rs=np.random.RandomState(1)
A=env.action_space.shape[-1]
pc=pr=0
for t in range(T):
    a=rs.uniform(-1,1,(N,A)).astype(np.float32)
    env.step(a)
    c,r=lib.pmge_cyl_contact_count(), lib.pmge_cyl_redo_count()
    print(t, c-pc, r-pr, flush=True); pc,pr=c,r
