import sys, numpy as np, pickle
sys.path.insert(0,'tests'); sys.path.insert(0,'tools')
import oracle_lib
task='chest_pick_and_place'; kw={'num_block':2}; N=1024; T=50
o64=oracle_lib.OracleEnv(task,N,seed_base=0,seed_stride=1,threads=8,**kw); o64.reset(); o64.reset()
o32=oracle_lib.OracleEnv(task,N,seed_base=0,seed_stride=1,threads=8,f32=True,**kw); o32.reset(); o32.reset()
rs=np.random.RandomState(12345); A=o64.dims.action_dim
bad=[]
for t in range(T):
    a=rs.uniform(-1,1,(N,A)).astype(np.float32)
    s0=o64.get_state().copy()
    o32.set_state(s0); o32.step(a); s32=o32.get_state()
    o64.step(a); s64=o64.get_state()
    e=np.abs(s32[:,48]-s64[:,48])           # door q
    eq=np.abs(s32[:,:7]-s64[:,:7]).max(1)
    for i in np.nonzero((e>1e-3)|(eq>1e-3))[0]:
        bad.append((t,int(i),float(e[i]),float(eq[i]),s0[i].copy(),a[i].copy()))
print(len(bad),'bad steps; first few:',[(b[0],b[1],round(b[2],5),round(b[3],5)) for b in bad[:12]])
pickle.dump(bad,open('/tmp/bad_steps.pkl','wb'))
