import sys, numpy as np, pickle
sys.path.insert(0,'tests')
import oracle_lib
bad=pickle.load(open('/tmp/bad_steps.pkl','rb'))
t,i,e,eq,s0,a=bad[int(sys.argv[1])]
f32 = sys.argv[2]=='f32'
env=oracle_lib.OracleEnv('chest_pick_and_place',1,seed_base=0,seed_stride=1,threads=1,f32=f32,num_block=2); env.reset(); env.reset()
env.set_state(s0[None]); env.step(a[None])
s=env.get_state()[0]
print('door', s[48], 'q', s[:7], file=sys.stdout)
