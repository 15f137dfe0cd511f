import sys, os, pickle
sys.path.insert(0,'/root/repo/tools'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import oracle_lib
oracle_lib.ODIR=os.environ.get('ORC_DIR','/tmp/orc_exp')
task,kw,cases=pickle.load(open(sys.argv[1],'rb'))
t,i,s0,a,err=cases[int(sys.argv[2])]
which=sys.argv[3]
e=oracle_lib.OracleEnv(task,1,seed_base=0,seed_stride=1,threads=1,f32=(which=='f32'),**kw); e.reset(); e.reset()
e.set_state(s0[None,:].copy())
if which=='floor': oracle_lib.set_prior('state_f32_per_substep',1.0)
sys.stderr.write('BEGIN\n'); sys.stderr.flush()
e.step(a[None,:].copy())
s=e.get_state()
print(' '.join('%.9g'%x for x in s[0,:18]))
