import sys, os, subprocess, ctypes as C, numpy as np
sys.path.insert(0,'/root/repo/tests')
import oracle_lib
oracle_lib.ODIR='/tmp/orc_exp'
pkl=sys.argv[1]; idx=sys.argv[2]; sub=int(sys.argv[3]); which=sys.argv[4]; k=int(sys.argv[5]) if len(sys.argv)>5 else -1
env=dict(os.environ, EXP_TRACE='2')
p=subprocess.run([sys.executable,'/tmp/trace_case.py',pkl,idx,'f32'],env=env,capture_output=True,text=True)
lines=p.stderr.split('BEGIN\n')[-1].splitlines()
# CYL lines precede the SUB line of their substep
subs=[]; cur=[]
for l in lines:
    if l.startswith(' CYL'): cur.append(l)
    elif l.startswith('SUB'): subs.append(cur); cur=[]
cyl=subs[sub]
print(len(cyl),'cyl_box calls in substep',sub)
lib=oracle_lib.load(which=='f32')
for j,l in enumerate(cyl):
    if k>=0 and j!=k: continue
    v=np.array([float(x) for x in l.split()[1:]])
    cc=v[0:3]; Rc=v[3:12]; rad=v[12]; hl=v[13]; cb=v[14:17]; Rb=v[17:26]; hb=v[26:29]
    print('call',j,'cc',cc,'a',Rc[[2,5,8]],'rad',rad,'hl',hl,'cb',cb,'hb',hb)
    sys.stdout.flush()
    out=np.zeros(40)
    arr=lambda x: np.ascontiguousarray(x,np.float64).ctypes.data_as(C.c_void_p)
    os.environ['EXP_TRACE']='3'
    n=lib.pmgo_cyl_box(arr(cc),arr(Rc),C.c_double(rad),C.c_double(hl),arr(cb),arr(Rb),arr(hb),C.c_double(0.002),out.ctypes.data_as(C.c_void_p))
    print(n,out[:10*n])
