import sys, subprocess, os, collections
import numpy as np
import concurrent.futures as cf
pkl=sys.argv[1]; n=int(sys.argv[2]); total=int(sys.argv[3])
def one(i):
    r=[]
    for which in ('f64','f32'):
        env=dict(os.environ, EXP_TRACE='1')
        p=subprocess.run([sys.executable,'/tmp/trace_case.py',pkl,str(i),which],env=env,capture_output=True,text=True)
        l=[x for x in p.stderr.split('BEGIN\n')[-1].splitlines() if x.startswith('IK')][0].split()
        r.append((int(l[1].split('=')[1]), np.array([float(x) for x in l[2:]])))
    return i, r[0][0], r[1][0], np.abs(r[0][1]-r[1][1]).max()
idx=list(range(0,total,max(1,total//n)))
with cf.ThreadPoolExecutor(8) as ex:
    res=list(ex.map(one,idx))
print('iters64 iters32 maxdiff')
d=np.array([r[3] for r in res])
print('n',len(res),'ik target diff >1e-4:',(d>1e-4).sum(),'>1e-5:',(d>1e-5).sum(),'iters differ:',sum(r[1]!=r[2] for r in res))
print(collections.Counter((r[1],r[2]) for r in res).most_common(8))
print(np.percentile(d,[50,90,99,100]))
