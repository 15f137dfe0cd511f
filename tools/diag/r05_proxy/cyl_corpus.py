import sys, os, subprocess, pickle, ctypes as C
import numpy as np
sys.path.insert(0,'/root/repo/tests')
import oracle_lib
oracle_lib.ODIR='/tmp/orc_exp'
pkl=sys.argv[1]; idxs=[int(x) for x in sys.argv[2].split(',')]
L64=oracle_lib.load(False); L32=oracle_lib.load(True)
def call(lib,v):
    cc=v[0:3]; Rc=v[3:12]; rad=v[12]; hl=v[13]; cb=v[14:17]; Rb=v[17:26]; hb=v[26:29]
    out=np.zeros(40)
    arr=lambda x: np.ascontiguousarray(x,np.float64).ctypes.data_as(C.c_void_p)
    lib.pmgo_cyl_box.restype=C.c_int
    n=lib.pmgo_cyl_box(arr(cc),arr(Rc),C.c_double(rad),C.c_double(hl),arr(cb),arr(Rb),arr(hb),C.c_double(0.002),out.ctypes.data_as(C.c_void_p))
    return n,out[:10*n].reshape(n,10).copy()
corpus=[]
tot=0
for i in idxs:
    env=dict(os.environ, EXP_TRACE='2')
    p=subprocess.run([sys.executable,'/tmp/trace_case.py',pkl,str(i),'f32'],env=env,capture_output=True,text=True)
    for line in p.stderr.split('BEGIN\n')[-1].splitlines():
        if line.startswith(' CYL'):
            v=np.array([float(x) for x in line.split()[1:]])
            tot+=1
            n64,o64=call(L64,v); n32,o32=call(L32,v)
            bad=None
            if n64!=n32: bad='count %d %d'%(n64,n32)
            elif n64:
                dn=np.abs(o64[:,6:9]-o32[:,6:9]).max(); dd=np.abs(o64[:,9]-o32[:,9]).max(); dp=np.abs(o64[:,0:3]-o32[:,0:3]).max()
                if dn>1e-4 or dd>2e-5 or dp>1e-3: bad='val dn=%.1e dd=%.1e dp=%.1e'%(dn,dd,dp)
            if bad: corpus.append((v,bad,o64,o32))
print(tot,'cyl_box calls,',len(corpus),'differ')
import collections
print(collections.Counter(c[1].split()[0] for c in corpus))
for c in corpus[:10]: print(c[1]); print(c[2]); print(c[3])
pickle.dump(corpus,open('/tmp/cyl_corpus.pkl','wb'))
