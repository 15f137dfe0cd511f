import sys, os, pickle, ctypes as C
import numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
emu=C.CDLL(sys.argv[1] if len(sys.argv)>1 else '/root/repo/tests/emu/libpmg_emu.so')
emu.pmge_probe_narrowphase.restype=C.c_int
c=pickle.load(open('/tmp/cyl_corpus_base.pkl','rb'))
fp=lambda x: x.ctypes.data_as(C.c_void_p)
big=0; bad=0; small_bad=0
for v,tag,o64,o32 in c:
    cc=np.float32(v[0:3]); Rc=np.float32(v[3:12]); ha=np.float32([v[12],v[12],v[13]]); cb=np.float32(v[14:17]); Rb=np.float32(v[17:26]); hb=np.float32(v[26:29])
    out=np.zeros(40,np.float32)
    n=emu.pmge_probe_narrowphase(1,fp(cc),fp(Rc),fp(ha),fp(cb),fp(Rb),fp(hb),C.c_float(0.002),fp(out))
    got=out.reshape(4,10)[:n]
    isbig = tag.startswith('count') or float(tag.split('dp=')[1])>1e-3 or float(tag.split('dn=')[1].split()[0])>3e-3
    big+=isbig
    if n!=len(o64): d=1.0
    else: d=max(np.abs(got[:,6:9]-o64[:,6:9]).max(), np.abs(got[:,0:3]-o64[:,0:3]).max()*3) if n else 0
    if d>3e-3:
        if isbig: bad+=1
        else: small_bad+=1
print('corpus',len(c),'big(f32 oracle vs f64)',big,' device still far from f64 on big:',bad,' on small:',small_bad)
