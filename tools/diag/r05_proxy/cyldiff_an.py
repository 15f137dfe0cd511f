import sys, os, ctypes as C, collections
import numpy as np
sys.path.insert(0,'/root/repo/tests')
import oracle_lib
oracle_lib.ODIR='/tmp/orc_exp'
L64=oracle_lib.load(False)
rows=[]
for l in open('/tmp/cyldiff.txt'):
    if l.startswith('CYLDIFF'):
        p=l.split(); rows.append((int(p[1]),int(p[2]),np.array([float(x) for x in p[3:]])))
print(len(rows))
def call(cc,Rc,rad,hl,cb,Rb,hb):
    out=np.zeros(40)
    arr=lambda x: np.ascontiguousarray(x,np.float64).ctypes.data_as(C.c_void_p)
    L64.pmgo_cyl_box.restype=C.c_int
    n=L64.pmgo_cyl_box(arr(cc),arr(Rc),C.c_double(rad),C.c_double(hl),arr(cb),arr(Rb),arr(hb),C.c_double(0.002),out.ctypes.data_as(C.c_void_p))
    return n,out[:10*n].reshape(n,10).copy()
cnt=collections.Counter()
np.set_printoptions(precision=7,suppress=True,linewidth=220)
shown=0
for n0,n1,v in rows:
    a0=v[0:3];A0=v[3:12];b0=v[12:15];B0=v[15:24];a1=v[24:27];A1=v[27:36];b1=v[36:39];B1=v[39:48];rad,hl=v[48],v[49];hb=v[50:53]
    key=(round(rad,4),round(hl,4),tuple(np.round(hb,4)))
    cnt[(key,n0,n1)]+=1
    if shown<6 and cnt[(key,n0,n1)]<=2:
        shown+=1
        r0=call(a0,A0,rad,hl,b0,B0,hb); r1=call(a1,A1,rad,hl,b1,B1,hb)
        print(key,'pose diff: cyl c',np.abs(a0-a1).max(),'R',np.abs(A0-A1).max(),'box c',np.abs(b0-b1).max(),'R',np.abs(B0-B1).max())
        print(' float poses:',r0[0]); print(r0[1]); print(' double poses:',r1[0]); print(r1[1])
for k,v in cnt.most_common(12): print(v,k)
