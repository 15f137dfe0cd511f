import sys, os, ctypes as C
import numpy as np
sys.path.insert(0,'/root/repo/tests')
import oracle_lib
oracle_lib.ODIR='/tmp/orc_exp'
rows=[l.split() for l in open('/tmp/cyldiff.txt') if l.startswith('CYLDIFF')]
hbsel=tuple(float(x) for x in sys.argv[1].split(',')); nth=int(sys.argv[2]); which=int(sys.argv[3])
k=0
for p in rows:
    v=np.array([float(x) for x in p[3:]])
    if tuple(np.round(v[50:53],4))==hbsel:
        if k==nth: break
        k+=1
lib=oracle_lib.load(which==0)
o=24*which
cc=v[o:o+3];Rc=v[o+3:o+12];cb=v[o+12:o+15];Rb=v[o+15:o+24];rad,hl=v[48],v[49];hb=v[50:53]
out=np.zeros(40)
arr=lambda x: np.ascontiguousarray(x,np.float64).ctypes.data_as(C.c_void_p)
print('a',Rc[[2,5,8]],'cc',cc,'cb',cb,'hb',hb); sys.stdout.flush()
n=lib.pmgo_cyl_box(arr(cc),arr(Rc),C.c_double(rad),C.c_double(hl),arr(cb),arr(Rb),arr(hb),C.c_double(0.002),out.ctypes.data_as(C.c_void_p))
np.set_printoptions(precision=7,suppress=True,linewidth=220)
print(n,out[:10*n].reshape(n,10))
